"""Base class, dict-like config and the optional sqlite embedding cache.

Mirrors src/comorag/embedding_model/base.py: `EmbeddingConfig` (:22-104), `make_cache_embed`
(:112-187), `BaseEmbeddingModel` (:189-218).  Same public names and call conventions; new code.
"""
from __future__ import annotations

import hashlib
import json
import sqlite3
from typing import Any, Dict, List, Optional

import numpy as np

from ..utils.config_utils import BaseConfig


class EmbeddingConfig:
    """Attribute- and item-addressable bag of settings (reference: a dataclass wrapping `_data`)."""

    def __init__(self):
        object.__setattr__(self, "_data", {})

    def __getattr__(self, key: str) -> Any:
        if key.startswith(("_ipython_", "_repr_")) or key not in self._data:
            raise AttributeError(f"'{self.__class__.__name__}' object has no attribute '{key}'")
        return self._data[key]

    def __setattr__(self, key: str, value: Any) -> None:
        self._data[key] = value

    def __delattr__(self, key: str) -> None:
        if key not in self._data:
            raise AttributeError(f"'{self.__class__.__name__}' object has no attribute '{key}'")
        del self._data[key]

    def __getitem__(self, key: str) -> Any:
        if key not in self._data:
            raise KeyError(f"'{key}' not found in configuration.")
        return self._data[key]

    def __setitem__(self, key: str, value: Any) -> None:
        self._data[key] = value

    def __delitem__(self, key: str) -> None:
        if key not in self._data:
            raise KeyError(f"'{key}' not found in configuration.")
        del self._data[key]

    def __contains__(self, key: str) -> bool:
        return key in self._data

    def batch_upsert(self, updates: Dict[str, Any]) -> None:
        self._data.update(updates)

    def to_dict(self) -> Dict[str, Any]:
        return self._data

    def to_json(self) -> str:
        return json.dumps(self._data)

    @classmethod
    def from_dict(cls, config_dict: Dict[str, Any]) -> "EmbeddingConfig":
        inst = cls()
        inst.batch_upsert(config_dict)
        return inst

    @classmethod
    def from_json(cls, json_str: str) -> "EmbeddingConfig":
        return cls.from_dict(json.loads(json_str))

    def __str__(self) -> str:
        return json.dumps(self._data, indent=4, default=str)


def make_cache_embed(encode_func, cache_file_name, device):
    """Per-prompt sqlite cache around an `encode(**kwargs)` callable (base.py:112-187): key =
    sha256 of {"instruction", "promps", "max_length"} (the reference's spelling, kept so existing
    cache files hit), value = fp32 blob.  Keyword-only like the reference wrapper; returns a torch
    tensor [n, D] on `device`."""
    import torch
    from filelock import FileLock

    lock_file = cache_file_name + ".lock"

    def wrapper(**kwargs):
        instruction = kwargs.get("instruction", "")
        max_length = kwargs.get("max_length", "")
        prompts = kwargs["prompts"]
        keys = [hashlib.sha256(json.dumps({"instruction": instruction, "promps": p, "max_length": max_length},
                                          sort_keys=True, default=str).encode("utf-8")).hexdigest() for p in prompts]
        out: List[Optional[np.ndarray]] = [None] * len(prompts)
        with FileLock(lock_file), sqlite3.connect(cache_file_name) as conn:
            conn.execute("CREATE TABLE IF NOT EXISTS embeddings (hash TEXT PRIMARY KEY, embedding BLOB)")
            for i, h in enumerate(keys):
                row = conn.execute("SELECT embedding FROM embeddings WHERE hash = ?", (h,)).fetchone()
                if row:
                    out[i] = np.frombuffer(row[0], dtype=np.float32).copy()
        missed = [i for i, e in enumerate(out) if e is None]
        if missed:
            kw = dict(kwargs)
            kw["prompts"] = [prompts[i] for i in missed]
            fresh = encode_func(**kw)
            fresh = fresh.detach().float().cpu().numpy() if hasattr(fresh, "detach") else np.asarray(fresh, np.float32)
            with FileLock(lock_file), sqlite3.connect(cache_file_name) as conn:
                for j, i in enumerate(missed):
                    out[i] = np.ascontiguousarray(fresh[j], dtype=np.float32)
                    conn.execute("INSERT OR REPLACE INTO embeddings (hash, embedding) VALUES (?, ?)", (keys[i], out[i].tobytes()))
        return torch.from_numpy(np.stack(out)).to(device)

    return wrapper


class BaseEmbeddingModel:
    global_config: BaseConfig
    embedding_model_name: str
    embedding_config: EmbeddingConfig
    embedding_dim: int

    def __init__(self, global_config: Optional[BaseConfig] = None) -> None:
        self.global_config = BaseConfig() if global_config is None else global_config
        self.embedding_model_name = self.global_config.embedding_model_name

    def batch_encode(self, texts: List[str], **kwargs) -> np.ndarray:
        raise NotImplementedError

    def get_query_doc_scores(self, query_vec: np.ndarray, doc_vecs: np.ndarray):
        return np.dot(query_vec, doc_vecs.T)   # base.py:212-218 (unused by callers)
