"""OpenAIEmbeddingModel — remote HTTP embeddings (src/comorag/embedding_model/OpenAI.py:17-128).

Not on the GPU path; kept so the factory covers the same names and an `EmbeddingStore` / the retrieval hooks run on it
unchanged.  Behaviour of the reference, restated:
  * client from the configuration: `OpenAI(base_url=embedding_base_url, api_key=embedding_api_key)`, or `AzureOpenAI` when
    `azure_embedding_endpoint` is set (OpenAI.py:32-42); a ready client may be injected (`client=`: tests, proxies);
  * `encode(texts)`: newlines -> spaces, '' -> ' ', ONE embeddings.create call, `np.array` of the returned Python floats,
    i.e. **float64** (OpenAI.py:77-85) — the store persists what it is handed and casts on read;
  * `batch_encode(texts, **kw)`: str -> [str]; `batch_size` from the call, else `global_config.embedding_batch_size`; one
    call when everything fits a batch, else one per batch, concatenated; rows L2-normalised iff
    `global_config.embedding_return_as_normalized` (the `norm=` keyword is ignored, as in the reference: OpenAI.py:87-128).
Conscious deviation: a batch that fails raises.  The reference logs the exception and `pass`es (OpenAI.py:109-117), which
silently drops the batch's rows and misaligns every row behind it in the store.
Needs the `openai` package unless a client is injected.
"""
from __future__ import annotations

from copy import deepcopy
from typing import List, Optional

import numpy as np

from .base import BaseEmbeddingModel, EmbeddingConfig


class OpenAIEmbeddingModel(BaseEmbeddingModel):
    def __init__(self, global_config=None, embedding_model_name: Optional[str] = None, client=None) -> None:
        super().__init__(global_config=global_config)
        if embedding_model_name is not None:
            self.embedding_model_name = embedding_model_name
        self._init_embedding_config()
        if client is None:
            cfg = self.global_config
            if getattr(cfg, "azure_embedding_endpoint", None) is None:
                from openai import OpenAI  # ImportError here is the loud failure
                client = OpenAI(base_url=getattr(cfg, "embedding_base_url", None), api_key=getattr(cfg, "embedding_api_key", None))
            else:
                from openai import AzureOpenAI
                ep = cfg.azure_embedding_endpoint
                client = AzureOpenAI(api_version=ep.split("api-version=")[1], azure_endpoint=ep, api_key=getattr(cfg, "embedding_api_key", None))
        self.client = client
        self.embedding_dim = 1536          # text-embedding-3-small (the only name the factory maps here)

    def _init_embedding_config(self) -> None:
        cfg = self.global_config
        self.embedding_config = EmbeddingConfig.from_dict({
            "embedding_model_name": self.embedding_model_name,
            "norm": getattr(cfg, "embedding_return_as_normalized", True),
            "model_init_params": {"pretrained_model_name_or_path": self.embedding_model_name, "trust_remote_code": True, "device_map": "auto"},
            "encode_params": {"max_length": getattr(cfg, "embedding_max_seq_len", 2048), "instruction": "",
                              "batch_size": getattr(cfg, "embedding_batch_size", 32), "num_workers": 32},
        })

    def encode(self, texts: List[str]):
        texts = [t.replace("\n", " ") for t in texts]
        texts = [t if t != "" else " " for t in texts]
        response = self.client.embeddings.create(input=texts, model=self.embedding_model_name)
        return np.array([v.embedding for v in response.data])

    def batch_encode(self, texts, **kwargs) -> np.ndarray:
        if isinstance(texts, str):
            texts = [texts]
        params = deepcopy(self.embedding_config.encode_params)
        if kwargs:
            params.update(kwargs)
        batch_size = params.pop("batch_size", 16)
        if len(texts) <= batch_size:
            results = self.encode(texts)
        else:
            results = np.concatenate([self.encode(texts[i:i + batch_size]) for i in range(0, len(texts), batch_size)])
        if self.embedding_config.norm:
            results = (results.T / np.linalg.norm(results, axis=1)).T
        return results
