"""Import the *reference* ComoRAG (read-only, /root/reference) with in-memory stubs.

TEST INFRASTRUCTURE ONLY.  Used by ``oracle/make_golden.py`` and by the
``-m "not gpu"`` pinning tests when ``/root/reference`` exists (this container);
the GPU box has no reference tree, so nothing on the product / bench / gpu-test
path may import this module.

Recipe follows SURVEY.md §8(c): eight third-party packages the reference imports
at module scope are absent from this image (cv2 igraph openai tenacity wandb umap
tiktoken vllm); they are replaced by ``types.ModuleType`` stubs whose attributes
resolve to ``MagicMock``.  ``PYTHONDONTWRITEBYTECODE`` is forced so the read-only
tree gets no ``__pycache__``.
"""
from __future__ import annotations

import importlib
import importlib.machinery
import os
import sys
import types
from unittest import mock

REFERENCE_ROOT = os.environ.get("COMORAG_REFERENCE_ROOT", "/root/reference")
_STUBS = ("cv2", "igraph", "openai", "tenacity", "wandb", "umap", "tiktoken", "vllm",
          "faiss", "sentence_transformers", "chonkie", "dspy", "litellm", "gritlm")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "src", "comorag", "ComoRAG.py"))


def _stub(name: str) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []  # behave like a package so "from x.y import z" resolves

    def _getattr(attr, _name=name):
        if attr.startswith("__"):
            raise AttributeError(attr)
        return mock.MagicMock(name=f"{_name}.{attr}")

    m.__getattr__ = _getattr  # type: ignore[attr-defined]
    return m


def prepare_stubs():
    """Install the stubs and put the reference tree on sys.path WITHOUT importing it — for callers that alias modules
    first (comorag_amd.hooks.patch_reference_modules must run before ``src.comorag.ComoRAG`` is imported)."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    for name in _STUBS:
        try:
            importlib.import_module(name)
        except Exception:
            sys.modules[name] = _stub(name)
            for sub in ("sampling_params", "openai", "_types"):
                sys.modules.setdefault(f"{name}.{sub}", _stub(f"{name}.{sub}"))
    ten = sys.modules["tenacity"]
    if isinstance(getattr(ten, "__getattr__", None), types.FunctionType):
        ten.retry = lambda *a, **k: (lambda f: f)  # decorator passthrough
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def load_reference():
    """Returns the imported ``src.comorag`` package of the reference."""
    prepare_stubs()
    return importlib.import_module("src.comorag")


def ref_modules():
    """dict of the reference modules on the hot path."""
    load_reference()
    imp = importlib.import_module
    return {
        "ComoRAG": imp("src.comorag.ComoRAG"),
        "embedding_store": imp("src.comorag.embedding_store"),
        "bge": imp("src.comorag.embedding_model.BGEEmbedding"),
        "emb_base": imp("src.comorag.embedding_model.base"),
        "emb_init": imp("src.comorag.embedding_model"),
        "embed_utils": imp("src.comorag.utils.embed_utils"),
        "memory_utils": imp("src.comorag.utils.memory_utils"),
        "misc_utils": imp("src.comorag.utils.misc_utils"),
        "rerank": imp("src.comorag.rerank"),
        "config_utils": imp("src.comorag.utils.config_utils"),
    }
