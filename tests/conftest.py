import hashlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun / the driver's GPU tier)")


def _has_gpu() -> bool:
    # A missing / stale libcomorag_hip.so must FAIL the run (ImportError / AttributeError propagate),
    # never silently skip the GPU tests; only "no device visible" skips them.
    from comorag_amd import _lib as L
    return L.device_count() > 0


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


class FakeEmbedder:
    """Deterministic text → unit vector (md5-seeded gaussian).  Same construction as
    oracle/make_golden.py:FakeEmbedder, which produced the store/summaries/cinderella fixtures."""

    def __init__(self, dim=32):
        self.embedding_dim = dim
        self.calls = []

    def _vec(self, text):
        seed = int.from_bytes(hashlib.md5(text.encode()).digest()[:8], "little")
        v = np.random.default_rng(seed).standard_normal(self.embedding_dim).astype(np.float32)
        return v / np.linalg.norm(v)

    def batch_encode(self, texts, **kw):
        if isinstance(texts, str):
            texts = [texts]
        self.calls.append(list(texts))
        return np.stack([self._vec(t) for t in texts]).astype(np.float32)

    def encode(self, texts, **kw):
        import torch
        return torch.from_numpy(self.batch_encode(texts))


@pytest.fixture
def fake_embedder():
    return FakeEmbedder(32)


class NumpyIndex:
    """numpy stand-in with DenseIndex's call surface (append / scores / search / sorted_scores / len / close), fp32,
    exported tie rule.  TEST INFRASTRUCTURE: lets the CPU tier drive the binding glue (comorag_amd.hooks, retrieval.*)
    on the real reference classes; the product never sees it."""

    def __init__(self, dim, dtype="f32", device=0, **kw):
        self.dim, self.dtype, self.device = dim, dtype, device
        self._x = np.empty((0, dim), np.float32)

    def append(self, rows):
        rows = np.asarray(rows, np.float32).reshape(-1, self.dim)
        self._x = np.concatenate([self._x, rows])

    def __len__(self):
        return len(self._x)

    def close(self):
        pass

    def scores(self, q):
        q = np.asarray(q, np.float32).reshape(-1, self.dim)
        return (q @ self._x.T).astype(np.float32)

    def search(self, q, k, with_minmax=True):
        s = self.scores(q)
        k = min(k, s.shape[1])
        ids = np.stack([np.lexsort((np.arange(s.shape[1]), -r))[:k] for r in s])
        sc = np.take_along_axis(s, ids, axis=1)
        return ids.astype(np.int64), sc, (s.min(1) if with_minmax else None), (s.max(1) if with_minmax else None)

    def sorted_scores(self, q):
        s = self.scores(q)
        ids = np.stack([np.lexsort((np.arange(s.shape[1]), -r)) for r in s])
        return ids.astype(np.int64), np.take_along_axis(s, ids, axis=1), s.min(1), s.max(1)


@pytest.fixture
def numpy_index_cls():
    return NumpyIndex
