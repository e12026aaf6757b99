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
