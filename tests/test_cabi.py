"""The C-ABI library loads on a CPU-only host and exports every symbol include/comorag_hip.h
declares; compute entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from comorag_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "comorag_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cmr_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = L.lib()
    syms = _header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/comorag_hip.h but not exported"
    assert sorted(L.SIGNATURES) == syms, "ctypes binding table and header drifted"
    assert lib.cmr_abi_version() == 2


def test_host_merge_is_pure_host_code():
    from comorag_amd.index import merge_topk
    ids = np.array([[[5, 9, -1], [0, 1, 2]], [[7, 3, 8], [4, -1, -1]]], dtype=np.int64)     # [S=2, nq=2, k=3]
    sc = np.array([[[0.9, 0.2, -np.inf], [0.5, 0.5, 0.1]], [[0.9, 0.8, 0.1], [0.5, -np.inf, -np.inf]]], dtype=np.float32)
    oi, os_ = merge_topk(ids, sc)
    assert oi.tolist() == [[5, 7, 3], [0, 1, 4]]            # ties: lower id first (5<7 at 0.9; 0<1<4 at 0.5)
    assert np.allclose(os_, [[0.9, 0.9, 0.8], [0.5, 0.5, 0.5]])


def test_compute_calls_fail_loudly_without_gpu():
    if L.device_count() > 0:
        pytest.skip("GPU present")
    from comorag_amd.index import DenseIndex
    with pytest.raises(L.CmrError) as e:
        DenseIndex(16, "bf16")
    assert e.value.code == L.CMR_ERR_NO_DEVICE and "no CPU fallback" in str(e.value)
    out = np.zeros((1, 4), dtype=np.float32)
    rc = L.lib().cmr_pool_l2norm(0, C.c_void_p(out.ctypes.data), 0, C.c_void_p(out.ctypes.data), 1, 1, 4, 1,
                                 C.c_void_p(out.ctypes.data), None)
    assert rc == L.CMR_ERR_NO_DEVICE
    with pytest.raises(RuntimeError):
        from comorag_amd.embedding_model.bge import HipBGEEmbeddingModel
        HipBGEEmbeddingModel(embedding_model_name="bge-x", model=object(), tokenizer=object())


def test_product_never_imports_oracle():
    """comorag_amd/ must not import, call or link anything under oracle/."""
    for dp, _, fs in os.walk(os.path.join(ROOT, "comorag_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "retrieval_np" not in src and "ref_loader" not in src, f


def test_new_entry_points_reject_null_handles_without_a_device():
    """The round's new symbols (options, id block table, communicator info, threshold search on device buffers) fail with
    CMR_ERR_INVALID on NULL handles — no device call is made before the argument check, so this holds on a CPU-only host."""
    lib = L.lib()
    v = C.c_int64(0)
    assert lib.cmr_index_set_option(None, b"scan_no_wide", 1) == L.CMR_ERR_INVALID
    assert lib.cmr_index_get_option(None, b"pipe_scan_cus", C.byref(v)) == L.CMR_ERR_INVALID
    a = np.zeros(2, np.int64)
    assert lib.cmr_index_set_id_blocks(None, 2, a.ctypes.data, a.ctypes.data) == L.CMR_ERR_INVALID
    w, r, n = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    assert lib.cmr_comm_info(None, C.byref(w), C.byref(r), C.byref(n)) == L.CMR_ERR_INVALID
    assert lib.cmr_index_search_min_score_dev(None, None, 1, 1, 0.5, None, None, None) == L.CMR_ERR_INVALID
    assert b"NULL" in lib.cmr_last_error()
