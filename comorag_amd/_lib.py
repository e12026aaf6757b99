"""ctypes binding of libcomorag_hip.so (include/comorag_hip.h).

There is no CPU fallback: if the shared library is missing this module raises at
import of the symbol table (`lib()`), and every compute entry point returns
CMR_ERR_NO_DEVICE on a host without a gfx950 GPU (raised as `CmrError`).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("COMORAG_HIP_LIB", os.path.join(_HERE, "lib", "libcomorag_hip.so"))

CMR_OK = 0
CMR_ERR_INVALID, CMR_ERR_NO_DEVICE, CMR_ERR_HIP, CMR_ERR_OOM, CMR_ERR_NONFINITE, CMR_ERR_UNSUPPORTED = -1, -2, -3, -4, -5, -6
CMR_F32, CMR_BF16, CMR_F16 = 0, 1, 2
CMR_FLAG_KEEP_F32 = 1
CMR_MAX_K = 128
ABI_VERSION = 2        # include/comorag_hip.h: CMR_ABI_VERSION
CMR_MAX_K_2PASS = 4096
DTYPES = {"f32": CMR_F32, "fp32": CMR_F32, "float32": CMR_F32, "bf16": CMR_BF16, "bfloat16": CMR_BF16,
          "f16": CMR_F16, "fp16": CMR_F16, "float16": CMR_F16}


class CmrError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libcomorag_hip error {code}: {msg}")
        self.code = code


_i32, _i64, _u32, _f32, _f64 = C.c_int32, C.c_int64, C.c_uint32, C.c_float, C.c_double
_p = C.c_void_p
_P = C.POINTER

# name -> (restype, argtypes); mirrors include/comorag_hip.h one to one (tests check the set)
SIGNATURES = {
    "cmr_abi_version": (_i32, []),
    "cmr_last_error": (C.c_char_p, []),
    "cmr_device_count": (_i32, [_P(_i32)]),
    "cmr_device_info": (_i32, [_i32, C.c_char_p, _i32, _P(_i32), _P(_i64)]),
    "cmr_index_create": (_i32, [_i32, _i32, _i32, _i64, _u32, _P(_p)]),
    "cmr_index_destroy": (_i32, [_p]),
    "cmr_index_size": (_i32, [_p, _P(_i64)]),
    "cmr_index_info": (_i32, [_p, _P(_i32), _P(_i32), _P(_i64), _P(_i64)]),
    "cmr_index_append": (_i32, [_p, _p, _i64]),
    "cmr_index_append_dev": (_i32, [_p, _p, _i64, _p]),
    "cmr_index_search": (_i32, [_p, _p, _i32, _i32, _p, _p, _p, _p]),
    "cmr_index_search_min_score": (_i32, [_p, _p, _i32, _i32, _f32, _p, _p]),
    "cmr_index_search_min_score_dev": (_i32, [_p, _p, _i32, _i32, _f32, _p, _p, _p]),
    "cmr_index_search_min_score_pipelined": (_i32, [_p, _p, _i32, _i32, _f32, _p, _p, _p, _P(_p)]),
    "cmr_index_search_dev": (_i32, [_p, _p, _i32, _i32, _p, _p, _p, _p, _p]),
    "cmr_index_search_pipelined": (_i32, [_p, _p, _i32, _i32, _p, _p, _p, _p, _p, _P(_p)]),
    "cmr_index_set_id_base": (_i32, [_p, _i64]),
    "cmr_index_set_id_blocks": (_i32, [_p, _i32, _p, _p]),
    "cmr_index_set_option": (_i32, [_p, C.c_char_p, _i64]),
    "cmr_index_get_option": (_i32, [_p, C.c_char_p, _P(_i64)]),
    "cmr_index_pipeline_stream": (_i32, [_p, _i32, _P(_p)]),
    "cmr_index_query_status": (_i32, [_p, _P(_i32)]),
    "cmr_stream_wait_event": (_i32, [_p, _p]),
    "cmr_event_synchronize": (_i32, [_p]),
    "cmr_index_scores": (_i32, [_p, _p, _i32, _p, _i64]),
    "cmr_index_scores_dev": (_i32, [_p, _p, _i32, _p, _i64, _p]),
    "cmr_index_sorted_scores": (_i32, [_p, _p, _i32, _p, _p, _p, _p]),
    "cmr_index_rescore": (_i32, [_p, _p, _i32, _p, _i32, _i32, _p, _p]),
    "cmr_index_get_rows": (_i32, [_p, _p, _i64, _p]),
    "cmr_merge_topk": (_i32, [_p, _p, _i32, _i32, _i32, _p, _p]),
    "cmr_merge_topk_dev": (_i32, [_i32, _p, _p, _i32, _i32, _i32, _p, _p, _p]),
    "cmr_graph_create": (_i32, [_i32, _i64, _i64, _p, _p, _p, _P(_p)]),
    "cmr_graph_destroy": (_i32, [_p]),
    "cmr_graph_set_passage_vertices": (_i32, [_p, _p, _i64]),
    "cmr_graph_ppr": (_i32, [_p, _p, _f64, _f64, _i32, _p, _P(_i32)]),
    "cmr_index_ppr": (_i32, [_p, _p, _p, _p, _p, _i32, _f64, _f64, _f64, _i32, _p, _P(_i32)]),
    "cmr_pack_candidates_dev": (_i32, [_p, _p, _i64, _p, _p]),
    "cmr_merge_keys_dev": (_i32, [_p, _i32, _i32, _i32, _p, _p, _p]),
    "cmr_comm_unique_id": (_i32, [_p]),
    "cmr_comm_create": (_i32, [_i32, _i32, _p, _i32, _P(_p)]),
    "cmr_comm_destroy": (_i32, [_p]),
    "cmr_comm_info": (_i32, [_p, _P(_i32), _P(_i32), _P(_i32)]),
    "cmr_comm_allgather_merge": (_i32, [_p, _p, _p, _i32, _i32, _p, _p, _p]),
    "cmr_mindex_create": (_i32, [_i32, _p, _i32, _i32, _i64, _u32, _P(_p)]),
    "cmr_mindex_destroy": (_i32, [_p]),
    "cmr_mindex_size": (_i32, [_p, _P(_i64)]),
    "cmr_mindex_info": (_i32, [_p, _P(_i32), _p, _p, _P(_i64)]),
    "cmr_mindex_shard": (_i32, [_p, _i32, _P(_p)]),
    "cmr_mindex_set_option": (_i32, [_p, C.c_char_p, _i64]),
    "cmr_mindex_append": (_i32, [_p, _p, _i64]),
    "cmr_mindex_append_dev": (_i32, [_p, _p, _i64, _i32, _p]),
    "cmr_mindex_search": (_i32, [_p, _p, _i32, _i32, _p, _p, _p, _p]),
    "cmr_mindex_search_min_score": (_i32, [_p, _p, _i32, _i32, _f32, _p, _p]),
    "cmr_mindex_scores": (_i32, [_p, _p, _i32, _p, _i64]),
    "cmr_mindex_sorted_scores": (_i32, [_p, _p, _i32, _p, _p, _p, _p]),
    "cmr_mindex_rescore": (_i32, [_p, _p, _i32, _p, _i32, _i32, _p, _p]),
    "cmr_mindex_get_rows": (_i32, [_p, _p, _i64, _p]),
    "cmr_mindex_search_pipelined": (_i32, [_p, _p, _i32, _i32, _P(_p)]),
    "cmr_mindex_collect": (_i32, [_p, _p, _p, _p, _p, _p]),
    "cmr_mindex_profile": (_i32, [_p, _i32, _P(_i64), _P(_f64), _P(_f64), _P(_f64)]),
    "cmr_mindex_plan_append": (_i32, [_p, _i32, _i32, _i64, _i64, _i64, _i32, _p, _p, _P(_i32), _P(_i32), _P(_i64)]),
    "cmr_pool_l2norm": (_i32, [_i32, _p, _i32, _p, _i32, _i32, _i32, _i32, _p, _p]),
    "cmr_encoder_embed_layernorm": (_i32, [_i32, _p, _p, _p, _p, _p, _p, _p, _f32, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p]),
    "cmr_encoder_embed_layernorm_ragged": (_i32, [_i32, _p, _p, _p, _p, _p, _p, _p, _f32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p]),
    "cmr_encoder_attention": (_i32, [_i32, _p, _i32, _p, _i32, _i32, _i32, _i32, _p, _p]),
    "cmr_encoder_add_layernorm": (_i32, [_i32, _p, _p, _p, _p, _p, _f32, _i64, _i32, _i32, _p, _p]),
    "cmr_encoder_add_layernorm_pool": (_i32, [_i32, _p, _p, _p, _p, _p, _f32, _i32, _i32, _i32, _i32, _p, _i32, _p, _p, _p]),
    "cmr_profile_enable": (_i32, [_p, _i32]),
    "cmr_profile_collect": (_i32, [_p, _P(_i64), _P(_f64), _P(_f64)]),
}

_lib = None
_lock = threading.Lock()


def lib() -> C.CDLL:
    """Load (once) and type the shared library.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH) and "COMORAG_HIP_LIB" not in os.environ:
            # a fresh checkout (the .so is git-ignored): build it once, under a file lock so that the
            # ranks of a multi-process launch do not race.  This is a BUILD, not a fallback.
            try:
                from filelock import FileLock
                from . import build as _build
                os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
                with FileLock(LIB_PATH + ".lock"):
                    if not os.path.exists(LIB_PATH):
                        _build.build(force=False, verbose=False)
            except Exception as e:  # hipcc missing etc.
                raise ImportError(f"{LIB_PATH} not found and could not be built ({e}). Build it with "
                                  "`python -m comorag_amd.build` (hipcc --offload-arch=gfx950). "
                                  "comorag_amd has no CPU fallback.") from e
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found. Build it with `python -m comorag_amd.build` "
                "(hipcc --offload-arch=gfx950). comorag_amd has no CPU fallback.")
        # One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64 (soname
        # without the ".7"), so loading this library first would bring up /opt/rocm's runtime and
        # torch's second runtime then finds no GPU.  When torch is installed, let it load its runtime
        # (RTLD_GLOBAL) first: this library's HIP symbols then bind to that same runtime, and torch
        # streams / tensors can be handed across the C-ABI.  Without torch the library uses the ROCm
        # install it was linked against.
        if not os.environ.get("COMORAG_HIP_NO_TORCH"):
            import importlib.util
            if importlib.util.find_spec("torch") is not None:
                import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError = header/library drift: fail loudly
            fn.restype = res
            fn.argtypes = args
        if l.cmr_abi_version() != ABI_VERSION:
            raise ImportError(f"ABI version mismatch: library {l.cmr_abi_version()} != binding {ABI_VERSION}")
        _lib = l
        return l


def check(rc: int) -> None:
    if rc != CMR_OK:
        msg = lib().cmr_last_error()
        raise CmrError(rc, msg.decode("utf-8", "replace") if msg else "")


def device_count() -> int:
    n = _i32(0)
    check(lib().cmr_device_count(C.byref(n)))
    return n.value


def device_info(device_id: int = 0) -> dict:
    name = C.create_string_buffer(256)
    ncu, hbm = _i32(0), _i64(0)
    check(lib().cmr_device_info(device_id, name, 256, C.byref(ncu), C.byref(hbm)))
    return {"name": name.value.decode(), "n_cu": ncu.value, "hbm_bytes": hbm.value}
