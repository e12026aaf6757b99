"""DenseIndex — Python handle on one HBM-resident shard (`cmr_index_t`).

Host-side mirror of the numeric half of ComoRAG's retrieval path: where the
reference keeps `np.array(store.get_embeddings(keys))` matrices on the host
(src/comorag/ComoRAG.py:896-900) and runs np.dot + argsort per query (:937-967),
this keeps the matrix in HBM (MFMA-fragment-major panels) and asks the HIP library
for top-k / full scores.  numpy in, numpy out; torch tensors (device) accepted by
the *_dev methods.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _lib as L


def _f32c(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a: np.ndarray) -> int:
    # the array's address as an int (c_void_p parameters take one): `a.ctypes.data_as(...)` builds a ctypes helper object
    # per call, 2 us each — a third of a 35 us single-query search went into marshalling five arrays
    return a.__array_interface__["data"][0]


class DenseIndex:
    def __init__(self, dim: int, dtype: str = "bf16", device: int = 0, capacity_hint: int = 0,
                 keep_f32: bool = False, options: Optional[dict] = None):
        self._h = C.c_void_p()
        self.dim = int(dim)
        self.dtype = dtype
        self.device = int(device)
        self.keep_f32 = bool(keep_f32)
        L.check(L.lib().cmr_index_create(self.device, self.dim, L.DTYPES[dtype], int(capacity_hint),
                                         L.CMR_FLAG_KEEP_F32 if keep_f32 else 0, C.byref(self._h)))
        for name, value in (options or {}).items():
            self.set_option(name, value)

    @classmethod
    def _borrow(cls, handle, dim: int, dtype: str, device: int, owner=None) -> "DenseIndex":
        """A view of a cmr_index_t somebody else owns (a shard of a MultiDeviceIndex): never destroyed from here."""
        self = cls.__new__(cls)
        self._h, self.dim, self.dtype, self.device, self.keep_f32 = handle, int(dim), dtype, int(device), False
        self._borrowed, self._owner = True, owner
        return self

    def set_option(self, name: str, value: int) -> None:
        """Route selector (cmr_index_set_option): picks between implementations that return the same results."""
        L.check(L.lib().cmr_index_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        """Read-only pipeline facts (cmr_index_get_option): "pipe_dual_scan_active", "pipe_cu_mask_active", "pipe_scan_cus"."""
        v = C.c_int64(0)
        L.check(L.lib().cmr_index_get_option(self._h, name.encode(), C.byref(v)))
        return v.value

    # -- lifetime
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            if not getattr(self, "_borrowed", False):
                L.lib().cmr_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        n = C.c_int64(0)
        L.check(L.lib().cmr_index_size(self._h, C.byref(n)))
        return n.value

    @property
    def device_bytes(self) -> int:
        b = C.c_int64(0)
        L.check(L.lib().cmr_index_info(self._h, None, None, None, C.byref(b)))
        return b.value

    # -- append
    def append(self, rows) -> None:
        rows = _f32c(rows)
        if rows.ndim == 1:
            rows = rows[None, :]
        if rows.shape[0] == 0:
            return
        if rows.ndim != 2 or rows.shape[1] != self.dim:
            raise ValueError(f"rows must be [n,{self.dim}], got {rows.shape}")
        L.check(L.lib().cmr_index_append(self._h, _ptr(rows), rows.shape[0]))

    def append_dev(self, rows_t, stream: Optional[int] = None) -> None:
        """rows_t: torch float32 CUDA tensor [n, dim], contiguous, on this index's device."""
        import torch
        assert rows_t.is_cuda and rows_t.dtype == torch.float32 and rows_t.is_contiguous()
        if rows_t.device.index != self.device:
            # the encoder may live on another GPU than the index (cfg.device vs the store's index device): a foreign
            # device pointer must not reach cmr_index_append_dev — go through the host
            self.append(rows_t.cpu().numpy())
            return
        if stream is None:
            stream = torch.cuda.current_stream(rows_t.device).cuda_stream
        L.check(L.lib().cmr_index_append_dev(self._h, C.c_void_p(rows_t.data_ptr()), rows_t.shape[0], C.c_void_p(stream)))

    # -- search
    def search(self, q, k: int, with_minmax: bool = True
               ) -> Tuple[np.ndarray, np.ndarray, Optional[np.ndarray], Optional[np.ndarray]]:
        """q [nq,dim] → (ids int64 [nq,k'], raw scores fp32 [nq,k'], min [nq], max [nq]) with
        k' = min(k, len(self)); order: score desc, row asc."""
        q = _f32c(q)
        if q.ndim == 1:
            q = q[None, :]
        if q.shape[1] != self.dim:
            raise ValueError(f"q must be [nq,{self.dim}], got {q.shape}")
        nq = q.shape[0]
        ids = np.empty((nq, k), dtype=np.int64)
        sc = np.empty((nq, k), dtype=np.float32)
        mn = np.empty(nq, dtype=np.float32) if with_minmax else None
        mx = np.empty(nq, dtype=np.float32) if with_minmax else None
        L.check(L.lib().cmr_index_search(self._h, _ptr(q), nq, k, _ptr(ids), _ptr(sc),
                                         _ptr(mn) if with_minmax else None, _ptr(mx) if with_minmax else None))
        kk = min(k, len(self))
        return ids[:, :kk], sc[:, :kk], mn, mx

    def search_min_score(self, q, k: int, min_score: float) -> Tuple[np.ndarray, np.ndarray]:
        """The k best rows among those with raw score >= min_score: (ids [nq,k], scores [nq,k]), -1 / -inf padded."""
        q = _f32c(q)
        if q.ndim == 1:
            q = q[None, :]
        nq = q.shape[0]
        ids = np.empty((nq, k), dtype=np.int64)
        sc = np.empty((nq, k), dtype=np.float32)
        L.check(L.lib().cmr_index_search_min_score(self._h, _ptr(q), nq, k, float(min_score), _ptr(ids), _ptr(sc)))
        return ids, sc

    def search_min_score_dev(self, q_t, k: int, min_score: float, out_ids=None, out_scores=None, stream: Optional[int] = None):
        """`search_min_score` on torch CUDA tensors, enqueued on torch's current stream without synchronising."""
        import torch
        assert q_t.is_cuda and q_t.dtype == torch.float32 and q_t.is_contiguous() and q_t.shape[1] == self.dim
        nq, dev = q_t.shape[0], q_t.device
        if out_ids is None:
            out_ids = torch.empty((nq, k), dtype=torch.int64, device=dev)
        if out_scores is None:
            out_scores = torch.empty((nq, k), dtype=torch.float32, device=dev)
        if stream is None:
            stream = torch.cuda.current_stream(dev).cuda_stream
        L.check(L.lib().cmr_index_search_min_score_dev(self._h, C.c_void_p(q_t.data_ptr()), nq, k, float(min_score),
                                                       C.c_void_p(out_ids.data_ptr()), C.c_void_p(out_scores.data_ptr()), C.c_void_p(stream)))
        return out_ids, out_scores

    def search_min_score_pipelined(self, q_t, k: int, min_score: float, out_ids, out_scores, wait_event=None):
        """`search_min_score` in throughput mode (the index's own streams; packing, scan and candidate merge of consecutive blocks
        overlap): returns the done-event handle of this block — `sync(handle)` of the LAST block covers every earlier one."""
        import torch
        assert q_t.is_cuda and q_t.dtype == torch.float32 and q_t.is_contiguous() and q_t.shape[1] == self.dim
        done = C.c_void_p()
        we = C.c_void_p(wait_event.cuda_event) if wait_event is not None else None
        L.check(L.lib().cmr_index_search_min_score_pipelined(self._h, C.c_void_p(q_t.data_ptr()), q_t.shape[0], k, float(min_score),
                                                             C.c_void_p(out_ids.data_ptr()), C.c_void_p(out_scores.data_ptr()), we, C.byref(done)))
        return done

    def search_dev(self, q_t, k: int, out_ids=None, out_scores=None, out_min=None, out_max=None,
                   stream: Optional[int] = None):
        """Asynchronous search on torch CUDA tensors, enqueued on torch's current stream (also when that is the default
        stream, handle 0): ordered after the kernels that produced `q_t`, before later readers of the outputs."""
        import torch
        assert q_t.is_cuda and q_t.dtype == torch.float32 and q_t.is_contiguous() and q_t.shape[1] == self.dim
        nq = q_t.shape[0]
        dev = q_t.device
        if out_ids is None:
            out_ids = torch.empty((nq, k), dtype=torch.int64, device=dev)
        if out_scores is None:
            out_scores = torch.empty((nq, k), dtype=torch.float32, device=dev)
        if stream is None:
            stream = torch.cuda.current_stream(dev).cuda_stream
        L.check(L.lib().cmr_index_search_dev(
            self._h, C.c_void_p(q_t.data_ptr()), nq, k, C.c_void_p(out_ids.data_ptr()), C.c_void_p(out_scores.data_ptr()),
            C.c_void_p(out_min.data_ptr()) if out_min is not None else None,
            C.c_void_p(out_max.data_ptr()) if out_max is not None else None, C.c_void_p(stream)))
        return out_ids, out_scores

    def search_pipelined(self, q_t, k: int, out_ids, out_scores, out_min=None, out_max=None, wait_event=None):
        """Throughput mode (cmr_index_search_pipelined): enqueue on the index's own two streams and
        return an opaque done-event handle.  `wait_event`: a torch.cuda.Event (inputs ready) or None.
        Use `wait(handle, stream)` / `sync(handle)` before reading the outputs."""
        import torch
        assert q_t.is_cuda and q_t.dtype == torch.float32 and q_t.is_contiguous() and q_t.shape[1] == self.dim
        done = C.c_void_p()
        we = C.c_void_p(wait_event.cuda_event) if wait_event is not None else None
        L.check(L.lib().cmr_index_search_pipelined(
            self._h, C.c_void_p(q_t.data_ptr()), q_t.shape[0], k, C.c_void_p(out_ids.data_ptr()), C.c_void_p(out_scores.data_ptr()),
            C.c_void_p(out_min.data_ptr()) if out_min is not None else None,
            C.c_void_p(out_max.data_ptr()) if out_max is not None else None, we, C.byref(done)))
        return done

    def query_status(self) -> bool:
        """True if a `search_dev` / `search_pipelined` call since the last check saw a NaN/Inf query (those entry
        points cannot raise without a sync; this one synchronises their streams)."""
        f = C.c_int32(0)
        L.check(L.lib().cmr_index_query_status(self._h, C.byref(f)))
        return bool(f.value)

    def set_id_base(self, base: int) -> None:
        """Offset added to every returned row id (global id of local row 0 of a row shard)."""
        L.check(L.lib().cmr_index_set_id_base(self._h, int(base)))

    def set_id_blocks(self, local_starts, global_starts) -> None:
        """Block table of a shard that took incremental appends: local rows [local_starts[b], local_starts[b+1]) carry the
        global ids global_starts[b] + 0, 1, ... (cmr_index_set_id_blocks)."""
        ls = np.ascontiguousarray(local_starts, dtype=np.int64)
        gs = np.ascontiguousarray(global_starts, dtype=np.int64)
        if ls.shape != gs.shape or ls.ndim != 1 or len(ls) == 0:
            raise ValueError("local_starts / global_starts must be equally long, non-empty 1-d arrays")
        L.check(L.lib().cmr_index_set_id_blocks(self._h, len(ls), _ptr(ls), _ptr(gs)))

    def pipeline_stream(self, which: int = 2):
        """The pipeline's pre / scan / post stream as a torch.cuda.ExternalStream."""
        import torch
        s = C.c_void_p()
        L.check(L.lib().cmr_index_pipeline_stream(self._h, which, C.byref(s)))
        return torch.cuda.ExternalStream(s.value, device=torch.device("cuda", self.device))

    @staticmethod
    def wait(done_handle, torch_stream) -> None:
        """Make a torch stream wait for a pipelined search's outputs."""
        L.check(L.lib().cmr_stream_wait_event(C.c_void_p(torch_stream.cuda_stream), done_handle))

    @staticmethod
    def sync(done_handle) -> None:
        L.check(L.lib().cmr_event_synchronize(done_handle))

    def scores(self, q) -> np.ndarray:
        """All raw scores [nq, N] (fp32)."""
        q = _f32c(q)
        if q.ndim == 1:
            q = q[None, :]
        n = len(self)
        out = np.empty((q.shape[0], n), dtype=np.float32)
        if n:
            L.check(L.lib().cmr_index_scores(self._h, _ptr(q), q.shape[0], _ptr(out), n))
        return out

    def scores_dev(self, q_t, out=None, stream: Optional[int] = None):
        import torch
        assert q_t.is_cuda and q_t.dtype == torch.float32 and q_t.is_contiguous()
        n = len(self)
        if out is None:
            out = torch.empty((q_t.shape[0], n), dtype=torch.float32, device=q_t.device)
        if stream is None:
            stream = torch.cuda.current_stream(q_t.device).cuda_stream
        L.check(L.lib().cmr_index_scores_dev(self._h, C.c_void_p(q_t.data_ptr()), q_t.shape[0],
                                             C.c_void_p(out.data_ptr()), out.stride(0), C.c_void_p(stream)))
        return out

    def sorted_scores(self, q):
        """Complete ranking: (ids int64 [nq,N], raw scores fp32 [nq,N] descending, min [nq], max [nq])."""
        q = _f32c(q)
        if q.ndim == 1:
            q = q[None, :]
        n, nq = len(self), q.shape[0]
        ids = np.empty((nq, n), dtype=np.int64)
        sc = np.empty((nq, n), dtype=np.float32)
        mn = np.empty(nq, dtype=np.float32)
        mx = np.empty(nq, dtype=np.float32)
        if n:
            L.check(L.lib().cmr_index_sorted_scores(self._h, _ptr(q), nq, _ptr(ids), _ptr(sc), _ptr(mn), _ptr(mx)))
        return ids, sc, mn, mx

    def rescore(self, q, cand, k: int) -> Tuple[np.ndarray, np.ndarray]:
        q = _f32c(q)
        if q.ndim == 1:
            q = q[None, :]
        cand = np.ascontiguousarray(cand, dtype=np.int64)
        if cand.ndim == 1:
            cand = cand[None, :]
        nq, nc = cand.shape
        k = min(k, nc)
        ids = np.empty((nq, k), dtype=np.int64)
        sc = np.empty((nq, k), dtype=np.float32)
        L.check(L.lib().cmr_index_rescore(self._h, _ptr(q), nq, _ptr(cand), nc, k, _ptr(ids), _ptr(sc)))
        return ids, sc

    def get_rows(self, ids) -> np.ndarray:
        ids = np.ascontiguousarray(ids, dtype=np.int64).ravel()
        out = np.empty((len(ids), self.dim), dtype=np.float32)
        if len(ids):
            L.check(L.lib().cmr_index_get_rows(self._h, _ptr(ids), len(ids), _ptr(out)))
        return out

    # -- measurement
    def profile(self, on) -> None:
        """HIP-event timing of the main scans: True / 1 = every scan, N > 1 = every N-th scan, False / 0 = off."""
        L.check(L.lib().cmr_profile_enable(self._h, int(on)))

    def profile_collect(self) -> dict:
        n, ms, b = C.c_int64(0), C.c_double(0), C.c_double(0)
        L.check(L.lib().cmr_profile_collect(self._h, C.byref(n), C.byref(ms), C.byref(b)))
        return {"launches": n.value, "total_ms": ms.value, "bytes_per_launch": b.value}


def merge_topk(ids: np.ndarray, scores: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Host-side final merge of per-shard candidates [S,nq,k] → [nq,k] (same tie rule)."""
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    S, nq, k = ids.shape
    oi = np.empty((nq, k), dtype=np.int64)
    os_ = np.empty((nq, k), dtype=np.float32)
    L.check(L.lib().cmr_merge_topk(_ptr(ids), _ptr(scores), S, nq, k, _ptr(oi), _ptr(os_)))
    return oi, os_
