"""MultiDeviceIndex — ONE process, the node's GPUs: row shards behind `DenseIndex`'s call surface (`cmr_mindex_t`).

ComoRAG answers its questions from one process whose 16-thread pool (src/comorag/ComoRAG.py:432-453) calls
`tri_retrieve` (:456-554) whenever an LLM reply comes back; an index that is sharded over the node's GPUs has to be
usable from exactly that loop — any thread, any time, host arrays in and out — which a one-process-per-GPU layout
(`comorag_amd/sharded.py`, where every call is a collective) is not.  This class is that index: `append / search / scores /
sorted_scores / rescore / get_rows / len` behave like one `DenseIndex` holding the same rows (global row ids dense in
append order, exported tie rule, results bit-identical — tests/test_multi_device_gpu.py holds S in {1, 2, 4, 8} logical
shards on one device against a single index).  `hooks.install`, `install_memory_pool` and `EmbeddingStore.device_index`
build one when `global_config.num_shards > 1` or `global_config.devices` is set (`make_index`).

Devices: `devices=[0, 1, ...]` names the GPU of every shard; a device may be named several times (logical shards).
`num_shards=S` alone spreads S shards over the visible devices round robin.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib as L
from .index import DenseIndex, _f32c, _ptr


MULTI_ONLY_OPTIONS = ("append_block_rows", "parallel_min_shards", "force_peer_staging")


def resolve_devices(num_shards: Optional[int] = None, devices: Optional[Sequence[int]] = None, device: int = 0) -> List[int]:
    """The device of every shard.  devices given: as they are; num_shards, when also given, must equal len(devices) or be a
    multiple of it (the list is cycled: logical shards) — anything else would silently drop or unbalance GPUs and raises.
    Only num_shards: shards go round the visible devices starting at `device`."""
    if devices is not None and len(devices) > 0:
        devs = [int(d) for d in devices]
        if num_shards and int(num_shards) != len(devs):
            if int(num_shards) % len(devs) != 0:
                raise ValueError(f"num_shards = {num_shards} is not a multiple of len(devices) = {len(devs)}: shards would drop or unbalance devices")
            devs = [devs[i % len(devs)] for i in range(int(num_shards))]
        return devs
    s = int(num_shards or 1)
    n = max(1, L.device_count())
    return [(int(device) + i) % n for i in range(s)]


class MultiDeviceIndex:
    def __init__(self, dim: int, dtype: str = "bf16", devices: Optional[Sequence[int]] = None, num_shards: Optional[int] = None,
                 device: int = 0, capacity_hint: int = 0, keep_f32: bool = False, options: Optional[dict] = None):
        self._h = C.c_void_p()
        self.dim, self.dtype, self.keep_f32 = int(dim), dtype, bool(keep_f32)
        self.devices = resolve_devices(num_shards, devices, device)
        self.n_shards = len(self.devices)
        self.device = self.devices[0]
        arr = (C.c_int32 * self.n_shards)(*self.devices)
        L.check(L.lib().cmr_mindex_create(self.n_shards, arr, self.dim, L.DTYPES[dtype], int(capacity_hint),
                                          L.CMR_FLAG_KEEP_F32 if keep_f32 else 0, C.byref(self._h)))
        self._tickets = {}
        for name, value in (options or {}).items():
            self.set_option(name, value)

    # -- lifetime / facts
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            L.lib().cmr_mindex_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        n = C.c_int64(0)
        L.check(L.lib().cmr_mindex_size(self._h, C.byref(n)))
        return n.value

    def set_option(self, name: str, value: int) -> None:
        """"append_block_rows" / "parallel_min_shards" / "force_peer_staging", or any route selector of `DenseIndex.set_option` (goes to
        every shard, with no search in flight)."""
        L.check(L.lib().cmr_mindex_set_option(self._h, name.encode(), int(value)))

    def shard_rows(self) -> List[int]:
        rows = (C.c_int64 * self.n_shards)()
        L.check(L.lib().cmr_mindex_info(self._h, None, None, rows, None))
        return list(rows)

    @property
    def device_bytes(self) -> int:
        b = C.c_int64(0)
        L.check(L.lib().cmr_mindex_info(self._h, None, None, None, C.byref(b)))
        return b.value

    def shard(self, s: int) -> DenseIndex:
        """Borrowed `DenseIndex` view of shard s (profiling, reading options).  Never append to it; while other threads search, set
        route selectors through `MultiDeviceIndex.set_option` (exclusive over the whole handle), not on this view."""
        h = C.c_void_p()
        L.check(L.lib().cmr_mindex_shard(self._h, int(s), C.byref(h)))
        return DenseIndex._borrow(h, self.dim, self.dtype, self.devices[s], owner=self)

    # -- append
    def append(self, rows) -> None:
        rows = _f32c(rows)
        if rows.ndim == 1:
            rows = rows[None, :]
        if rows.shape[0] == 0:
            return
        if rows.ndim != 2 or rows.shape[1] != self.dim:
            raise ValueError(f"rows must be [n,{self.dim}], got {rows.shape}")
        L.check(L.lib().cmr_mindex_append(self._h, _ptr(rows), rows.shape[0]))

    def append_dev(self, rows_t, stream: Optional[int] = None) -> None:
        """torch float32 CUDA tensor [n, dim] on ANY device (an encoder's output): chunks whose shard lives on that device
        are appended in place, the others are copied device to device first (cmr_mindex_append_dev)."""
        import torch
        if not (rows_t.is_cuda and rows_t.dtype == torch.float32 and rows_t.is_contiguous()):
            self.append(rows_t.detach().float().cpu().numpy())
            return
        if rows_t.shape[0] == 0:
            return
        if rows_t.ndim != 2 or rows_t.shape[1] != self.dim:
            raise ValueError(f"rows must be [n,{self.dim}], got {tuple(rows_t.shape)}")
        if stream is None:
            stream = torch.cuda.current_stream(rows_t.device).cuda_stream
        L.check(L.lib().cmr_mindex_append_dev(self._h, C.c_void_p(rows_t.data_ptr()), rows_t.shape[0], rows_t.device.index, C.c_void_p(stream)))

    # -- search (host arrays in / out, as DenseIndex)
    def search(self, q, k: int, with_minmax: bool = True):
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
        L.check(L.lib().cmr_mindex_search(self._h, _ptr(q), nq, k, _ptr(ids), _ptr(sc),
                                          _ptr(mn) if with_minmax else None, _ptr(mx) if with_minmax else None))
        kk = min(k, len(self))
        return ids[:, :kk], sc[:, :kk], mn, mx

    def search_min_score(self, q, k: int, min_score: float) -> Tuple[np.ndarray, np.ndarray]:
        q = _f32c(q)
        if q.ndim == 1:
            q = q[None, :]
        nq = q.shape[0]
        ids = np.empty((nq, k), dtype=np.int64)
        sc = np.empty((nq, k), dtype=np.float32)
        L.check(L.lib().cmr_mindex_search_min_score(self._h, _ptr(q), nq, k, float(min_score), _ptr(ids), _ptr(sc)))
        return ids, sc

    def scores(self, q) -> np.ndarray:
        q = _f32c(q)
        if q.ndim == 1:
            q = q[None, :]
        n = len(self)
        out = np.empty((q.shape[0], n), dtype=np.float32)
        if n:
            L.check(L.lib().cmr_mindex_scores(self._h, _ptr(q), q.shape[0], _ptr(out), n))
        return out

    def sorted_scores(self, q):
        q = _f32c(q)
        if q.ndim == 1:
            q = q[None, :]
        n, nq = len(self), q.shape[0]
        ids = np.empty((nq, n), dtype=np.int64)
        sc = np.empty((nq, n), dtype=np.float32)
        mn = np.empty(nq, dtype=np.float32)
        mx = np.empty(nq, dtype=np.float32)
        if n:
            L.check(L.lib().cmr_mindex_sorted_scores(self._h, _ptr(q), nq, _ptr(ids), _ptr(sc), _ptr(mn), _ptr(mx)))
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
        L.check(L.lib().cmr_mindex_rescore(self._h, _ptr(q), nq, _ptr(cand), nc, k, _ptr(ids), _ptr(sc)))
        return ids, sc

    def get_rows(self, ids) -> np.ndarray:
        ids = np.ascontiguousarray(ids, dtype=np.int64).ravel()
        out = np.empty((len(ids), self.dim), dtype=np.float32)
        if len(ids):
            L.check(L.lib().cmr_mindex_get_rows(self._h, _ptr(ids), len(ids), _ptr(out)))
        return out

    # -- throughput mode
    def place_queries(self, q) -> list:
        """One float32 CUDA copy of the batch per DEVICE that holds a shard: [tensor of shard 0's device, ...] per shard
        (shards of one device share their tensor).  Inputs of `search_pipelined`."""
        import torch
        q = torch.as_tensor(np.ascontiguousarray(q, dtype=np.float32)) if not hasattr(q, "is_cuda") else q
        per_dev = {}
        for d in self.devices:
            if d not in per_dev:
                per_dev[d] = q.to(torch.device("cuda", d), dtype=torch.float32).contiguous()
        for d, t in per_dev.items():
            torch.cuda.synchronize(t.device)
        return [per_dev[d] for d in self.devices]

    def search_pipelined(self, q_per_shard: Sequence, k: int):
        """Enqueue one batch on every shard (`q_per_shard[s]`: the batch on shard s's device, see `place_queries`) and
        return a ticket at once; `collect(ticket)` waits for the shards and merges on the host.  At most four tickets may
        be uncollected."""
        if len(q_per_shard) != self.n_shards:
            raise ValueError(f"need one query tensor per shard ({self.n_shards})")
        q0 = q_per_shard[0]
        nq = q0.shape[0]
        for s, t in enumerate(q_per_shard):
            assert t.is_cuda and t.dtype.is_floating_point and t.element_size() == 4 and t.is_contiguous() and tuple(t.shape) == (nq, self.dim)
            assert t.device.index == self.devices[s], f"shard {s} lives on cuda:{self.devices[s]}, its queries on {t.device}"
        arr = (C.c_void_p * self.n_shards)(*[t.data_ptr() for t in q_per_shard])
        ticket = C.c_void_p()
        L.check(L.lib().cmr_mindex_search_pipelined(self._h, arr, nq, int(k), C.byref(ticket)))
        self._tickets[ticket.value] = (nq, int(k), list(q_per_shard))      # the tensors stay alive until collected
        return ticket

    def collect(self, ticket, with_minmax: bool = False):
        nq, k, _ = self._tickets.pop(ticket.value)
        ids = np.empty((nq, k), dtype=np.int64)
        sc = np.empty((nq, k), dtype=np.float32)
        mn = np.empty(nq, dtype=np.float32) if with_minmax else None
        mx = np.empty(nq, dtype=np.float32) if with_minmax else None
        L.check(L.lib().cmr_mindex_collect(self._h, ticket, _ptr(ids), _ptr(sc), _ptr(mn) if with_minmax else None,
                                           _ptr(mx) if with_minmax else None))
        return (ids, sc, mn, mx) if with_minmax else (ids, sc)


def _host_profile(self, reset: bool = True) -> dict:
    """Host-side cost of the throughput mode since the last reset (cmr_mindex_profile)."""
    n, e, w, mg = C.c_int64(0), C.c_double(0), C.c_double(0), C.c_double(0)
    L.check(L.lib().cmr_mindex_profile(self._h, 1 if reset else 0, C.byref(n), C.byref(e), C.byref(w), C.byref(mg)))
    return {"batches": n.value, "enqueue_us_per_shard": e.value, "event_wait_us_per_batch": w.value, "host_merge_us_per_batch": mg.value}


MultiDeviceIndex.host_profile = _host_profile


DEFAULT_APPEND_BLOCK_ROWS = 65536       # = cmr_mindex::block_rows (csrc/multi.hip), the library's "append_block_rows" default


def plan_append(shard_rows: Sequence[int], m: int, block_rows: int = DEFAULT_APPEND_BLOCK_ROWS, cur: int = -1, room: int = 0):
    """The library's append routing on its own (cmr_mindex_plan_append; pure host arithmetic, runs without a GPU):
    -> ([(shard, rows), ...], new_cur, new_room).  `block_rows` defaults to the library's own "append_block_rows" default, so a
    model of the routing built with this function agrees with a real `MultiDeviceIndex` that was left at its defaults."""
    s = len(shard_rows)
    sizes = (C.c_int64 * s)(*[int(x) for x in shard_rows])
    cap = max(8, 2 * s + int(m) // max(1, int(block_rows)) + 4)
    out_s = (C.c_int32 * cap)()
    out_n = (C.c_int64 * cap)()
    n, ncur, nroom = C.c_int32(0), C.c_int32(0), C.c_int64(0)
    L.check(L.lib().cmr_mindex_plan_append(sizes, s, int(cur), int(room), int(m), int(block_rows), cap, out_s, out_n, C.byref(n),
                                           C.byref(ncur), C.byref(nroom)))
    return [(out_s[i], out_n[i]) for i in range(n.value)], ncur.value, nroom.value


def make_index(dim: int, dtype: str = "f32", device: int = 0, capacity_hint: int = 0, keep_f32: bool = False,
               options: Optional[dict] = None, num_shards: Optional[int] = None, devices: Optional[Sequence[int]] = None):
    """`DenseIndex` on one device, or `MultiDeviceIndex` when more than one shard is asked for (global_config.num_shards /
    global_config.devices)."""
    devs = resolve_devices(num_shards, devices, device) if (devices or (num_shards or 1) > 1) else [int(device)]
    if len(devs) <= 1:
        one = {k: v for k, v in (options or {}).items() if k not in MULTI_ONLY_OPTIONS}
        return DenseIndex(dim, dtype, device=devs[0], capacity_hint=capacity_hint, keep_f32=keep_f32, options=one)
    return MultiDeviceIndex(dim, dtype, devices=devs, capacity_hint=capacity_hint, keep_f32=keep_f32, options=options)


def shard_config(cfg) -> dict:
    """num_shards / devices / index_options of a configuration object (any object; reference defaults = one device)."""
    return {"num_shards": getattr(cfg, "num_shards", None) if cfg is not None else None,
            "devices": getattr(cfg, "devices", None) if cfg is not None else None,
            "options": getattr(cfg, "index_options", None) if cfg is not None else None}
