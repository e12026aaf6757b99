"""Row-sharded index across the GPUs of one node: one process per GPU (torch.distributed,
backend "nccl" = RCCL over xGMI), contiguous row blocks per rank, one all-gather of the per-shard
[B,k] (score, global id) candidates per query batch, then the final merge with the exported tie
rule — so the result equals a single-shard search by construction.  The reference
has no distributed path; nothing here mirrors reference code.

The candidate exchange is B*k*12 bytes per rank (15 KiB at B=64,k=20): latency-bound, so the
all-gather + merge of batch i run on a side stream while the scan of batch i+1 runs on the main
stream (`search_pipelined`).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np

from . import _lib as L


def shard_bounds(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row block of `rank`: [lo, hi).  ceil-sized blocks, last ones may be short/empty."""
    per = (n_total + world - 1) // world
    lo = min(rank * per, n_total)
    return lo, min(lo + per, n_total)


class ShardedIndex:
    def __init__(self, dim: int, dtype: str = "bf16", device: int = 0, rank: int = 0, world: int = 1,
                 group=None, base: int = 0, capacity_hint: int = 0, index=None, force_exchange: bool = False):
        self.dim, self.dtype, self.device = dim, dtype, device
        self.rank, self.world, self.group = rank, world, group
        self.base = int(base)          # global id of local row 0
        self.exchange = world > 1 or force_exchange   # force_exchange: run the RCCL path on a 1-rank group (tests)
        if index is None:
            from .index import DenseIndex
            index = DenseIndex(dim, dtype, device=device, capacity_hint=capacity_hint)
        self.local = index
        if hasattr(index, "set_id_base"):
            index.set_id_base(self.base)        # the library returns global ids from here on
        self._post = None
        self._bufs = {}

    def __len__(self):
        return len(self.local)

    # ---------------------------------------------------------------- host (numpy) path
    def search(self, q: np.ndarray, k: int, local_search: Optional[Callable] = None):
        """Synchronous search with host buffers; final merge on the host (cmr_merge_topk).
        `local_search(q,k)->(ids,scores)` may replace the GPU scan in CPU-only tests."""
        import torch
        import torch.distributed as dist
        from .index import merge_topk
        if local_search is not None:            # CPU-only tests: shard-local ids from the stand-in
            ids, sc = local_search(q, k)
            ids = np.where(ids >= 0, ids + self.base, -1)
        else:                                   # HIP scan: ids are already global (cmr_index_set_id_base)
            ids, sc = self.local.search(q, k, with_minmax=False)[:2]
        nq = q.shape[0]
        pid = np.full((nq, k), -1, dtype=np.int64)
        psc = np.full((nq, k), -np.inf, dtype=np.float32)
        kk = ids.shape[1]
        pid[:, :kk] = ids
        psc[:, :kk] = sc
        if self.world == 1:
            return merge_topk(pid[None], psc[None])
        dev = torch.device("cuda", self.device) if dist.get_backend(self.group) == "nccl" else torch.device("cpu")
        t_id = torch.from_numpy(pid).to(dev)
        t_sc = torch.from_numpy(psc).to(dev)
        # concatenated layout [world*nq, k] (accepted by both gloo and RCCL), viewed as [world, nq, k]
        g_id = torch.empty((self.world * nq, k), dtype=torch.int64, device=dev)
        g_sc = torch.empty((self.world * nq, k), dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(g_id, t_id, group=self.group)
        dist.all_gather_into_tensor(g_sc, t_sc, group=self.group)
        return merge_topk(g_id.cpu().numpy().reshape(self.world, nq, k), g_sc.cpu().numpy().reshape(self.world, nq, k))

    # ---------------------------------------------------------------- device path
    def _buffers(self, slot: int, nq: int, k: int, dev):
        import torch
        key = (slot, nq, k)
        if key not in self._bufs:
            self._bufs[key] = dict(
                ids=torch.empty((nq, k), dtype=torch.int64, device=dev),
                sc=torch.empty((nq, k), dtype=torch.float32, device=dev),
                g_ids=torch.empty((self.world * nq, k), dtype=torch.int64, device=dev),    # == [world, nq, k]
                g_sc=torch.empty((self.world * nq, k), dtype=torch.float32, device=dev),
                o_ids=None, o_sc=None, done=None)
            b = self._bufs[key]
            if self.exchange:
                b["o_ids"] = torch.empty((nq, k), dtype=torch.int64, device=dev)
                b["o_sc"] = torch.empty((nq, k), dtype=torch.float32, device=dev)
            else:
                b["o_ids"], b["o_sc"] = b["ids"], b["sc"]
        return self._bufs[key]

    def search_pipelined(self, q_t, k: int, slot: int, q_ready=None):
        """Enqueue one batch without blocking the host; alternate `slot` (0/1) between consecutive
        batches.  The index's own pipeline (cmr_index_search_pipelined) overlaps query packing +
        sampling passes of batch i+1 with the HBM-bound main scan of batch i (main scans serialised);
        row ids come out global (`cmr_index_set_id_base`).  For world > 1 the RCCL all-gather of the
        [B,k] candidates and the shard merge are enqueued on the pipeline's post stream, i.e. ordered
        after this batch's outputs and before the slot's buffers are rewritten — no extra streams or
        events (HIP multiplexes streams onto a few in-order hardware queues; every additional stream
        with wait packets risks blocking the scan queue: measured 429 vs 323 us/step).
        Returns the slot's buffers: `o_ids` / `o_sc` are valid after `done.synchronize()`.
        `q_t` must already be materialised on the device, or pass `q_ready` (a torch.cuda.Event
        recorded on a non-default stream; a null-stream event would serialise the batches)."""
        import ctypes as C
        import torch
        import torch.distributed as dist
        dev = q_t.device
        nq = q_t.shape[0]
        b = self._buffers(slot, nq, k, dev)
        handle = self.local.search_pipelined(q_t, k, b["ids"], b["sc"], wait_event=q_ready)
        if self.exchange:
            if self._post is None:
                self._post = self.local.pipeline_stream(2)
            with torch.cuda.stream(self._post):
                dist.all_gather_into_tensor(b["g_ids"], b["ids"], group=self.group)
                dist.all_gather_into_tensor(b["g_sc"], b["sc"], group=self.group)
                L.check(L.lib().cmr_merge_topk_dev(
                    self.device, C.c_void_p(b["g_ids"].data_ptr()), C.c_void_p(b["g_sc"].data_ptr()), self.world, nq, k,
                    C.c_void_p(b["o_ids"].data_ptr()), C.c_void_p(b["o_sc"].data_ptr()), C.c_void_p(self._post.cuda_stream)))
                ev = torch.cuda.Event()
                ev.record(self._post)
            b["done"] = ev
        else:
            b["done"] = _Done(self.local, handle)
        return b


class _Done:
    """Completion handle of a pipelined batch on a single shard (wraps the index's hipEvent)."""

    def __init__(self, index, handle):
        self._index, self._h = index, handle

    def synchronize(self):
        self._index.sync(self._h)

    def wait(self, stream=None):
        import torch
        self._index.wait(self._h, stream or torch.cuda.current_stream())
