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
                 group=None, base: int = 0, capacity_hint: int = 0, index=None):
        self.dim, self.dtype, self.device = dim, dtype, device
        self.rank, self.world, self.group = rank, world, group
        self.base = int(base)          # global id of local row 0
        if index is None:
            from .index import DenseIndex
            index = DenseIndex(dim, dtype, device=device, capacity_hint=capacity_hint)
        self.local = index
        self._side = None
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
        fn = local_search or (lambda qq, kk: self.local.search(qq, kk, with_minmax=False)[:2])
        ids, sc = fn(q, k)
        nq = q.shape[0]
        pid = np.full((nq, k), -1, dtype=np.int64)
        psc = np.full((nq, k), -np.inf, dtype=np.float32)
        kk = ids.shape[1]
        pid[:, :kk] = np.where(ids >= 0, ids + self.base, -1)
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
                o_ids=torch.empty((nq, k), dtype=torch.int64, device=dev),
                o_sc=torch.empty((nq, k), dtype=torch.float32, device=dev),
                done=torch.cuda.Event(), scanned=torch.cuda.Event(), used=False)
        return self._bufs[key]

    def search_pipelined(self, q_t, k: int, slot: int):
        """Enqueue one batch without blocking the host.  Buffers are double buffered (`slot` 0/1,
        alternate between consecutive batches):
          scan stream : query packing, sampling passes, the corpus scan, per-shard candidate merge
          side stream : RCCL all-gather of the [B,k] candidates + final shard merge (per slot)
        so the latency-bound collective + shard merge of batch i overlap the HBM-bound scan of
        batch i+1.  Returns the slot's buffer dict; `o_ids`/`o_sc` (global ids / raw scores) are
        valid after `bufs['done']` (a torch.cuda.Event)."""
        import ctypes as C
        import torch
        import torch.distributed as dist
        dev = q_t.device
        nq = q_t.shape[0]
        b = self._buffers(slot, nq, k, dev)
        caller = torch.cuda.current_stream(dev)
        if self._side is None:
            # ONE scan stream: two HBM-bound scans in flight at once only slow each other down
            # (measured: 2-stream scans 0.68 ms/step vs 0.43 ms/step serial at 1 M rows).
            sc = torch.cuda.Stream(device=dev)
            self._side = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
            self._scan = [sc, sc]
        scan, side = self._scan[slot], self._side[slot]
        ready = torch.cuda.Event()
        ready.record(caller)                      # q_t produced on the caller's stream
        scan.wait_event(ready)
        if b["used"]:
            scan.wait_event(b["done"])            # slot reuse: its previous merge has consumed ids/sc
        with torch.cuda.stream(scan):
            self.local.search_dev(q_t, k, out_ids=b["ids"], out_scores=b["sc"], stream=scan.cuda_stream)
            if self.base:
                b["ids"].add_(self.base * (b["ids"] >= 0))
            b["scanned"].record(scan)
        with torch.cuda.stream(side):
            side.wait_event(b["scanned"])
            if self.world > 1:
                dist.all_gather_into_tensor(b["g_ids"], b["ids"], group=self.group)
                dist.all_gather_into_tensor(b["g_sc"], b["sc"], group=self.group)
                L.check(L.lib().cmr_merge_topk_dev(
                    self.device, C.c_void_p(b["g_ids"].data_ptr()), C.c_void_p(b["g_sc"].data_ptr()), self.world, nq, k,
                    C.c_void_p(b["o_ids"].data_ptr()), C.c_void_p(b["o_sc"].data_ptr()), C.c_void_p(side.cuda_stream)))
            else:
                b["o_ids"].copy_(b["ids"], non_blocking=True)
                b["o_sc"].copy_(b["sc"], non_blocking=True)
            b["done"].record(side)
        b["used"] = True
        return b
