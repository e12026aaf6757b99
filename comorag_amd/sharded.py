"""Row-sharded index across the GPUs of one node: one process per GPU (torch.distributed,
backend "nccl" = RCCL over xGMI), contiguous row blocks per rank, ONE all-gather of the per-shard
[B,k] candidates per query batch — each candidate packed into a single u64 (order-preserving score
code | ~global row: unsigned order is the exported order) — then the final merge, so the result
equals a single-shard search by construction.  The reference has no distributed
path; nothing here mirrors reference code.

The candidate exchange is B*k*8 bytes per rank (10 KiB at B=64,k=20): latency-bound, so the
all-gather + merge of batch i run on the index pipeline's post stream while the scan of batch i+1
runs on its scan stream (`search_pipelined`).  Two bindings of the same exchange:
  exchange="torch"  torch.distributed.all_gather_into_tensor between cmr_pack_candidates_dev and cmr_merge_keys_dev
  exchange="cabi"   cmr_comm_allgather_merge: RCCL called by the library itself (no torch in the data path)
On a `gloo` group (CPU tests; two ranks sharing one GPU in the 1-GPU rehearsal of the N > 1 flow) the same packed keys
are staged through the host around the collective.

Incremental appends (`append`; BASELINE config 4 at N > 1): global ids stay dense in append order — what
EmbeddingStore._upsert (embedding_store.py:122-128) and MemoryPool.add_node (utils/memory_utils.py:294-300) rely on —
and appended rows go to the shards in blocks (a block of 8192 rows opens on the currently shortest shard and takes the appends
until it is full), so a shard holds several runs of consecutive global ids; the host-side table of those runs (`blocks`) is
mirrored into the library with cmr_index_set_id_blocks.
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
                 group=None, base: int = 0, capacity_hint: int = 0, index=None, force_exchange: bool = False,
                 exchange: str = "torch", timing: bool = False):
        self.dim, self.dtype, self.device = dim, dtype, device
        self.rank, self.world, self.group = rank, world, group
        self.base = int(base)          # global id of local row 0
        self.exchange = world > 1 or force_exchange   # force_exchange: run the RCCL path on a 1-rank group (tests)
        if exchange not in ("torch", "cabi"):
            raise ValueError("exchange must be 'torch' or 'cabi'")
        self.exchange_kind = exchange
        self.timing = timing           # record exchange / merge times of every pipelined batch (bench.py)
        self.times = []                # (event before exchange, after all-gather, after merge) per batch
        self._comm = None
        if index is None:
            from .index import DenseIndex
            index = DenseIndex(dim, dtype, device=device, capacity_hint=capacity_hint)
        self.local = index
        if hasattr(index, "set_id_base"):
            index.set_id_base(self.base)        # the library returns global ids from here on
        self._post = None
        self._bufs = {}
        # layout (host-side base table): this shard's runs of consecutive global ids [(local_start, global_start)], and —
        # once `sync_layout` ran — every shard's row count and the global row count, identical on every rank
        self.blocks = [(0, self.base)]
        self.sizes = None
        self.total = None
        self._cur = (None, 0)          # (shard taking appended rows, rows in its open block)
        self._owns_local = True

    def __len__(self):
        return len(self.local)

    # ---------------------------------------------------------------- layout / incremental append
    def sync_layout(self):
        """Collective, once after the bulk build: every rank learns every shard's (base, rows).  The bulk layout must be the
        contiguous one of `shard_bounds` (rank r starts where rank r-1 ends)."""
        mine = (self.base, len(self.local))
        if self.world > 1:
            import torch.distributed as dist
            box = [None] * self.world
            dist.all_gather_object(box, mine, group=self.group)
        else:
            box = [mine]
        at = box[0][0]
        for r, (b, n) in enumerate(box):
            if b != at:
                raise ValueError(f"shard {r} starts at global id {b}, expected {at}: bulk shards must be contiguous row blocks")
            at += n
        self.sizes = [n for _, n in box]
        self.total = at
        return self

    def _route(self, m: int, block_rows: int, commit: bool = False):
        """[(shard, n rows)] for m appended rows.  Rows keep going to the shard that took the previous ones until its current
        block holds `block_rows` rows, then the block after it opens on the shortest shard at that moment (lowest rank on
        ties): "round-robin blocks" of SURVEY §8e — shard sizes stay within one block of each other, and a shard's table of
        id runs grows by one entry per block_rows appended rows, not per append (a memory pool appends 25 rows per cycle).
        Deterministic from the replicated table, so every rank computes the same routing."""
        sizes = list(self.sizes)
        cur, fill = self._cur
        out = []
        while m > 0:
            if cur is None or fill >= block_rows:
                cur, fill = min(range(self.world), key=lambda r: (sizes[r], r)), 0
            n = min(m, block_rows - fill)
            out.append((cur, n))
            sizes[cur] += n
            fill += n
            m -= n
        if commit:
            self._cur = (cur, fill)
        return out

    def append(self, rows, block_rows: int = 8192) -> "np.ndarray":
        """Collective append of `rows` [m, dim] (numpy; every rank passes the same rows, only the owner of a chunk uploads
        it).  The rows get the global ids total .. total + m - 1 in order, exactly as one index would number them; returns
        those ids.  Blocks of `block_rows` rows go round robin to the shortest shard (`_route`)."""
        if not self._owns_local:
            # a view shares the shard and its block list but keeps its own copy of the running totals: appending through it would
            # leave the owner (and every other view) with a stale layout, and the ranks would route the next append differently
            raise RuntimeError("ShardedIndex.view() handles are read-only: append through the handle that owns the shard")
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        if rows.ndim == 1:
            rows = rows[None, :]
        if self.sizes is None:
            self.sync_layout()
        m = rows.shape[0]
        first = self.total
        at = 0
        changed = False
        for shard, n in self._route(m, block_rows, commit=True):
            if shard == self.rank:
                local_at = len(self.local)
                self.local.append(rows[at:at + n])
                l0, g0 = self.blocks[-1]
                if g0 + (local_at - l0) != self.total:      # not a continuation of this shard's last run: a new block
                    if local_at == l0:                      # (the last run is still empty: it simply starts elsewhere)
                        self.blocks[-1] = (local_at, self.total)
                    else:
                        self.blocks.append((local_at, self.total))
                    changed = True
            self.sizes[shard] += n
            self.total += n
            at += n
        if changed:
            self._push_blocks()
        return np.arange(first, first + m, dtype=np.int64)

    def _push_blocks(self):
        self.base = self.blocks[0][1]
        if hasattr(self.local, "set_id_blocks"):
            self.local.set_id_blocks([b[0] for b in self.blocks], [b[1] for b in self.blocks])

    def _to_global(self, ids: np.ndarray) -> np.ndarray:
        """Shard-local rows -> global ids through the block table (numpy twin of the library's remap; used where a stand-in
        replaces the HIP scan)."""
        ids = np.asarray(ids, np.int64)
        ls = np.array([b[0] for b in self.blocks], np.int64)
        gs = np.array([b[1] for b in self.blocks], np.int64)
        b = np.clip(np.searchsorted(ls, ids, side="right") - 1, 0, len(ls) - 1)
        return np.where(ids >= 0, ids - ls[b] + gs[b], -1)

    # ---------------------------------------------------------------- host (numpy) path
    def search(self, q: np.ndarray, k: int, local_search: Optional[Callable] = None):
        """Synchronous search with host buffers; final merge on the host (cmr_merge_topk).
        `local_search(q,k)->(ids,scores)` may replace the GPU scan in CPU-only tests."""
        import torch
        import torch.distributed as dist
        from .index import merge_topk
        if local_search is not None:            # CPU-only tests: shard-local ids from the stand-in
            ids, sc = local_search(q, k)
            ids = self._to_global(ids)
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
        t_key = torch.from_numpy(pack_candidates(pid, psc).view(np.int64)).to(dev)
        # concatenated layout [world*nq, k] (accepted by both gloo and RCCL), viewed as [world, nq, k]
        g_key = torch.empty((self.world * nq, k), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(g_key, t_key, group=self.group)                 # the one collective of the batch
        g_ids, g_sc = unpack_candidates(g_key.cpu().numpy().view(np.uint64).reshape(self.world, nq, k))
        return merge_topk(g_ids, g_sc)

    # ---------------------------------------------------------------- device path
    def _buffers(self, slot: int, nq: int, k: int, dev):
        import torch
        key = (slot, nq, k)
        if key not in self._bufs:
            self._bufs[key] = dict(
                ids=torch.empty((nq, k), dtype=torch.int64, device=dev),
                sc=torch.empty((nq, k), dtype=torch.float32, device=dev),
                keys=torch.empty((nq, k), dtype=torch.int64, device=dev),                   # packed candidates (u64 bit patterns)
                g_keys=torch.empty((self.world * nq, k), dtype=torch.int64, device=dev),    # == [world, nq, k]
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
                ps = C.c_void_p(self._post.cuda_stream)
                evs = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if self.timing else None
                if evs: evs[0].record(self._post)
                if self.exchange_kind == "cabi":
                    L.check(L.lib().cmr_comm_allgather_merge(
                        self.comm(), C.c_void_p(b["ids"].data_ptr()), C.c_void_p(b["sc"].data_ptr()), nq, k,
                        C.c_void_p(b["o_ids"].data_ptr()), C.c_void_p(b["o_sc"].data_ptr()), ps))
                    if evs: evs[1].record(self._post)
                else:
                    L.check(L.lib().cmr_pack_candidates_dev(C.c_void_p(b["ids"].data_ptr()), C.c_void_p(b["sc"].data_ptr()), nq * k,
                                                            C.c_void_p(b["keys"].data_ptr()), ps))
                    if self.world > 1 and dist.get_backend(self.group) == "gloo":
                        # ranks sharing one GPU (the 1-GPU rehearsal of the N > 1 flow) or a CPU collective: the packed keys go
                        # through the host around the all-gather — same keys, same single collective, same merge kernel
                        hk = b["keys"].cpu()                                  # synchronises the post stream
                        hg = torch.empty((self.world * nq, k), dtype=torch.int64)
                        dist.all_gather_into_tensor(hg, hk, group=self.group)
                        b["g_keys"].copy_(hg)
                    else:
                        dist.all_gather_into_tensor(b["g_keys"], b["keys"], group=self.group)     # the one collective of the batch
                    if evs: evs[1].record(self._post)
                    L.check(L.lib().cmr_merge_keys_dev(C.c_void_p(b["g_keys"].data_ptr()), self.world, nq, k,
                                                       C.c_void_p(b["o_ids"].data_ptr()), C.c_void_p(b["o_sc"].data_ptr()), ps))
                if evs:
                    evs[2].record(self._post)
                    self.times.append(evs)
                ev = torch.cuda.Event()
                ev.record(self._post)
            b["done"] = ev
        else:
            b["done"] = _Done(self.local, handle)
        return b


    # ---------------------------------------------------------------- C-ABI communicator
    def comm(self):
        """The library's own RCCL communicator (cmr_comm_*), created on first use: rank 0 draws the unique id, the
        group's object broadcast ships its 128 bytes (any channel would do), every rank joins."""
        if self._comm is None:
            import ctypes as C
            import torch.distributed as dist
            uid = (C.c_uint8 * 128)()
            if self.rank == 0:
                L.check(L.lib().cmr_comm_unique_id(uid))
            if self.world > 1:
                box = [bytes(uid)]
                dist.broadcast_object_list(box, src=0, group=self.group)
                uid = (C.c_uint8 * 128).from_buffer_copy(box[0])
            h = C.c_void_p()
            L.check(L.lib().cmr_comm_create(self.world, self.rank, uid, self.device, C.byref(h)))
            self._comm = h
        return self._comm

    def close(self):
        if self._comm is not None:
            L.lib().cmr_comm_destroy(self._comm)
            self._comm = None
        if self._owns_local:
            self.local.close()

    def comm_info(self) -> dict:
        """(world, rank, ranks RCCL itself reports) of the library's own communicator."""
        import ctypes as C
        w, r, n = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        L.check(L.lib().cmr_comm_info(self.comm(), C.byref(w), C.byref(r), C.byref(n)))
        return {"world": w.value, "rank": r.value, "rccl_ranks_seen": n.value}

    def view(self, exchange: str, timing: bool = False) -> "ShardedIndex":
        """A second, READ-ONLY handle on the SAME local shard with the other exchange binding (bench.py gives both bindings
        hardware time: B = 64 through one, B = 256 through the other): searches only — `append` raises; take the view after the
        appends (it copies the layout totals of that moment).  Closing the view leaves the shard alone."""
        v = ShardedIndex(self.dim, self.dtype, device=self.device, rank=self.rank, world=self.world, group=self.group, base=self.base,
                         index=_Borrowed(self.local), force_exchange=self.exchange and self.world == 1, exchange=exchange, timing=timing)
        v.local = self.local
        v.blocks, v.sizes, v.total = self.blocks, self.sizes, self.total
        v._owns_local = False
        return v

    def exchange_times_ms(self):
        """[(all-gather ms, merge ms)] of the batches recorded with timing=True (for the 'cabi' binding the first number
        is the whole pack + all-gather + merge call).  Synchronises the recorded events."""
        out = []
        for e0, e1, e2 in self.times:
            e2.synchronize()
            out.append((e0.elapsed_time(e1), e1.elapsed_time(e2)))
        self.times = []
        return out


def pack_candidates(ids: np.ndarray, scores: np.ndarray) -> np.ndarray:
    """(global row id, raw score) -> u64 keys whose unsigned order is the exported order (score desc, id asc); id < 0 -> 0.
    numpy twin of cmr_pack_candidates_dev (host / gloo path)."""
    ids = np.asarray(ids, np.int64)
    if ids.size and ids.max(initial=-1) >= 0xFFFFFFFF:
        raise ValueError("packed candidate exchange needs global row ids < 2^32 - 1")
    u = (np.asarray(scores, np.float32) + np.float32(0.0)).view(np.uint32)
    code = np.where(u & 0x80000000, ~u, u | 0x80000000).astype(np.uint64)
    key = (code << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - ids.clip(min=0).astype(np.uint64))
    return np.where(ids < 0, np.uint64(0), key)


def unpack_candidates(keys: np.ndarray):
    keys = np.asarray(keys, np.uint64)
    code = (keys >> np.uint64(32)).astype(np.uint32)
    u = np.where(code & 0x80000000, code & 0x7FFFFFFF, ~code).astype(np.uint32)
    ids = np.where(keys == 0, np.int64(-1), (np.uint64(0xFFFFFFFF) - (keys & np.uint64(0xFFFFFFFF))).astype(np.int64))
    sc = np.where(keys == 0, np.float32(-np.inf), u.view(np.float32))
    return ids, sc.astype(np.float32)


class _Borrowed:
    """Placeholder handed to ShardedIndex.__init__ by `view` (no set_id_base: the shard's ids are already set up)."""

    def __init__(self, index):
        self._i = index

    def __len__(self):
        return len(self._i)


class _Done:
    """Completion handle of a pipelined batch on a single shard (wraps the index's hipEvent)."""

    def __init__(self, index, handle):
        self._index, self._h = index, handle

    def synchronize(self):
        self._index.sync(self._h)

    def wait(self, stream=None):
        import torch
        self._index.wait(self._h, stream or torch.cuda.current_stream())
