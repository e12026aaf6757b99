"""world_size-2 `gloo` test of the N>1 path (runs on CPU): each rank owns a contiguous row block,
candidates are all-gathered with torch.distributed and merged with the library's host merge
(`cmr_merge_topk`, plain C++ — no device call).  The per-shard scan is substituted by the oracle
here because there is no GPU; on a GPU box the same code path runs the HIP scan (the logical-shard
equivalence of the HIP scan itself is covered by tests/test_dropin_gpu.py)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q_out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from comorag_amd.sharded import ShardedIndex, shard_bounds
    from oracle import retrieval_np as orc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X = orc.synthetic_corpus(4001, 64, seed=7)
        X[3000] = X[10]                                   # tie across the shard boundary
        Q = orc.synthetic_queries(6, 64, seed=8, planted=X)
        lo, hi = shard_bounds(len(X), world, rank)

        class Local:                                      # stands in for the DenseIndex of this rank
            def __len__(self): return hi - lo
        sh = ShardedIndex(64, "f32", rank=rank, world=world, base=lo, index=Local())
        def local_search(q, k):
            return orc.topk_rule(orc.exact_scores_f64(X[lo:hi], q).astype(np.float32), k)
        ids, sc = sh.search(Q, 20, local_search=local_search)
        want_i, want_s = orc.topk_rule(orc.exact_scores_f64(X, Q).astype(np.float32), 20)
        ok = bool(np.array_equal(ids, want_i) and np.allclose(sc, want_s))
        q_out.put((rank, ok, ids[:, :3].tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_gather_and_merge_equals_single_shard():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] == res[1][2]                          # every rank holds the same merged answer


def _append_worker(rank, world, port, q_out):
    """Incremental appends on a sharded index (BASELINE config 4 at N > 1): ids stay dense in append order,
    every append goes to the shortest shard, searches between the appends equal one index over everything so far."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from comorag_amd.sharded import ShardedIndex, shard_bounds
    from oracle import retrieval_np as orc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X = orc.synthetic_corpus(3001, 64, seed=7)               # bulk: shard 0 gets 1501 rows, shard 1 1500
        Q = orc.synthetic_queries(6, 64, seed=8, planted=X)
        lo, hi = shard_bounds(len(X), world, rank)

        class Local:                                             # numpy stand-in for this rank's DenseIndex (append + len)
            def __init__(self, x): self.x = x.copy()
            def __len__(self): return len(self.x)
            def append(self, rows): self.x = np.concatenate([self.x, rows])
        loc = Local(X[lo:hi])
        sh = ShardedIndex(64, "f32", rank=rank, world=world, base=lo, index=loc)
        local_search = lambda q, k: orc.topk_rule(orc.exact_scores_f64(loc.x, q).astype(np.float32), k)
        everything = X
        ok, log = True, []
        rng = np.random.default_rng(5)
        for step, m in enumerate((25, 25, 3, 700, 1)):           # 700 rows in chunks of 256: spread over both shards
            new = rng.standard_normal((m, 64)).astype(np.float32)
            new /= np.linalg.norm(new, axis=1, keepdims=True)
            if step == 1:
                new[0] = X[10]                                   # duplicate of a bulk row: tie across blocks, lower global id first
            got_ids = sh.append(new, block_rows=256)
            ok &= got_ids.tolist() == list(range(len(everything), len(everything) + m))
            everything = np.concatenate([everything, new])
            qq = np.concatenate([Q, new[:2]])
            ids, sc = sh.search(qq, 20, local_search=local_search)
            want_i, want_s = orc.topk_rule(orc.exact_scores_f64(everything, qq).astype(np.float32), 20)
            ok &= bool(np.array_equal(ids, want_i) and np.allclose(sc, want_s))
            log.append((list(sh.sizes), len(sh.blocks)))
        ok &= sum(sh.sizes) == len(everything) == sh.total and abs(sh.sizes[0] - sh.sizes[1]) <= 256
        q_out.put((rank, ok, log))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_incremental_append_keeps_dense_global_ids():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_append_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(ok for _, ok, _ in res), res
    by_rank = dict((r, log) for r, _, log in res)
    assert [s for s, _ in by_rank[0]] == [s for s, _ in by_rank[1]]        # the size table is identical on every rank
    assert by_rank[0][0][0] == [1501, 1525]                                # the first append went to the shorter shard


def test_route_is_deterministic_and_balances():
    from comorag_amd.sharded import ShardedIndex

    class Local:
        def __len__(self): return 0
    sh = ShardedIndex(8, "f32", rank=0, world=4, base=0, index=Local())
    sh.sizes, sh.total = [10, 7, 7, 12], 36
    assert sh._route(5, 8192) == [(1, 5)]                       # a new block opens on the shortest shard (lowest rank on ties)
    assert sh._route(20, 8) == [(1, 8), (2, 8), (0, 4)]         # full blocks move on to the then-shortest shard
    assert sh._route(3, 8, commit=True) == [(1, 3)]
    sh.sizes[1] += 3
    assert sh._route(4, 8) == [(1, 4)]                          # the open block keeps taking rows although shard 2 is shorter now
    assert sh._route(9, 8) == [(1, 5), (2, 4)]


def test_shard_bounds_cover_rows_exactly():
    from comorag_amd.sharded import shard_bounds
    for n in (0, 1, 7, 8, 10_000_000, 4001):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))


def test_views_are_read_only():
    """ShardedIndex.view() shares the shard but copies the layout totals: an append through it would leave the owning handle's
    routing state stale — it raises instead."""
    import pytest
    from comorag_amd.sharded import ShardedIndex

    class _Local:
        def __init__(self): self.rows = np.empty((0, 4), np.float32)
        def __len__(self): return len(self.rows)
        def append(self, r): self.rows = np.concatenate([self.rows, r])
        def set_id_base(self, b): pass
        def set_id_blocks(self, l, g): pass
        def close(self): pass

    sh = ShardedIndex(4, "f32", rank=0, world=1, index=_Local())
    sh.append(np.ones((3, 4), np.float32))
    v = sh.view("torch")
    with pytest.raises(RuntimeError, match="read-only"):
        v.append(np.ones((2, 4), np.float32))
    assert len(sh) == 3 and sh.total == 3
    sh.append(np.ones((2, 4), np.float32))
    assert sh.total == 5
    v.close()
    assert len(sh.local) == 5
