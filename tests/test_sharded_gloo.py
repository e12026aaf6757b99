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


def test_shard_bounds_cover_rows_exactly():
    from comorag_amd.sharded import shard_bounds
    for n in (0, 1, 7, 8, 10_000_000, 4001):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
