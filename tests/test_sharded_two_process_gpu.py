"""Two PROCESSES, one row shard each, real HIP scans: the N > 1 path of comorag_amd/sharded.py end to end — per-rank
DenseIndex with a global id base, packed candidate exchange over torch.distributed, final merge — on the one GPU a test
box has (both ranks share cuda:0, so the collective runs on `gloo`; RCCL needs one device per rank and is covered on a
1-rank group by tests/test_dropin_gpu.py::test_rccl_exchange_path_one_rank).  The merged result must equal a single
index over all rows bit for bit, on every rank, and the oracle."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q_out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        import torch.distributed as dist
        from comorag_amd.index import DenseIndex
        from comorag_amd.sharded import ShardedIndex, shard_bounds
        from oracle import retrieval_np as orc
        dist.init_process_group("gloo", rank=rank, world_size=world)
        try:
            n, d, k = 150_003, 128, 20
            X = orc.synthetic_corpus(n, d, seed=17)
            lo, hi = shard_bounds(n, world, rank)
            b = shard_bounds(n, world, 1)[0]                  # first row of shard 1
            X[b - 1] = X[b] = X[n - 1] = X[11]                # one row four times, on both sides of the shard boundary: cross-shard ties
            Q = orc.synthetic_queries(9, d, seed=18, planted=X)
            Q[0] = X[11]
            sh = ShardedIndex(d, "bf16", device=0, rank=rank, world=world, base=lo)
            sh.local.append(X[lo:hi])
            ids, sc = sh.search(Q, k)
            one = DenseIndex(d, "bf16")
            one.append(X)
            w_ids, w_sc = one.search(Q, k)[:2]
            same = bool(np.array_equal(ids, w_ids) and np.array_equal(sc, w_sc))
            exact = orc.exact_scores_f64(orc.bf16_round(X), orc.bf16_round(Q))
            ref_ids, _ = orc.topk_rule(exact, k)
            for i in range(len(Q)):
                orc.assert_topk_equivalent(ids[i], ref_ids[i], exact[i], 4e-6)
            tie = sorted(int(x) for x in ids[0][:4])
            # incremental appends between searches (BASELINE config 4 at N > 1): ids stay dense in append order, chunks go to
            # the shortest shard, the device path (pipelined scan -> remap through the block table -> pack -> gloo all-gather
            # staged through the host -> key merge) and the host path both equal ONE index that took the same appends
            import torch
            rng = np.random.default_rng(19)
            for step, m in enumerate((25, 25, 5000, 1)):
                new = rng.standard_normal((m, d)).astype(np.float32)
                new /= np.linalg.norm(new, axis=1, keepdims=True)
                if step == 0:
                    new[3] = X[11]                                # a fifth copy of row 11, in an appended block
                n_before = len(one)
                got = sh.append(new, block_rows=2048)
                one.append(new)
                same &= got.tolist() == list(range(n_before, n_before + m))
                qq = np.concatenate([Q, new[:3]])
                ids, sc = sh.search(qq, k)
                w_ids, w_sc = one.search(qq, k)[:2]
                same &= bool(np.array_equal(ids, w_ids) and np.array_equal(sc, w_sc))
                qt = torch.from_numpy(qq).cuda()
                torch.cuda.synchronize()
                b = sh.search_pipelined(qt, k, step & 1)
                b["done"].synchronize()
                same &= bool(np.array_equal(b["o_ids"].cpu().numpy(), w_ids) and np.array_equal(b["o_sc"].cpu().numpy(), w_sc))
            same &= sum(sh.sizes) == len(one) and abs(sh.sizes[0] - sh.sizes[1]) <= 2048 and len(sh.blocks) >= 2
            # ids coming IN are translated too: exact re-score of global candidates, row fetch by global id
            mine = [g for g in (n + 3, n + 30, n + 5050) if sh.local.get_rows(np.array([g]))[0].any()]
            same &= all(np.array_equal(sh.local.get_rows(np.array([g]))[0], one.get_rows(np.array([g]))[0]) for g in mine)
            one.close(); sh.close()
            q_out.put((rank, same, tie, ""))
        finally:
            dist.destroy_process_group()
    except Exception as e:   # surface the failure in the parent instead of a queue timeout
        import traceback
        q_out.put((rank, False, [], traceback.format_exc()[-1500:]))


@pytest.mark.timeout(300)
def test_two_processes_two_shards_equal_one_index():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(ok for _, ok, _, _ in res), res
    n = 150_003
    lo1 = (n + 1) // 2
    assert res[0][2] == res[1][2] == sorted([11, lo1 - 1, lo1, n - 1])      # the four copies of row 11, from both shards
