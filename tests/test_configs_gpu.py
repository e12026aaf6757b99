"""BASELINE configs 3, 4 and 5 as parity cases at their full sizes (configs 1-2 are covered in
test_oracle_pin.py and test_search_gpu.py::test_c2_size_properties; small-size sharding in the
logical-shard / gloo tests)."""
import numpy as np
import pytest

from oracle import retrieval_np as orc

pytestmark = pytest.mark.gpu


def test_config4_probe_loop_with_incremental_append():
    """5 reasoning cycles x 8 probing queries over a 2 M-chunk memory pool; after every cycle the
    pool grows (25 rows = 3 nodes x 8 probes + 1 fusion, utils/memory_utils.py:176,297-300; and one
    65,536-row burst that forces a capacity doubling via hipMemcpyAsync).  Checks: every search
    equals a bulk-built index of the same rows bit for bit, new rows are retrievable immediately,
    ids/scores agree with the oracle on the rounded inputs."""
    import torch
    from comorag_amd.index import DenseIndex
    n0, d, k = 2_000_000, 768, 20
    g = torch.Generator(device="cuda"); g.manual_seed(99)
    idx = DenseIndex(d, "bf16", capacity_hint=n0)           # exactly full: the first append must grow
    host = []
    for _ in range(8):
        x = torch.randn((n0 // 8, d), generator=g, device="cuda"); x = (x / x.norm(dim=1, keepdim=True)).contiguous()
        idx.append_dev(x); host.append(x.cpu().numpy())
    X = np.concatenate(host); del host
    rng = np.random.default_rng(5)
    cap0 = idx.device_bytes
    for cycle in range(5):
        probes = orc.synthetic_queries(8, d, seed=1000 + cycle, planted=X[rng.integers(0, len(X), 4)])
        ids, sc, mn, mx = idx.search(probes, k)
        Xr, Pr = orc.bf16_round(X), orc.bf16_round(probes)
        s32 = Pr @ Xr.T
        ref_ids, _ = orc.topk_rule(s32, k)
        for i in range(8):
            if not np.array_equal(ids[i], ref_ids[i]):
                cols = np.union1d(ids[i], ref_ids[i])
                ex = np.full(len(X), -np.inf); ex[cols] = Pr[i].astype(np.float64) @ Xr[cols].astype(np.float64).T
                orc.assert_topk_equivalent(ids[i], ref_ids[i], ex, 4e-6)
            np.testing.assert_allclose(sc[i], s32[i][ids[i]], atol=4e-6)
        np.testing.assert_allclose(mx, s32.max(1), atol=4e-6)
        n_new = 65_536 if cycle == 2 else 25
        new = orc.synthetic_corpus(n_new, d, seed=2000 + cycle)
        new[0] = probes[0]                                   # the "fused node" for probe 0
        idx.append(new)
        X = np.concatenate([X, new])
        hit, hsc, _, _ = idx.search(probes[:1], 1)
        assert hit[0, 0] == len(X) - n_new and hsc[0, 0] > 0.99    # retrievable immediately, right global row id
    assert len(idx) == len(X) and idx.device_bytes > cap0           # grew (capacity doubling path)
    bulk = DenseIndex(d, "bf16", capacity_hint=len(X)); bulk.append(X)
    pq = orc.synthetic_queries(8, d, seed=77)
    a, b = idx.search(pq, k), bulk.search(pq, k)
    assert all(np.array_equal(u, v) for u, v in zip(a, b))
    idx.close(); bulk.close()


def test_config5_encode_then_search_then_rescore():
    """BGE-large shape (1024-d) fp16: ~200 K tokens of narrative text (391 chunks x 512 tokens) →
    encoder (random init: no weights offline) → fp16 index with fp32 shadow → top-100 → exact fp32
    top-20.  Parity: pooled vectors are unit norm and equal the oracle pool on the same hidden
    states (test_dropin_gpu.py); here the search + rescore stages are checked against the oracle on
    the produced embeddings."""
    import torch
    from comorag_amd.embedding_model.bge import HipBGEEmbeddingModel
    from comorag_amd.index import DenseIndex
    from comorag_amd.utils.config_utils import BaseConfig
    from tools.synthetic import random_bert, synthetic_chunks, synthetic_wordpiece_tokenizer
    tok, words = synthetic_wordpiece_tokenizer(8000)
    cfg = BaseConfig(embedding_model_name="bge-large-random-init", embedding_batch_size=32, embedding_model_dtype="fp16")
    em = HipBGEEmbeddingModel(cfg, cfg.embedding_model_name, model=random_bert("large", vocab_size=len(tok)), tokenizer=tok)
    chunks = synthetic_chunks(words, 391, tokens_per_chunk=500)
    E = em.batch_encode(chunks)
    assert E.shape == (391, 1024) and E.dtype == np.float32
    np.testing.assert_allclose((E.astype(np.float64) ** 2).sum(1), 1.0, atol=1e-5)
    Q = em.batch_encode([c[:200] for c in chunks[:8]], is_query=True)
    idx = DenseIndex(1024, "f16", keep_f32=True); idx.append(E)
    cand, csc, _, _ = idx.search(Q, 100)
    ex16 = orc.exact_scores_f64(orc.f16_round(E), orc.f16_round(Q))
    ref100, _ = orc.topk_rule(ex16, 100)
    # random-init BERT vectors are nearly collinear: many near-ties → tie-aware compare
    for i in range(8):
        orc.assert_topk_equivalent(cand[i], ref100[i], ex16[i], 4e-6)
    ids, sc = idx.rescore(Q, cand, 20)
    ex32 = orc.exact_scores_f64(E, Q)
    for i in range(8):
        want = cand[i][np.lexsort((cand[i], -ex32[i][cand[i]]))][:20]
        orc.assert_topk_equivalent(ids[i], want, ex32[i], 1e-6)
        np.testing.assert_allclose(sc[i], ex32[i][ids[i]], atol=1e-6)
    idx.close()


def test_config3_full_size_eight_shards_equal_one_index():
    """BASELINE config 3 at its full size on one device: 10 M x 768 bf16 rows as 8 row shards of
    1.25 M (global ids via cmr_index_set_id_base), batch 256, k = 20, per-shard top-k + final merge.
    Size-independent properties: the merged result is bit-identical to ONE 10 M-row index (a row's
    score is the same fp32 chain wherever the row lives), planted neighbours come back first with
    their global id, scores are sorted, the global min/max are the extremes over the shards."""
    import torch
    from comorag_amd.index import DenseIndex, merge_topk
    rows, dim, S, B, k, blk = 10_000_000, 768, 8, 256, 20, 250_000
    per = rows // S
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(31337)
    single = DenseIndex(dim, "bf16", capacity_hint=rows)
    shards = [DenseIndex(dim, "bf16", capacity_hint=per) for _ in range(S)]
    for s, sh in enumerate(shards):
        sh.set_id_base(s * per)
    planted_ids, planted_rows = [], []
    for b0 in range(0, rows, blk):
        x = torch.randn((blk, dim), generator=g, device=dev)
        x = (x / x.norm(dim=1, keepdim=True)).contiguous()
        single.append_dev(x)
        shards[b0 // per].append_dev(x)
        planted_ids.append(b0 + 4321); planted_rows.append(x[4321].clone())
    del x
    q = torch.randn((B, dim), generator=g, device=dev)
    npl = len(planted_ids)                                           # 40 planted queries, 216 random ones
    q[:npl] = torch.stack(planted_rows) + 0.01 * q[:npl]           # |noise| ~ 0.28 -> cos ~ 0.96
    q = (q / q.norm(dim=1, keepdim=True)).cpu().numpy()
    ids1, sc1, mn1, mx1 = single.search(q, k)
    parts = [sh.search(q, k) for sh in shards]
    ids2, sc2 = merge_topk(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]))
    assert np.array_equal(ids1, ids2) and np.array_equal(sc1, sc2)
    assert np.array_equal(ids1[:npl, 0], np.asarray(planted_ids))
    assert np.all(sc1[:npl, 0] > 0.9)
    assert np.all(np.diff(sc1, axis=1) <= 0) and np.all(ids1 >= 0) and np.all(ids1 < rows)
    assert all(len(set(r.tolist())) == k for r in ids1)
    assert np.array_equal(mx1, sc1[:, 0]) and np.array_equal(mx1, np.max([p[3] for p in parts], axis=0))
    assert np.array_equal(mn1, np.min([p[2] for p in parts], axis=0)) and np.all(mn1 <= sc1[:, -1])
    single.close()
    for sh in shards:
        sh.close()


def _host_oracle_topk(q32, blocks_bf16, k, extra=8):
    """Top-(k+extra) candidates per query and block from a host fp32 GEMM over the bf16-rounded rows, then fp64 scores
    of the pooled candidates: returns (pool ids [nq, P], exact fp64 scores of the pool [nq, P])."""
    import torch
    nq = q32.shape[0]
    qt = torch.from_numpy(q32)
    pools = []
    base = 0
    for xb in blocks_bf16:
        s = qt @ xb.float().T                                    # [nq, rows] fp32 on the host cores
        top = torch.topk(s, min(k + extra, s.shape[1]), dim=1).indices.numpy() + base
        pools.append(top)
        base += xb.shape[0]
    pool = np.concatenate(pools, axis=1)
    return pool, base


def _exact_on(ids, q32, blocks_bf16, blk):
    """fp64 scores of rows `ids` (global) for one query."""
    rows = np.stack([blocks_bf16[i // blk][i % blk].float().numpy() for i in ids.tolist()])
    return rows.astype(np.float64) @ q32.astype(np.float64)


@pytest.fixture(scope="module")
def ten_million():
    """10 M x 768 bf16 rows in HBM + their bf16-rounded host copy (the oracle's input): built once for the full-size tests."""
    import torch
    from comorag_amd.index import DenseIndex
    rows, dim, blk = 10_000_000, 768, 250_000
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(20260925)
    idx = DenseIndex(dim, "bf16", capacity_hint=rows)
    host = []
    for b0 in range(0, rows, blk):
        x = torch.randn((blk, dim), generator=g, device=dev)
        x = (x / x.norm(dim=1, keepdim=True)).contiguous()
        idx.append_dev(x)
        host.append(x.to(torch.bfloat16).cpu())                  # RN-even, as the index rounds
    yield idx, host, g, blk
    idx.close()


def _check_against_host_oracle(ids, sc, qr, host, blk, k, rows):
    """ids / scores [B, k] of the HIP index vs the host fp32 GEMM over the bf16-rounded rows, fp64 arbitration."""
    B = len(qr)
    pool, n = _host_oracle_topk(qr, host, k)
    assert n == rows
    for i in range(B):
        cand = np.union1d(pool[i], ids[i])
        ex = _exact_on(cand, qr[i], host, blk)
        order = np.lexsort((cand, -ex))[:k]
        exact_full = {int(c): float(e) for c, e in zip(cand, ex)}
        ref = cand[order]
        if not np.array_equal(ids[i], ref):                       # only near-ties inside the fp32 accumulation bound may differ
            for a, b in zip(ids[i], ref):
                assert a == b or abs(exact_full[int(a)] - exact_full[int(b)]) < 4e-6, (i, a, b)
            assert abs(min(exact_full[int(a)] for a in ids[i]) - ex[order][-1]) < 4e-6
        np.testing.assert_allclose(sc[i], [exact_full[int(a)] for a in ids[i]], atol=4e-6)
        assert np.all(np.diff(sc[i]) <= 0)


def test_headline_config_pipelined_vs_host_oracle(ten_million):
    """The configuration bench.py times — 10 M x 768 bf16 rows, batch 64, k = 20, cmr_index_search_pipelined (narrow
    kernel, sampling thresholds, reserved CUs) — against an oracle at FULL size: host fp32 GEMM over the bf16-rounded
    corpus copied back from the device, fp64 arbitration of the candidates.  Also: pipelined == synchronous, bit for bit."""
    import torch
    idx, host, g, blk = ten_million
    rows, dim, B, k = 10_000_000, 768, 64, 20
    dev = torch.device("cuda", 0)
    q = torch.randn((B, dim), generator=g, device=dev)
    q[:8] = torch.stack([host[5 * i][77 + i].float().to(dev) for i in range(8)]) + 0.02 * q[:8]     # planted neighbours
    q = (q / q.norm(dim=1, keepdim=True)).contiguous()
    outs = [(torch.empty((B, k), dtype=torch.int64, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev),
             torch.empty(B, dtype=torch.float32, device=dev), torch.empty(B, dtype=torch.float32, device=dev)) for _ in range(2)]
    h = None
    for i in range(4):                                            # steady state of the pipeline: both slots used
        o = outs[i & 1]
        h = idx.search_pipelined(q, k, o[0], o[1], o[2], o[3])
    idx.sync(h); torch.cuda.synchronize()
    ids, sc, mn, mx = (t.cpu().numpy() for t in outs[1])
    ids0, sc0 = outs[0][0].cpu().numpy(), outs[0][1].cpu().numpy()
    assert np.array_equal(ids, ids0) and np.array_equal(sc, sc0)                  # both slots
    qh = q.cpu().numpy()
    sid, ssc, smn, smx = idx.search(qh, k)                                          # synchronous host API
    assert np.array_equal(ids, sid) and np.array_equal(sc, ssc) and np.array_equal(mn, smn) and np.array_equal(mx, smx)
    assert not idx.query_status()
    qr = q.to(torch.bfloat16).float().cpu().numpy()                                 # queries are rounded to the index dtype
    _check_against_host_oracle(ids, sc, qr, host, blk, k, rows)
    assert np.all(sc[:8, 0] > 0.8) and [int(ids[i, 0]) for i in range(8)] == [5 * i * blk + 77 + i for i in range(8)]
    assert np.all(mx == sc[:, 0])


@pytest.mark.timeout(900)
def test_config3_batch256_wide_pass_vs_narrow_passes_and_host_oracle(ten_million):
    """BASELINE config 3's batch at FULL size on one device: 10 M x 768 bf16 rows, B = 256, k = 20, ONE corpus pass of the
    wide kernel in pipelined mode (what bench.py's config3_batch256 row times) — bit for bit against (a) four passes of the
    narrow kernel (scan_no_wide = 1), (b) the query-split grid of the narrow kernel (wide_mode = 2), (c) the synchronous
    host API; and against the host oracle (fp32 GEMM over the bf16-rounded rows, fp64 arbitration) like the headline."""
    import torch
    idx, host, g, blk = ten_million
    rows, dim, B, k = 10_000_000, 768, 256, 20
    dev = torch.device("cuda", 0)
    q = torch.randn((B, dim), generator=g, device=dev)
    q[:8] = torch.stack([host[3 * i + 1][1234 + i].float().to(dev) for i in range(8)]) + 0.02 * q[:8]     # planted neighbours
    q[200] = host[39][249_999].float().to(dev)                                                           # the corpus' last row, exactly
    q = (q / q.norm(dim=1, keepdim=True)).contiguous()

    def pipelined():
        outs = [(torch.empty((B, k), dtype=torch.int64, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev),
                 torch.empty(B, dtype=torch.float32, device=dev), torch.empty(B, dtype=torch.float32, device=dev)) for _ in range(2)]
        h = None
        for i in range(3):
            o = outs[i & 1]
            h = idx.search_pipelined(q, k, o[0], o[1], o[2], o[3])
        idx.sync(h); torch.cuda.synchronize()
        a, b = [t.cpu().numpy() for t in outs[0]], [t.cpu().numpy() for t in outs[1]]
        assert all(np.array_equal(x, y) for x, y in zip(a, b))                     # both slots
        return a

    try:
        wide = pipelined()
        idx.set_option("scan_no_wide", 1)
        narrow = pipelined()
        idx.set_option("scan_no_wide", 0); idx.set_option("wide_mode", 2)
        grid = pipelined()
    finally:
        idx.set_option("scan_no_wide", 0); idx.set_option("wide_mode", 0)
    for x, y, z in zip(wide, narrow, grid):
        assert np.array_equal(x, y) and np.array_equal(x, z)
    sync = idx.search(q.cpu().numpy(), k)
    assert all(np.array_equal(x, y) for x, y in zip(wide, sync))
    ids, sc, mn, mx = wide
    qr = q.to(torch.bfloat16).float().cpu().numpy()
    _check_against_host_oracle(ids, sc, qr, host, blk, k, rows)
    assert [int(ids[i, 0]) for i in range(8)] == [(3 * i + 1) * blk + 1234 + i for i in range(8)] and int(ids[200, 0]) == rows - 1
    assert np.all(mx == sc[:, 0])


def test_fp32_index_one_million_rows_vs_host_oracle():
    """north_star's 'bit-exact top-k indices at k <= 20 for fp32' at 1 M x 768: fp32 index (v_mfma_f32_32x32x2_f32, exact
    k-ordered fmaf chains) vs the host fp32 GEMM the reference runs (np.dot -> OpenBLAS), ids identical up to ties
    inside the fp32 accumulation bound (arbitrated in fp64), normalised scores within 2e-6."""
    from comorag_amd.index import DenseIndex
    n, d, b, k = 1_000_000, 768, 32, 20
    X = np.concatenate([orc.synthetic_corpus(250_000, d, seed=4321, block=i) for i in range(4)])
    Q = orc.synthetic_queries(b, d, seed=78, planted=X[::50_000])
    idx = DenseIndex(d, "f32", capacity_hint=n); idx.append(X)
    ids, sc, mn, mx = idx.search(Q, k)
    s32 = Q @ X.T                                                 # what ComoRAG.dense_passage_retrieval computes (:958-962)
    ref_ids, _ = orc.topk_rule(s32, k)
    n_diff = 0
    for i in range(b):
        if not np.array_equal(ids[i], ref_ids[i]):
            n_diff += 1
            cols = np.union1d(ids[i], ref_ids[i])
            ex = np.full(n, -np.inf); ex[cols] = X[cols].astype(np.float64) @ Q[i].astype(np.float64)
            orc.assert_topk_equivalent(ids[i], ref_ids[i], ex, 4e-6)
        np.testing.assert_allclose(sc[i], s32[i][ids[i]], atol=2e-6)
        norm = (sc[i] - mn[i]) / (mx[i] - mn[i])
        np.testing.assert_allclose(norm, orc.min_max_normalize(s32[i])[ids[i]], atol=2e-6)
    assert n_diff <= 2, n_diff            # identical ids is the rule; a swap needs two scores within fp32 rounding of each other
    idx.close()


@pytest.mark.timeout(600)
def test_fp32_index_ten_million_rows_vs_host_oracle():
    """The same fp32 claim at the HEADLINE size: 10 M x 768 fp32 rows (30.7 GB of HBM), k = 20, B = 8, against the host
    fp32 GEMM the reference runs (np.dot -> OpenBLAS) taken in 250 K-row blocks with a running top-k; ids identical up to
    ties inside the fp32 accumulation bound (arbitrated in fp64 on the rows in question).  Rows are generated on the device
    (seeded, per block) and copied back block by block, so the host never holds more than one block."""
    import torch
    from comorag_amd.index import DenseIndex
    n, d, b, k, blk = 10_000_000, 768, 8, 20, 250_000
    dev = torch.device("cuda", 0)
    idx = DenseIndex(d, "f32", capacity_hint=n)
    rng = np.random.default_rng(99)
    Q = rng.standard_normal((b, d)).astype(np.float32); Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    best_s = np.full((b, 0), -np.inf, np.float32); best_i = np.zeros((b, 0), np.int64)
    keep_rows = {}
    gmin, gmax = np.full(b, np.inf, np.float32), np.full(b, -np.inf, np.float32)
    for bi in range(n // blk):
        g = torch.Generator(device=dev); g.manual_seed(777_000 + bi)
        x = torch.randn((blk, d), generator=g, device=dev, dtype=torch.float32)
        x = (x / x.norm(dim=1, keepdim=True)).contiguous()
        if bi % 5 == 0:                                          # a planted near-neighbour of query (bi / 5) % b in this block
            qi = (bi // 5) % b
            x[1234 + bi] = torch.from_numpy(Q[qi]).to(dev) + 0.05 * x[1234 + bi]
            x[1234 + bi] /= x[1234 + bi].norm()
        idx.append_dev(x)
        xh = x.cpu().numpy()
        s = Q @ xh.T                                             # the reference's arithmetic (ComoRAG.py:958-962) on this block
        gmin, gmax = np.minimum(gmin, s.min(1)), np.maximum(gmax, s.max(1))
        part = np.argpartition(-s, 2 * k, axis=1)[:, :2 * k]
        cs = np.concatenate([best_s, np.take_along_axis(s, part, 1)], 1); ci = np.concatenate([best_i, part + bi * blk], 1)
        order = np.lexsort((ci, -cs), axis=1)[:, :2 * k]         # score descending, id ascending: the exported rule
        best_s, best_i = np.take_along_axis(cs, order, 1), np.take_along_axis(ci, order, 1)
        for r in np.unique(best_i):
            if r // blk == bi:
                keep_rows[int(r)] = xh[r - bi * blk].copy()
    ids, sc, mn, mx = idx.search(Q, k)
    n_diff = 0
    for i in range(b):
        ref = best_i[i, :k]
        if not np.array_equal(ids[i], ref):
            n_diff += 1
            cols = np.union1d(ids[i], best_i[i])
            assert all(int(c) in keep_rows for c in ids[i]), "the index returned a row the host ranking never had among its 40 best"
            ex = {int(c): float(keep_rows[int(c)].astype(np.float64) @ Q[i].astype(np.float64)) for c in cols if int(c) in keep_rows}
            exv = np.full(n, -np.inf); exv[list(ex)] = list(ex.values())
            orc.assert_topk_equivalent(ids[i], ref, exv, 4e-6)
        np.testing.assert_allclose(sc[i], best_s[i, :k], atol=2e-6)
    assert n_diff <= 1, n_diff
    np.testing.assert_allclose(mn, gmin, atol=2e-6); np.testing.assert_allclose(mx, gmax, atol=2e-6)
    assert np.all(sc[:, 0] > 0.9)                                # every query has planted neighbours
    idx.close()
