"""The scan with the finishing stage (scan_kernel MODE_FIN: a synchronous caller's handful of queries on a corpus beyond the
single-launch path — thresholds from the first panels and the final selection inside the scan launch) against the oracle and,
bit for bit, against the sampling / scan / merge chain it replaces (option scan_fin = 0)."""
import numpy as np
import pytest

from oracle import retrieval_np as orc

pytestmark = pytest.mark.gpu

ROUND = {"bf16": orc.bf16_round, "f16": orc.f16_round, "f32": lambda x: np.asarray(x, np.float32)}
ERR = 4e-6


def _mk(n, d, nq, seed=0):
    X = orc.synthetic_corpus(n, d, seed=300 + seed)
    Q = orc.synthetic_queries(nq, d, seed=400 + seed, planted=X)
    return X, Q


def _index(dtype, X, options=None, **kw):
    from comorag_amd.index import DenseIndex
    options = dict(options or {})
    if options.get("scan_fin", 1):
        options.setdefault("scan_fin_queries", 32)       # the stage takes batches up to a query tile; the default route stops earlier (measured)
    idx = DenseIndex(X.shape[1], dtype, options=options, **kw)
    idx.append(X)
    return idx


def _oracle_check(dtype, X, Q, k, out):
    ids, sc, mn, mx = out
    rnd = ROUND[dtype]
    exact = orc.exact_scores_f64(rnd(X), rnd(Q))
    ref_ids, _ = orc.topk_rule(exact, k)
    for i in range(Q.shape[0]):
        orc.assert_topk_equivalent(ids[i], ref_ids[i], exact[i], ERR)
        np.testing.assert_allclose(sc[i], exact[i][ids[i]], atol=ERR, rtol=0)
        assert np.all(np.diff(sc[i]) <= 0)
    np.testing.assert_allclose(mn, exact.min(axis=1), atol=ERR)
    np.testing.assert_allclose(mx, exact.max(axis=1), atol=ERR)


def _same(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("dtype,n,d,nq,k", [("bf16", 200_000, 128, 1, 20), ("bf16", 300_000, 64, 8, 20), ("f16", 262_144 + 7, 256, 3, 1),
                                            ("f32", 220_001, 64, 2, 64), ("bf16", 250_000, 384, 5, 33), ("f32", 200_000, 200, 1, 20),
                                            ("bf16", 1_000_003, 64, 8, 20), ("bf16", 600_000, 768, 1, 20)])
def test_finishing_stage_equals_the_chain_and_the_oracle(dtype, n, d, nq, k):
    X, Q = _mk(n, d, nq, seed=(n + nq + k) % 997)
    fin = _index(dtype, X)
    chain = _index(dtype, X, {"scan_fin": 0})
    a = fin.search(Q, k)
    b = chain.search(Q, k)
    assert _same(a, b)
    if n <= 300_000:
        _oracle_check(dtype, X, Q, k, a)
    # again on the same workspace (the control words re-arm themselves), other batch sizes in between
    for m in (1, nq, max(1, nq - 1)):
        assert _same(fin.search(Q[:m], k), chain.search(Q[:m], k))
    # the compiler-counted ring of the same kernel
    slow = _index(dtype, X, {"scan_asm_ring": 0})
    assert _same(slow.search(Q, k), a)
    for i in (fin, chain, slow):
        i.close()


def test_finishing_stage_dense_list_overflow_falls_back_to_the_merge_launch():
    X, Q = _mk(400_000, 64, 4, seed=11)
    tight = _index("bf16", X, {"scan_fin_dense": 8})        # every dense list overflows (k = 20 > 8 keys): state 2, the merge launch decides
    chain = _index("bf16", X, {"scan_fin": 0})
    for k in (20, 5):
        assert _same(tight.search(Q, k), chain.search(Q, k))
    # one list overflows, the others do not: a query that matches thousands of duplicated rows next to ordinary ones
    X2 = X.copy()
    X2[1000:9000] = X2[7]
    Q2 = Q.copy(); Q2[1] = X2[7]
    some = _index("bf16", X2, {"scan_fin_dense": 4096})
    chain2 = _index("bf16", X2, {"scan_fin": 0})
    a = some.search(Q2, 20)
    assert _same(a, chain2.search(Q2, 20))
    assert a[0][1].tolist() == [7] + list(range(1000, 1019))       # ties in index order
    for i in (tight, chain, some, chain2):
        i.close()


def test_finishing_stage_batches_up_to_one_query_tile():
    X, Q = _mk(300_000, 128, 32, seed=5)
    wide = _index("bf16", X, {"scan_fin_queries": 32})
    chain = _index("bf16", X, {"scan_fin": 0})
    for m in (9, 17, 32):
        assert _same(wide.search(Q[:m], 20), chain.search(Q[:m], 20))
    wide.close(); chain.close()


def test_finishing_stage_ties_few_rows_above_threshold_and_global_ids():
    from comorag_amd.index import DenseIndex
    n, d = 230_000, 64
    X = orc.synthetic_corpus(n, d, seed=21)
    X[100_000] = X[7]; X[229_999] = X[7]; X[31] = X[7]
    q = X[7:8].copy()
    for dtype in ("bf16", "f32"):
        idx = _index(dtype, X)
        ids, sc, _, _ = idx.search(q, 6)
        assert ids[0, :4].tolist() == [7, 31, 100_000, 229_999], ids
        assert sc[0, 0] == sc[0, 1] == sc[0, 2] == sc[0, 3]
        idx.close()
    # every row identical: the first k rows in index order, min == max
    Z = np.tile(X[3:4], (200_000, 1))
    idx = _index("bf16", Z)
    ids, sc, mn, mx = idx.search(q, 10)
    assert ids[0].tolist() == list(range(10)) and mn[0] == mx[0]
    idx.close()
    # a row shard: ids come back global (id_base), also through an id-block table
    sh = DenseIndex(d, "bf16", options={"scan_fin_queries": 32})
    sh.set_id_base(5_000_000)
    sh.append(X)
    ids, _, _, _ = sh.search(q, 4)
    assert ids[0].tolist() == [5_000_007, 5_000_031, 5_100_000, 5_229_999]
    sh.set_id_blocks([0, 100_000], [1_000, 9_000_000])
    ids, _, _, _ = sh.search(q, 4)
    assert ids[0].tolist() == [1_007, 1_031, 9_000_000, 9_129_999]
    sh.close()


def test_finishing_stage_from_sixteen_threads():
    from concurrent.futures import ThreadPoolExecutor
    X, Q = _mk(260_000, 64, 32, seed=9)
    idx = _index("bf16", X)
    chain = _index("bf16", X, {"scan_fin": 0})
    want = chain.search(Q, 10)[0]
    def work(i):
        return idx.search(Q[i:i + 2], 10)[0]
    with ThreadPoolExecutor(16) as ex:
        outs = list(ex.map(work, list(range(0, 32, 2)) * 6))
    for j, o in enumerate(outs):
        i = (j % 16) * 2
        assert np.array_equal(o, want[i:i + 2])
    idx.close(); chain.close()


def test_finishing_stage_on_the_callers_stream_and_through_the_multi_device_index():
    import torch
    from comorag_amd.multi_index import MultiDeviceIndex
    X, Q = _mk(450_000, 64, 4, seed=13)
    one = _index("bf16", X)
    chain = _index("bf16", X, {"scan_fin": 0})
    want = chain.search(Q, 20)
    # device buffers, the caller's stream (cmr_index_search_dev): same single-stream chain, same finishing stage
    qd = torch.from_numpy(Q).cuda()
    ids, sc = one.search_dev(qd, 20)
    torch.cuda.synchronize()
    assert np.array_equal(ids.cpu().numpy(), want[0]) and np.array_equal(sc.cpu().numpy(), want[1])
    m = MultiDeviceIndex(64, "bf16", devices=[0, 0], options={"scan_fin_queries": 32})
    m.append(X)
    got = m.search(Q, 20)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    m.close(); one.close(); chain.close()


def test_finishing_stage_keeps_the_key_a_compaction_made_the_threshold():
    """k = 1 (and small k): a list compacted before the published thresholds arrive has its own k-th best AS its threshold — the
    hand-over must keep that key.  Whether a wave compacts depends on timing: many repetitions, every one against the chain."""
    X, Q = _mk(400_000, 64, 8, seed=31)
    X[399_990] = Q[0]; X[17] = Q[3]; X[123_456] = Q[7]
    fin = _index("bf16", X)
    chain = _index("bf16", X, {"scan_fin": 0})
    for k in (1, 2, 5):
        want = chain.search(Q, k)
        one = chain.search(Q[:1], k)
        for _ in range(15):
            assert _same(fin.search(Q, k), want)
            assert _same(fin.search(Q[:1], k), one)
    assert fin.search(Q[:1], 1)[0][0, 0] == 399_990
    fin.close(); chain.close()


@pytest.mark.parametrize("n,d", [(6, 64), (900, 128), (40_000, 64), (400_000, 64)])
def test_polled_done_word_returns_what_the_stream_wait_returns(n, d):
    """The synchronous host API polls a word in its mapped result buffer that the search's last kernel sets behind its results
    (single-launch search and scan with the finishing stage; option sync_poll, default on) instead of waiting for the stream: call after call
    on one workspace, batch sizes — hence buffer layouts — changing in between, must return what the stream wait returns (a done word that
    overtook the results would hand back the previous call's bytes), a NaN query must still be reported, and the call after it must be clean."""
    from comorag_amd._lib import CmrError, CMR_ERR_NONFINITE
    X, Q = _mk(n, d, 16, seed=n % 89)
    poll = _index("bf16", X, {"sync_poll": 1})
    wait = _index("bf16", X, {"sync_poll": 0})
    k = min(20, n)
    rng = np.random.default_rng(5)
    for rep in range(120):
        m = int(rng.integers(1, 17))
        q = Q[rng.permutation(16)[:m]] * np.float32(1.0 + 0.01 * rep)
        assert _same(poll.search(q, k), wait.search(q, k)), (rep, m)
        if rep % 3 == 0:        # all N scores per query (single launch up to 192 K rows: every workgroup's rows, then the last one's word)
            assert np.array_equal(poll.scores(q), wait.scores(q)), (rep, m)
        if rep % 40 == 7:
            bad = q.copy(); bad[m // 2, 3] = np.nan
            with pytest.raises(CmrError) as ei:
                poll.search(bad, k)
            assert ei.value.code == CMR_ERR_NONFINITE
            with pytest.raises(CmrError) as ei:
                poll.scores(bad)
            assert ei.value.code == CMR_ERR_NONFINITE
    for i in (poll, wait):
        i.close()


def test_polled_done_word_from_sixteen_threads_and_on_overflow():
    import threading
    X, Q = _mk(300_000, 64, 16, seed=3)
    poll = _index("bf16", X, {"sync_poll": 1})
    ref = [_index("bf16", X, {"sync_poll": 0, "scan_fin": 0}).search(Q[i:i + 1 + i % 4], 20) for i in range(12)]
    errs = []
    def work(t):
        try:
            for rep in range(25):
                i = (t + rep) % 12
                if not _same(poll.search(Q[i:i + 1 + i % 4], 20), ref[i]): errs.append((t, rep, i))
        except Exception as e:      # noqa: BLE001
            errs.append(repr(e))
    th = [threading.Thread(target=work, args=(t,)) for t in range(16)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errs, errs[:4]
    # state 2 (every dense list overflows): the merge launch is issued by the polling host, only then
    tight = _index("bf16", X, {"sync_poll": 1, "scan_fin_dense": 8})
    for i in range(12):
        assert _same(tight.search(Q[i:i + 1 + i % 4], 20), ref[i])
    poll.close(); tight.close()


def test_polled_calls_by_the_ten_thousand_never_wait_for_the_stream():
    """40 000 polled calls in a row on one workspace — the runtime never sees a stream wait from this thread and has to recycle its launch
    resources (signals, kernel-argument chunks) on its own — and the last call still returns what the first one did."""
    X, Q = _mk(2000, 64, 4, seed=9)
    idx = _index("bf16", X, {"sync_poll": 1})
    first = idx.search(Q[:1], 10)
    first_sc = idx.scores(Q[:1])
    for _ in range(20_000):
        idx.search(Q[:1], 10)
        idx.scores(Q[:1])
    assert _same(idx.search(Q[:1], 10), first)
    assert np.array_equal(idx.scores(Q[:1]), first_sc)
    idx.close()
