"""GPU parity of the HIP scan / top-k path against the oracle (oracle/retrieval_np.py) and the
reference-generated golden fixtures.  Everything goes through the C-ABI (ctypes)."""
import os

import numpy as np
import pytest

from oracle import retrieval_np as orc

pytestmark = pytest.mark.gpu

ROUND = {"bf16": orc.bf16_round, "f16": orc.f16_round, "f32": lambda x: np.asarray(x, np.float32)}
# |fp32-accumulated dot - exact| for unit-norm rows: D * 2^-24 * sum|a_i b_i| <= ~D*6e-8*1 ; generous
ERR = 4e-6


def _mk(n, d, nq, seed=0):
    X = orc.synthetic_corpus(n, d, seed=100 + seed)
    Q = orc.synthetic_queries(nq, d, seed=200 + seed, planted=X)
    return X, Q


def _check(index_dtype, X, Q, k, env=None):
    from comorag_amd.index import DenseIndex
    # `env` names a route the library would not pick by itself ("CMR_SCAN_NO_WIDE": "1" = option scan_no_wide = 1): the
    # shipped library reads nothing from the environment, the options go through cmr_index_set_option
    idx = DenseIndex(X.shape[1], index_dtype, options={kk[4:].lower(): int(v) for kk, v in (env or {}).items()})
    idx.append(X)
    ids, sc, mn, mx = idx.search(Q, k)
    rnd = ROUND[index_dtype]
    exact = orc.exact_scores_f64(rnd(X), rnd(Q))
    ref_ids, ref_sc = orc.topk_rule(exact, k)
    assert ids.shape == ref_ids.shape
    for i in range(Q.shape[0]):
        orc.assert_topk_equivalent(ids[i], ref_ids[i], exact[i], ERR)
        np.testing.assert_allclose(sc[i], exact[i][ids[i]], atol=ERR, rtol=0)
        assert np.all(np.diff(sc[i]) <= 0), "scores not descending"
    np.testing.assert_allclose(mn, exact.min(axis=1), atol=ERR)
    np.testing.assert_allclose(mx, exact.max(axis=1), atol=ERR)
    idx.close()
    return ids, sc


@pytest.mark.parametrize("dtype", ["bf16", "f16", "f32"])
@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 64, 65, 257, 1000])
def test_small_shapes(dtype, n):
    X, Q = _mk(n, 48, 5, seed=n)
    _check(dtype, X, Q, 5)


@pytest.mark.parametrize("dtype", ["bf16", "f16", "f32"])
@pytest.mark.parametrize("n,nq,k", [(1, 1, 1), (6, 1, 5), (6, 3, 20), (33, 8, 20), (1000, 1, 20), (1024, 16, 20), (1024, 9, 128), (700, 16, 7),
                                    (257, 2, 20), (513, 1, 20)])
def test_tiny_single_launch_path_equals_general_path(dtype, n, nq, k):
    """<= 1024 rows: one single-workgroup launch (tiny_search_kernel) — must equal the oracle and, bit for bit, the
    general pack / scan / merge path (CMR_SCAN_NO_TINY=1)."""
    X, Q = _mk(n, 768 if dtype != "f32" else 200, nq, seed=n + nq)
    if n > 40:
        X[n // 2] = X[3]; Q[0] = X[3]                            # a tie, and an exact hit
    a_ids, a_sc = _check(dtype, X, Q, k)
    b_ids, b_sc = _check(dtype, X, Q, k, env={"CMR_SCAN_NO_TINY": "1"})
    assert np.array_equal(a_ids, b_ids) and np.array_equal(a_sc, b_sc)
    if n > 256:                                                    # more than 8 panels: up to four workgroups scan, the last one to arrive selects
        c_ids, c_sc = _check(dtype, X, Q, k, env={"CMR_TINY_MULTI": "0"})
        assert np.array_equal(a_ids, c_ids) and np.array_equal(a_sc, c_sc)


@pytest.mark.parametrize("dtype,n,d,nq,k", [("bf16", 1025, 768, 1, 20), ("bf16", 3000, 768, 1, 20), ("bf16", 10_000, 768, 1, 20), ("f16", 10_000, 768, 16, 20),
                                            ("bf16", 13_057, 128, 5, 20), ("bf16", 13_058, 128, 5, 21), ("f32", 20_000, 100, 3, 5), ("bf16", 40_000, 256, 2, 20),
                                            ("bf16", 65_536, 64, 1, 16), ("bf16", 65_537, 64, 1, 16), ("bf16", 30_000, 64, 9, 64), ("bf16", 8_000, 64, 2, 65),
                                            ("f32", 50_000, 1024, 1, 20), ("bf16", 2_000, 768, 1, 1), ("bf16", 100_000, 128, 1, 20),
                                            ("bf16", 196_608, 64, 3, 20), ("bf16", 196_609, 64, 3, 20), ("f16", 190_000, 96, 16, 64), ("f32", 9_000, 1000, 8, 20),
                                            ("bf16", 150_000, 64, 2, 33), ("bf16", 60_000, 64, 4, 64), ("bf16", 60_000, 64, 5, 64)])
def test_small_corpus_single_launch_hierarchical_selection(dtype, n, d, nq, k):
    """1025 rows .. 192 K rows, <= 16 queries, k <= 64: ONE launch — every workgroup scans its panels and selects the k best of
    its rows (in chunks of 960 with the running k best carried along when it has more than 1024), the last workgroup to
    arrive selects among the workgroups' candidates the same way (tiny_search_kernel).  Must equal the oracle and, bit for
    bit, the general pack / sample / scan / merge chain (CMR_SCAN_NO_SMALL=1); sizes just past the limits (196 609 rows,
    k = 65) take the general chain anyway."""
    X, Q = _mk(n, d, nq, seed=n % 997 + nq + k)
    X[n - 1] = X[5]; X[n // 2] = X[5]; X[min(n - 1, 300)] = X[5]; Q[0] = X[5]      # one row four times: ties across workgroups and the last panel
    a_ids, a_sc = _check(dtype, X, Q, k)
    b_ids, b_sc = _check(dtype, X, Q, k, env={"CMR_SCAN_NO_SMALL": "1"})
    assert np.array_equal(a_ids, b_ids) and np.array_equal(a_sc, b_sc)
    assert sorted(a_ids[0][:4].tolist()) == sorted({5, min(n - 1, 300), n // 2, n - 1}) or k < 4


@pytest.mark.parametrize("dtype,n,d,nq", [("bf16", 1, 768, 1), ("bf16", 6, 768, 1), ("bf16", 1000, 768, 3), ("f32", 1025, 200, 1), ("f16", 10_000, 768, 16),
                                          ("bf16", 33_333, 128, 2), ("bf16", 196_608, 64, 1), ("f32", 100_001, 64, 2), ("bf16", 50_000, 64, 16)])
def test_scores_single_launch_equals_general_path(dtype, n, d, nq):
    """cmr_index_scores on a small corpus (what dense_passage_retrieval / get_fact_scores call, one query at a time): one launch
    that packs, scans and writes the scores into a mapped host buffer.  Must equal, bit for bit, the general pack + scan
    (operands swapped: D[query][row]) + copy path (CMR_SCAN_NO_SMALL=1), and the oracle; a NaN query is still reported."""
    from comorag_amd import _lib
    from comorag_amd.index import DenseIndex
    X, Q = _mk(n, d, nq, seed=n % 991 + nq)
    outs = []
    for opts in ({}, {"scan_no_small": 1}):
        idx = DenseIndex(d, dtype, options=opts)
        idx.append(X)
        outs.append(idx.scores(Q))
        if not opts:
            bad = Q.copy(); bad[0, 1] = np.inf
            with pytest.raises(_lib.CmrError):
                idx.scores(bad)
            assert np.array_equal(idx.scores(Q), outs[0])
        idx.close()
    assert outs[0].shape == (nq, n) and np.array_equal(outs[0], outs[1])
    rnd = ROUND[dtype]
    np.testing.assert_allclose(outs[0], orc.exact_scores_f64(rnd(X), rnd(Q)), atol=ERR, rtol=0)


@pytest.mark.parametrize("n,nq,k", [(6, 1, 5), (1000, 3, 20), (5000, 1, 20), (140_000, 1, 20), (140_000, 8, 20), (20_000, 64, 20), (3000, 2, 100)])
def test_zero_copy_host_api_equals_copy_path(n, nq, k):
    """The synchronous host API maps queries / results / the non-finite flag from pinned host memory (no copies around the
    kernels); CMR_ZERO_COPY=0 is the one-copy-each-way path.  Both must equal the oracle and each other bit for bit; a
    NaN query must still be reported through the mapped flag, and the call after it must be clean."""
    from comorag_amd import _lib
    from comorag_amd.index import DenseIndex
    X, Q = _mk(n, 768 if n < 100_000 else 128, nq, seed=n + nq)
    a_ids, a_sc = _check("bf16", X, Q, k)
    b_ids, b_sc = _check("bf16", X, Q, k, env={"CMR_ZERO_COPY": "0"})
    assert np.array_equal(a_ids, b_ids) and np.array_equal(a_sc, b_sc)
    idx = DenseIndex(X.shape[1], "bf16")
    idx.append(X)
    bad = Q.copy(); bad[-1, 3] = np.nan
    with pytest.raises(_lib.CmrError):
        idx.search(bad, k)
    ids, sc, _, _ = idx.search(Q, k)
    assert np.array_equal(ids, a_ids) and np.array_equal(sc, a_sc)
    idx.close()


@pytest.mark.parametrize("dtype,n,d,nq", [("bf16", 140_000, 128, 1), ("bf16", 140_000, 128, 8), ("f32", 131_072 + 5, 64, 3), ("bf16", 300_000, 64, 2),
                                          ("f16", 200_000, 256, 1), ("f32", 150_001, 128, 2)])
def test_small_batch_single_level_sampling(dtype, n, d, nq):
    """<= 8 queries on >= 128 Ki rows: one sampling level of 128 panels instead of two (CMR_SAMPLE_SINGLE=0 restores the
    two-level scheme).  A sample threshold is a lower bound of the true k-th best whatever the sample: same results."""
    X, Q = _mk(n, d, nq, seed=n % 1000 + nq)
    a_ids, a_sc = _check(dtype, X, Q, 20)
    b_ids, b_sc = _check(dtype, X, Q, 20, env={"CMR_SAMPLE_SINGLE": "0"})
    assert np.array_equal(a_ids, b_ids) and np.array_equal(a_sc, b_sc)
    # the single level's thresholds are derived by the main scan's workgroups from the sample lists (default) or by a merge
    # launch between the two scans (sample_tau_in_scan = 0): any valid lower bound gives the same results; also with duplicated
    # best rows (ties at the threshold), k = 1 and k = 32, and a corpus whose last sampled panel is partial
    c_ids, c_sc = _check(dtype, X, Q, 20, env={"CMR_SAMPLE_TAU_IN_SCAN": "0"})
    assert np.array_equal(a_ids, c_ids) and np.array_equal(a_sc, c_sc)
    X2 = X.copy(); X2[5::7919] = X2[3]
    for k in (1, 32):
        d_ids, d_sc = _check(dtype, X2, Q, k)
        e_ids, e_sc = _check(dtype, X2, Q, k, env={"CMR_SAMPLE_TAU_IN_SCAN": "0"})
        assert np.array_equal(d_ids, e_ids) and np.array_equal(d_sc, e_sc)


@pytest.mark.parametrize("dtype,d", [("bf16", 8), ("bf16", 128), ("bf16", 768), ("bf16", 1024), ("f16", 1024),
                                     ("f32", 768), ("f32", 100), ("bf16", 200)])
def test_dims(dtype, d):
    X, Q = _mk(3001, d, 9, seed=d)
    _check(dtype, X, Q, 20)


@pytest.mark.parametrize("nq", [1, 31, 32, 33, 64, 65, 130])
def test_batch_sizes(nq):
    X, Q = _mk(5000, 256, nq, seed=nq)
    _check("bf16", X, Q, 20)
    if nq in (1, 33, 65):
        _check("f32", X, Q, 20)


@pytest.mark.parametrize("dtype,d,nq", [("bf16", 768, 65), ("bf16", 768, 256), ("bf16", 768, 300), ("f16", 768, 200),
                                         ("bf16", 1024, 128), ("f16", 1024, 97), ("bf16", 700, 130)])
def test_wide_batch_register_resident_queries(dtype, d, nq):
    """nq > 64 at 768-d / 1024-d: one pass of the wide kernel (queries in registers) per 256 / 128
    queries; must equal the oracle and, bit for bit, the multi-pass narrow kernel."""
    X, Q = _mk(40_000 + 17, d, nq, seed=d + nq)
    X[30_000] = X[11]; Q[3] = X[11]
    a_ids, a_sc = _check(dtype, X, Q, 20, env={"CMR_WIDE_MODE": "1"})      # the register-resident kernel, whatever the default route of this size is
    b_ids, b_sc = _check(dtype, X, Q, 20, env={"CMR_SCAN_NO_WIDE": "1"})
    assert np.array_equal(a_ids, b_ids) and np.array_equal(a_sc, b_sc)
    c_ids, c_sc = _check(dtype, X, Q, 20)                                  # default: 65 .. 128 queries over a short scan take the query-split grid
    assert np.array_equal(a_ids, c_ids) and np.array_equal(a_sc, c_sc)


@pytest.mark.parametrize("dtype,d,nq", [("bf16", 384, 256), ("f32", 768, 128), ("f32", 200, 97), ("bf16", 512, 65), ("f16", 1536, 200), ("bf16", 768, 256),
                                         ("bf16", 1024, 256), ("f16", 768, 193)])
def test_query_split_grid_of_the_narrow_kernel(dtype, d, nq):
    """Batches of more than one narrow pass on shapes WITHOUT a register-resident wide kernel (any dim, fp32 indexes) run on the
    query-split grid of the narrow kernel by default: up to four query tiles walk the same panel ranges in ONE corpus pass.
    Bit for bit against the narrow passes (scan_no_wide = 1), and against the oracle; for the shapes that do have a wide kernel
    (768-d / 1024-d, 16-bit) the grid is forced with wide_mode = 2 and held against the wide kernel as well."""
    X, Q = _mk(40_000 + 17, d, nq, seed=d + nq + 1)
    X[30_000] = X[11]; Q[3] = X[11]
    a_ids, a_sc = _check(dtype, X, Q, 20, env={"CMR_WIDE_MODE": "2"})
    b_ids, b_sc = _check(dtype, X, Q, 20, env={"CMR_SCAN_NO_WIDE": "1"})
    assert np.array_equal(a_ids, b_ids) and np.array_equal(a_sc, b_sc)
    c_ids, c_sc = _check(dtype, X, Q, 20)                        # the default route of this shape
    assert np.array_equal(a_ids, c_ids) and np.array_equal(a_sc, c_sc)
    d_ids, d_sc = _check(dtype, X, Q, 100, env={"CMR_WIDE_MODE": "2", "CMR_STREAM_NT": "1"})      # k > 32: 256-entry lists; non-temporal loads
    e_ids, e_sc = _check(dtype, X, Q, 100, env={"CMR_SCAN_NO_WIDE": "1"})
    assert np.array_equal(d_ids, e_ids) and np.array_equal(d_sc, e_sc)


def test_query_split_grid_sampling_levels_and_pipelined_mode():
    """300 K rows: both sampling levels run per query group; the pipelined entry point with the grid forced equals the
    synchronous wide kernel."""
    import torch
    from comorag_amd.index import DenseIndex
    X, Q = _mk(300_000, 768, 256, seed=6)
    a_ids, a_sc = _check("bf16", X, Q, 20, env={"CMR_WIDE_MODE": "2"})
    idx = DenseIndex(768, "bf16", options={"wide_mode": 2}); idx.append(X)
    qd = torch.from_numpy(Q).cuda()
    outs = [(torch.empty((256, 20), dtype=torch.int64, device="cuda"), torch.empty((256, 20), dtype=torch.float32, device="cuda")) for _ in range(2)]
    for i in range(4):
        h = idx.search_pipelined(qd, 20, outs[i & 1][0], outs[i & 1][1])
    idx.sync(h); torch.cuda.synchronize()
    for o in outs:
        assert np.array_equal(o[0].cpu().numpy(), a_ids) and np.array_equal(o[1].cpu().numpy(), a_sc)
    idx.close()
    b_ids, b_sc = _check("bf16", X, Q, 20, env={"CMR_WIDE_MODE": "1"})
    assert np.array_equal(a_ids, b_ids) and np.array_equal(a_sc, b_sc)


@pytest.mark.parametrize("env", [{}, {"CMR_SCAN_NO_SAMPLE": "1"}])
@pytest.mark.parametrize("n", [60_000, 9_000])
def test_wide_batch_ascending_scores_force_compaction(n, env):
    """Adversarial order for the running top-k: every query's score grows with the row index, so each panel beats
    the threshold and the per-workgroup candidate lists overflow and compact (wide kernel's wide_compact); with
    and without the sampling thresholds; also at sizes where a workgroup scans a single panel."""
    rng = np.random.default_rng(n)
    u = rng.standard_normal(768).astype(np.float32); u /= np.linalg.norm(u)
    X = 0.25 * rng.standard_normal((n, 768)).astype(np.float32) / np.sqrt(768) + np.linspace(0.0, 1.0, n, dtype=np.float32)[:, None] * u
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    Q = u[None, :] + 0.05 * rng.standard_normal((200, 768)).astype(np.float32) / np.sqrt(768)
    Q = (Q / np.linalg.norm(Q, axis=1, keepdims=True)).astype(np.float32)
    a_ids, a_sc = _check("bf16", X, Q, 20, env=env)
    b_ids, b_sc = _check("bf16", X, Q, 20, env={**env, "CMR_SCAN_NO_WIDE": "1"})
    assert np.array_equal(a_ids, b_ids) and np.array_equal(a_sc, b_sc)


@pytest.mark.parametrize("n", [1, 31, 100, 1025])
def test_wide_batch_tiny_corpora(n):
    X, Q = _mk(n, 768, 256, seed=n)
    _check("bf16", X, Q, 20)


def test_wide_batch_large_k_and_sampling():
    X, Q = _mk(300_000, 768, 256, seed=5)          # large enough for both sampling levels
    _check("bf16", X, Q, 100)


@pytest.mark.parametrize("k", [1, 5, 20, 32, 33, 100, 128])
def test_k_values(k):
    X, Q = _mk(4000, 128, 7, seed=k)
    _check("bf16", X, Q, k)


@pytest.mark.parametrize("dtype,n,k", [("bf16", 5000, 129), ("bf16", 5000, 2047), ("f32", 3000, 500), ("bf16", 4097, 4096),
                                       ("bf16", 300, 2047), ("f16", 70000, 1000)])
def test_large_k_two_pass(dtype, n, k):
    """k above CMR_MAX_K: device score block + per-row radix select / ordered compaction / bitonic sort."""
    X, Q = _mk(n, 64, 7, seed=n + k)
    X[n // 2] = X[3]; X[n - 1] = X[3]                 # ties straddling the selection threshold region
    Q[0] = X[3]
    _check(dtype, X, Q, k)


def test_large_k_all_equal_rows():
    from comorag_amd.index import DenseIndex
    X = np.tile(orc.synthetic_corpus(1, 32, seed=1), (1000, 1))
    idx = DenseIndex(32, "bf16"); idx.append(X)
    ids, sc, mn, mx = idx.search(X[:2], 300)
    assert ids[0].tolist() == list(range(300)) and ids[1].tolist() == list(range(300)) and mn[0] == mx[0]
    idx.close()


@pytest.mark.parametrize("env", [{"CMR_SCAN_ASM_RING": "0"}, {"CMR_SCAN_ASM_RING": "0", "CMR_SCAN_RING": "8"},
                                 {"CMR_SCAN_RING": "8"}, {"CMR_SCAN_RING": "16"}, {"CMR_SCAN_GRID": "3"}])
def test_ring_variants_agree(env):
    X, Q = _mk(70001, 768, 64, seed=7)
    a_ids, a_sc = _check("bf16", X, Q, 20)
    b_ids, b_sc = _check("bf16", X, Q, 20, env=env)
    # same arithmetic in every variant (same MFMA order) -> bitwise equal
    assert np.array_equal(a_ids, b_ids) and np.array_equal(a_sc, b_sc)


@pytest.mark.parametrize("dtype,nq", [("bf16", 1), ("bf16", 8), ("f32", 5), ("f16", 32)])
def test_one_tile_kernels_hand_counted_ring_equals_compiler_counted(dtype, nq):
    """<= 32 queries run the one-tile kernels; their hand-counted load ring only exists since the accumulator rides along with the
    reloads (LLVM sank the single MFMA of a slot below them and read the slot through a copy made before its wait).
    General chain forced (no single-launch path, no finishing stage), both ring builds, top-k and all-scores modes."""
    from comorag_amd.index import DenseIndex
    X, Q = _mk(90_001, 384, nq, seed=17 + nq)
    outs = []
    for asm in (1, 0):
        idx = DenseIndex(384, dtype, options={"scan_no_tiny": 1, "scan_fin": 0, "scan_asm_ring": asm, "zero_copy": 0})
        idx.append(X)
        outs.append((idx.search(Q, 20), idx.scores(Q)))
        idx.close()
    assert all(np.array_equal(a, b) for a, b in zip(outs[0][0], outs[1][0]))
    assert np.array_equal(outs[0][1], outs[1][1])
    rnd = ROUND[dtype]
    exact = orc.exact_scores_f64(rnd(X), rnd(Q))
    np.testing.assert_allclose(outs[0][1], exact, atol=ERR, rtol=0)


def test_ties_index_ascending():
    from comorag_amd.index import DenseIndex
    X = orc.synthetic_corpus(500, 64, seed=5)
    X[100] = X[7]; X[300] = X[7]; X[499] = X[7]
    q = X[7:8].copy()
    for dtype in ("bf16", "f32"):
        idx = DenseIndex(64, dtype); idx.append(X)
        ids, sc, _, _ = idx.search(q, 6)
        assert ids[0, :4].tolist() == [7, 100, 300, 499], ids
        assert sc[0, 0] == sc[0, 1] == sc[0, 2] == sc[0, 3]
        idx.close()
    # all-equal scores: every row identical -> first k rows in index order
    Z = np.tile(X[3:4], (200, 1))
    idx = DenseIndex(64, "bf16"); idx.append(Z)
    ids, sc, mn, mx = idx.search(q, 10)
    assert ids[0].tolist() == list(range(10)) and mn[0] == mx[0]
    idx.close()


def test_k_larger_than_n_and_empty():
    from comorag_amd.index import DenseIndex
    X, Q = _mk(7, 32, 3)
    idx = DenseIndex(32, "bf16"); idx.append(X)
    ids, sc, mn, mx = idx.search(Q, 20)
    assert ids.shape == (3, 7) and len(set(ids[0].tolist())) == 7
    e = DenseIndex(32, "bf16")
    ids, sc, mn, mx = e.search(Q, 5)
    assert ids.shape == (3, 0)
    idx.close(); e.close()


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_incremental_append_equals_bulk(dtype):
    from comorag_amd.index import DenseIndex
    X, Q = _mk(3000, 96, 8, seed=3)
    a = DenseIndex(96, dtype, capacity_hint=16); b = DenseIndex(96, dtype)
    b.append(X)
    for lo, hi in ((0, 1), (1, 2), (2, 33), (33, 64), (64, 65), (65, 1000), (1000, 1025), (1025, 3000)):
        a.append(X[lo:hi])       # forces several capacity doublings (hipMemcpyAsync grow)
    assert len(a) == len(b) == 3000
    ia, sa, mna, mxa = a.search(Q, 20); ib, sb, mnb, mxb = b.search(Q, 20)
    assert np.array_equal(ia, ib) and np.array_equal(sa, sb) and np.array_equal(mna, mnb) and np.array_equal(mxa, mxb)
    rows = a.get_rows([0, 31, 32, 2999])
    np.testing.assert_array_equal(rows, ROUND[dtype](X[[0, 31, 32, 2999]]))
    a.close(); b.close()


@pytest.mark.parametrize("dtype", ["bf16", "f16", "f32"])
def test_full_scores(dtype):
    from comorag_amd.index import DenseIndex
    X, Q = _mk(2077, 192, 37, seed=11)
    idx = DenseIndex(192, dtype); idx.append(X)
    s = idx.scores(Q)
    exact = orc.exact_scores_f64(ROUND[dtype](X), ROUND[dtype](Q))
    np.testing.assert_allclose(s, exact, atol=ERR, rtol=0)
    # top-k of the full-score output equals the fused top-k (same arithmetic order? no: operand
    # roles are swapped but each score is the same k-ordered fp32 chain) -> ids must agree
    ids, sc, _, _ = idx.search(Q, 10)
    rid, _ = orc.topk_rule(s, 10)
    for i in range(len(Q)):
        orc.assert_topk_equivalent(ids[i], rid[i], exact[i], ERR)
    idx.close()


@pytest.mark.parametrize("dtype,n", [("bf16", 1), ("bf16", 2047), ("bf16", 2049), ("f32", 70001), ("bf16", 300017)])
def test_sorted_scores_full_ranking(dtype, n):
    """cmr_index_sorted_scores: all N rows, score descending, ties by ascending row — bit-identical
    to sorting the library's own full-score vector with the exported rule."""
    from comorag_amd.index import DenseIndex
    d = 64
    X, Q = _mk(n, d, 3, seed=21)
    if n > 100:                         # exact ties: duplicate rows far apart, plus a block of identical rows
        X[n // 2] = X[7]; X[n - 1] = X[7]; X[40:60] = X[40]
    idx = DenseIndex(d, dtype); idx.append(X)
    s = idx.scores(Q)
    ids, sc, mn, mx = idx.sorted_scores(Q)
    assert ids.shape == (3, n) and sc.shape == (3, n)
    for i in range(3):
        order = np.lexsort((np.arange(n), -s[i].astype(np.float64)))      # score desc, row asc
        np.testing.assert_array_equal(ids[i], order)
        np.testing.assert_array_equal(sc[i], s[i][order])
        assert mx[i] == s[i].max() and mn[i] == s[i].min()
    exact = orc.exact_scores_f64(ROUND[dtype](X), ROUND[dtype](Q))
    np.testing.assert_allclose(sc, np.take_along_axis(exact, ids, 1), atol=ERR, rtol=0)
    idx.close()


def test_dense_passage_retrieval_device_sort_matches_reference_lines():
    """retrieval.dense_passage_retrieval above DEVICE_SORT_MIN_ROWS (device sort) returns what the
    reference's normalise + argsort lines give on the same scores (ComoRAG.py:950-967), ties aside."""
    from comorag_amd import retrieval
    from comorag_amd.index import DenseIndex
    n, d = 40009, 96
    assert n >= retrieval.DEVICE_SORT_MIN_ROWS
    X, Q = _mk(n, d, 1, seed=22)
    idx = DenseIndex(d, "f32"); idx.append(X)
    ids, sc = retrieval.dense_passage_retrieval(idx, Q[0])
    ref_ids, ref_sc = orc.dense_passage_retrieval(X, Q[0])
    assert ids.shape == (n,) and ids.dtype == np.int64
    np.testing.assert_allclose(sc, ref_sc, atol=2e-6, rtol=0)
    exact = orc.exact_scores_f64(X, Q)[0]
    orc.assert_topk_equivalent(ids, ref_ids, exact, 1e-6)
    assert sc[0] == 1.0 and sc[-1] == 0.0 and np.all(np.diff(sc) <= 0)
    assert sorted(ids.tolist()) == list(range(n))
    idx.close()


def test_nonfinite_rejected():
    from comorag_amd.index import DenseIndex
    from comorag_amd._lib import CmrError, CMR_ERR_NONFINITE
    X, Q = _mk(100, 32, 2)
    idx = DenseIndex(32, "bf16"); idx.append(X)
    bad = X[:3].copy(); bad[1, 5] = np.nan
    with pytest.raises(CmrError) as ei:
        idx.append(bad)
    assert ei.value.code == CMR_ERR_NONFINITE and len(idx) == 100
    qb = Q.copy(); qb[0, 0] = np.inf
    with pytest.raises(CmrError):
        idx.search(qb, 3)
    ids, _, _, _ = idx.search(Q, 3)   # still usable
    assert ids.shape == (2, 3)
    idx.close()


def test_k100_on_five_million_rows():
    """k in 33..128 on > 4.19 M rows: the level-1 sample used to produce more candidate lists than merge_query_kernel
    accepts (W > 4096 -> CMR_ERR_HIP).  5 M x 64 bf16 rows, k = 100 and 128, narrow and pipelined paths."""
    import torch
    from comorag_amd.index import DenseIndex
    n, d = 5_000_000, 64
    X = orc.synthetic_corpus(n, d, seed=55)
    Q = orc.synthetic_queries(6, d, seed=56, planted=X[::500_000])
    idx = DenseIndex(d, "bf16", capacity_hint=n); idx.append(X)
    Xr, Qr = orc.bf16_round(X), orc.bf16_round(Q)
    exact = Qr.astype(np.float64) @ Xr.astype(np.float64).T
    for k in (100, 128, 33):
        ids, sc, mn, mx = idx.search(Q, k)
        ref_ids, _ = orc.topk_rule(exact, k)
        for i in range(len(Q)):
            orc.assert_topk_equivalent(ids[i], ref_ids[i], exact[i], ERR)
    dev = torch.device("cuda", 0)
    qt = torch.from_numpy(Q).to(dev)
    oi = torch.empty((6, 100), dtype=torch.int64, device=dev); os_ = torch.empty((6, 100), dtype=torch.float32, device=dev)
    idx.sync(idx.search_pipelined(qt, 100, oi, os_))
    a_ids, a_sc = idx.search(Q, 100)[:2]
    assert np.array_equal(oi.cpu().numpy(), a_ids) and np.array_equal(os_.cpu().numpy(), a_sc)
    idx.close()


def test_dev_api_on_torch_default_stream_is_ordered():
    """search_dev / scores_dev with torch's default stream (handle 0): the queries are PRODUCED by torch kernels on that
    stream and the outputs CONSUMED by torch kernels on it, with no device synchronisation in between — the library must
    enqueue on the stream it was handed, not on a private one."""
    import torch
    from comorag_amd.index import DenseIndex
    X, Q = _mk(200_000, 256, 16, seed=3)
    idx = DenseIndex(256, "bf16"); idx.append(X)
    want_ids, want_sc = idx.search(Q, 10)[:2]
    want_full = idx.scores(Q[:2])
    dev = torch.device("cuda", 0)
    qh = torch.from_numpy(Q)
    big = torch.randn((4096, 4096), device=dev)
    for rep in range(5):
        assert torch.cuda.current_stream(dev).cuda_stream == 0
        junk = big @ big                                          # keeps the stream busy in front of the query producer
        q_t = (qh.to(dev, non_blocking=True) * 2.0) / 2.0         # produced on the default stream, after `junk`
        ids_t, sc_t = idx.search_dev(q_t, 10)
        ids_c = ids_t.clone(); sc_c = sc_t * 1.0                  # consumers on the same stream
        full = idx.scores_dev(q_t[:2].contiguous())
        full_c = full + 0.0
        del junk
        assert np.array_equal(ids_c.cpu().numpy(), want_ids) and np.array_equal(sc_c.cpu().numpy(), want_sc)
        assert np.array_equal(full_c.cpu().numpy(), want_full)
    idx.close()


def test_rescore_and_get_rows_take_global_ids():
    """A row shard with an id base: search returns global ids; rescore / get_rows must accept exactly those."""
    from comorag_amd.index import DenseIndex
    from comorag_amd.rerank import search_then_rescore
    X, Q = _mk(5000, 128, 3, seed=31)
    a = DenseIndex(128, "bf16", keep_f32=True); a.append(X)
    b = DenseIndex(128, "bf16", keep_f32=True); b.append(X); b.set_id_base(1_000_000)
    ia, sa = a.search(Q, 50)[:2]
    ib, sb = b.search(Q, 50)[:2]
    assert np.array_equal(ib, ia + 1_000_000) and np.array_equal(sa, sb)
    ra, rsa = a.rescore(Q, ia, 10)
    rb, rsb = b.rescore(Q, ib, 10)
    assert np.array_equal(rb, ra + 1_000_000) and np.array_equal(rsa, rsb) and np.all(ra >= 0)
    assert np.array_equal(b.get_rows(ib[0, :5]), a.get_rows(ia[0, :5]))
    ja, _ = search_then_rescore(a, Q, 50, 10)
    jb, _ = search_then_rescore(b, Q, 50, 10)
    assert np.array_equal(jb, ja + 1_000_000)
    a.close(); b.close()


def test_query_status_reports_nonfinite_dev_queries():
    import torch
    from comorag_amd.index import DenseIndex
    X, Q = _mk(3000, 64, 4, seed=41)
    idx = DenseIndex(64, "bf16"); idx.append(X)
    dev = torch.device("cuda", 0)
    q = torch.from_numpy(Q).to(dev)
    idx.search_dev(q, 5); torch.cuda.synchronize()
    assert idx.query_status() is False
    qb = q.clone(); qb[1, 3] = float("nan")
    idx.search_dev(qb, 5); torch.cuda.synchronize()
    assert idx.query_status() is True and idx.query_status() is False        # reported once, then re-armed
    oi = torch.empty((4, 5), dtype=torch.int64, device=dev); os_ = torch.empty((4, 5), dtype=torch.float32, device=dev)
    qb[1, 3] = float("inf")
    idx.sync(idx.search_pipelined(qb, 5, oi, os_))
    assert idx.query_status() is True
    idx.sync(idx.search_pipelined(q, 5, oi, os_))
    assert idx.query_status() is False
    idx.close()


def test_golden_dpr_fp32(golden_dir):
    """fp32 index vs the REFERENCE's dense_passage_retrieval / get_fact_scores outputs."""
    from comorag_amd.index import DenseIndex
    for tag in ("small", "mid", "d768", "n2"):
        g = np.load(os.path.join(golden_dir, f"dpr_{tag}.npz"))
        X, F, Q = g["X"], g["F"], g["Q"]
        idx = DenseIndex(X.shape[1], "f32"); idx.append(X)
        fidx = DenseIndex(F.shape[1], "f32"); fidx.append(F)
        exact = orc.exact_scores_f64(X, Q)
        k = min(20, len(X))
        ids, sc, mn, mx = idx.search(Q, k)
        full = fidx.scores(Q)
        for i in range(len(Q)):
            ref_ids, ref_sc = g[f"dpr_ids_{i}"], g[f"dpr_scores_{i}"]
            orc.assert_topk_equivalent(ids[i], ref_ids[:k], exact[i], 1e-6)
            norm = (sc[i] - mn[i]) / (mx[i] - mn[i])        # misc_utils.py:141-150 from out_min/out_max
            np.testing.assert_allclose(norm, ref_sc[:k], atol=2e-6)
            fs = orc.min_max_normalize(full[i])
            np.testing.assert_allclose(fs, g[f"fact_scores_{i}"], atol=2e-6)
        idx.close(); fidx.close()


def test_rescore_exact():
    from comorag_amd.index import DenseIndex
    X, Q = _mk(3000, 1024, 4, seed=21)
    idx = DenseIndex(1024, "f16", keep_f32=True); idx.append(X)
    ids, _, _, _ = idx.search(Q, 100)
    rid, rsc = idx.rescore(Q, ids, 20)
    exact = orc.exact_scores_f64(X, Q)           # fp32 shadow: un-rounded rows, fp32 queries
    for i in range(len(Q)):
        cand = ids[i]
        order = cand[np.lexsort((cand, -exact[i][cand]))][:20]
        orc.assert_topk_equivalent(rid[i], order, exact[i], 1e-6)
        np.testing.assert_allclose(rsc[i], exact[i][rid[i]], atol=1e-6)
    idx.close()


def test_threads_concurrent_search():
    """16 Python threads on one index (ComoRAG.try_answer's pool, ComoRAG.py:436-441)."""
    from concurrent.futures import ThreadPoolExecutor
    from comorag_amd.index import DenseIndex
    X, Q = _mk(20000, 128, 32, seed=9)
    idx = DenseIndex(128, "bf16"); idx.append(X)
    want = idx.search(Q, 10)[0]
    def work(i):
        return idx.search(Q[i:i + 2], 10)[0]
    with ThreadPoolExecutor(16) as ex:
        outs = list(ex.map(work, list(range(0, 32, 2)) * 4))
    for j, o in enumerate(outs):
        i = (j % 16) * 2
        assert np.array_equal(o, want[i:i + 2])
    idx.close()


def test_concurrent_append_and_search():
    """Writers and readers on one index (cmr_index's shared_mutex: searches shared, append exclusive — ComoRAG's 16
    question threads search while insert_strings appends).  Appends force several capacity doublings while
    8 threads search; every answer must be the exact top-k of SOME prefix of the rows (the index is never seen
    half-appended), planted rows that exist from the start always come first, and the end state equals a bulk build."""
    import threading
    from comorag_amd.index import DenseIndex
    d, k = 128, 10
    X = orc.synthetic_corpus(60_000, d, seed=71)
    Q = orc.synthetic_queries(6, d, seed=72)
    Q[0] = X[5]; Q[1] = X[777]                                    # present from the first chunk on
    Xr, Qr = orc.bf16_round(X), orc.bf16_round(Q)
    exact = Qr.astype(np.float64) @ Xr.astype(np.float64).T
    chunks = [(0, 2000)] + [(a, min(a + 3500, len(X))) for a in range(2000, len(X), 3500)]
    bounds = [b for _, b in chunks]
    idx = DenseIndex(d, "bf16", capacity_hint=16)                 # tiny hint: the matrix is reallocated again and again
    idx.append(X[:2000])
    stop, errors, seen = threading.Event(), [], []

    def reader(t):
        try:
            while not stop.is_set():
                n0 = len(idx)
                ids, sc, mn, mx = idx.search(Q, k)
                n1 = len(idx)
                ok = False
                for n in [b for b in bounds if n0 <= b <= n1]:   # the state searched is one of the committed prefixes
                    ref_ids, _ = orc.topk_rule(exact[:, :n], k)
                    try:
                        for i in range(len(Q)):
                            orc.assert_topk_equivalent(ids[i], ref_ids[i], exact[i], ERR)
                        ok = True
                        break
                    except AssertionError:
                        continue
                if not ok:
                    errors.append((t, n0, n1, ids[:, :3].tolist()))
                if ids[0, 0] != 5 or ids[1, 0] != 777:
                    errors.append((t, "planted", ids[:2, 0].tolist()))
                seen.append(n1)
        except Exception as e:              # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=reader, args=(t,)) for t in range(8)]
    for th in threads:
        th.start()
    for a, b in chunks[1:]:
        idx.append(X[a:b])
    stop.set()
    for th in threads:
        th.join()
    assert not errors, errors[:3]
    assert len(idx) == len(X) and len(set(seen)) >= 3            # the readers really overlapped the appends
    ids, sc, _, _ = idx.search(Q, k)
    bulk = DenseIndex(d, "bf16"); bulk.append(X)
    bi, bs, _, _ = bulk.search(Q, k)
    assert np.array_equal(ids, bi) and np.array_equal(sc, bs)
    idx.close(); bulk.close()


def test_c2_size_properties():
    """BASELINE config 2 size: 1M x 768 bf16, B=64, k=20 — oracle on the full size (numpy fp32 BLAS,
    fp64 arbitration on disagreements) + planted rows + shard-merge equivalence."""
    from comorag_amd.index import DenseIndex, merge_topk
    n, d, b, k = 1_000_000, 768, 64, 20
    X = np.concatenate([orc.synthetic_corpus(250_000, d, seed=1234, block=i) for i in range(4)])
    Q = orc.synthetic_queries(b, d, seed=77, planted=X[::1000])
    Q[-1] = X[999_999]; Q[-2] = X[0]                        # exact rows: must come back first
    idx = DenseIndex(d, "bf16", capacity_hint=n); idx.append(X)
    ids, sc, mn, mx = idx.search(Q, k)
    assert ids[-1, 0] == 999_999 and ids[-2, 0] == 0
    Xr, Qr = orc.bf16_round(X), orc.bf16_round(Q)
    s32 = Qr @ Xr.T
    ref_ids, _ = orc.topk_rule(s32, k)
    for i in range(b):
        if not np.array_equal(ids[i], ref_ids[i]):
            cols = np.union1d(ids[i], ref_ids[i])
            ex = np.full(n, -np.inf); ex[cols] = Qr[i].astype(np.float64) @ Xr[cols].astype(np.float64).T
            orc.assert_topk_equivalent(ids[i], ref_ids[i], ex, ERR)
        np.testing.assert_allclose(sc[i], s32[i][ids[i]], atol=ERR)
    np.testing.assert_allclose(mx, s32.max(axis=1), atol=ERR)
    np.testing.assert_allclose(mn, s32.min(axis=1), atol=ERR)
    # |cos_bf16 - cos_fp32| <= 1e-3 (north_star tolerance) and recall@20 vs the fp32 reference ranking
    s_fp32 = Q @ X.T
    assert np.max(np.abs(np.take_along_axis(s_fp32, ids, 1) - sc)) <= 1e-3
    ref32, _ = orc.topk_rule(s_fp32, k)
    recall = np.mean([len(set(ids[i]) & set(ref32[i])) / k for i in range(b)])
    assert recall >= 0.95, recall
    # two logical shards merged == one index
    h = n // 2
    a = DenseIndex(d, "bf16", capacity_hint=h); a.append(X[:h])
    c = DenseIndex(d, "bf16", capacity_hint=n - h); c.append(X[h:])
    ia, sa, _, _ = a.search(Q, k); ic, sc2, _, _ = c.search(Q, k)
    mi, ms = merge_topk(np.stack([ia, ic + h]), np.stack([sa, sc2]))
    assert np.array_equal(mi, ids) and np.array_equal(ms, sc)
    idx.close(); a.close(); c.close()
