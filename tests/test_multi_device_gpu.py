"""MultiDeviceIndex (cmr_mindex_t: row shards over the node's GPUs driven from ONE process) against ONE DenseIndex holding
the same rows — bit for bit, through every entry point of the drop-in surface.  A 1-GPU box rehearses the layout with S
logical shards on device 0 (SURVEY.md §4 "multi-shard": S in {1,2,4,8} logical shards on one device == S = 1).

What a caller of the reference sees through it: ComoRAG.dense_passage_retrieval / get_fact_scores (ComoRAG.py:937-967) via
hooks.install, get_similar_summaries (utils/embed_utils.py:109-161) via EmbeddingStore.device_index, MemoryPool's
incremental appends (utils/memory_utils.py:188-235, 294-300), from up to 16 threads (ComoRAG.py:432-453)."""
import os
import threading
import types

import numpy as np
import pytest

from oracle import retrieval_np as orc

pytestmark = pytest.mark.gpu


def _pair(X, dtype, S, block_rows=1024, **kw):
    from comorag_amd.index import DenseIndex
    from comorag_amd.multi_index import MultiDeviceIndex
    one = DenseIndex(X.shape[1], dtype, **kw)
    # (the default block of 65536 rows would keep these test corpora on one shard)
    multi = MultiDeviceIndex(X.shape[1], dtype, devices=[0] * S, options={"append_block_rows": block_rows}, **kw)
    if len(X):
        one.append(X)
        multi.append(X)
    return one, multi


def _same_search(one, multi, Q, k):
    a = one.search(Q, k)
    b = multi.search(Q, k)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    return a


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("S", [1, 2, 4, 8])
def test_every_entry_point_equals_one_index(S, dtype):
    d = 768 if dtype == "bf16" else 200
    X = orc.synthetic_corpus(40_017, d, seed=51)
    Q = orc.synthetic_queries(70, d, seed=52, planted=X)
    X[39_000] = X[5]; X[20_000] = X[5]                    # ties across shards
    Q[0] = X[5]
    one, multi = _pair(X, dtype, S, keep_f32=(dtype == "bf16"))
    rows = multi.shard_rows()
    assert sum(rows) == len(X) == len(multi) and max(rows) - min(rows) <= S                  # a bulk append: contiguous, equal blocks
    for nq, k in ((1, 1), (1, 20), (8, 20), (64, 20), (70, 100), (3, 128)):
        _same_search(one, multi, Q[:nq], k)
    a = one.search(Q[:5], 2047); b = multi.search(Q[:5], 2047)                               # the large-k route (retrieve_knn's k)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.array_equal(one.scores(Q[:3]), multi.scores(Q[:3]))
    for x, y in zip(one.sorted_scores(Q[:2]), multi.sorted_scores(Q[:2])):
        assert np.array_equal(x, y)
    for x, y in zip(one.search_min_score(Q[:9], 50, 0.2), multi.search_min_score(Q[:9], 50, 0.2)):
        assert np.array_equal(x, y)
    cand = one.search(Q[:4], 100)[0]
    cand[1, 7] = -1; cand[2, 3] = len(X) + 5                                                   # a skipped and a foreign id
    for x, y in zip(one.rescore(Q[:4], cand, 20), multi.rescore(Q[:4], cand, 20)):
        assert np.array_equal(x, y)
    ids = np.array([0, 5, 20_000, len(X) - 1, 12_345, len(X) + 3, -1], np.int64)
    assert np.array_equal(one.get_rows(ids), multi.get_rows(ids))
    one.close(); multi.close()


@pytest.mark.parametrize("S", [2, 3, 8])
def test_incremental_appends_keep_global_ids_dense(S):
    """BASELINE config 4 on a sharded index: a bulk build, then many small appends (a memory pool's 25 rows per cycle, a few
    bigger ones); blocks of 64 rows here so that every shard collects several runs of global ids."""
    d = 256
    X = orc.synthetic_corpus(9_000, d, seed=61)
    Q = orc.synthetic_queries(16, d, seed=62, planted=X)
    one, multi = _pair(X[:5_000], "bf16", S, block_rows=64)
    assert min(multi.shard_rows()) > 0
    at = 5_000
    rng = np.random.default_rng(3)
    step = 0
    while at < len(X):
        n = int(rng.choice([1, 25, 25, 25, 64, 130, 700]))
        n = min(n, len(X) - at)
        one.append(X[at:at + n]); multi.append(X[at:at + n])
        at += n
        step += 1
        if step % 7 == 0 or at == len(X):
            Q[1] = X[at - 1]                                   # the newest row must come back first, under its append-order id
            ids = _same_search(one, multi, Q, 20)[0]
            assert ids[1, 0] == at - 1
    rows = multi.shard_rows()
    assert sum(rows) == len(X) and max(rows) - min(rows) <= 700 + 64
    assert np.array_equal(one.scores(Q[:2]), multi.scores(Q[:2]))
    for x, y in zip(one.sorted_scores(Q[:1]), multi.sorted_scores(Q[:1])):
        assert np.array_equal(x, y)
    ids = np.arange(4_990, 5_300, 7, dtype=np.int64)
    assert np.array_equal(one.get_rows(ids), multi.get_rows(ids))
    cand = one.search(Q[:3], 64)[0]
    for x, y in zip(one.rescore(Q[:3], cand, 10), multi.rescore(Q[:3], cand, 10)):
        assert np.array_equal(x, y)
    one.close(); multi.close()


def test_failed_append_rolls_every_shard_back():
    from comorag_amd._lib import CMR_ERR_NONFINITE, CmrError
    d = 128
    X = orc.synthetic_corpus(3_000, d, seed=71)
    Q = orc.synthetic_queries(4, d, seed=72, planted=X)
    one, multi = _pair(X[:1_000], "f32", 4)
    multi.set_option("append_block_rows", 100)
    before = multi.search(Q, 10)
    bad = X[1_000:1_900].copy()
    bad[650, 3] = np.nan                                      # lands in a later chunk: earlier chunks are already on their shards
    with pytest.raises(CmrError) as e:
        multi.append(bad)
    assert e.value.code == CMR_ERR_NONFINITE
    assert len(multi) == 1_000 and sum(multi.shard_rows()) == 1_000
    after = multi.search(Q, 10)
    for x, y in zip(before, after):
        assert np.array_equal(x, y)
    one.append(X[1_000:]); multi.append(X[1_000:])            # the index is as usable as before
    _same_search(one, multi, Q, 10)
    one.close(); multi.close()


def test_sixteen_threads_search_while_one_appends():
    """ComoRAG's thread pool (ComoRAG.py:436-441) on one sharded index: 16 threads search (each answer checked against a
    single index holding the rows that existed when the batch of appends began: results may only differ by rows appended
    meanwhile, which carry ids >= that size), one thread appends."""
    d = 256
    X = orc.synthetic_corpus(30_000, d, seed=81)
    Q = orc.synthetic_queries(64, d, seed=82, planted=X[:20_000])
    one, multi = _pair(X[:20_000], "bf16", 4)
    multi.set_option("append_block_rows", 256)
    want = [one.search(Q[i:i + 1 + i % 3], 10) for i in range(0, 60)]
    errs = []

    def searcher(t):
        try:
            for rep in range(25):
                i = (t * 7 + rep) % 60
                ids, sc, mn, mx = multi.search(Q[i:i + 1 + i % 3], 10)
                w = want[i]
                old = ids < 20_000
                # rows that were there from the start keep their order and scores; newcomers only ever push in
                for r in range(ids.shape[0]):
                    got = ids[r][old[r]]
                    assert np.array_equal(got, w[0][r][:len(got)]), (i, r)
                assert np.all(mx >= w[3]) and np.all(mn <= w[2])
        except Exception as ex:      # noqa: BLE001
            errs.append(repr(ex))

    def appender():
        try:
            for at in range(20_000, 30_000, 125):
                multi.append(X[at:at + 125])
        except Exception as ex:      # noqa: BLE001
            errs.append(repr(ex))

    th = [threading.Thread(target=searcher, args=(t,)) for t in range(16)] + [threading.Thread(target=appender)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs[:3]
    one.append(X[20_000:])
    _same_search(one, multi, Q, 20)
    one.close(); multi.close()


@pytest.mark.parametrize("B", [64, 256])
def test_pipelined_batches_equal_synchronous_search(B):
    from comorag_amd._lib import CmrError
    d = 768
    X = orc.synthetic_corpus(150_000, d, seed=91)
    one, multi = _pair(X, "bf16", 4)
    batches = [orc.synthetic_queries(B, d, seed=100 + j, planted=X) for j in range(6)]
    placed = [multi.place_queries(q) for q in batches]
    tickets = []
    got = []
    for j, p in enumerate(placed):
        tickets.append(multi.search_pipelined(p, 20))
        if len(tickets) == 3:
            got.append(multi.collect(tickets.pop(0), with_minmax=True))
    while tickets:
        got.append(multi.collect(tickets.pop(0), with_minmax=True))
    for q, g in zip(batches, got):
        w = one.search(q, 20)
        for x, y in zip(w, g):
            assert np.array_equal(x, y)
    # the ticket ring holds four uncollected batches
    ts = [multi.search_pipelined(placed[0], 20) for _ in range(4)]
    with pytest.raises(CmrError):
        multi.search_pipelined(placed[0], 20)
    for t in ts:
        multi.collect(t)
    one.close(); multi.close()


def test_hooks_store_and_memory_pool_with_num_shards(golden_dir, tmp_path, fake_embedder):
    """The drop-in layer builds MultiDeviceIndexes when global_config.num_shards says so, and every caller-visible result
    equals the one-shard run's: hooks.install (ComoRAG.py:876-967), EmbeddingStore.device_index + get_similar_summaries
    (utils/embed_utils.py:109-161), install_memory_pool (utils/memory_utils.py:188-235)."""
    import sys
    from comorag_amd import hooks, retrieval
    from comorag_amd.embedding_store import EmbeddingStore
    from comorag_amd.multi_index import MultiDeviceIndex
    g = np.load(os.path.join(golden_dir, "dpr_mid.npz"))
    X, F, S_, Q = g["X"], g["F"], g["S"], g["Q"]

    class Enc:
        def batch_encode(self, text, **kw):
            return Q[int(text[1:]):int(text[1:]) + 1]

    def rag_with(shards):
        class Rag:
            def __init__(self):
                self.global_config = types.SimpleNamespace(need_cluster=True, index_dtype="f32", num_shards=shards, devices=[0] * shards,
                                                           index_options={"append_block_rows": 16})
                self.embedding_model = Enc()
                self.ready_to_retrieve = False
            def prepare_retrieval_objects(self):
                self.query_to_embedding = {"triple": {}, "passage": {}}
                self.passage_embeddings, self.fact_embeddings, self.summary_embeddings = X, F, S_
                self.ready_to_retrieve = True
        sys.modules[Rag.__module__].get_query_instruction = lambda k: k
        r = hooks.install(Rag(), patch_module_functions=False)
        r.prepare_retrieval_objects()
        return r

    base = rag_with(1)
    for shards in (2, 4, 8):
        rag = rag_with(shards)
        assert isinstance(rag._hip["passage"], MultiDeviceIndex) and rag._hip["passage"].n_shards == shards
        assert min(rag._hip["passage"].shard_rows()) > 0
        for i in range(len(Q)):
            a, b = rag.dense_passage_retrieval(f"q{i}"), base.dense_passage_retrieval(f"q{i}")
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
            assert np.array_equal(rag.get_fact_scores(f"q{i}"), base.get_fact_scores(f"q{i}"))
            a, b = rag.dense_passage_retrieval(f"q{i}", need_cluster=True), base.dense_passage_retrieval(f"q{i}", need_cluster=True)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # a store's device mirror, sharded, keeps following insert_strings
    texts = [f"window {i}: once upon a time {i * 7}" for i in range(300)]
    outs = []
    for shards in (1, 4):
        fake_embedder.global_config = types.SimpleNamespace(index_dtype="f32", num_shards=shards, devices=[0] * shards, index_options={"append_block_rows": 16})
        st = EmbeddingStore(fake_embedder, str(tmp_path / f"s{shards}"), 8, "level_0")
        st.insert_strings(texts[:200])
        ix = st.device_index()
        st.insert_strings(texts[200:])
        assert len(ix) == 300 and (shards == 1 or isinstance(ix, MultiDeviceIndex))
        outs.append(retrieval.get_similar_summaries("where is the glass slipper?", st, fake_embedder, top_k=7))
    assert outs[0] == outs[1]
    # a memory pool's appendable index
    class Node:
        def __init__(self, v): self.embedding = v
    class Pool:
        def __init__(self): self.pool, self.embedding_model = [], fake_embedder
        def compute_probe_note_embeddings(self): pass
    picks = []
    for shards in (1, 3):
        pool = hooks.install_memory_pool(Pool(), index_dtype="f32", num_shards=shards, devices=[0] * shards)
        if shards > 1:
            pass
        sel = []
        for cycle in range(5):
            pool.pool.extend(Node(fake_embedder._vec(f"note {cycle} {j}")) for j in range(25))
            if shards > 1 and cycle == 0:
                pool.retrieve_similar_nodes("probe 0")
                pool._hip_state["index"].set_option("append_block_rows", 8)
            sel.append([pool.pool.index(n) for n in pool.retrieve_similar_nodes(f"probe {cycle}", 0.3)])
        picks.append(sel)
    assert picks[0] == picks[1]


def test_ppr_on_a_sharded_passage_index_equals_the_fused_path():
    from comorag_amd.index import DenseIndex
    from comorag_amd.multi_index import MultiDeviceIndex
    from comorag_amd.ppr import DeviceGraph, ppr_passage_scores
    rng = np.random.default_rng(5)
    n_p, n_e, d = 3_000, 700, 128
    X = orc.synthetic_corpus(n_p, d, seed=95)
    q = orc.synthetic_queries(1, d, seed=96, planted=X)[0]
    src = rng.integers(0, n_p + n_e, 12_000); dst = rng.integers(0, n_p + n_e, 12_000)
    keep = src != dst
    g = DeviceGraph(n_p + n_e, src[keep], dst[keep], rng.random(keep.sum()) + 0.1)
    g.set_passage_vertices(np.arange(n_e, n_e + n_p))
    pw = np.zeros(n_p + n_e); pw[rng.integers(0, n_e, 5)] = rng.random(5)
    one = DenseIndex(d, "f32"); one.append(X)
    multi = MultiDeviceIndex(d, "f32", devices=[0] * 4, options={"append_block_rows": 256}); multi.append(X)
    a = ppr_passage_scores(one, g, q, pw, 0.05, 0.5)
    b = ppr_passage_scores(multi, g, q, pw, 0.05, 0.5)
    np.testing.assert_allclose(a, b, rtol=0, atol=1e-13)
    assert np.array_equal(np.argsort(a)[::-1][:50], np.argsort(b)[::-1][:50])
    one.close(); multi.close(); g.close()


def test_device_rows_append_equals_host_rows_append():
    """cmr_mindex_append_dev: rows that live on a device (an encoder's output tensor) are routed like host rows — chunks of
    shards on that device appended in place — and land under the same global ids."""
    import torch
    from comorag_amd.multi_index import MultiDeviceIndex
    d = 256
    X = orc.synthetic_corpus(6_000, d, seed=97)
    Q = orc.synthetic_queries(8, d, seed=98, planted=X)
    a = MultiDeviceIndex(d, "bf16", devices=[0] * 4, options={"append_block_rows": 128})
    b = MultiDeviceIndex(d, "bf16", devices=[0] * 4, options={"append_block_rows": 128})
    at = 0
    for n in (3_000, 25, 25, 1_000, 1, 1_949):
        a.append(X[at:at + n])
        b.append_dev(torch.from_numpy(X[at:at + n]).cuda())
        at += n
    assert a.shard_rows() == b.shard_rows() and len(b) == len(X)
    for x, y in zip(a.search(Q, 20), b.search(Q, 20)):
        assert np.array_equal(x, y)
    assert np.array_equal(a.get_rows(np.arange(0, 6_000, 97)), b.get_rows(np.arange(0, 6_000, 97)))
    a.close(); b.close()


def test_device_rows_append_through_the_peer_staging_branch():
    """The cross-device branch of cmr_mindex_append_dev (wait for the source stream, hipMalloc of a staging buffer on the shard's
    device, hipMemcpyPeer, local append, hipFree) has no second GPU to run on here; option "force_peer_staging" sends EVERY chunk
    through it with the same device on both ends (legal for hipMemcpyPeer).  Same rows, same global ids, same results as the
    in-place route — incl. rows produced on a side stream that the branch must wait for."""
    import torch
    from comorag_amd.multi_index import MultiDeviceIndex
    d = 256
    X = orc.synthetic_corpus(6_000, d, seed=197)
    Q = orc.synthetic_queries(8, d, seed=198, planted=X)
    a = MultiDeviceIndex(d, "bf16", devices=[0] * 4, options={"append_block_rows": 128})
    b = MultiDeviceIndex(d, "bf16", devices=[0] * 4, options={"append_block_rows": 128, "force_peer_staging": 1})
    side = torch.cuda.Stream()
    at = 0
    for n in (3_000, 25, 25, 1_000, 1, 1_949):
        a.append(X[at:at + n])
        with torch.cuda.stream(side):                 # the rows are still being written on `side` when append_dev is called
            t = torch.from_numpy(X[at:at + n]).cuda(non_blocking=True) * 1.0
        b.append_dev(t, stream=side.cuda_stream)
        at += n
    assert a.shard_rows() == b.shard_rows() and len(b) == len(X)
    for x, y in zip(a.search(Q, 20), b.search(Q, 20)):
        assert np.array_equal(x, y)
    assert np.array_equal(a.get_rows(np.arange(0, 6_000, 97)), b.get_rows(np.arange(0, 6_000, 97)))
    b.set_option("force_peer_staging", 0)              # and back: the in-place route on the same handle
    b.append_dev(torch.from_numpy(X[:300]).cuda()); a.append(X[:300])
    for x, y in zip(a.search(Q, 20), b.search(Q, 20)):
        assert np.array_equal(x, y)
    a.close(); b.close()
