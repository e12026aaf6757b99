"""GPU parity of the host-side mirrors (store mirror, retrieval functions, hooks, pool kernel,
encoder tail, re-score, logical shards) against reference-generated fixtures and the oracle."""
import json
import os
import types

import numpy as np
import pytest

from oracle import retrieval_np as orc

pytestmark = pytest.mark.gpu


def test_retrieval_functions_vs_reference_outputs(golden_dir):
    from comorag_amd import retrieval
    from comorag_amd.index import DenseIndex
    for tag in ("small", "mid", "d768"):
        g = np.load(os.path.join(golden_dir, f"dpr_{tag}.npz"))
        X, F, S, Q = g["X"], g["F"], g["S"], g["Q"]
        ix, fx, sx = (DenseIndex(M.shape[1], "f32") for M in (X, F, S))
        ix.append(X); fx.append(F); sx.append(S)
        ex, es = orc.exact_scores_f64(X, Q), orc.exact_scores_f64(S, Q)
        for i in range(len(Q)):
            ids, sc = retrieval.dense_passage_retrieval(ix, Q[i:i + 1])
            assert ids.shape == g[f"dpr_ids_{i}"].shape and sc.dtype == g[f"dpr_scores_{i}"].dtype
            orc.assert_topk_equivalent(ids, g[f"dpr_ids_{i}"], ex[i], 1e-6)          # ALL N ids, tie-aware
            np.testing.assert_allclose(sc, g[f"dpr_scores_{i}"], atol=2e-6)
            idc, scc = retrieval.dense_passage_retrieval(sx, Q[i:i + 1])
            orc.assert_topk_equivalent(idc, g[f"dprc_ids_{i}"], es[i], 1e-6)
            np.testing.assert_allclose(retrieval.get_fact_scores(fx, Q[i:i + 1]), g[f"fact_scores_{i}"], atol=2e-6)
            tid, tsc = retrieval.dense_passage_topk(ix, Q[i:i + 1], 5)
            orc.assert_topk_equivalent(tid[0], g[f"dpr_ids_{i}"][:5], ex[i], 1e-6)
            np.testing.assert_allclose(tsc[0], g[f"dpr_scores_{i}"][:5], atol=2e-6)
        for z in (ix, fx, sx):
            z.close()
    # N = 2 corner (np.squeeze paths) and the reference's N = 1 guard
    g = np.load(os.path.join(golden_dir, "dpr_n2.npz"))
    ix = DenseIndex(g["X"].shape[1], "f32"); ix.append(g["X"])
    ids, sc = retrieval.dense_passage_retrieval(ix, g["Q"][:1])
    assert ids.tolist() == g["dpr_ids_0"].tolist() and np.allclose(sc, g["dpr_scores_0"], atol=2e-6)
    ix.close()


def test_store_device_mirror_summaries_and_cinderella(golden_dir, tmp_path, fake_embedder):
    from comorag_amd import retrieval
    from comorag_amd.embedding_store import EmbeddingStore
    s = json.load(open(os.path.join(golden_dir, "summaries.json")))
    lv = EmbeddingStore(fake_embedder, str(tmp_path), 8, "level_0")
    lv.insert_strings(s["summaries"][:4])
    idx = lv.device_index()
    lv.insert_strings(s["summaries"][4:] + s["summaries"][:2])     # mirror keeps up with appends, dups skipped
    assert len(idx) == len(s["summaries"]) == len(lv.hash_ids)
    fake_embedder.calls.clear()
    texts, scores = retrieval.get_similar_summaries(s["query"], lv, fake_embedder, top_k=3)
    assert texts == s["top_texts"] and fake_embedder.calls == s["encode_calls"]
    np.testing.assert_allclose(scores, s["top_scores"], atol=2e-6)


def test_config1_cinderella_on_the_hip_index(golden_dir):
    """BASELINE config 1 on the GPU: the six cinderella chunks' vectors (fixture: the fake embedder's outputs the
    reference store held; the chunk texts live in the reference tree only) in an f32 DenseIndex; the complete ranking
    and the top-5 of each question must be the reference's (ComoRAG.dense_passage_retrieval, oracle/make_golden.py).
    The CPU-tier twin with the real texts through EmbeddingStore is tests/test_binding_reference.py."""
    from comorag_amd import retrieval
    from comorag_amd.index import DenseIndex
    c = json.load(open(os.path.join(golden_dir, "cinderella.json")))
    X = np.asarray(c["doc_vecs"], np.float32)
    assert X.shape[0] == c["n_docs"] == len(c["keys"]) == 6
    ix = DenseIndex(X.shape[1], "f32"); ix.append(X)
    for want_ids, want_sc, qv in zip(c["top5_ids"], c["top5_scores"], c["question_vecs"]):
        q = np.asarray(qv, np.float32)[None]
        ids, sc = retrieval.dense_passage_retrieval(ix, q)
        assert ids[:5].tolist() == want_ids and len(ids) == 6
        np.testing.assert_allclose(sc[:5], want_sc, atol=2e-6)
        tid, tsc = retrieval.dense_passage_topk(ix, q, 5)
        assert tid[0].tolist() == want_ids
        np.testing.assert_allclose(tsc[0], want_sc, atol=2e-6)
    ix.close()


def test_install_memory_pool_on_a_pool_shaped_object(fake_embedder):
    """GPU twin of tests/test_binding_reference.py::test_install_memory_pool_on_a_real_pool: append-then-search on the
    HIP index behind MemoryPool.retrieve_similar_nodes, against the reference's python loop restated in the oracle."""
    from comorag_amd import hooks

    class Node:
        def __init__(self, probe, cue):
            self.probe, self.cue, self.embedding = probe, cue, None

    class Pool:                                  # utils/memory_utils.py:149-186 as far as the hook reads it
        def __init__(self, em):
            self.pool, self.embedding_model = [], em
        def compute_probe_note_embeddings(self):
            todo = [n for n in self.pool if n.embedding is None]
            if todo:
                for n, e in zip(todo, self.embedding_model.encode([f"{n.probe} {n.cue}" for n in todo]).numpy()):
                    n.embedding = e

    pool = hooks.install_memory_pool(Pool(fake_embedder))
    for cycle in range(3):
        for j in range(5):
            pool.pool.append(Node(f"probe {cycle}", f"cue {cycle}-{j}"))
        if cycle == 1:
            pool.pool.append(Node("probe 1", "cue 1-0"))           # duplicate content: pool order decides
        for pct in (0.5, 0.25, 1.0):
            got = pool.retrieve_similar_nodes(f"probe {cycle}", top_percent=pct)
            embs = [n.embedding for n in pool.pool]
            want = orc.retrieve_similar_nodes(embs, fake_embedder._vec(f"probe {cycle}"), pct)
            assert [pool.pool.index(n) for n in got] == want
    assert len(pool._hip_state["index"]) == len(pool.pool) == 16
    pool._hip_state["index"].close()


def test_retrieve_knn_vs_reference_and_large_k(golden_dir):
    from comorag_amd import retrieval
    g = np.load(os.path.join(golden_dir, "knn.npz"))
    E = g["E"]
    ids = [f"e{i}" for i in range(len(E))]
    exact = orc.exact_scores_f64(orc._l2n(E), orc._l2n(E))
    got = retrieval.retrieve_knn(ids, ids, E, E, k=10, query_batch_size=64, key_batch_size=100)
    for i, q in enumerate(ids):
        gi = np.array([int(s[1:]) for s in got[q][0]])
        orc.assert_topk_equivalent(gi, g["knn_ids"][i], exact[i], 1e-6)
        np.testing.assert_allclose(got[q][1], g["knn_scores"][i], atol=2e-6)
    assert got["e3"][0][:2] == ["e3", "e17"]                       # exact duplicate: lower index first
    # k above CMR_MAX_K (synonymy_edge_topk style) → full-score path
    big = retrieval.retrieve_knn(ids, ids, E, E, k=250, query_batch_size=128)
    ref = orc.retrieve_knn(ids, ids, E, E, k=250, query_batch_size=128)
    for i, q in enumerate(ids[:40]):
        a = np.array([int(s[1:]) for s in big[q][0]]); b = np.array([int(s[1:]) for s in ref[q][0]])
        orc.assert_topk_equivalent(a, b, exact[i], 1e-6)
    assert retrieval.retrieve_knn([], [], np.zeros((0, 4)), np.zeros((0, 4))) == {}


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_retrieve_knn_threshold_filter_feeds_the_synonymy_consumer_identically(dtype):
    """retrieve_knn(min_score=0.8) — fused kernel started at the threshold — vs the oracle's retrieve_knn
    (k = 2047) cut where ComoRAG.add_synonymy_edges stops reading (:696-699).  Entities with near-duplicates, one with
    more than 128 of them (forces the exact re-run), and one with none."""
    from comorag_amd import retrieval
    rng = np.random.default_rng(12)
    base = orc.synthetic_corpus(3000, 768, seed=12)
    E = [base]
    for j, copies in enumerate([3, 1, 150, 20]):                 # clusters of near-duplicates around base[j]
        E.append(base[j] + 0.02 * rng.standard_normal((copies, 768)).astype(np.float32) / np.sqrt(768) * np.sqrt(768) * 0.1)
    E = np.concatenate(E).astype(np.float32)
    ids = [f"e{i}" for i in range(len(E))]
    rnd = orc.bf16_round if dtype == "bf16" else (lambda a: a)
    got = retrieval.retrieve_knn(ids, ids, E, E, k=2047, query_batch_size=512, index_dtype=dtype, min_score=0.8)
    En = rnd(orc._l2n(E))
    exact = orc.exact_scores_f64(En, En)
    n_full = 0
    for i, qid in enumerate(ids):
        want = np.flatnonzero(exact[i] >= 0.8 - 4e-6)
        order = want[np.lexsort((want, -exact[i][want]))]
        gi = np.array([int(x[1:]) for x in got[qid][0]], dtype=np.int64)
        sure = order[exact[i][order] >= 0.8 + 4e-6]               # rows within fp32 rounding of the threshold may go either way
        assert set(sure.tolist()) <= set(gi.tolist()) <= set(order.tolist()), qid
        assert np.all(np.diff(got[qid][1]) <= 0) and (len(gi) == 0 or got[qid][1][-1] >= 0.8 - 4e-6)
        np.testing.assert_allclose(got[qid][1], exact[i][gi], atol=4e-6)
        n_full += len(gi) > 128
    assert n_full >= 150 and len(got["e2999"][0]) == 1               # the big cluster took the exact path; a lone entity finds itself
    # what the consumer reads (self excluded, <= 101 neighbours) is the reference's
    ref = orc.retrieve_knn(ids[:40] + ids[3004:3010], ids, E[list(range(40)) + list(range(3004, 3010))], E, k=2047) if dtype == "f32" else None
    if ref is not None:
        for qid in ref:
            cut = [(n, s) for n, s in zip(*ref[qid]) if s >= 0.8][:103]
            mine = list(zip(*got[qid]))[:len(cut)]
            assert [n for n, _ in mine] == [n for n, _ in cut] or all(abs(a[1] - b[1]) < 4e-6 for a, b in zip(mine, cut))


def test_memory_pool_numeric(golden_dir, fake_embedder):
    from comorag_amd import retrieval
    from comorag_amd.index import DenseIndex
    m = json.load(open(os.path.join(golden_dir, "mempool.json")))
    embs = np.stack([fake_embedder._vec(c) for c in m["contents"]])
    idx = DenseIndex(embs.shape[1], "f32"); idx.append(embs)
    assert retrieval.retrieve_similar_rows(idx, fake_embedder._vec(m["probe"]), len(embs), 0.5) == m["selected"]
    idx.close()


@pytest.mark.parametrize("dtype", ["float32", "bfloat16", "float16"])
@pytest.mark.parametrize("shape", [(5, 19, 40), (32, 512, 768), (3, 7, 1024), (1, 1, 8), (64, 128, 1000)])
def test_pool_kernel(golden_dir, dtype, shape):
    import torch
    from comorag_amd.embedding_model.bge import pool_l2norm
    if shape == (5, 19, 40) and dtype == "float32":
        g = np.load(os.path.join(golden_dir, "pool.npz"))          # torch-CPU outputs of the reference code
        h, m = torch.from_numpy(g["hidden"]).cuda(), torch.from_numpy(g["mask"]).cuda()
        np.testing.assert_allclose(pool_l2norm(h, m).cpu().numpy(), g["normed"], atol=1e-6)
        np.testing.assert_allclose(pool_l2norm(h, m, normalize=False).cpu().numpy(), g["pooled"], atol=1e-6)
    b, l, d = shape
    rng = np.random.default_rng(b * l + d)
    hid = rng.standard_normal(shape).astype(np.float32)
    lens = rng.integers(1, l + 1, size=b); lens[0] = l
    mask = (np.arange(l)[None, :] < lens[:, None]).astype(np.int64)
    ht = torch.from_numpy(hid).cuda().to(getattr(torch, dtype))
    got = pool_l2norm(ht, torch.from_numpy(mask).cuda()).cpu().numpy()
    want = orc.mean_pool_l2norm(ht.float().cpu().numpy(), mask)     # oracle on the same (rounded) inputs
    np.testing.assert_allclose(got, want, atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose((got ** 2).sum(1), 1.0, atol=1e-5)


def test_encoder_end_to_end_vs_oracle():
    """tokenise → BERT forward (torch-ROCm) → HIP pool+norm  ==  oracle restatement of the reference
    `batch_encode` (torch fp32), same seed-initialised model and synthetic vocabulary."""
    import copy
    import torch
    from comorag_amd.embedding_model import _get_embedding_model_class
    from comorag_amd.utils.config_utils import BaseConfig
    from oracle import encode_torch as enc
    model, tok = enc.tiny_bert(hidden=128, layers=2, heads=4, inter=256, max_pos=64)
    cfg = BaseConfig(embedding_model_name="bge-tiny-random", embedding_batch_size=4, embedding_max_seq_len=2048)
    cls = _get_embedding_model_class(cfg.embedding_model_name)
    em = cls(global_config=cfg, embedding_model_name=cfg.embedding_model_name, model=copy.deepcopy(model), tokenizer=tok)
    assert em.embedding_dim == 128
    texts = [f"the prince and the golden slipper number {i} " + "and the bird in the tree " * (i % 5) for i in range(11)]
    texts.append("she was good and pious " * 40)                     # > 64 positions: clamped, not a crash
    got = em.batch_encode(texts, instruction="ignored", norm=True)
    want = enc.batch_encode(model, tok, texts, batch_size=4, max_length=64)
    assert got.shape == want.shape == (12, 128) and got.dtype == np.float32
    np.testing.assert_allclose(got, want, atol=2e-5)                 # GPU vs CPU forward: fp32 reduction order
    one = em.batch_encode("midnight")
    np.testing.assert_allclose(one, enc.batch_encode(model, tok, "midnight", batch_size=4, max_length=64), atol=2e-5)
    t = em.encode(["what did the mother wish", "midnight"])         # positional, torch tensor (memory_utils.py:176)
    assert isinstance(t, torch.Tensor) and t.shape == (2, 128)
    np.testing.assert_allclose(em.encode_queries(["midnight"]), one, atol=1e-6)
    # length-bucketed mini-batches (default) vs the reference's arrival-order mini-batches: same rows, same order
    cfg2 = BaseConfig(embedding_model_name="bge-tiny-random", embedding_batch_size=4, embedding_max_seq_len=2048, embedding_length_bucketing=False)
    em2 = cls(global_config=cfg2, embedding_model_name=cfg2.embedding_model_name, model=copy.deepcopy(model), tokenizer=tok)
    assert em._bucket and not em2._bucket
    np.testing.assert_allclose(em2.batch_encode(texts), got, atol=2e-5)
    # bucketing within windows of ONE reference chunk (three windows here, the next ones tokenised while one is on the GPU),
    # and the same with the tokenizer in worker processes: same rows, same order
    # ... and with worker processes started lazily by the first corpus-sized call (-1: the call that starts them goes on with threads,
    # the next one finds them answering)
    for extra in ({"embedding_bucket_window": 1}, {"embedding_bucket_window": 2, "embedding_tokenizer_processes": 2},
                  {"embedding_bucket_window": 1, "embedding_tokenizer_processes": -1}):
        cfg3 = BaseConfig(embedding_model_name="bge-tiny-random", embedding_batch_size=4, embedding_max_seq_len=2048, **extra)
        em3 = cls(global_config=cfg3, embedding_model_name=cfg3.embedding_model_name, model=copy.deepcopy(model), tokenizer=tok)
        np.testing.assert_allclose(em3.batch_encode(texts), got, atol=2e-5)
        if extra.get("embedding_tokenizer_processes") == -1 and len(texts) >= 2 * 4:
            assert em3._tok_procs_starting is not None                  # two windows of texts: the start was triggered
            em3._tok_procs_starting.join(120.0)
            assert em3._tok_procs is not None
            np.testing.assert_allclose(em3.batch_encode(texts), got, atol=2e-5)      # now through the worker processes
        em3.close()


def test_store_insert_appends_the_encoder_device_tensor(tmp_path):
    """EmbeddingStore.insert_strings with an HBM mirror in place: the encoder's device tensor is appended as it is
    (batch_encode_dev -> cmr_index_append_dev), the host matrix gets its copy from the same tensor — index rows, host rows
    and a plain batch_encode of the same texts agree bit for bit; ids stay in insertion order."""
    import copy
    from comorag_amd.embedding_model import _get_embedding_model_class
    from comorag_amd.embedding_store import EmbeddingStore
    from comorag_amd.utils.config_utils import BaseConfig
    from oracle import encode_torch as enc
    model, tok = enc.tiny_bert(hidden=128, layers=2, heads=4, inter=256, max_pos=64)
    cfg = BaseConfig(embedding_model_name="bge-tiny-random", embedding_batch_size=4)
    em = _get_embedding_model_class(cfg.embedding_model_name)(global_config=cfg, embedding_model_name=cfg.embedding_model_name, model=copy.deepcopy(model), tokenizer=tok)
    store = EmbeddingStore(em, str(tmp_path / "s"), 4, "chunk")
    first = [f"the bird in the tree number {i}" for i in range(6)]
    store.insert_strings(first)
    idx = store.device_index("f32")
    calls = []
    orig = idx.append_dev
    idx.append_dev = lambda t, *a, **k: (calls.append(tuple(t.shape)), orig(t, *a, **k))[1]
    more = [f"the golden slipper number {i} " + "and the prince " * (i % 4) for i in range(9)]
    store.insert_strings(more + first[:2])                       # two known texts are skipped
    assert calls == [(9, 128)] and len(idx) == 15 and len(store.hash_ids) == 15
    host = np.array(store.get_embeddings(store.get_all_ids()))
    np.testing.assert_array_equal(idx.get_rows(np.arange(15)), host)
    np.testing.assert_array_equal(host[6:], em.batch_encode(more))
    t = em.batch_encode_dev(more)
    assert t.is_cuda and tuple(t.shape) == (9, 128)
    np.testing.assert_array_equal(t.cpu().numpy(), host[6:])
    em.close()


def test_hooks_on_a_comorag_shaped_object(golden_dir):
    """install() rebinds the numeric call sites on an object with ComoRAG's attribute names."""
    from comorag_amd import hooks
    g = np.load(os.path.join(golden_dir, "dpr_mid.npz"))
    X, F, S, Q = g["X"], g["F"], g["S"], g["Q"]

    class Enc:
        n = 0
        def batch_encode(self, text, **kw):
            Enc.n += 1
            return Q[int(text[1:]):int(text[1:]) + 1]

    class Rag:                                     # the attributes ComoRAG.prepare_retrieval_objects sets
        def __init__(self):
            self.global_config = types.SimpleNamespace(need_cluster=True, index_dtype="f32")
            self.embedding_model = Enc()
            self.ready_to_retrieve = False
        def prepare_retrieval_objects(self):
            self.query_to_embedding = {"triple": {}, "passage": {}}
            self.passage_embeddings, self.fact_embeddings, self.summary_embeddings = X, F, S
            self.ready_to_retrieve = True
    import sys
    mod = sys.modules[Rag.__module__]
    mod.get_query_instruction = lambda k: k
    rag = hooks.install(Rag(), patch_module_functions=False)
    rag.prepare_retrieval_objects(); rag.prepare_retrieval_objects()
    ex = orc.exact_scores_f64(X, Q)
    for i in range(len(Q)):
        ids, sc = rag.dense_passage_retrieval(f"q{i}")
        orc.assert_topk_equivalent(ids, g[f"dpr_ids_{i}"], ex[i], 1e-6)
        np.testing.assert_allclose(sc, g[f"dpr_scores_{i}"], atol=2e-6)
        np.testing.assert_allclose(rag.get_fact_scores(f"q{i}"), g[f"fact_scores_{i}"], atol=2e-6)
        idc, scc = rag.dense_passage_retrieval(f"q{i}", need_cluster=True)
        np.testing.assert_allclose(scc, g[f"dprc_scores_{i}"], atol=2e-6)
    n0 = Enc.n
    rag.get_query_embeddings("q1"); rag.dense_passage_retrieval("q1")
    assert Enc.n == n0                                               # full query memoised: no re-encode


def test_exact_rescorer_shape():
    from comorag_amd.index import DenseIndex
    from comorag_amd.rerank import ExactRescorer, search_then_rescore
    X = orc.synthetic_corpus(2000, 1024, seed=3); Q = orc.synthetic_queries(2, 1024, seed=4, planted=X)
    idx = DenseIndex(1024, "f16", keep_f32=True); idx.append(X)
    cand = idx.search(Q[:1], 100)[0][0].tolist()
    items = [("s", "p", str(c)) for c in cand]
    keep, kept_items, info = ExactRescorer(idx).rerank(Q[0], items, cand, len_after_rerank=5)
    exact = orc.exact_scores_f64(X, Q[:1])[0]
    want = sorted(cand, key=lambda r: (-exact[r], r))[:5]
    orc.assert_topk_equivalent(np.array(keep), np.array(want), exact, 1e-6)
    assert kept_items == [("s", "p", str(r)) for r in keep] and len(info["confidence"]) == 5
    ids, sc = search_then_rescore(idx, Q, 100, 20)
    assert ids.shape == (2, 20)
    idx.close()


@pytest.mark.parametrize("S", [2, 4, 8])
def test_logical_shards_equal_single_index(S):
    """Multi-GPU oracle: S row shards + candidate merge == one index, bit for bit."""
    from comorag_amd.index import DenseIndex, merge_topk
    from comorag_amd.sharded import shard_bounds
    X = orc.synthetic_corpus(10_007, 256, seed=31); Q = orc.synthetic_queries(9, 256, seed=32, planted=X)
    X[9000] = X[5]                                                    # a cross-shard tie
    one = DenseIndex(256, "bf16"); one.append(X)
    wi, ws, _, _ = one.search(Q, 20)
    ids, scs = [], []
    for r in range(S):
        lo, hi = shard_bounds(len(X), S, r)
        sh = DenseIndex(256, "bf16"); sh.append(X[lo:hi])
        i, s, _, _ = sh.search(Q, 20)
        ids.append(i + lo); scs.append(s); sh.close()
    mi, ms = merge_topk(np.stack(ids), np.stack(scs))
    assert np.array_equal(mi, wi) and np.array_equal(ms, ws)
    # device-side merge kernel gives the same
    import ctypes as C, torch
    from comorag_amd import _lib as L
    gi, gs = torch.from_numpy(np.stack(ids)).cuda(), torch.from_numpy(np.stack(scs)).cuda()
    oi, os_ = torch.empty((9, 20), dtype=torch.int64, device="cuda"), torch.empty((9, 20), dtype=torch.float32, device="cuda")
    L.check(L.lib().cmr_merge_topk_dev(0, C.c_void_p(gi.data_ptr()), C.c_void_p(gs.data_ptr()), S, 9, 20,
                                       C.c_void_p(oi.data_ptr()), C.c_void_p(os_.data_ptr()),
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert np.array_equal(oi.cpu().numpy(), wi) and np.array_equal(os_.cpu().numpy(), ws)
    # the packed exchange format: one u64 per candidate (what the single all-gather ships), then the key merge
    from comorag_amd.sharded import pack_candidates, unpack_candidates
    keys = torch.empty((S, 9, 20), dtype=torch.int64, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(L.lib().cmr_pack_candidates_dev(C.c_void_p(gi.data_ptr()), C.c_void_p(gs.data_ptr()), S * 9 * 20, C.c_void_p(keys.data_ptr()), st))
    L.check(L.lib().cmr_merge_keys_dev(C.c_void_p(keys.data_ptr()), S, 9, 20, C.c_void_p(oi.data_ptr()), C.c_void_p(os_.data_ptr()), st))
    torch.cuda.synchronize()
    assert np.array_equal(oi.cpu().numpy(), wi) and np.array_equal(os_.cpu().numpy(), ws)
    hk = keys.cpu().numpy().view(np.uint64)
    assert np.array_equal(hk, pack_candidates(np.stack(ids), np.stack(scs)))          # numpy twin (gloo / host path) packs identically
    ui, us = unpack_candidates(hk)
    assert np.array_equal(ui, np.stack(ids)) and np.array_equal(us, np.stack(scs))
    one.close()


def test_index_pipelined_equals_plain():
    import torch
    from comorag_amd.index import DenseIndex
    X = orc.synthetic_corpus(300_000, 128, seed=51)
    idx = DenseIndex(128, "bf16"); idx.append(X)
    for nq in (5, 64, 130):                                   # 130 = three passes, each its own pipeline slot
        Q = orc.synthetic_queries(nq, 128, seed=nq, planted=X)
        wi, ws, wmn, wmx = idx.search(Q, 20)
        q = torch.from_numpy(Q).cuda()
        res = []
        for i in range(4):                                    # back-to-back batches, alternating output buffers
            oi = torch.empty((nq, 20), dtype=torch.int64, device="cuda"); os_ = torch.empty((nq, 20), dtype=torch.float32, device="cuda")
            mn = torch.empty(nq, device="cuda"); mx = torch.empty(nq, device="cuda")
            h = idx.search_pipelined(q, 20, oi, os_, mn, mx)
            res.append((h, oi, os_, mn, mx))
        for h, oi, os_, mn, mx in res:
            idx.sync(h)
            assert np.array_equal(oi.cpu().numpy(), wi) and np.array_equal(os_.cpu().numpy(), ws)
            assert np.array_equal(mn.cpu().numpy(), wmn) and np.array_equal(mx.cpu().numpy(), wmx)
    idx.close()


def test_pipelined_search_single_rank():
    import torch
    from comorag_amd.index import DenseIndex
    from comorag_amd.sharded import ShardedIndex
    X = orc.synthetic_corpus(50_000, 128, seed=41); Q = orc.synthetic_queries(16, 128, seed=42, planted=X)
    plain = DenseIndex(128, "bf16"); plain.append(X)
    want_i, want_s, _, _ = plain.search(Q, 20)
    sh = ShardedIndex(128, "bf16", base=1000)                 # the library offsets row ids by the shard base
    sh.local.append(X)
    q = torch.from_numpy(Q).cuda()
    torch.cuda.synchronize()
    outs = []
    for i in range(5):
        b = sh.search_pipelined(q, 20, i & 1)
        b["done"].synchronize()
        outs.append((b["o_ids"].cpu().numpy().copy(), b["o_sc"].cpu().numpy().copy()))
    for oi, os_ in outs:
        assert np.array_equal(oi, want_i + 1000) and np.array_equal(os_, want_s)
    hi, hs = sh.search(Q, 20)
    assert np.array_equal(hi, want_i + 1000) and np.array_equal(hs, want_s)
    plain.close()


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_id_block_table_translates_every_id_path(dtype):
    """cmr_index_set_id_blocks (a row shard that took incremental appends: several runs of consecutive global ids).  Every
    entry point that returns ids — single-launch search, general chain, wide batch, large k, threshold search, complete
    ranking, pipelined — must return exactly the ids of the same index WITHOUT a table mapped through the table on the
    host, and the entry points that take ids (re-score, row fetch) must accept the global ones."""
    import torch
    from comorag_amd import _lib as L
    from comorag_amd.index import DenseIndex
    n, d = 70_000, 128
    X = orc.synthetic_corpus(n, d, seed=31); Q = orc.synthetic_queries(70, d, seed=32, planted=X)
    ls = np.array([0, 40_000, 40_025, 65_000], np.int64)          # local starts
    gs = np.array([1000, 900_000, 900_050, 2_000_000], np.int64)  # global starts (ascending, gaps between the runs)
    def to_global(ids):
        b = np.searchsorted(ls, ids, side="right") - 1
        return np.where(ids >= 0, ids - ls[b] + gs[b], -1)
    plain = DenseIndex(d, dtype, keep_f32=True); plain.append(X)
    blk = DenseIndex(d, dtype, keep_f32=True); blk.append(X); blk.set_id_blocks(ls, gs)
    small_p = DenseIndex(d, dtype); small_p.append(X[:900])
    small_b = DenseIndex(d, dtype); small_b.append(X[:900]); small_b.set_id_blocks([0, 500], [7, 5000])
    si, ss = small_p.search(Q[:3], 10)[:2]; bi, bs = small_b.search(Q[:3], 10)[:2]                  # single launch
    assert np.array_equal(bi, np.where(si < 500, si + 7, si - 500 + 5000)) and np.array_equal(bs, ss)
    for nq, k in ((1, 20), (9, 20), (70, 20), (4, 300)):                                            # general chain, wide batch (bf16: 70 > 64), large k
        pi, ps = plain.search(Q[:nq], k)[:2]; gi, gsc = blk.search(Q[:nq], k)[:2]
        assert np.array_equal(gi, to_global(pi)) and np.array_equal(gsc, ps), (nq, k)
    pi, ps = plain.search_min_score(Q[:8], 16, 0.5); gi, gsc = blk.search_min_score(Q[:8], 16, 0.5)
    assert np.array_equal(gi, to_global(pi)) and np.array_equal(gsc, ps)
    pi, ps = plain.sorted_scores(Q[:2])[:2]; gi, gsc = blk.sorted_scores(Q[:2])[:2]
    assert np.array_equal(gi, to_global(pi)) and np.array_equal(gsc, ps)
    qt = torch.from_numpy(Q[:64]).cuda(); oi = torch.empty((64, 20), dtype=torch.int64, device="cuda"); osc = torch.empty((64, 20), device="cuda")
    torch.cuda.synchronize()
    blk.sync(blk.search_pipelined(qt, 20, oi, osc))
    pi, ps = plain.search(Q[:64], 20)[:2]
    assert np.array_equal(oi.cpu().numpy(), to_global(pi)) and np.array_equal(osc.cpu().numpy(), ps)
    cand_l = plain.search(Q[:5], 100)[0]
    ri, rs = plain.rescore(Q[:5], cand_l, 20); gi, gsc = blk.rescore(Q[:5], to_global(cand_l), 20)   # ids IN are global
    assert np.array_equal(gi, to_global(ri)) and np.array_equal(gsc, rs)
    rows_l = np.array([0, 39_999, 40_000, 40_024, 40_025, 69_999])
    assert np.array_equal(blk.get_rows(to_global(rows_l)), plain.get_rows(rows_l))
    assert not blk.get_rows(np.array([999, 41_000 + 1000, 900_025 + 10])).any()                     # ids in the gaps: no such row
    # validation: overlapping runs, a table beyond the 32-bit row of the packed exchange, an id base that overflows it
    with pytest.raises(L.CmrError):
        blk.set_id_blocks([0, 10], [100, 105])
    with pytest.raises(L.CmrError):
        blk.set_id_blocks([0, 10], [100, 0xFFFFFFFF - 5])
    with pytest.raises(L.CmrError):
        plain.set_id_base(0xFFFFFFFF - 10)
    for i in (plain, blk, small_p, small_b):
        i.close()


def test_rccl_exchange_path_one_rank(tmp_path):
    """The N>1 code path (ExternalStream on the pipeline's post stream → cmr_pack_candidates_dev → ONE RCCL
    all_gather_into_tensor → cmr_merge_keys_dev; and the library's own cmr_comm_allgather_merge) run mechanically
    on a 1-rank group in a subprocess (a single-GPU box cannot host two RCCL ranks).  Numerics of multi-shard merging are covered by the logical-shard
    and gloo tests."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", RANK="0", WORLD_SIZE="1")
        import numpy as np, torch, torch.distributed as dist
        from oracle import retrieval_np as orc
        from comorag_amd.index import DenseIndex
        from comorag_amd.sharded import ShardedIndex
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        X = orc.synthetic_corpus(40_000, 128, seed=1); Q = orc.synthetic_queries(16, 128, seed=2, planted=X)
        plain = DenseIndex(128, "bf16"); plain.append(X)
        wi, ws, _, _ = plain.search(Q, 20)
        sh = ShardedIndex(128, "bf16", rank=0, world=1, base=500, force_exchange=True)
        sh.local.append(X)
        q = torch.from_numpy(Q).cuda(); torch.cuda.synchronize()
        for i in range(6):
            b = sh.search_pipelined(q, 20, i & 1)
        b["done"].synchronize()
        assert np.array_equal(b["o_ids"].cpu().numpy(), wi + 500) and np.array_equal(b["o_sc"].cpu().numpy(), ws)
        hi, hs = sh.search(Q, 20)
        assert np.array_equal(hi, wi + 500)
        # the same exchange through the library's own RCCL communicator (cmr_comm_*: no torch collective in the path)
        sh2 = ShardedIndex(128, "bf16", rank=0, world=1, base=500, force_exchange=True, exchange="cabi", timing=True, index=sh.local)
        for i in range(6):
            b2 = sh2.search_pipelined(q, 20, i & 1)
        b2["done"].synchronize()
        assert np.array_equal(b2["o_ids"].cpu().numpy(), wi + 500) and np.array_equal(b2["o_sc"].cpu().numpy(), ws)
        t = sh2.exchange_times_ms()
        assert len(t) == 6 and all(a >= 0 and m >= 0 for a, m in t)
        assert sh2.comm_info() == {"world": 1, "rank": 0, "rccl_ranks_seen": 1}, sh2.comm_info()
        sh2.close()
        dist.destroy_process_group()
        print("EXCHANGE_OK")
    """ % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "EXCHANGE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("n", [28_671, 28_672])
def test_full_ranking_with_duplicate_rows_on_both_sides_of_the_device_sort_threshold(n):
    """retrieval.dense_passage_retrieval ranks ALL rows (ComoRAG.py:958-966).  Below DEVICE_SORT_MIN_ROWS the reference's own
    `np.argsort(scores)[::-1]` runs on the GPU scores; from there on the device radix sort.  Rows with EQUAL scores
    (duplicated chunks) come out in numpy's introsort order (unspecified, not stable) in the first regime and by ascending
    row id in the second (the exported tie rule) — the documented deviation of include/comorag_hip.h.  Pinned here: on both
    sides every duplicate group sits together with identical scores, the ranking is a permutation that agrees with the
    reference's own ranking up to the order INSIDE groups of equal scores, and above the threshold that order is ascending."""
    from comorag_amd import retrieval
    from comorag_amd.index import DenseIndex
    assert retrieval.DEVICE_SORT_MIN_ROWS == 28_672
    d = 128
    X = orc.synthetic_corpus(n, d, seed=777)
    groups = [[10, 500, 9_000, n - 1], [77, 20_000], [3, 4, 5]]
    for g in groups:
        for r in g[1:]:
            X[r] = X[g[0]]
    q = orc.synthetic_queries(1, d, seed=778)[0]
    q = (q + 0.7 * X[10] + 0.4 * X[77]); q /= np.linalg.norm(q)
    idx = DenseIndex(d, "f32", capacity_hint=n); idx.append(X)
    ids, sc = retrieval.dense_passage_retrieval(idx, q)
    ref_ids, ref_sc = orc.dense_passage_retrieval(X, q[None, :])          # the reference's lines on the host (np.dot + argsort)
    assert sorted(ids.tolist()) == list(range(n)) and np.all(np.diff(sc) <= 0)
    np.testing.assert_allclose(sc, ref_sc, atol=2e-6)
    pos = np.empty(n, np.int64); pos[ids] = np.arange(n)
    rpos = np.empty(n, np.int64); rpos[ref_ids] = np.arange(n)
    for g in groups:
        p = np.sort(pos[g])
        assert np.array_equal(p, np.arange(p[0], p[0] + len(g))), g                      # the group is contiguous ...
        assert len(set(sc[p].tolist())) == 1                                              # ... with one score
        assert np.ptp(ref_sc[rpos[g]]) <= 2e-6                                            # (the host BLAS may round a row's copies differently)
        if n >= retrieval.DEVICE_SORT_MIN_ROWS:
            assert ids[p].tolist() == sorted(g)                                           # exported tie rule: ascending row id
    # outside equal-score groups the two rankings agree (near-ties inside the fp32 rounding bound may swap)
    ex = orc.exact_scores_f64(X, q[None, :])[0]
    orc.assert_topk_equivalent(ids[:200], ref_ids[:200], ex, 1e-6)
    idx.close()


def test_tri_retrieve_layers_on_the_hip_index_vs_the_reference_fixture(golden_dir, tmp_path, fake_embedder):
    """tests/golden/tri_retrieve.json holds what the reference's own ComoRAG.tri_retrieve (ComoRAG.py:456-554) returned in the build
    container for a seeded toy corpus (oracle/make_golden.py; the CPU tier runs that very method on this package's stores and hooks,
    tests/test_binding_reference.py).  The reference tree is not on the GPU box, so here the three layers are assembled from the same
    pieces on a ComoRAG-shaped object — hooks.install's rebound numeric calls on HIP indexes, this package's EmbeddingStore, retrieval.
    get_similar_summaries on the level store's HBM mirror — and must be the fixture's, text for text."""
    from comorag_amd import hooks, retrieval
    from comorag_amd.embedding_store import EmbeddingStore
    gold = json.load(open(os.path.join(golden_dir, "tri_retrieve.json")))
    cfg = gold["config"]
    st = {}
    for ns, texts in gold["corpus"].items():
        st[ns] = EmbeddingStore(fake_embedder, str(tmp_path / ns), 8, ns)
        st[ns].insert_strings(texts)
    held = {kind: {st[ns].text_to_hash_id[gold["corpus"][ns][i]] for ns, i in items} for kind, items in gold["pool"].items()}

    class Rag:                                     # the attributes ComoRAG.prepare_retrieval_objects sets (:876-907), from the stores
        def __init__(self):
            self.global_config = types.SimpleNamespace(**cfg)
            self.embedding_model = fake_embedder
            self.ready_to_retrieve = False
        def prepare_retrieval_objects(self):
            self.query_to_embedding = {"triple": {}, "passage": {}}
            self.passage_node_keys, self.fact_node_keys, self.summary_node_keys = (list(st[n].get_all_ids()) for n in ("chunk", "fact", "summary"))
            self.passage_embeddings = np.array(st["chunk"].get_embeddings(self.passage_node_keys))
            self.fact_embeddings = np.array(st["fact"].get_embeddings(self.fact_node_keys))
            self.summary_embeddings = np.array(st["summary"].get_embeddings(self.summary_node_keys))
            self.ready_to_retrieve = True
    import sys
    sys.modules[Rag.__module__].get_query_instruction = lambda k: k
    rag = hooks.install(Rag(), patch_module_functions=False)

    def layer(store, keys, ids, top_k, pool_hashes, by_store_order):
        texts = [store.get_row(keys[i])["content"] for i in ids[:top_k]]
        texts = [t for t in texts if store.text_to_hash_id[t] not in pool_hashes]
        if by_store_order:
            order = store.get_hash_id_to_order()
            texts = sorted(texts, key=lambda t: order.get(store.text_to_hash_id[t], float("inf")))
        return texts

    for q, want in zip(gold["queries"], gold["docs"]):
        if not rag.ready_to_retrieve:
            rag.prepare_retrieval_objects()
        rag.get_query_embeddings(q)
        assert len(rag.get_fact_scores(q)) == len(gold["corpus"]["fact"])          # the link step's input; the reranker (an LLM) kept no fact
        ids, _ = rag.dense_passage_retrieval(q)
        ver = layer(st["chunk"], rag.passage_node_keys, ids, cfg["qa_ver_top_k"], held["VER"], True)
        sids, _ = rag.dense_passage_retrieval(q, need_cluster=True)
        sem = layer(st["summary"], rag.summary_node_keys, sids, cfg["qa_sem_top_k"], held["SEM"], False)
        epi_texts, _ = retrieval.get_similar_summaries(query=q, level_store=st["level_0"], embedding_model=fake_embedder, top_k=cfg["qa_epi_top_k"])
        lv = st["level_0"]
        epi = [t for t in epi_texts[:cfg["qa_epi_top_k"]] if lv.text_to_hash_id[t] not in held["EPI"]]
        order = lv.get_hash_id_to_order()
        epi = sorted(epi, key=lambda t: order.get(lv.text_to_hash_id[t], float("inf")))
        assert {"veridical": ver, "semantic": sem, "episodic": epi} == want, q
