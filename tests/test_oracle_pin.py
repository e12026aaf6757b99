"""Pin the oracle (oracle/*.py): (1) against the committed fixtures that oracle/make_golden.py
produced by RUNNING THE REFERENCE's functions, (2) against the live reference when the tree is here."""
import json
import os

import numpy as np
import pytest

from oracle import retrieval_np as orc


@pytest.mark.parametrize("tag", ["small", "mid", "d768", "n2"])
def test_dpr_and_fact_scores_match_reference_outputs(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, f"dpr_{tag}.npz"))
    X, F, S, Q = g["X"], g["F"], g["S"], g["Q"]
    exact = orc.exact_scores_f64(X, Q)
    for i in range(len(Q)):
        ids, sc = orc.dense_passage_retrieval(X, Q[i:i + 1])
        # same numpy build → bit-identical; a different BLAS may reorder near-ties → tie-aware check
        if not np.array_equal(ids, g[f"dpr_ids_{i}"]):
            orc.assert_topk_equivalent(ids, g[f"dpr_ids_{i}"], exact[i], 1e-6)
        np.testing.assert_allclose(sc, g[f"dpr_scores_{i}"], atol=1e-6)
        ids_c, sc_c = orc.dense_passage_retrieval(S, Q[i:i + 1])
        np.testing.assert_allclose(sc_c, g[f"dprc_scores_{i}"], atol=1e-6)
        np.testing.assert_allclose(orc.get_fact_scores(F, Q[i:i + 1]), g[f"fact_scores_{i}"], atol=1e-6)


def test_minmax_mdhash(golden_dir):
    g = np.load(os.path.join(golden_dir, "minmax.npz"))
    assert np.array_equal(orc.min_max_normalize(g["v"]), g["v_out"]) and np.array_equal(orc.min_max_normalize(g["c"]), g["c_out"])
    j = json.load(open(os.path.join(golden_dir, "mdhash.json")))
    assert [orc.compute_mdhash_id(s, prefix=p) for s in j["strings"] for p in ("", "chunk-", "entity-")] == j["ids"]


def test_pool_matches_torch_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "pool.npz"))
    np.testing.assert_allclose(orc.mean_pool_l2norm(g["hidden"], g["mask"], normalize=False), g["pooled"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(orc.mean_pool_l2norm(g["hidden"], g["mask"]), g["normed"], rtol=1e-6, atol=1e-6)


def test_knn_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "knn.npz"))
    E = g["E"]
    ids = [f"e{i}" for i in range(len(E))]
    got = orc.retrieve_knn(ids, ids, E, E, k=10, query_batch_size=64, key_batch_size=100)
    exact = orc.exact_scores_f64(orc._l2n(E), orc._l2n(E))
    for i, q in enumerate(ids):
        gi = np.array([int(s[1:]) for s in got[q][0]])
        orc.assert_topk_equivalent(gi, g["knn_ids"][i], exact[i], 1e-6)   # torch.topk tie order is unspecified
        np.testing.assert_allclose(got[q][1], g["knn_scores"][i], atol=1e-6)


def test_mempool_and_summaries(golden_dir, fake_embedder):
    m = json.load(open(os.path.join(golden_dir, "mempool.json")))
    embs = [fake_embedder._vec(c) for c in m["contents"]]
    assert orc.retrieve_similar_nodes(embs, fake_embedder._vec(m["probe"]), 0.5) == m["selected"]
    s = json.load(open(os.path.join(golden_dir, "summaries.json")))
    S = np.stack([fake_embedder._vec(t) for t in s["summaries"]])
    texts, scores = orc.get_similar_summaries(S, s["summaries"], fake_embedder._vec(s["query"])[None], top_k=3)
    assert texts == s["top_texts"]
    np.testing.assert_allclose(scores, s["top_scores"], atol=1e-6)
    assert s["encode_calls"] == [[s["query"]]]


def test_insert_plan_matches_store_fixture(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "store.json")))
    miss, texts = orc.insert_plan([], g["batch1"], "chunk")
    assert miss == g["ids_after_1"] and texts == ["alpha", "beta", "gamma"]
    miss2, texts2 = orc.insert_plan(miss, g["batch2"], "chunk")
    assert miss + miss2 == g["ids_after_2"] and texts2 == ["delta", "epsilon"]


def test_cinderella_fixture(golden_dir, fake_embedder):
    """BASELINE config 1 (plumbing, CPU): the ORACLE on the six cinderella chunk vectors must give the top-5 the
    reference's store + dense_passage_retrieval gave (the product's run over the same data: tests/test_binding_reference.py
    on CPU with the real texts, tests/test_dropin_gpu.py::test_config1_cinderella_on_the_hip_index on the GPU)."""
    c = json.load(open(os.path.join(golden_dir, "cinderella.json")))
    assert c["n_docs"] == 6 and c["keys"] == ["cinder-" + h for h in c["doc_md5"]]
    X = np.asarray(c["doc_vecs"], np.float32)
    for q, want_ids, want_sc, qv in zip(c["questions"], c["top5_ids"], c["top5_scores"], c["question_vecs"]):
        np.testing.assert_array_equal(fake_embedder._vec(q), np.asarray(qv, np.float32))
        ids, sc = orc.dense_passage_retrieval(X, np.asarray(qv, np.float32)[None])
        assert ids[:5].tolist() == want_ids
        np.testing.assert_allclose(sc[:5], want_sc, atol=1e-6)


def test_live_reference_functions_agree():
    from oracle.ref_loader import reference_available, ref_modules
    if not reference_available():
        pytest.skip("reference tree not present (GPU box)")
    m = ref_modules()
    rng = np.random.default_rng(5)
    X = orc.synthetic_corpus(777, 96, seed=1); Q = orc.synthetic_queries(3, 96, seed=2, planted=X)
    obj = m["ComoRAG"].ComoRAG.__new__(m["ComoRAG"].ComoRAG)
    obj.passage_embeddings = obj.fact_embeddings = obj.summary_embeddings = X
    obj.query_to_embedding = {"triple": {"q": Q[:1]}, "passage": {"q": Q[:1]}}
    ids, sc = obj.dense_passage_retrieval("q")
    oi, os_ = orc.dense_passage_retrieval(X, Q[:1])
    assert np.array_equal(ids, oi) and np.array_equal(sc, os_)
    assert np.array_equal(obj.get_fact_scores("q"), orc.get_fact_scores(X, Q[:1]))
    v = rng.standard_normal(50).astype(np.float32)
    assert np.array_equal(m["misc_utils"].min_max_normalize(v), orc.min_max_normalize(v))
    assert m["misc_utils"].compute_mdhash_id("héllo", prefix="x-") == orc.compute_mdhash_id("héllo", prefix="x-")
    import torch
    h = rng.standard_normal((3, 9, 20)).astype(np.float32); mk = np.array([[1] * 9, [1] * 4 + [0] * 5, [1] + [0] * 8])
    ref = torch.nn.functional.normalize(m["bge"].mean_pooling(torch.from_numpy(h), torch.from_numpy(mk)), p=2, dim=1).numpy()
    np.testing.assert_allclose(orc.mean_pool_l2norm(h, mk), ref, rtol=1e-6, atol=1e-7)


def test_encode_oracle_equals_live_reference_class():
    """oracle/encode_torch.py vs the reference BGEEmbeddingModel run verbatim on a bare instance with
    the same seed-initialised BERT + synthetic tokenizer (SURVEY.md §8c recipe)."""
    from oracle.ref_loader import reference_available, ref_modules
    if not reference_available():
        pytest.skip("reference tree not present (GPU box)")
    from oracle import encode_torch as enc
    m = ref_modules()
    model, tok = enc.tiny_bert()
    Ref = m["bge"].BGEEmbeddingModel
    r = Ref.__new__(Ref)
    r.tokenizer, r.embedding_model = tok, model
    r.embedding_config = m["emb_base"].EmbeddingConfig.from_dict({"norm": True, "encode_params": {
        "max_length": 64, "query_instruction": enc.BGE_PREFIX, "passage_instruction": enc.BGE_PREFIX, "batch_size": 2}})
    r.encode = r._encode
    texts = ["she was good and pious", "the prince and the golden slipper", "midnight", "a bird in the tree",
             "what did the mother wish"]
    ref = r.batch_encode(texts, instruction="ignored by the reference", norm=True)
    got = enc.batch_encode(model, tok, texts, batch_size=2, max_length=64)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-6)
    one = r.batch_encode("midnight")
    np.testing.assert_allclose(enc.batch_encode(model, tok, "midnight", batch_size=2, max_length=64), one, atol=1e-6)
