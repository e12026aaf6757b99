"""Generate tests/golden/*.npz|json by running the REFERENCE's own functions.

TEST INFRASTRUCTURE ONLY.  Runs in the build container (needs /root/reference);
the fixtures it writes are committed so the GPU box (no reference tree) can check
the oracle and the HIP path against reference outputs.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

Every fixture stores the inputs *and* the reference outputs.  Inputs are seeded.
"""
from __future__ import annotations

import json
import os
import shutil
import sys
import tempfile

import numpy as np

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.ref_loader import ref_modules  # noqa: E402
from oracle import retrieval_np as orc  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


class FakeEmbedder:
    """Deterministic text → unit vector (md5-seeded gaussian); stands in for BGE,
    whose weights are not on disk.  Same class is re-implemented in tests/."""

    def __init__(self, dim=32):
        self.embedding_dim = dim
        self.calls = []

    def _vec(self, text):
        seed = int.from_bytes(__import__("hashlib").md5(text.encode()).digest()[:8], "little")
        v = np.random.default_rng(seed).standard_normal(self.embedding_dim).astype(np.float32)
        return v / np.linalg.norm(v)

    def batch_encode(self, texts, **kw):
        if isinstance(texts, str):
            texts = [texts]
        self.calls.append(list(texts))
        return np.stack([self._vec(t) for t in texts]).astype(np.float32)

    def encode(self, texts, **kw):
        import torch
        return torch.from_numpy(self.batch_encode(texts))


TRI_CORPUS = {
    "chunk": [f"chunk {i}: " + w for i, w in enumerate(
        ["cinders by the hearth", "a glass slipper on the stairs", "the ball at midnight", "two stepsisters at the mirror",
         "a pumpkin coach and six mice", "the prince's search through the town", "a hazel tree on the grave", "white doves at the window",
         "the stepmother's orders", "lentils in the ashes", "the third night of the feast", "the wedding procession"])],
    "entity": ["cinderella", "prince", "stepmother", "slipper", "pumpkin", "doves"],
    "fact": [str(t) for t in [("cinderella", "lost", "slipper"), ("prince", "found", "slipper"), ("stepmother", "hid", "cinderella"),
                              ("pumpkin", "became", "coach"), ("doves", "helped", "cinderella"), ("prince", "married", "cinderella"),
                              ("stepsisters", "envied", "cinderella")]],
    "summary": ["summary: the ball", "summary: the search", "summary: the wedding", "summary: the household"],
    "level_0": [f"timeline window {i}: " + w for i, w in enumerate(["before the ball", "the three nights", "midnight", "the search", "the fitting", "the wedding"])],
}
TRI_QUERIES = ["who lost a slipper?", "what became a coach?", "who helped cinderella?", "where did the doves sit?"]
TRI_CONFIG = {"need_cluster": True, "index_dtype": "f32", "linking_top_k": 5, "qa_ver_top_k": 5, "qa_sem_top_k": 2, "qa_epi_top_k": 3}
TRI_POOL = {"VER": [("chunk", 1), ("chunk", 5)], "SEM": [("summary", 1)], "EPI": [("level_0", 3)]}      # what the memory pool already holds: (store, text index)


def tri_bare_rag(ComoRAG, Store, emb, tmp):
    """A real ComoRAG instance without its constructor (LLM clients, igraph, OpenIE) carrying exactly what ComoRAG.tri_retrieve
    (ComoRAG.py:456-554) reads: five stores, the graph's vertex names, the timeline summarizer's embedder — and a reranker (an LLM
    filter in the reference, rerank.py) that keeps no fact, i.e. the loop's dense-retrieval branch (:489-491)."""
    import types
    st = {}
    for ns, texts in TRI_CORPUS.items():
        st[ns] = Store(emb, os.path.join(tmp, ns), 8, ns)
        st[ns].insert_strings(texts)
    rag = ComoRAG.__new__(ComoRAG)
    rag.global_config = types.SimpleNamespace(**TRI_CONFIG)
    rag.embedding_model = emb
    rag.ver_embedding_store, rag.entity_embedding_store = st["chunk"], st["entity"]
    rag.fact_embedding_store, rag.sem_embedding_store = st["fact"], st["summary"]
    rag.graph = types.SimpleNamespace(vs=[{"name": n} for n in st["entity"].get_all_ids() + st["chunk"].get_all_ids()])
    rag.ready_to_retrieve = False
    rag.level_store = st["level_0"]
    rag.timeline_summarizer = types.SimpleNamespace(summary_store=types.SimpleNamespace(embedding_model=emb))
    rag.rerank_facts = lambda query, scores: ([], [], {"facts_before_rerank": [], "facts_after_rerank": []})
    return rag, st


def tri_pool(NodeType, st):
    import types
    held = {getattr(NodeType, kind): [st[ns].text_to_hash_id[TRI_CORPUS[ns][i]] for ns, i in items] for kind, items in TRI_POOL.items()}
    return types.SimpleNamespace(get_all_hashes=lambda: held)


def golden_tri_retrieve(m):
    """tests/golden/tri_retrieve.json: what the reference's OWN tri_retrieve returns for a seeded toy corpus (real ComoRAG class, real
    EmbeddingStore, real get_similar_summaries, numpy scores; the fake embedder for BGE).  The fixture is data: texts in, texts out."""
    import tempfile, shutil
    tmp = tempfile.mkdtemp(prefix="cmr_tri_")
    try:
        emb = FakeEmbedder(32)
        rag, st = tri_bare_rag(m["ComoRAG"].ComoRAG, m["embedding_store"].EmbeddingStore, emb, tmp)
        pool = tri_pool(m["memory_utils"].NodeType, st)
        out = {"corpus": TRI_CORPUS, "queries": TRI_QUERIES, "config": TRI_CONFIG, "pool": {k: [list(x) for x in v] for k, v in TRI_POOL.items()}, "docs": []}
        for q in TRI_QUERIES:
            docs, _ = rag.tri_retrieve(q, pool)
            out["docs"].append(docs)
        with open(os.path.join(OUT, "tri_retrieve.json"), "w") as f:
            json.dump(out, f, indent=1)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    os.makedirs(OUT, exist_ok=True)
    m = ref_modules()
    if "--only-tri-retrieve" in sys.argv:
        golden_tri_retrieve(m)
        print("wrote", os.path.join(OUT, "tri_retrieve.json"))
        return
    ComoRAG = m["ComoRAG"].ComoRAG
    rng = np.random.default_rng(20250829)

    # ---- a1/a2: dense_passage_retrieval / get_fact_scores on a bare instance
    for tag, n, d, nq in (("small", 257, 48, 5), ("mid", 1061, 256, 4), ("d768", 300, 768, 3),
                          ("n2", 2, 16, 2)):
        X = orc.synthetic_corpus(n, d, seed=11)
        F = orc.synthetic_corpus(max(2, n // 2), d, seed=12)
        S = orc.synthetic_corpus(max(2, n // 3), d, seed=13)
        Q = orc.synthetic_queries(nq, d, seed=14, planted=X)
        obj = ComoRAG.__new__(ComoRAG)
        obj.passage_embeddings, obj.fact_embeddings, obj.summary_embeddings = X, F, S
        obj.query_to_embedding = {"triple": {}, "passage": {}}
        rec = {"X": X, "F": F, "S": S, "Q": Q}
        for i in range(nq):
            qn = f"q{i}"
            obj.query_to_embedding["triple"][qn] = Q[i:i + 1]
            obj.query_to_embedding["passage"][qn] = Q[i:i + 1]
            ids, sc = obj.dense_passage_retrieval(qn)
            ids_c, sc_c = obj.dense_passage_retrieval(qn, need_cluster=True)
            fs = obj.get_fact_scores(qn)
            rec[f"dpr_ids_{i}"], rec[f"dpr_scores_{i}"] = ids.astype(np.int64), sc
            rec[f"dprc_ids_{i}"], rec[f"dprc_scores_{i}"] = ids_c.astype(np.int64), sc_c
            rec[f"fact_scores_{i}"] = fs
        np.savez_compressed(os.path.join(OUT, f"dpr_{tag}.npz"), **rec)

    # ---- a3 min_max_normalize (both copies) incl. range-0
    mm = m["misc_utils"].min_max_normalize
    mm2 = m["embed_utils"].min_max_normalize
    v = rng.standard_normal(33).astype(np.float32)
    c = np.full(7, 0.25, dtype=np.float32)
    assert np.array_equal(mm(v), mm2(v)) and np.array_equal(mm(c), mm2(c))
    np.savez_compressed(os.path.join(OUT, "minmax.npz"), v=v, v_out=mm(v), c=c, c_out=mm(c))

    # ---- a7 compute_mdhash_id
    strs = ["", "Cinderella", "a glass slipper", "naïve café ☕", "x" * 1000]
    h = m["misc_utils"].compute_mdhash_id
    with open(os.path.join(OUT, "mdhash.json"), "w") as f:
        json.dump({"strings": strs,
                   "ids": [h(s, prefix=p) for s in strs for p in ("", "chunk-", "entity-")]}, f)

    # ---- a5 mean_pooling + F.normalize (torch CPU)
    import torch
    hid = rng.standard_normal((5, 19, 40)).astype(np.float32)
    mask = np.zeros((5, 19), dtype=np.int64)
    for b, l in enumerate((19, 1, 7, 12, 18)):
        mask[b, :l] = 1
    pooled = m["bge"].mean_pooling(torch.from_numpy(hid), torch.from_numpy(mask))
    normed = torch.nn.functional.normalize(pooled, p=2, dim=1)
    np.savez_compressed(os.path.join(OUT, "pool.npz"), hidden=hid, mask=mask,
                        pooled=pooled.numpy(), normed=normed.numpy())

    # ---- a10 retrieve_knn (torch CPU; no GPU visible here)
    E = orc.synthetic_corpus(300, 24, seed=21)
    E[17] = E[3]  # exact duplicate → tie
    ids = [f"entity-{i}" for i in range(len(E))]
    knn = m["embed_utils"].retrieve_knn(ids, ids, E, E, k=10, query_batch_size=64, key_batch_size=100)
    np.savez_compressed(os.path.join(OUT, "knn.npz"), E=E,
                        knn_ids=np.array([[int(s.split("-")[1]) for s in knn[q][0]] for q in ids]),
                        knn_scores=np.array([knn[q][1] for q in ids], dtype=np.float32))

    # ---- a6 EmbeddingStore behaviour + a9 get_similar_summaries + C1 cinderella plumbing
    Store = m["embedding_store"].EmbeddingStore
    tmp = tempfile.mkdtemp(prefix="golden_store_")
    try:
        emb = FakeEmbedder(32)
        st = Store(emb, tmp, 8, "chunk")
        batch1 = ["alpha", "beta", "alpha", "gamma"]
        batch2 = ["beta", "delta", "epsilon", "delta"]
        r1 = st.insert_strings(batch1)
        ids_after_1 = list(st.hash_ids)
        r2 = st.insert_strings(batch2)
        r3 = st.insert_strings(["alpha"])
        r4 = st.insert_strings([])
        missing = st.get_missing_string_hash_ids(["alpha", "zeta", "zeta"])
        st2 = Store(emb, tmp, 8, "chunk")  # reload from parquet
        rec = {
            "batch1": batch1, "batch2": batch2,
            "ids_after_1": ids_after_1, "ids_after_2": list(st.hash_ids), "texts": list(st.texts),
            "ret": [repr(r1), repr(r2), repr(r3), repr(r4)],
            "missing": missing, "encode_calls": emb.calls,
            "reload_ids": list(st2.hash_ids), "reload_texts": list(st2.texts),
            "reload_emb_type": type(st2.embeddings[0]).__name__,
            "reload_emb_dtype": str(st2.embeddings[0].dtype),
            "hash_id_to_idx": st.get_hash_id_to_order(),
        }
        E2 = st.get_embeddings(list(st.hash_ids))
        one = st.get_embedding(st.hash_ids[2])
        with open(os.path.join(OUT, "store.json"), "w") as f:
            json.dump(rec, f, indent=1)
        np.savez_compressed(os.path.join(OUT, "store_emb.npz"), all=E2, one=one)

        # get_similar_summaries over a level store
        lv = Store(emb, tmp, 8, "level_0")
        summaries = [f"summary window {i}: " + w for i, w in enumerate(
            ["the ball", "the stepmother", "the pumpkin coach", "midnight", "the slipper fits",
             "the prince searches", "the wedding"])]
        lv.insert_strings(summaries)
        emb.calls.clear()
        texts, scores = m["embed_utils"].get_similar_summaries("who lost a slipper?", lv, emb, top_k=3)
        with open(os.path.join(OUT, "summaries.json"), "w") as f:
            json.dump({"summaries": summaries, "query": "who lost a slipper?", "top_texts": texts,
                       "top_scores": scores, "encode_calls": emb.calls}, f, indent=1)

        # C1: cinderella corpus → store → dense retrieval top-5 (fake embedder)
        # same parsing as main_openai.py:13-19 (skip blank lines, docs = 'contents')
        cdir = "/root/reference/dataset/cinderella/cinderella_1"
        corpus = [json.loads(l) for l in open(f"{cdir}/corpus.jsonl", encoding="utf-8") if l.strip()]
        qas = [json.loads(l) for l in open(f"{cdir}/qas.jsonl", encoding="utf-8") if l.strip()]
        docs = [d["contents"] for d in corpus]
        cs = Store(emb, tmp, 8, "cinder")
        cs.insert_strings(docs)
        keys = list(cs.get_all_ids())
        obj = ComoRAG.__new__(ComoRAG)
        obj.passage_embeddings = np.array(cs.get_embeddings(keys))
        obj.query_to_embedding = {"triple": {}, "passage": {}}
        obj.embedding_model = emb
        out = {"doc_md5": [orc.compute_mdhash_id(d) for d in docs], "n_docs": len(docs),
               "keys": keys, "questions": [], "top5_ids": [], "top5_scores": [],
               # the fake embedder's vectors of the six chunks / three questions, so the GPU tier (no reference tree,
               # hence no chunk texts) can rebuild the same index
               "doc_vecs": np.array(cs.get_embeddings(keys)).tolist(),
               "question_vecs": [emb._vec(qa["question"]).tolist() for qa in qas]}
        for qa in qas:
            q = qa["question"]
            ids_, sc_ = obj.dense_passage_retrieval(q)
            out["questions"].append(q)
            out["top5_ids"].append(ids_[:5].tolist())
            out["top5_scores"].append(sc_[:5].tolist())
        with open(os.path.join(OUT, "cinderella.json"), "w") as f:
            json.dump(out, f, indent=1)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)

    # ---- a11 MemoryPool.retrieve_similar_nodes numeric part
    mu = m["memory_utils"]
    pool = mu.MemoryPool.__new__(mu.MemoryPool)
    emb = FakeEmbedder(32)
    pool.embedding_model = emb
    nodes = []
    for i in range(9):
        nd = mu.MemoryNode.__new__(mu.MemoryNode)
        nd.probe, nd.cue, nd.embedding = f"probe {i % 3}", f"cue number {i}", None
        nodes.append(nd)
    nodes[5].cue = nodes[2].cue  # identical content → identical similarity (stable order)
    pool.pool = nodes
    sel = pool.retrieve_similar_nodes("probe 1", top_percent=0.5)
    with open(os.path.join(OUT, "mempool.json"), "w") as f:
        json.dump({"contents": [f"{n.probe} {n.cue}" for n in nodes], "probe": "probe 1",
                   "selected": [nodes.index(s) for s in sel]}, f, indent=1)

    golden_tri_retrieve(m)
    print("golden fixtures written to", OUT, sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
