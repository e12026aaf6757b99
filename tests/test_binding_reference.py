"""The binding glue (comorag_amd/hooks.py, embedding_store.py, retrieval.py) on the REAL reference classes — CPU tier.

Needs the reference tree (this container; the GPU box has none, tests skip there — its twins with the HIP index are in
tests/test_dropin_gpu.py).  The device index is replaced by conftest.NumpyIndex through the hooks' `index_factory`
seam, so what is tested here is exactly the glue: which names get rebound, on which objects, with which semantics.
    ComoRAG.py:92-124   five stores + embedding-model factory          (module aliasing)
    ComoRAG.py:876-967  prepare_retrieval_objects / get_query_embeddings / get_fact_scores / dense_passage_retrieval
    utils/memory_utils.py:149-235  MemoryPool.compute_probe_note_embeddings / retrieve_similar_nodes
    main_openai.py:8-44 → BASELINE config 1 (cinderella, fake embedder): index + dense retrieval top-5
"""
import json
import os
import subprocess
import sys
import types

import numpy as np
import pytest

from oracle.ref_loader import REFERENCE_ROOT, reference_available

pytestmark = pytest.mark.skipif(not reference_available(), reason="reference tree not present")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_patch_reference_modules_binds_the_real_comorag_module():
    """alias the three replaced modules, THEN import the reference's ComoRAG.  Runs in a fresh
    interpreter (module aliasing is process-global)."""
    code = r'''
import sys
sys.path.insert(0, %r)
from oracle import ref_loader
ref_loader.prepare_stubs()
from comorag_amd import hooks, embedding_store, embedding_model, retrieval
hooks.patch_reference_modules("src.comorag")
import importlib
C = importlib.import_module("src.comorag.ComoRAG")
assert C.EmbeddingStore is embedding_store.EmbeddingStore, C.EmbeddingStore
assert C._get_embedding_model_class is embedding_model._get_embedding_model_class
assert C.retrieve_knn is retrieval.retrieve_knn
tl = importlib.import_module("src.comorag.utils.timeline_utils")        # the other EmbeddingStore consumer
assert tl.EmbeddingStore is embedding_store.EmbeddingStore
assert getattr(tl, "get_similar_summaries", retrieval.get_similar_summaries) is retrieval.get_similar_summaries
# untouched: everything else is still the reference's
assert C.ComoRAG.__module__ == "src.comorag.ComoRAG" and C.MemoryPool.__module__ == "src.comorag.utils.memory_utils"
print("BOUND")
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env={**os.environ, "PYTHONDONTWRITEBYTECODE": "1", "COMORAG_HIP_NO_TORCH": "1"})
    assert r.returncode == 0 and "BOUND" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def _stores(tmp_path, emb, Store):
    chunks = [f"chunk {i}: " + w for i, w in enumerate(["cinders", "a glass slipper", "the ball at midnight", "two stepsisters",
                                                         "a pumpkin coach", "the prince's search", "a hazel tree", "white doves"])]
    ents = ["cinderella", "prince", "stepmother", "slipper", "pumpkin"]
    facts = [str(t) for t in [("cinderella", "lost", "slipper"), ("prince", "found", "slipper"), ("stepmother", "hid", "cinderella"),
                              ("pumpkin", "became", "coach"), ("doves", "helped", "cinderella"), ("prince", "married", "cinderella")]]
    sums = ["summary: the ball", "summary: the search", "summary: the wedding"]
    out = {}
    for ns, texts in (("chunk", chunks), ("entity", ents), ("fact", facts), ("summary", sums)):
        st = Store(emb, str(tmp_path / f"{ns}_{Store.__module__.split('.')[0]}"), 8, ns)
        st.insert_strings(texts)
        out[ns] = st
    return out


def _bare_rag(ComoRAG, stores, emb):
    """A real ComoRAG instance without its constructor (LLM clients, igraph): only what :876-967 read."""
    rag = ComoRAG.__new__(ComoRAG)
    rag.global_config = types.SimpleNamespace(need_cluster=True, index_dtype="f32")
    rag.embedding_model = emb
    rag.ver_embedding_store, rag.entity_embedding_store = stores["chunk"], stores["entity"]
    rag.fact_embedding_store, rag.sem_embedding_store = stores["fact"], stores["summary"]
    names = stores["entity"].get_all_ids() + stores["chunk"].get_all_ids()
    rag.graph = types.SimpleNamespace(vs=[{"name": n} for n in names])        # igraph's vertex sequence, as far as :889 goes
    rag.ready_to_retrieve = False
    return rag


def test_install_on_a_real_comorag_instance(tmp_path, fake_embedder, numpy_index_cls):
    from oracle.ref_loader import ref_modules
    m = ref_modules()
    ComoRAG = m["ComoRAG"].ComoRAG
    from comorag_amd import hooks
    from comorag_amd.embedding_store import EmbeddingStore

    built = []

    def factory(mat, dtype, device):
        mat = np.asarray(mat, np.float32)
        if mat.ndim != 2 or len(mat) == 0:
            return None
        ix = numpy_index_cls(mat.shape[1], dtype, device)
        ix.append(mat)
        built.append(ix)
        return ix

    ours = _bare_rag(ComoRAG, _stores(tmp_path, fake_embedder, EmbeddingStore), fake_embedder)
    ref = _bare_rag(ComoRAG, _stores(tmp_path, fake_embedder, m["embedding_store"].EmbeddingStore), fake_embedder)
    hooks.install(ours, index_factory=factory, patch_module_functions=False)
    assert ours.prepare_retrieval_objects.__func__.__module__ == "comorag_amd.hooks"
    ours.prepare_retrieval_objects()                 # runs the reference's own :876-907 inside, then builds the indexes
    ours.prepare_retrieval_objects()                 # once only
    ref.prepare_retrieval_objects()
    assert len(built) == 3 and ours.ready_to_retrieve
    assert ours.passage_node_keys == ref.passage_node_keys and ours.fact_node_keys == ref.fact_node_keys
    assert np.array_equal(ours.passage_embeddings, ref.passage_embeddings) and np.array_equal(ours.fact_embeddings, ref.fact_embeddings)
    assert ours.passage_node_idxs == ref.passage_node_idxs
    for q in ["who lost a slipper?", "what became a coach?", "who helped cinderella?"]:
        n0 = len(fake_embedder.calls)
        ours.get_query_embeddings(q)                 # tri_retrieve passes a str (:470): encoded once per kind, not per character
        assert fake_embedder.calls[n0:] == [[q], [q]]
        a_ids, a_sc = ours.dense_passage_retrieval(q)
        assert len(fake_embedder.calls) == n0 + 2    # memoised
        b_ids, b_sc = ref.dense_passage_retrieval(q)
        assert a_ids.tolist() == b_ids.tolist()
        np.testing.assert_allclose(a_sc, b_sc, atol=2e-6)
        np.testing.assert_allclose(ours.get_fact_scores(q), ref.get_fact_scores(q), atol=2e-6)
        c_ids, c_sc = ours.dense_passage_retrieval(q, need_cluster=True)
        d_ids, d_sc = ref.dense_passage_retrieval(q, need_cluster=True)
        assert c_ids.tolist() == d_ids.tolist()
        np.testing.assert_allclose(c_sc, d_sc, atol=2e-6)


def test_install_memory_pool_on_a_real_pool(fake_embedder, numpy_index_cls):
    """BASELINE config 4's shape on the reference's own MemoryPool: three cycles of (add nodes, retrieve); the rebound
    method must select the nodes the reference method selects on an identical pool, appending only the new rows."""
    from oracle.ref_loader import ref_modules
    mu = ref_modules()["memory_utils"]
    from comorag_amd import hooks

    def mk():
        return mu.MemoryPool(embedding_model=fake_embedder, agent=None)

    ours, ref = mk(), mk()
    appended = []

    class Ix(numpy_index_cls):
        def append(self, rows):
            appended.append(len(np.asarray(rows).reshape(-1, self.dim)))
            super().append(rows)

    hooks.install_memory_pool(ours, index_factory=lambda dim, dtype, device: Ix(dim, dtype, device))
    assert ours.retrieve_similar_nodes.__func__.__module__ == "comorag_amd.hooks"
    cue = 0
    for cycle, probe in enumerate(["where is the slipper?", "who is the stepmother?", "what happened at midnight?"]):
        for pool in (ours, ref):
            c = cue
            for kind in (mu.NodeType.VER, mu.NodeType.SEM, mu.NodeType.EPI):
                for j in range(2):
                    pool.add_to_temp_pool(mu.MemoryNode(probe=probe, node_type=kind, original_content=[f"text {c}"], cue=f"cue number {c}"))
                    c += 1
            if cycle == 1:          # identical content twice: equal similarity, the stable sort keeps pool order
                pool.add_to_temp_pool(mu.MemoryNode(probe=probe, node_type=mu.NodeType.SEM, original_content=["dup"], cue=f"cue number {cue}"))
            pool.merge_temp_to_main()
        cue = c
        for pct in (0.5, 0.2, 1.0):
            a = ours.retrieve_similar_nodes(probe, top_percent=pct)
            b = ref.retrieve_similar_nodes(probe, top_percent=pct)
            assert [ours.pool.index(n) for n in a] == [ref.pool.index(n) for n in b], (cycle, pct)
        ours.add_fused_node(probe, f"fused note {cycle}", a); ref.add_fused_node(probe, f"fused note {cycle}", b)
        ours.merge_temp_to_main(); ref.merge_temp_to_main()
    assert appended == [6, 8, 7] and len(ours._hip_state["rows"]) == len(ours.pool) - 1      # the last fused node is not indexed yet


def test_config1_cinderella_plumbing(tmp_path, golden_dir, fake_embedder, numpy_index_cls):
    """BASELINE config 1: the cinderella corpus through OUR EmbeddingStore (md5 ids, dedup, order) and OUR
    dense_passage_retrieval; ids and normalised scores must be the ones the reference produced (tests/golden/
    cinderella.json: reference store + reference ComoRAG.dense_passage_retrieval, oracle/make_golden.py)."""
    from comorag_amd import retrieval
    from comorag_amd.embedding_store import EmbeddingStore
    c = json.load(open(os.path.join(golden_dir, "cinderella.json")))
    cdir = os.path.join(REFERENCE_ROOT, "dataset", "cinderella", "cinderella_1")
    docs = [json.loads(l)["contents"] for l in open(os.path.join(cdir, "corpus.jsonl"), encoding="utf-8") if l.strip()]   # main_openai.py:13-19
    st = EmbeddingStore(fake_embedder, str(tmp_path / "cinder"), 8, "cinder")
    st.insert_strings(docs)
    st.insert_strings(docs[:2])                                    # re-insert: nothing new
    keys = st.get_all_ids()
    assert keys == c["keys"] and len(keys) == c["n_docs"]
    rows = np.asarray(st.get_embeddings(keys), np.float32)
    np.testing.assert_array_equal(rows, np.asarray(c["doc_vecs"], np.float32))
    ix = numpy_index_cls(rows.shape[1]); ix.append(rows)
    for q, want_ids, want_sc, qv in zip(c["questions"], c["top5_ids"], c["top5_scores"], c["question_vecs"]):
        vec = fake_embedder.batch_encode(q)
        np.testing.assert_array_equal(vec[0], np.asarray(qv, np.float32))
        ids, sc = retrieval.dense_passage_retrieval(ix, vec)
        assert ids[:5].tolist() == want_ids
        np.testing.assert_allclose(sc[:5], want_sc, atol=2e-6)
        tid, tsc = retrieval.dense_passage_topk(ix, vec, 5)
        assert tid[0].tolist() == want_ids
        np.testing.assert_allclose(tsc[0], want_sc, atol=2e-6)


class _VS:
    """igraph's vertex sequence as ComoRAG uses it: iteration yields vertices (v["name"], ComoRAG.py:890), vs["name"] the list (:1005)."""

    def __init__(self, names):
        self._n = names

    def __iter__(self):
        return iter([{"name": n} for n in self._n])

    def __getitem__(self, k):
        assert k == "name"
        return self._n

    def __len__(self):
        return len(self._n)


class _FakeIGraph:
    """What ComoRAG touches of its igraph.Graph on the PPR path: vs['name'], vcount, get_edgelist, es['weight'], and
    personalized_pagerank (ComoRAG.py:1092-1099) — the latter answered by the oracle's direct solve."""

    def __init__(self, names, src, dst, w):
        self._names, self._src, self._dst, self._w = list(names), list(src), list(dst), list(w)
        self.vs = _VS(self._names)
        self.es = {"weight": self._w}

    def vcount(self):
        return len(self._names)

    def get_edgelist(self):
        return list(zip(self._src, self._dst))

    def personalized_pagerank(self, vertices=None, damping=0.85, directed=True, weights=None, reset=None, implementation="prpack"):
        from oracle import ppr_np
        assert directed is False and weights == "weight" and implementation == "prpack"
        return ppr_np.personalized_pagerank(len(self._names), self._src, self._dst, self._w, reset, damping).tolist()


class _NumpyGraph:
    """numpy stand-in for comorag_amd.ppr.DeviceGraph (conftest.NumpyIndex's sibling): same call surface, oracle arithmetic."""

    def __init__(self, g, device=0):
        e = g.get_edgelist()
        self.n_vertices, self._src, self._dst, self._w = g.vcount(), [a for a, _ in e], [b for _, b in e], list(g.es["weight"])
        self.rows = None

    def set_passage_vertices(self, idxs):
        self.rows = list(idxs); self.n_rows = len(self.rows)

    def ppr(self, reset, damping=0.5, **kw):
        from oracle import ppr_np
        return ppr_np.personalized_pagerank(self.n_vertices, self._src, self._dst, self._w, reset, damping)

    def passage_scores(self, index, q, phrase_w, pnw, damping):          # what cmr_index_ppr fuses on the device
        from oracle import ppr_np
        from oracle import retrieval_np as orc
        s = index.scores(np.asarray(q, np.float32).reshape(1, -1))[0]
        r = np.asarray(phrase_w, np.float64).copy()
        r[self.rows] = orc.min_max_normalize(s).astype(np.float64) * pnw
        return self.ppr(r, damping)[self.rows]


def test_device_ppr_hooks_on_a_real_comorag_instance(tmp_path, fake_embedder, numpy_index_cls):
    """run_ppr and graph_search_with_fact_entities rebound on a real ComoRAG instance (§8 f4's glue): same ranking and
    scores as the reference's own methods over a graph whose personalized_pagerank the oracle answers."""
    from oracle.ref_loader import ref_modules
    m = ref_modules()
    ComoRAG = m["ComoRAG"].ComoRAG
    mdhash = m["misc_utils"].compute_mdhash_id
    from comorag_amd import hooks
    from comorag_amd.embedding_store import EmbeddingStore

    def factory(mat, dtype, device):
        mat = np.asarray(mat, np.float32)
        if mat.ndim != 2 or len(mat) == 0:
            return None
        ix = numpy_index_cls(mat.shape[1], dtype, device); ix.append(mat)
        return ix

    def build(Store, sub):
        st = _stores(tmp_path / sub, fake_embedder, Store)
        rag = _bare_rag(ComoRAG, st, fake_embedder)
        names = st["entity"].get_all_ids() + st["chunk"].get_all_ids()
        ne, nc = len(st["entity"].get_all_ids()), len(st["chunk"].get_all_ids())
        rng = np.random.default_rng(5)
        src = [int(rng.integers(0, ne)) for _ in range(3 * nc)] + [0, 1, 2]
        dst = [ne + i // 3 for i in range(3 * nc)] + [1, 2, 3]
        w = rng.uniform(0.5, 2.0, len(src)).tolist()
        rag.graph = _FakeIGraph(names, src, dst, w)
        rag.ent_node_to_num_chunk = {k: (i % 3) for i, k in enumerate(st["entity"].get_all_ids())}
        return rag

    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    ours = build(EmbeddingStore, "a")
    ref = build(m["embedding_store"].EmbeddingStore, "b")
    hooks.install(ours, index_factory=factory, graph_factory=lambda g, device: _NumpyGraph(g, device), patch_module_functions=False)
    ours.prepare_retrieval_objects(); ref.prepare_retrieval_objects()
    assert isinstance(ours._hip["graph"], _NumpyGraph) and ours._hip["graph"].rows == ref.passage_node_idxs
    nv = ref.graph.vcount()
    reset = np.zeros(nv); reset[[0, 3, nv - 1]] = [1.0, 0.5, 0.25]; reset[2] = -1.0
    a_ids, a_sc = ours.run_ppr(reset.copy(), damping=0.5)
    b_ids, b_sc = ref.run_ppr(reset.copy(), damping=0.5)
    assert a_ids.tolist() == b_ids.tolist()
    np.testing.assert_allclose(a_sc, b_sc, atol=1e-12)
    facts = [("Cinderella", "lost", "Slipper"), ("prince", "found", "slipper"), ("stepmother", "hid", "pumpkin")]   # phrases are lower-cased (:1005-1007)
    for q in ["who lost a slipper?", "what became a coach?"]:
        fs = ref.get_fact_scores(q)
        idxs = np.argsort(fs)[-3:][::-1].tolist()                 # rerank_facts (:1073): the link_top_k best facts, all with a positive score
        want = ref.graph_search_with_fact_entities(q, 3, fs, facts, idxs, passage_node_weight=0.05)
        got = ours.graph_search_with_fact_entities(q, 3, ours.get_fact_scores(q), facts, idxs, passage_node_weight=0.05)
        assert got[0].tolist() == want[0].tolist() and got[2] == want[2]
        np.testing.assert_allclose(got[1], want[1], atol=1e-9)


def test_embedding_cache_files_are_interchangeable_with_the_reference_wrapper(tmp_path):
    """a14 make_cache_embed (embedding_model/base.py:112-187): a cache file written by the reference's wrapper must hit
    in ours and vice versa (same sha256 key over {"instruction", "promps" [sic], "max_length"}, same fp32 blobs)."""
    import torch
    from oracle.ref_loader import ref_modules
    ref_make = ref_modules()["emb_base"].make_cache_embed
    from comorag_amd.embedding_model.base import make_cache_embed as our_make
    calls = []

    def enc(**kw):
        calls.append(list(kw["prompts"]))
        return torch.tensor([[float(len(p)), float(ord(p[0])), 0.5] for p in kw["prompts"]])

    f1, f2 = str(tmp_path / "ref.db"), str(tmp_path / "ours.db")
    ref_w, our_w = ref_make(enc, f1, "cpu"), our_make(enc, f1, "cpu")
    a = ref_w(prompts=["alpha", "be"], instruction="I", max_length=64)          # reference writes
    calls.clear()
    b = our_w(prompts=["be", "gamma", "alpha"], instruction="I", max_length=64)  # ours reads its entries, adds one
    assert calls == [["gamma"]]
    assert torch.equal(b[0], a[1]) and torch.equal(b[2], a[0]) and b.shape == (3, 3)
    calls.clear()
    c = ref_w(prompts=["gamma", "alpha"], instruction="I", max_length=64)        # reference reads ours
    assert calls == [] and torch.equal(c[0], b[1])
    our_w2, ref_w2 = our_make(enc, f2, "cpu"), ref_make(enc, f2, "cpu")          # and a file created by ours
    d = our_w2(prompts=["delta"], instruction="", max_length=16)
    calls.clear()
    e = ref_w2(prompts=["delta"], instruction="", max_length=16)
    assert calls == [] and torch.equal(d, e)
    our_w2(prompts=["delta"], instruction="other", max_length=16)                # a different instruction is a different key
    assert calls == [["delta"]]


def test_the_real_tri_retrieve_loop_runs_on_the_hooks(tmp_path, golden_dir, fake_embedder, numpy_index_cls, monkeypatch):
    """ComoRAG.tri_retrieve (ComoRAG.py:456-554) — the reference's OWN method, unmodified — on an instance whose stores are this package's
    EmbeddingStore and whose numeric call sites are rebound by hooks.install (the device index replaced by conftest.NumpyIndex, the module's
    get_similar_summaries by retrieval's): the three layers it returns must be the ones the untouched reference returns on its own stores,
    live and as recorded in tests/golden/tri_retrieve.json (the fixture the GPU tier checks the HIP index against)."""
    from oracle.ref_loader import ref_modules
    from oracle.make_golden import tri_bare_rag, tri_pool, TRI_QUERIES
    from comorag_amd import hooks
    from comorag_amd.embedding_store import EmbeddingStore
    m = ref_modules()
    ComoRAG, NodeType = m["ComoRAG"].ComoRAG, m["memory_utils"].NodeType
    gold = json.load(open(os.path.join(golden_dir, "tri_retrieve.json")))
    assert gold["queries"] == TRI_QUERIES

    def factory(mat, dtype, device):
        mat = np.asarray(mat, np.float32)
        if mat.ndim != 2 or len(mat) == 0:
            return None
        ix = numpy_index_cls(mat.shape[1], dtype, device)
        ix.append(mat)
        return ix

    # (the level store's HBM mirror — EmbeddingStore.device_index, what get_similar_summaries scores against — is the numpy stand-in too)
    monkeypatch.setattr("comorag_amd.multi_index.make_index", lambda dim, dtype, **kw: numpy_index_cls(dim, dtype, kw.get("device", 0)))
    ref, ref_st = tri_bare_rag(ComoRAG, m["embedding_store"].EmbeddingStore, fake_embedder, str(tmp_path / "ref"))
    ours, our_st = tri_bare_rag(ComoRAG, EmbeddingStore, fake_embedder, str(tmp_path / "ours"))
    mod = sys.modules[ComoRAG.__module__]
    keep = mod.get_similar_summaries
    try:
        want = [ref.tri_retrieve(q, tri_pool(NodeType, ref_st))[0] for q in TRI_QUERIES]          # the untouched reference, live
        assert want == gold["docs"]
        hooks.install(ours, index_factory=factory, patch_module_functions=True)                        # rebinds the module's get_similar_summaries too
        assert mod.get_similar_summaries is not keep
        got = [ours.tri_retrieve(q, tri_pool(NodeType, our_st))[0] for q in TRI_QUERIES]
        assert got == gold["docs"]
        assert ours.tri_retrieve.__func__ is ComoRAG.tri_retrieve                                      # the loop itself is the reference's
    finally:
        mod.get_similar_summaries = keep
        if hasattr(mod, "retrieve_knn"):
            mod.retrieve_knn = m["embed_utils"].retrieve_knn
