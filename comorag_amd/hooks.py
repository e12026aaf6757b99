"""Put an UNMODIFIED reference `ComoRAG` instance on the HIP engine.

`install(rag)` rebinds, on that instance (and on the two helper names ComoRAG.py imported at module
scope), exactly the numeric call sites of SURVEY.md §8a:
    prepare_retrieval_objects   ComoRAG.py:876-907   → also builds the HBM indexes, once, under a lock
    get_query_embeddings        :909-935             → memoises the full query string
    get_fact_scores             :937-948
    dense_passage_retrieval     :950-967
    retrieve_knn                utils/embed_utils.py:8-97     (name in the ComoRAG module)
    get_similar_summaries       utils/embed_utils.py:109-161  (name in the ComoRAG module)
    MemoryPool.retrieve_similar_nodes  utils/memory_utils.py:188-235 (instance-level, optional)
    run_ppr                     ComoRAG.py:1086-1105         → power iteration on a CSR copy of the graph in HBM (no igraph call)
    graph_search_with_fact_entities  :992-1053               → its passage loop + run_ppr fused on the device (cmr_index_ppr)
Everything else (LLM calls, graph construction, clustering) keeps running the reference's code.
"""
from __future__ import annotations

import sys
import threading
import types
from typing import Optional

import numpy as np

from . import retrieval
from .index import DenseIndex


def patch_reference_modules(package: str = "src.comorag") -> None:
    """Alias the three replaced reference modules to this package BEFORE `package`.ComoRAG is
    imported, so its `from .embedding_store import EmbeddingStore`, `from .embedding_model import
    _get_embedding_model_class` and `from .utils.embed_utils import ...` bind to the HIP-backed
    classes/functions.  ComoRAG.py itself is not edited."""
    import importlib
    from . import embedding_model, embedding_store, retrieval
    sys.modules[f"{package}.embedding_store"] = embedding_store
    sys.modules[f"{package}.embedding_model"] = embedding_model
    shim = types.ModuleType(f"{package}.utils.embed_utils")
    shim.retrieve_knn = retrieval.retrieve_knn
    shim.get_similar_summaries = retrieval.get_similar_summaries
    shim.min_max_normalize = retrieval.min_max_normalize
    shim.EmbeddingStore = embedding_store.EmbeddingStore
    sys.modules[f"{package}.utils.embed_utils"] = shim
    importlib.invalidate_caches()


def _matrix_index(mat, dtype: str, device: int, num_shards=None, devices=None, options=None):
    mat = np.asarray(mat, dtype=np.float32)
    if mat.ndim != 2 or mat.shape[0] == 0:
        return None
    from .multi_index import make_index
    idx = make_index(mat.shape[1], dtype, device=device, capacity_hint=mat.shape[0], num_shards=num_shards, devices=devices, options=options)
    idx.append(mat)
    return idx


def install(rag, index_dtype: Optional[str] = None, device: int = 0, patch_module_functions: bool = True, index_factory=None,
            graph_factory=None, ppr_on_device: bool = True, knn_threshold_filter: bool = True, num_shards: Optional[int] = None,
            devices=None):
    """`num_shards` / `devices` (default: `global_config.num_shards` / `.devices`, i.e. one shard): more than one shard puts
    the passage / fact / summary matrices on a `MultiDeviceIndex` — row shards over the node's GPUs driven from this one
    process, same results (comorag_amd/multi_index.py).
    `index_factory(matrix, dtype, device) -> index` builds the HBM mirror of a host matrix (default: a `DenseIndex`
    filled with `append`); anything with DenseIndex's `scores` / `search` / `sorted_scores` / `__len__` serves — the
    CPU-tier binding tests pass a numpy stand-in there to exercise this glue on the real reference classes without a GPU.
    `graph_factory(igraph_like, device) -> graph` likewise (default `comorag_amd.ppr.DeviceGraph.from_igraph`); it must offer
    `set_passage_vertices(idxs)` and `ppr(reset, damping)`; the fused path additionally goes through
    `comorag_amd.ppr.ppr_passage_scores` unless the graph object brings its own `passage_scores(index, q, phrase_w, pnw, damping)`."""
    cfg = getattr(rag, "global_config", None)
    dtype = index_dtype or getattr(cfg, "index_dtype", None) or "f32"
    n_sh = num_shards if num_shards is not None else getattr(cfg, "num_shards", None)
    devs = devices if devices is not None else getattr(cfg, "devices", None)
    opts = getattr(cfg, "index_options", None)        # route selectors for every index built here (DenseIndex.set_option / "append_block_rows")
    make_index = index_factory or (lambda mat, dt, dev: _matrix_index(mat, dt, dev, n_sh, devs, opts))
    lock = threading.Lock()
    orig_prepare = rag.prepare_retrieval_objects
    rag._hip = {"passage": None, "summary": None, "fact": None, "graph": None, "dtype": dtype}

    def _make_graph(self):
        g = getattr(self, "graph", None)
        if not ppr_on_device or g is None or not hasattr(g, "get_edgelist"):
            return None
        if graph_factory is not None:
            dg = graph_factory(g, device)
        else:
            from .ppr import DeviceGraph
            dg = DeviceGraph.from_igraph(g, device=device)
        dg.set_passage_vertices(self.passage_node_idxs)
        return dg

    def prepare_retrieval_objects(self):
        with lock:                                    # once-only (the reference races here, :467-468)
            if getattr(self, "ready_to_retrieve", False) and self._hip["passage"] is not None:
                return
            orig_prepare()
            self._hip["passage"] = make_index(self.passage_embeddings, dtype, device)
            self._hip["fact"] = make_index(self.fact_embeddings, dtype, device)
            if getattr(self.global_config, "need_cluster", False) and hasattr(self, "summary_embeddings"):
                self._hip["summary"] = make_index(self.summary_embeddings, dtype, device)
            self._hip["graph"] = _make_graph(self)

    def _query_vec(self, kind: str, query: str, instruction_key: str):
        vec = self.query_to_embedding[kind].get(query, None)
        if vec is None:
            from importlib import import_module
            gqi = import_module(type(self).__module__).get_query_instruction
            vec = self.embedding_model.batch_encode(query, instruction=gqi(instruction_key), norm=True)
            self.query_to_embedding[kind][query] = vec      # memoise the FULL string (fixes :470 waste)
        return vec

    def get_query_embeddings(self, queries):
        if isinstance(queries, str):                  # tri_retrieve passes a str (:470): encode it once,
            queries = [queries]                       # not per character — same cached vectors result
        kinds = (("triple", "query_to_fact"), ("passage", "query_to_passage"))
        for q in queries:
            q = getattr(q, "question", q)
            em = self.embedding_model
            if getattr(em, "instruction_is_ignored", False) and all(self.query_to_embedding[k].get(q, None) is None for k, _ in kinds):
                # The reference encodes a new question once per instruction (:937-948) — and its BGE model ignores the instruction it is handed
                # (BGEEmbedding.py:150-155 overwrite it with the fixed prefix because no caller passes is_query; HipBGEEmbeddingModel keeps the
                # quirk and says so): both calls run the same prompt through the same deterministic forward.  One forward serves both kinds,
                # bit for bit what two would return — a short query's encode is ~84 dependent launches, 0.75 ms at BGE-base.
                from importlib import import_module
                gqi = import_module(type(self).__module__).get_query_instruction
                vec = em.batch_encode(q, instruction=gqi(kinds[0][1]), norm=True)
                for k, _ in kinds:
                    self.query_to_embedding[k][q] = vec
                continue
            for k, key in kinds:
                _query_vec(self, k, q, key)

    def get_fact_scores(self, query: str) -> np.ndarray:
        return retrieval.get_fact_scores(self._hip["fact"], _query_vec(self, "triple", query, "query_to_fact"))

    def dense_passage_retrieval(self, query: str, need_cluster: bool = False):
        idx = self._hip["summary"] if need_cluster else self._hip["passage"]
        return retrieval.dense_passage_retrieval(idx, _query_vec(self, "passage", query, "query_to_passage"))

    orig_run_ppr = getattr(rag, "run_ppr", None)
    orig_graph_search = getattr(rag, "graph_search_with_fact_entities", None)

    def run_ppr(self, reset_prob, damping: float = 0.5):
        g = self._hip["graph"]
        if g is None:
            return orig_run_ppr(reset_prob, damping)
        from . import ppr
        return ppr.run_ppr(g, reset_prob, self.passage_node_idxs, damping)

    def graph_search_with_fact_entities(self, query, link_top_k, query_fact_scores, top_k_facts, top_k_fact_indices, passage_node_weight: float = 0.05):
        """ComoRAG.py:992-1053.  Phrase weights as the reference computes them (fact score of the facts that mention the
        phrase, divided by the number of chunks the entity occurs in, top `link_top_k` phrases kept by the reference's own
        get_top_k_weights); the passage loop + run_ppr run fused on the device.  (The text -> score map the reference also
        fills for every passage is dropped there after trimming: dead code, not rebuilt.)"""
        g, index = self._hip["graph"], self._hip["passage"]
        if g is None or index is None:
            return orig_graph_search(query, link_top_k, query_fact_scores, top_k_facts, top_k_fact_indices, passage_node_weight)
        from importlib import import_module
        mdhash = import_module(type(self).__module__).compute_mdhash_id
        phrase_weights = np.zeros(len(self.node_name_to_vertex_idx) if not hasattr(g, "n_vertices") else g.n_vertices)
        seen: dict = {}
        used_phrases_with_scores = {}
        for rank, fact in enumerate(top_k_facts):
            fact_score = query_fact_scores[top_k_fact_indices[rank]] if query_fact_scores.ndim > 0 else query_fact_scores
            for phrase in (fact[0].lower(), fact[2].lower()):
                key = mdhash(content=phrase, prefix="entity-")
                vid = self.node_name_to_vertex_idx.get(key, None)
                if vid is not None:
                    phrase_weights[vid] = fact_score          # float64 slot first, THEN the division: as ComoRAG.py:1019-1021
                    if self.ent_node_to_num_chunk[key] != 0:
                        phrase_weights[vid] /= self.ent_node_to_num_chunk[key]
                    if phrase_weights[vid] > 0:
                        used_phrases_with_scores[phrase] = phrase_weights[vid]
                seen.setdefault(phrase, []).append(fact_score)
        linking_score_map = {p: float(np.mean(v)) for p, v in seen.items()}
        if link_top_k:
            phrase_weights, linking_score_map = self.get_top_k_weights(link_top_k, phrase_weights, linking_score_map)
        q = _query_vec(self, "passage", query, "query_to_passage")
        # ComoRAG.py:1051 `assert sum(node_weights) > 0`: the passage part sums to > 0 exactly when there are passages and a
        # positive passage_node_weight (the best passage normalises to 1.0), so the test needs no score from the device
        assert phrase_weights.sum() > 0 or (len(index) > 0 and passage_node_weight > 0), \
            f'No phrases found in the graph for the given facts: {top_k_facts}'
        if hasattr(g, "passage_scores"):
            doc_scores = g.passage_scores(index, q, phrase_weights, passage_node_weight, 0.5)
        else:
            from . import ppr
            doc_scores = ppr.ppr_passage_scores(index, g, q, phrase_weights, passage_node_weight, 0.5)
        sorted_doc_ids = np.argsort(doc_scores)[::-1]
        sorted_doc_scores = doc_scores[sorted_doc_ids.tolist()]
        assert len(sorted_doc_ids) == len(self.passage_node_idxs)
        return sorted_doc_ids, sorted_doc_scores, used_phrases_with_scores

    fns = [prepare_retrieval_objects, get_query_embeddings, get_fact_scores, dense_passage_retrieval]
    if ppr_on_device and orig_run_ppr is not None:
        fns.append(run_ppr)
    if ppr_on_device and orig_graph_search is not None:
        fns.append(graph_search_with_fact_entities)
    for fn in fns:
        setattr(rag, fn.__name__, types.MethodType(fn, rag))

    if patch_module_functions:
        mod = sys.modules.get(type(rag).__module__)
        if mod is not None:
            def _knn(query_ids, key_ids, query_vecs, key_vecs, k=2047, query_batch_size=1000, key_batch_size=10000):
                # the module's one caller (add_synonymy_edges, :670-712) stops at synonymy_edge_sim_threshold: hand it to the
                # kernel as the starting threshold instead of materialising 2047 neighbours per entity
                thr = getattr(cfg, "synonymy_edge_sim_threshold", None) if knn_threshold_filter else None
                return retrieval.retrieve_knn(query_ids, key_ids, query_vecs, key_vecs, k=k, query_batch_size=query_batch_size,
                                              key_batch_size=key_batch_size, index_dtype=dtype, device=device, min_score=thr)
            if hasattr(mod, "retrieve_knn"):
                mod.retrieve_knn = _knn
            if hasattr(mod, "get_similar_summaries"):
                mod.get_similar_summaries = retrieval.get_similar_summaries
    return rag


def install_memory_pool(pool, index_dtype: str = "f32", device: int = 0, index_factory=None, num_shards: Optional[int] = None, devices=None):
    """Rebind `MemoryPool.retrieve_similar_nodes` (utils/memory_utils.py:188-235) on ONE pool
    instance: node embeddings live in an appendable HBM index (rows = pool order; nodes added since
    the last call are encoded with the reference's own `compute_probe_note_embeddings` and appended —
    BASELINE config 4's incremental append), the probe is scored against all of them in one scan.
    Selection semantics are the reference's: cosine, descending, ties in pool order, keep
    max(1, int(n * top_percent)).  Nodes without an embedding are skipped like the reference does."""
    state = {"index": None, "rows": []}      # rows[i] = pool position of index row i
    lock = threading.Lock()

    def _to_np(x):
        if hasattr(x, "detach"):
            x = x.detach().cpu().numpy()
        return np.asarray(x, dtype=np.float32)

    def retrieve_similar_nodes(self, current_probe: str, top_percent: float = 0.5):
        if not self.embedding_model:
            raise ValueError("Embedding model not provided")
        self.compute_probe_note_embeddings()
        probe = _to_np(self.embedding_model.encode([current_probe])[0])
        with lock:
            have = {id(self.pool[i]) for i in state["rows"] if i < len(self.pool)}
            fresh = [(i, n) for i, n in enumerate(self.pool) if n.embedding is not None and id(n) not in have]
            if fresh:
                mat = np.stack([_to_np(n.embedding) for _, n in fresh])
                mat = mat / np.maximum(np.linalg.norm(mat, axis=1, keepdims=True), 1e-12)   # cosine == dot of unit rows
                if state["index"] is None:      # index_factory(dim, dtype, device): test seam, as in install()
                    if index_factory:
                        state["index"] = index_factory(mat.shape[1], index_dtype, device)
                    else:
                        from .multi_index import make_index
                        state["index"] = make_index(mat.shape[1], index_dtype, device=device, num_shards=num_shards, devices=devices)
                state["index"].append(mat)
                state["rows"].extend(i for i, _ in fresh)
            if state["index"] is None:
                return []
            order = retrieval.retrieve_similar_rows(state["index"], probe, len(state["rows"]), top_percent)
            return [self.pool[state["rows"][r]] for r in order]

    pool.retrieve_similar_nodes = types.MethodType(retrieve_similar_nodes, pool)
    pool._hip_state = state
    return pool
