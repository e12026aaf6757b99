"""Device-side PPR seeding + personalised PageRank — host mirror of
ComoRAG.graph_search_with_fact_entities' passage loop (src/comorag/ComoRAG.py:1034-1045) and ComoRAG.run_ppr (:1086-1105).

The reference copies all N (passage id, normalised score) pairs to the host, scatters them one by one into a vertex
vector and hands that to igraph's prpack PageRank.  Here the graph lives in HBM as a CSR copy (`DeviceGraph`), and
`ppr_passage_scores` keeps scan -> min-max -> scatter -> power iteration -> gather on the device: only n_passages
doubles come back.  igraph itself is not needed (it is absent from this image): `DeviceGraph.from_igraph` only reads an
edge list + weights from anything that offers `vcount() / get_edgelist() / es['weight']`.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _lib as L


class DeviceGraph:
    def __init__(self, n_vertices: int, src, dst, weight=None, device: int = 0):
        src = np.ascontiguousarray(src, dtype=np.int32)
        dst = np.ascontiguousarray(dst, dtype=np.int32)
        if src.shape != dst.shape:
            raise ValueError("src / dst length mismatch")
        w = None if weight is None else np.ascontiguousarray(weight, dtype=np.float64)
        if w is not None and w.shape != src.shape:
            raise ValueError("weight length mismatch")
        self.n_vertices, self.device = int(n_vertices), int(device)
        self.n_rows = 0
        self._h = C.c_void_p()
        L.check(L.lib().cmr_graph_create(self.device, self.n_vertices, len(src), src.ctypes.data_as(C.c_void_p), dst.ctypes.data_as(C.c_void_p),
                                         w.ctypes.data_as(C.c_void_p) if w is not None else None, C.byref(self._h)))

    @classmethod
    def from_igraph(cls, graph, device: int = 0, weight_attr: str = "weight"):
        """Anything with igraph's vcount() / get_edgelist() / es[attr] (ComoRAG.graph)."""
        edges = graph.get_edgelist()
        src = [e[0] for e in edges]
        dst = [e[1] for e in edges]
        try:
            w = list(graph.es[weight_attr]) if len(edges) else []
        except Exception:
            w = None
        return cls(graph.vcount(), src, dst, w, device=device)

    def set_passage_vertices(self, passage_node_idxs: Sequence[int]) -> None:
        v = np.ascontiguousarray(passage_node_idxs, dtype=np.int32)
        L.check(L.lib().cmr_graph_set_passage_vertices(self._h, v.ctypes.data_as(C.c_void_p), len(v)))
        self.n_rows = len(v)
        self.passage_vertices = v

    def ppr(self, reset, damping: float = 0.5, tol: float = 1e-12, max_iter: int = 200) -> np.ndarray:
        """personalized_pagerank(reset=...) over every vertex (negative / NaN reset entries count as 0)."""
        r = np.ascontiguousarray(reset, dtype=np.float64)
        if r.shape != (self.n_vertices,):
            raise ValueError(f"reset must have {self.n_vertices} entries")
        out = np.empty(self.n_vertices, dtype=np.float64)
        it = C.c_int32(0)
        L.check(L.lib().cmr_graph_ppr(self._h, r.ctypes.data_as(C.c_void_p), float(damping), float(tol), int(max_iter),
                                      out.ctypes.data_as(C.c_void_p), C.byref(it)))
        self.last_iters = it.value
        return out

    def close(self) -> None:
        if getattr(self, "_h", None):
            L.lib().cmr_graph_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def run_ppr(graph: DeviceGraph, reset_prob, passage_node_idxs, damping: Optional[float] = 0.5) -> Tuple[np.ndarray, np.ndarray]:
    """ComoRAG.run_ppr (ComoRAG.py:1086-1105) with the PageRank itself on the device; the last three lines are the reference's."""
    if damping is None:
        damping = 0.5
    pagerank_scores = graph.ppr(np.asarray(reset_prob, dtype=np.float64), damping=damping)
    doc_scores = np.array([pagerank_scores[idx] for idx in passage_node_idxs])
    sorted_doc_ids = np.argsort(doc_scores)[::-1]
    sorted_doc_scores = doc_scores[sorted_doc_ids.tolist()]
    return sorted_doc_ids, sorted_doc_scores


def ppr_passage_scores(index, graph: DeviceGraph, query_embedding, phrase_weights=None, passage_node_weight: float = 0.05,
                       damping: float = 0.5, tol: float = 1e-12, max_iter: int = 200) -> np.ndarray:
    """The fused path for one query: doc_scores[i] = pagerank[vertex of passage row i], with the DPR scores scattered into
    the reset vector on the device.  `phrase_weights`: dense [n_vertices] array (only its non-zero entries are shipped)
    or a (vertices, weights) pair.  `graph.set_passage_vertices(...)` must map every row of `index`."""
    q = np.ascontiguousarray(np.asarray(query_embedding, dtype=np.float32).reshape(-1))
    if hasattr(index, "n_shards") or not hasattr(index, "_h"):
        # a row-sharded index (MultiDeviceIndex): its shards live on several devices, the graph on one — the N scores come to
        # the host once (4 N bytes), the reference's own lines build the reset vector (ComoRAG.py:1034-1045: min-max, score x
        # passage_node_weight into the passages' vertices, product in float64 as numpy 1.26 forms it), PageRank runs on the device
        from .utils.misc_utils import min_max_normalize
        s = index.scores(q[None, :])[0]
        norm = min_max_normalize(s)
        reset = np.zeros(graph.n_vertices, dtype=np.float64)
        if phrase_weights is not None:
            if isinstance(phrase_weights, tuple):
                np.add.at(reset, np.asarray(phrase_weights[0], np.int64), np.asarray(phrase_weights[1], np.float64))
            else:
                reset += np.asarray(phrase_weights, dtype=np.float64)
        reset[graph.passage_vertices] += norm.astype(np.float64) * float(passage_node_weight)
        return graph.ppr(reset, damping=damping, tol=tol, max_iter=max_iter)[graph.passage_vertices]
    if phrase_weights is None:
        sv, sw = np.empty(0, np.int32), np.empty(0, np.float64)
    elif isinstance(phrase_weights, tuple):
        sv, sw = np.ascontiguousarray(phrase_weights[0], np.int32), np.ascontiguousarray(phrase_weights[1], np.float64)
    else:
        pw = np.asarray(phrase_weights, dtype=np.float64)
        sv = np.flatnonzero(pw != 0).astype(np.int32)
        sw = np.ascontiguousarray(pw[sv])
    out = np.empty(graph.n_rows, dtype=np.float64)
    it = C.c_int32(0)
    L.check(L.lib().cmr_index_ppr(index._h, graph._h, q.ctypes.data_as(C.c_void_p), sv.ctypes.data_as(C.c_void_p), sw.ctypes.data_as(C.c_void_p),
                                  len(sv), float(passage_node_weight), float(damping), float(tol), int(max_iter),
                                  out.ctypes.data_as(C.c_void_p), C.byref(it)))
    return out


def ppr_passage_ranking(index, graph: DeviceGraph, query_embedding, phrase_weights=None, passage_node_weight: float = 0.05,
                        damping: float = 0.5) -> Tuple[np.ndarray, np.ndarray]:
    """(sorted_doc_ids, sorted_doc_scores) exactly as ComoRAG.run_ppr returns them (ComoRAG.py:1101-1105)."""
    doc_scores = ppr_passage_scores(index, graph, query_embedding, phrase_weights, passage_node_weight, damping)
    sorted_doc_ids = np.argsort(doc_scores)[::-1]
    return sorted_doc_ids, doc_scores[sorted_doc_ids.tolist()]
