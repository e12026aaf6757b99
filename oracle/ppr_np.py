"""CPU restatement (numpy, fp64) of ComoRAG's PPR seeding + personalised PageRank — oracle for comorag_amd/csrc/ppr.hip.

TEST INFRASTRUCTURE ONLY (see oracle/retrieval_np.py header).  Follows
  src/comorag/ComoRAG.py:1034-1045  passage_weights[vertex(passage)] = min_max_normalize(dpr score) * passage_node_weight,
                                    node_weights = phrase_weights + passage_weights
  src/comorag/ComoRAG.py:1086-1105  run_ppr: reset[nan | < 0] = 0; graph.personalized_pagerank(damping=0.5, directed=False,
                                    weights='weight', reset=reset, implementation='prpack'); doc_scores = pagerank[passage_node_idxs];
                                    argsort descending.
Pinning: python-igraph (and with it prpack) is ABSENT from this image and the reference ships no PPR fixture.
`personalized_pagerank` solves the defining linear system x = d (P^T + r 1_dangling^T) x + (1 - d) r directly
(numpy.linalg.solve — not a power iteration, so it is independent of the device algorithm).  tests/test_ppr.py pins it
(1) to closed forms (two vertices: x = (1/(1+d), d/(1+d)); a star; an isolated seed) and (2) to a THIRD-PARTY
implementation that is in the image: networkx 3.4.2 `pagerank(G, alpha=d, personalization=r, weight='weight',
dangling=r)` on random weighted graphs with isolated vertices, seeds on isolated vertices and negative / NaN reset entries
(agreement 1e-12).  The conventions are those igraph documents for igraph_personalized_pagerank with the PRPACK solver
(the reference pins igraph==0.11.8 / python-igraph==0.11.8, requirements.txt:56,138 — igraph C core 0.10): the reset
vector is normalised to sum 1; a walker leaves vertex i along edge (i, j) with probability w_ij / strength(i), an
undirected edge serving both directions, parallel edges adding up; a vertex without edges ('dangling') restarts
according to the RESET distribution (PRPACK's personalised u = v), which is what `dangling=r` selects in networkx —
networkx's default (dangling = personalization when given) is the same rule.
"""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np

from .retrieval_np import min_max_normalize


def transition_matrix(n: int, src: Sequence[int], dst: Sequence[int], weight=None) -> Tuple[np.ndarray, np.ndarray]:
    """Dense column-stochastic-where-possible matrix M[j, i] = w_ij / strength(i) and the dangling indicator."""
    W = np.zeros((n, n), dtype=np.float64)
    w = np.ones(len(src)) if weight is None else np.asarray(weight, np.float64)
    for u, v, x in zip(src, dst, w):
        W[u, v] += x
        if u != v:
            W[v, u] += x
    s = W.sum(axis=1)
    M = np.zeros_like(W)
    nz = s > 0
    M[:, nz] = (W[nz, :] / s[nz, None]).T
    return M, ~nz


def personalized_pagerank(n: int, src, dst, weight, reset, damping: float = 0.5) -> np.ndarray:
    r = np.asarray(reset, np.float64).copy()
    r = np.where(np.isnan(r) | (r < 0), 0.0, r)                   # ComoRAG.py:1090
    r = r / r.sum() if r.sum() > 0 else np.full(n, 1.0 / n)
    M, dang = transition_matrix(n, src, dst, weight)
    A = np.eye(n) - damping * (M + np.outer(r, dang.astype(np.float64)))
    return np.linalg.solve(A, (1.0 - damping) * r)


def passage_weights(dpr_sorted_doc_ids, dpr_sorted_doc_scores, passage_node_idxs, n_vertices: int, passage_node_weight: float) -> np.ndarray:
    """ComoRAG.py:1034-1040 (the dict of texts it also fills is dead code: trimmed and dropped)."""
    out = np.zeros(n_vertices)
    norm = min_max_normalize(np.asarray(dpr_sorted_doc_scores))
    for i, doc in enumerate(np.asarray(dpr_sorted_doc_ids).tolist()):
        out[passage_node_idxs[doc]] = norm[i] * passage_node_weight
    return out


def run_ppr(n: int, src, dst, weight, reset_prob, passage_node_idxs, damping: float = 0.5):
    """ComoRAG.py:1086-1105."""
    pr = personalized_pagerank(n, src, dst, weight, reset_prob, damping)
    doc_scores = np.array([pr[i] for i in passage_node_idxs])
    order = np.argsort(doc_scores)[::-1]
    return order, doc_scores[order.tolist()]
