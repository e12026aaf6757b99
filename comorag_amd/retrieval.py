"""The five retrieval call sites of ComoRAG as functions over a `DenseIndex`.

Each function names the reference code it replaces; scoring runs on the GPU (C-ABI), the
surrounding formulae (min-max normalisation, argsort direction, squeeze corner cases, return
types) are the reference's, textually, so callers see the same objects.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .index import DenseIndex
from .utils.misc_utils import min_max_normalize

# Complete ranking of one query: below this, numpy's argsort of the N GPU scores (single-launch cmr_index_scores + ~9 ns per
# row) beats the device radix sort's 26 launches + 12 N bytes of D2H (measured, profiles/r2_full_ranking_crossover.txt:
# 32 / 85 / 181 us against 189 / 242 / 251 us at 1 K / 8 K / 20 K rows; 363 against 270 us at 40 K)
DEVICE_SORT_MIN_ROWS = 28_672
QUERY_INSTRUCTION_SUMMARIES = "Given a question, retrieve relevant documents that best answer the question."


def _as_query(q) -> np.ndarray:
    q = np.asarray(q, dtype=np.float32)
    return q[None, :] if q.ndim == 1 else q


def full_scores(index: DenseIndex, query_embedding) -> np.ndarray:
    """`np.dot(M, q.T)` + squeeze of ComoRAG.py:944-945 / :958-962: one query → shape (N,)
    (0-d when N == 1, as np.squeeze gives)."""
    s = index.scores(_as_query(query_embedding))           # [1, N]
    s = s.T                                                # (N, 1), what np.dot(M, q.T) returns
    return np.squeeze(s) if s.ndim == 2 else s


def dense_passage_retrieval(index: DenseIndex, query_embedding) -> Tuple[np.ndarray, np.ndarray]:
    """ComoRAG.dense_passage_retrieval (ComoRAG.py:950-967): ALL N ids by descending min-max
    normalised score + the scores in that order.  The N·D inner products come from the GPU; the
    normalise + argsort lines are the reference's.
    Rows with EQUAL scores (duplicated chunks): below DEVICE_SORT_MIN_ROWS they come in the order of numpy's
    `argsort(x)[::-1]` (introsort: unspecified, not stable — whatever the reference's own call yields on these scores),
    from DEVICE_SORT_MIN_ROWS on by ascending row id (the library's exported tie rule; include/comorag_hip.h)."""
    if len(index) < DEVICE_SORT_MIN_ROWS:          # tiny corpora: the reference's own lines on GPU scores
        query_doc_scores = full_scores(index, query_embedding)
        query_doc_scores = min_max_normalize(query_doc_scores)
        sorted_doc_ids = np.argsort(query_doc_scores)[::-1]
        sorted_doc_scores = query_doc_scores[sorted_doc_ids]      # (the reference indexes with .tolist(): same values, N Python ints)
        return sorted_doc_ids, sorted_doc_scores
    ids, raw, mn, mx = index.sorted_scores(_as_query(query_embedding)[:1])     # scan + stable radix sort on the GPU
    rng = mx[0] - mn[0]
    # min_max_normalize on the sorted vector (elementwise, order-independent): same fp32 formula
    sorted_doc_scores = np.ones_like(raw[0]) if rng == 0 else (raw[0] - mn[0]) / rng
    return ids[0], sorted_doc_scores


def dense_passage_topk(index: DenseIndex, query_embeddings, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """Fast path for callers that only consume the head (tri_retrieve's qa_*_top_k slices): fused
    scan + top-k on the GPU; scores min-max normalised from the kernel's global min/max with the
    reference formula.  ids [nq,k'], scores [nq,k'] (descending; ties: lower row id first)."""
    ids, sc, mn, mx = index.search(_as_query(query_embeddings), k)
    rng = (mx - mn)[:, None]
    norm = np.where(rng == 0, np.ones_like(sc), (sc - mn[:, None]) / np.where(rng == 0, 1, rng))
    return ids, norm.astype(np.float32)


def get_fact_scores(index: Optional[DenseIndex], query_embedding) -> np.ndarray:
    """ComoRAG.get_fact_scores (ComoRAG.py:937-948): full normalised vector over the fact matrix."""
    if index is None or len(index) == 0:
        return np.array([])
    return min_max_normalize(full_scores(index, query_embedding))


def link_top_k(query_fact_scores: np.ndarray, k: int) -> List[int]:
    """ComoRAG.py:1073 / :475."""
    return np.argsort(query_fact_scores)[-k:][::-1].tolist()


def get_similar_summaries(query: str, level_store, embedding_model, top_k: int = 3,
                          instruction: Optional[str] = None) -> Tuple[List[str], List[float]]:
    """utils/embed_utils.py:109-161 — same signature and returns; the store's device mirror replaces
    `get_embeddings(all ids)` (a full N×D copy) + np.dot."""
    level_ids = level_store.get_all_ids()
    if not level_ids:
        return [], []
    level_texts = [level_store.hash_id_to_text[id] for id in level_ids]
    query_embedding = embedding_model.batch_encode(query, instruction=QUERY_INSTRUCTION_SUMMARIES, norm=True)
    if hasattr(level_store, "device_index"):
        similarity_scores = full_scores(level_store.device_index(), query_embedding)
    else:   # a reference-class store: build a throw-away index from its rows
        rows = np.asarray(level_store.get_embeddings(level_ids), dtype=np.float32)
        if len(rows) == 0:
            return [], []
        tmp = DenseIndex(rows.shape[1], "f32", capacity_hint=len(rows))
        try:
            tmp.append(rows)
            similarity_scores = full_scores(tmp, query_embedding)
        finally:
            tmp.close()
    similarity_scores = min_max_normalize(similarity_scores)
    sorted_indices = np.argsort(similarity_scores)[::-1][:top_k]
    sorted_scores = similarity_scores[sorted_indices]
    return [level_texts[i] for i in sorted_indices], sorted_scores.tolist()


def _l2n(x: np.ndarray) -> np.ndarray:
    n = np.sqrt((x * x).sum(axis=1, keepdims=True, dtype=np.float32))
    return x / np.maximum(n, np.float32(1e-12))


def retrieve_knn(query_ids: Sequence[str], key_ids: Sequence[str], query_vecs, key_vecs, k: int = 2047,
                 query_batch_size: int = 1000, key_batch_size: int = 10000, index_dtype: str = "f32",
                 device: int = 0, min_score: Optional[float] = None, max_neighbours: int = 128) -> Dict[str, Tuple[List[str], List[float]]]:
    """utils/embed_utils.py:8-97 — fp32 re-normalise both sides, exact top-k of every query over all keys,
    `{query_id: ([key ids], [scores])}`.  The reference's key-block loop + merge computes the global top-k, so one pass
    over a single HBM index gives the same answer; k <= 128 uses the fused scan+top-k kernel, larger k
    (synonymy_edge_topk = 2047, up to 4096) materialises the score block in HBM and selects per row on the device; only
    k > 4096 falls back to a host select of GPU scores.  Equal scores: lower key index first (torch.topk's tie order is
    unspecified).

    `min_score` (not in the reference signature): the caller will stop at the first neighbour below this score — as the
    only caller does, ComoRAG.add_synonymy_edges (:696-699: `score < synonymy_edge_sim_threshold or num_nns > 100`).  Then
    each list holds the neighbours with score >= min_score only, best first, found by the fused kernel with its threshold
    started at min_score (cmr_index_search_min_score) — no [nq, N] score block, no 2047-wide selection.  A query with
    more than `max_neighbours` (<= 128) such neighbours is re-run through the exact large-k path, so every list is a
    prefix-complete answer: what the consumer reads is identical."""
    if len(key_vecs) == 0:
        return {}
    q = _l2n(np.asarray(query_vecs, dtype=np.float32))
    kx = _l2n(np.asarray(key_vecs, dtype=np.float32))
    index = DenseIndex(kx.shape[1], index_dtype, device=device, capacity_hint=len(kx))
    try:
        index.append(kx)
        kk = min(k, len(kx))
        results: Dict[str, Tuple[List[str], List[float]]] = {}
        from ._lib import CMR_MAX_K, CMR_MAX_K_2PASS
        thr_k = min(max_neighbours, CMR_MAX_K, kk)
        thr_all = None
        if min_score is not None and kk > thr_k and len(q) > query_batch_size:
            # many query blocks (the synonymy self-join: every entity against every entity): queries go up ONCE, every block is
            # enqueued on one stream without a synchronisation in between, the [nq, thr_k] results come back ONCE
            import torch
            dev = torch.device("cuda", device)
            with torch.cuda.device(dev):
                q_t = torch.from_numpy(q).to(dev)
                ids_t = torch.empty((len(q), thr_k), dtype=torch.int64, device=dev)
                sc_t = torch.empty((len(q), thr_k), dtype=torch.float32, device=dev)
                torch.cuda.synchronize(dev)                     # the uploads above are done before the index's own streams read them
                done = None
                for s in range(0, len(q), query_batch_size):   # throughput mode: block i + 1 is packed and block i - 1 merged beside the scan of block i
                    e = min(s + query_batch_size, len(q))
                    done = index.search_min_score_pipelined(q_t[s:e], thr_k, min_score, ids_t[s:e], sc_t[s:e])
                index.sync(done)
                torch.cuda.synchronize(dev)
                if index.query_status():
                    from ._lib import CMR_ERR_NONFINITE, CmrError
                    raise CmrError(CMR_ERR_NONFINITE, "query contains NaN/Inf")
                thr_all = (ids_t.cpu().numpy(), sc_t.cpu().numpy())
                del q_t, ids_t, sc_t
        for s in range(0, len(q), query_batch_size):
            qb = q[s:s + query_batch_size]
            if min_score is not None and kk > thr_k:
                ids, sc = (thr_all[0][s:s + query_batch_size], thr_all[1][s:s + query_batch_size]) if thr_all is not None else index.search_min_score(qb, thr_k, min_score)
                full = np.flatnonzero(ids[:, -1] >= 0)                  # lists that filled up: more neighbours may exist
                redo = dict(zip(full.tolist(), zip(*index.search(qb[full], kk, with_minmax=False)[:2]))) if len(full) and kk <= CMR_MAX_K_2PASS else {}
                for r in range(len(qb)):
                    if r in redo:
                        ri, rs = redo[r]
                        keep = rs >= np.float32(min_score)
                        ri, rs = ri[keep], rs[keep]
                    else:
                        keep = ids[r] >= 0
                        ri, rs = ids[r][keep], sc[r][keep]
                    results[query_ids[s + r]] = ([key_ids[j] for j in ri], rs.tolist())
                continue
            if kk <= CMR_MAX_K_2PASS:      # fused kernel (k <= 128) or device scores + per-row select
                ids, sc, _, _ = index.search(qb, kk, with_minmax=False)
            else:
                full = index.scores(qb)
                part = np.argpartition(-full, kk - 1, axis=1)[:, :kk] if kk < full.shape[1] else np.tile(np.arange(full.shape[1]), (len(qb), 1))
                ps = np.take_along_axis(full, part, axis=1)
                order = np.lexsort((part, -ps), axis=1)
                ids = np.take_along_axis(part, order, axis=1)
                sc = np.take_along_axis(ps, order, axis=1)
            for r in range(len(qb)):
                results[query_ids[s + r]] = ([key_ids[j] for j in ids[r]], sc[r].tolist())
        return results
    finally:
        index.close()


def retrieve_similar_rows(index: DenseIndex, probe_embedding, n_nodes: int, top_percent: float = 0.5) -> List[int]:
    """Numeric part of MemoryPool.retrieve_similar_nodes (utils/memory_utils.py:213-235) for a pool
    whose (unit-norm) node embeddings live in `index`: cosine of the probe with every node, keep
    max(1, int(n*top_percent)), ties keep pool order (the reference's stable sort)."""
    p = _l2n(_as_query(probe_embedding))
    keep = max(1, int(n_nodes * top_percent))
    if keep <= 4096:
        return index.search(p, keep, with_minmax=False)[0][0].tolist()
    s = index.scores(p)[0]
    return np.argsort(-s, kind="stable")[:keep].tolist()
