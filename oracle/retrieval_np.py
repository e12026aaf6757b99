"""CPU restatement (numpy) of ComoRAG's embedding / dense-retrieval hot path.

TEST INFRASTRUCTURE ONLY — this is the *oracle* the HIP path is checked against.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it; nothing under ``comorag_amd/`` does.

Pinning status: the reference ships **no** tests, golden vectors or fixtures for
this path, so the pin is constructed: every function below is
compared, in ``tests/test_oracle_pin.py``, against (1) the committed fixtures in
``tests/golden/`` that ``oracle/make_golden.py`` produced by importing and
running the reference's own functions in the build container, and (2) — when
``/root/reference`` is present — the live reference functions.  Third-party
arithmetic under the reference (OpenBLAS sgemv via ``np.dot``, torch CPU
``mm``/``topk``) is not vendored; accumulation order there is unspecified, which
is why an fp64 arbiter (`exact_scores_f64`) and a tie/rounding-aware comparator
(`assert_topk_equivalent`) are part of the oracle.

All ``file:line`` citations are relative to ``/root/reference/src/comorag``.
"""
from __future__ import annotations

from hashlib import md5
from typing import Dict, List, Sequence, Tuple

import numpy as np


# --------------------------------------------------------------------------- a7
def compute_mdhash_id(content: str, prefix: str = "") -> str:
    """utils/misc_utils.py:152-163 — ``prefix + md5(utf8).hexdigest()``."""
    return prefix + md5(content.encode()).hexdigest()


# --------------------------------------------------------------------------- a3
def min_max_normalize(x: np.ndarray) -> np.ndarray:
    """utils/misc_utils.py:141-150 — ``(x-min)/(max-min)``; range 0 → ones."""
    min_val = np.min(x)
    max_val = np.max(x)
    range_val = max_val - min_val
    if range_val == 0:
        return np.ones_like(x)
    return (x - min_val) / range_val


def _squeeze_scores(s: np.ndarray) -> np.ndarray:
    # ComoRAG.py:945 / :962 — squeeze only a 2-d product
    return np.squeeze(s) if s.ndim == 2 else s


# --------------------------------------------------------------------------- a1
def dense_passage_retrieval(matrix: np.ndarray, query_embedding: np.ndarray
                            ) -> Tuple[np.ndarray, np.ndarray]:
    """ComoRAG.py:950-967 with the embedding lookup factored out.

    ``matrix`` is ``passage_embeddings`` or ``summary_embeddings`` [N,D] fp32
    (ComoRAG.py:897,900), ``query_embedding`` is what ``batch_encode(query)``
    returned: shape [1,D].  Returns all N ids (descending score) and the
    min-max-normalised scores in that order.
    """
    s = np.dot(matrix, query_embedding.T)
    s = _squeeze_scores(s)
    s = min_max_normalize(s)
    ids = np.argsort(s)[::-1]
    return ids, s[ids.tolist()]


# --------------------------------------------------------------------------- a2
def get_fact_scores(fact_embeddings: np.ndarray, query_embedding: np.ndarray) -> np.ndarray:
    """ComoRAG.py:937-948 — full normalised score vector over the fact matrix."""
    s = np.dot(fact_embeddings, query_embedding.T)
    s = _squeeze_scores(s)
    return min_max_normalize(s)


def link_top_k(query_fact_scores: np.ndarray, k: int) -> List[int]:
    """ComoRAG.py:1073 (dup :475) — ``argsort(s)[-k:][::-1]``."""
    return np.argsort(query_fact_scores)[-k:][::-1].tolist()


# --------------------------------------------------------------------------- a9
def get_similar_summaries(summary_embeddings: np.ndarray, level_texts: Sequence[str],
                          query_embedding: np.ndarray, top_k: int = 3
                          ) -> Tuple[List[str], List[float]]:
    """utils/embed_utils.py:109-161 with store / encoder access factored out."""
    if len(level_texts) == 0 or len(summary_embeddings) == 0:
        return [], []
    s = np.dot(summary_embeddings, query_embedding.T)
    s = _squeeze_scores(s)
    s = min_max_normalize(s)
    idx = np.argsort(s)[::-1][:top_k]
    return [level_texts[i] for i in idx], s[idx].tolist()


# --------------------------------------------------------------------------- a10
def _l2n(x: np.ndarray, eps: float = 1e-12) -> np.ndarray:
    # torch.nn.functional.normalize(dim=1): x / max(||x||, eps)
    n = np.sqrt((x.astype(np.float32) ** 2).sum(axis=1, keepdims=True, dtype=np.float32))
    return x / np.maximum(n, np.float32(eps))


def retrieve_knn(query_ids: Sequence[str], key_ids: Sequence[str], query_vecs, key_vecs,
                 k: int = 2047, query_batch_size: int = 1000, key_batch_size: int = 10000
                 ) -> Dict[str, Tuple[List[str], List[float]]]:
    """utils/embed_utils.py:8-97 — fp32 re-normalise, blocked sim + top-k, merge.

    torch.topk's order among equal scores is unspecified; this restatement uses
    score-desc / index-asc (the engine's exported tie rule).
    """
    if len(key_vecs) == 0:
        return {}
    q = _l2n(np.asarray(query_vecs, dtype=np.float32))
    kx = _l2n(np.asarray(key_vecs, dtype=np.float32))
    out: Dict[str, Tuple[List[str], List[float]]] = {}
    for qs in range(0, len(q), query_batch_size):
        qb = q[qs:qs + query_batch_size]
        cand_s, cand_i = [], []
        for ks in range(0, len(kx), key_batch_size):
            kb = kx[ks:ks + key_batch_size]
            sim = qb @ kb.T
            kk = min(k, kb.shape[0])
            order = np.argsort(-sim, axis=1, kind="stable")[:, :kk]
            cand_s.append(np.take_along_axis(sim, order, axis=1))
            cand_i.append(order + ks)
        cs = np.concatenate(cand_s, axis=1)
        ci = np.concatenate(cand_i, axis=1)
        kk = min(k, cs.shape[1])
        # merge: score desc, then global key index asc
        order = np.lexsort((ci, -cs), axis=1)[:, :kk]
        fs = np.take_along_axis(cs, order, axis=1)
        fi = np.take_along_axis(ci, order, axis=1)
        for r in range(qb.shape[0]):
            out[query_ids[qs + r]] = ([key_ids[j] for j in fi[r]], fs[r].tolist())
    return out


# --------------------------------------------------------------------------- a11
def retrieve_similar_nodes(node_embeddings: Sequence[np.ndarray], probe_embedding: np.ndarray,
                           top_percent: float = 0.5) -> List[int]:
    """utils/memory_utils.py:188-235, numeric part: python-loop cosine, *stable*
    ``list.sort(reverse=True)`` (equal similarities keep pool order), keep
    ``max(1, int(n*top_percent))``.  Returns pool indices."""
    sims = []
    for i, e in enumerate(node_embeddings):
        if e is None:
            continue
        sim = np.dot(probe_embedding, e) / (np.linalg.norm(probe_embedding) * np.linalg.norm(e))
        sims.append((i, sim))
    sims.sort(key=lambda t: t[1], reverse=True)
    k = max(1, int(len(sims) * top_percent))
    return [i for i, _ in sims[:k]]


# --------------------------------------------------------------------------- a5
def mean_pool_l2norm(token_embeddings: np.ndarray, mask: np.ndarray, normalize: bool = True,
                     eps: float = 1e-12) -> np.ndarray:
    """embedding_model/BGEEmbedding.py:15-28 (``mean_pooling``) followed by
    ``F.normalize(p=2, dim=1)`` (:126-127, eps 1e-12).  fp32 arithmetic."""
    h = np.asarray(token_embeddings, dtype=np.float32)
    m = np.asarray(mask).astype(bool)
    h = np.where(m[..., None], h, np.float32(0))
    s = h.sum(axis=1, dtype=np.float32) / m.sum(axis=1).astype(np.float32)[..., None]
    if normalize:
        n = np.sqrt((s * s).sum(axis=1, keepdims=True, dtype=np.float32))
        s = s / np.maximum(n, np.float32(eps))
    return s.astype(np.float32)


# --------------------------------------------------------------------------- a6
def insert_plan(existing_ids: Sequence[str], texts: Sequence[str], namespace: str
                ) -> Tuple[List[str], List[str]]:
    """embedding_store.py:63-86 — dict-dedup (first occurrence keeps its slot),
    skip ids already present; returns (missing_ids, texts_to_encode) in order."""
    nodes: Dict[str, str] = {}
    for t in texts:
        nodes[compute_mdhash_id(t, prefix=namespace + "-")] = t
    have = set(existing_ids)
    missing = [h for h in nodes if h not in have]
    return missing, [nodes[h] for h in missing]


# --------------------------------------------------------------------------- arbiter
def exact_scores_f64(matrix: np.ndarray, queries: np.ndarray) -> np.ndarray:
    """[nq,N] inner products accumulated in fp64 (inputs taken as given)."""
    return np.asarray(queries, dtype=np.float64) @ np.asarray(matrix, dtype=np.float64).T


def topk_rule(scores: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """The engine's exported order: score descending, then row index ascending.
    ``scores`` [nq,N] → (ids int64[nq,k'], scores[nq,k']) with k' = min(k,N)."""
    scores = np.atleast_2d(scores)
    n = scores.shape[1]
    kk = min(k, n)
    order = np.argsort(-scores, axis=1, kind="stable")[:, :kk]
    return order.astype(np.int64), np.take_along_axis(scores, order, axis=1)


def assert_topk_equivalent(got_ids: np.ndarray, ref_ids: np.ndarray, exact: np.ndarray,
                           err_bound: float) -> int:
    """Tie/rounding-aware id comparison (SURVEY.md §7 "hard parts").

    ``got_ids``/``ref_ids`` [k] for one query, ``exact`` [N] fp64 scores of that
    query.  Positions may differ only where the fp64 scores of the two ids are
    within ``err_bound`` (the accumulation-order error of the two fp32 paths);
    as *sets*, any id in one list and not the other must be within ``err_bound``
    of the k-th exact score.  Returns the number of tolerated swaps; raises
    AssertionError otherwise.
    """
    got_ids = np.asarray(got_ids).ravel()
    ref_ids = np.asarray(ref_ids).ravel()
    assert got_ids.shape == ref_ids.shape, (got_ids.shape, ref_ids.shape)
    assert len(set(got_ids.tolist())) == len(got_ids), "duplicate ids in result"
    swaps = 0
    for pos, (g, r) in enumerate(zip(got_ids, ref_ids)):
        if g == r:
            continue
        d = abs(exact[g] - exact[r])
        assert d <= err_bound, (f"pos {pos}: got id {g} (exact {exact[g]!r}) vs ref id {r} "
                                f"(exact {exact[r]!r}) differ by {d:.3e} > {err_bound:.3e}")
        swaps += 1
    only = set(got_ids.tolist()) ^ set(ref_ids.tolist())
    if only:
        kth = np.sort(exact)[::-1][len(ref_ids) - 1]
        for i in only:
            assert abs(exact[i] - kth) <= err_bound, (i, exact[i], kth)
    return swaps


# --------------------------------------------------------------------------- inputs
def synthetic_corpus(n: int, d: int, seed: int = 1234, block: int | None = None) -> np.ndarray:
    """seeded standard-normal rows, L2-normalised in fp32."""
    rng = np.random.default_rng(seed if block is None else [seed, block])
    x = rng.standard_normal((n, d), dtype=np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x


def synthetic_queries(b: int, d: int, seed: int = 4321, planted: np.ndarray | None = None,
                      noise: float = 0.1) -> np.ndarray:
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((b, d), dtype=np.float32)
    if planted is not None and len(planted):
        m = min(len(planted), max(1, b // 10))
        q[:m] = planted[:m] + noise * rng.standard_normal((m, d), dtype=np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q


def bf16_round(x: np.ndarray) -> np.ndarray:
    """fp32 → bf16 (round-to-nearest-even) → fp32, NaN-free inputs."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    return ((u + r) & np.uint32(0xFFFF0000)).view(np.float32)


def f16_round(x: np.ndarray) -> np.ndarray:
    return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)


def retrieve_knn_torch_cpu(query_vecs, key_vecs, k: int = 2047, query_batch_size: int = 1000, key_batch_size: int = 10000):
    """utils/embed_utils.py:8-97 with the reference's own primitives forced onto the CPU (torch.mm + torch.topk per
    key block, concat, final torch.topk) — the batched CPU comparator of bench.py's ``cpu_baseline`` (BASELINE.md §3 ii).
    Returns (indices [nq, k'], scores [nq, k']); tie order is torch.topk's (unspecified), so parity checks use
    ``retrieve_knn`` above, timing uses this one."""
    import torch
    q = torch.nn.functional.normalize(torch.as_tensor(np.asarray(query_vecs, dtype=np.float32)), dim=1)      # :27-31
    kx = torch.nn.functional.normalize(torch.as_tensor(np.asarray(key_vecs, dtype=np.float32)), dim=1)
    out_i, out_s = [], []
    for qs in range(0, len(q), query_batch_size):                                                           # :40-78
        qb = q[qs:qs + query_batch_size]
        cs, ci = [], []
        for ks in range(0, len(kx), key_batch_size):
            kb = kx[ks:ks + key_batch_size]
            sim = torch.mm(qb, kb.T)
            s, i = torch.topk(sim, min(k, kb.shape[0]), dim=1)
            cs.append(s); ci.append(i + ks)
        cs, ci = torch.cat(cs, dim=1), torch.cat(ci, dim=1)
        s, o = torch.topk(cs, min(k, cs.shape[1]), dim=1)
        out_s.append(s); out_i.append(torch.gather(ci, 1, o))
    return torch.cat(out_i).numpy(), torch.cat(out_s).numpy()
