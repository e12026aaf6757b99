"""Numeric re-score stage behind the reference's rerank call shape.

src/comorag/rerank.py:97-123 (`DSPyFilter.rerank`) is an LLM filter: it returns
`(sorted_indices[:n], sorted_items[:n], {'confidence': None})` and contains no numeric scoring.
BASELINE config 5 asks for "cross-scores on top-100 candidates"; this module supplies that as an
exact fp32 re-score of low-precision candidates (cmr_index_rescore) with the same return triple,
`confidence` carrying the scores.  The LLM filter itself stays the reference's (out of scope).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

from .index import DenseIndex


class ExactRescorer:
    def __init__(self, index: DenseIndex):
        self.index = index

    def __call__(self, *args, **kwargs):
        return self.rerank(*args, **kwargs)

    def rerank(self, query_embedding, candidate_items: Sequence, candidate_indices: Sequence[int],
               len_after_rerank: int = None) -> Tuple[List[int], List, dict]:
        """query_embedding [D] fp32 (the reference passes the query *string* to its LLM; the numeric
        stage needs the vector), candidate_items / candidate_indices as in rerank.py:100-104."""
        n = len(candidate_indices)
        if n == 0:
            return [], [], {"confidence": []}
        k = n if len_after_rerank is None else min(len_after_rerank, n)
        ids, sc = self.index.rescore(np.asarray(query_embedding, np.float32), np.asarray(candidate_indices, np.int64), k)
        pos = {int(r): i for i, r in reversed(list(enumerate(candidate_indices)))}
        keep = [int(r) for r in ids[0] if r >= 0]
        return keep, [candidate_items[pos[r]] for r in keep], {"confidence": sc[0][:len(keep)].tolist()}


def search_then_rescore(index: DenseIndex, queries, k_candidates: int = 100, k: int = 20):
    """BASELINE config 5 flow: low-precision top-`k_candidates` → exact fp32 top-`k`."""
    q = np.asarray(queries, np.float32)
    cand, _, _, _ = index.search(q, k_candidates, with_minmax=False)
    return index.rescore(q, cand, k)
