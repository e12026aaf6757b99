"""Bit-identical restatements of the two helpers the hot path shares with the rest of ComoRAG.

Reference: src/comorag/utils/misc_utils.py:141-163 (and the duplicate min_max_normalize at
src/comorag/utils/embed_utils.py:99-107, which additionally returns an empty input unchanged).
"""
from hashlib import md5

import numpy as np


def compute_mdhash_id(content: str, prefix: str = "") -> str:
    """``prefix + md5(utf-8 bytes).hexdigest()`` — row identity across runs (misc_utils.py:152-163)."""
    return prefix + md5(content.encode()).hexdigest()


def min_max_normalize(x):
    """(x - min) / (max - min); a zero range yields ones (misc_utils.py:141-150)."""
    x = np.asarray(x)
    if x.size == 0:            # embed_utils.py:101-102
        return x
    min_val = np.min(x)
    max_val = np.max(x)
    range_val = max_val - min_val
    if range_val == 0:
        return np.ones_like(x)
    return (x - min_val) / range_val
