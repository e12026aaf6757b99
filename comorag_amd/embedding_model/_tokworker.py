"""Worker side of the tokenizer PROCESS pool of HipBGEEmbeddingModel (embedding_tokenizer_processes > 0).

The HF fast tokenizer's `encode_batch` holds the GIL for the whole call, so tokenizer THREADS share one core with the
thread that launches the encoder's kernels (measured ceiling ~4.7 K chunks/s of 512 tokens per process); worker
processes lift that.  A worker imports nothing but `tokenizers` (no torch, no HIP: spawn-safe) and rebuilds the
tokenizer from its JSON; what it returns — truncated, unpadded id arrays — holds exactly what
`tokenize_ragged` (bge.py) returns in-process (tests/test_store_host.py holds the two against each other)."""
from __future__ import annotations

_TOK = None
_TRUNC = {"strategy": "longest_first", "direction": "right"}
_LEN = None


def init(tokenizer_json: str, direction: str = "right", strategy: str = "longest_first") -> None:
    """direction / strategy: the in-process path's truncation settings (bge.py:_backend honours tokenizer.truncation_side) —
    a left-truncating tokenizer must yield the same ids from a worker process."""
    global _TOK
    import os
    if os.environ.get("CMR_TOKWORKER_RAYON"):          # probe knob (tools/tok_mode_probe.py): rayon threads of a worker's encode_batch
        os.environ["RAYON_NUM_THREADS"] = os.environ["CMR_TOKWORKER_RAYON"]
    from tokenizers import Tokenizer
    _TOK = Tokenizer.from_str(tokenizer_json)
    _TOK.no_padding()
    _TRUNC["direction"], _TRUNC["strategy"] = direction, strategy


def ragged(prompts, max_length: int):
    """Token ids per prompt as int32 arrays (a pickled array crosses the pipe as one buffer; a list of 512 Python ints is
    512 objects to rebuild on the side that also launches the encoder's kernels)."""
    import numpy as np
    global _LEN
    if _LEN != int(max_length):          # configured once per length, not per call
        _TOK.enable_truncation(int(max_length), stride=0, strategy=_TRUNC["strategy"], direction=_TRUNC["direction"])
        _LEN = int(max_length)
    enc = _TOK.encode_batch_fast(list(prompts)) if hasattr(_TOK, "encode_batch_fast") else _TOK.encode_batch(list(prompts))      # no offsets: ~5x
    return [np.asarray(e.ids, dtype=np.int32) for e in enc]
