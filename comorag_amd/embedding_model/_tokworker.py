"""Worker side of the tokenizer PROCESS pool of HipBGEEmbeddingModel (embedding_tokenizer_processes > 0).

The HF fast tokenizer's `encode_batch` holds the GIL for the whole call, so tokenizer THREADS share one core with the
thread that launches the encoder's kernels (measured ceiling ~4.7 K chunks/s of 512 tokens per process); worker
processes lift that.  A worker imports nothing but `tokenizers` (no torch, no HIP: spawn-safe) and rebuilds the
tokenizer from its JSON; what it returns — truncated, unpadded id arrays — holds exactly what
`tokenize_ragged` (bge.py) returns in-process (tests/test_store_host.py holds the two against each other)."""
from __future__ import annotations

_TOK = None


def init(tokenizer_json: str) -> None:
    global _TOK
    from tokenizers import Tokenizer
    _TOK = Tokenizer.from_str(tokenizer_json)
    _TOK.no_padding()


def ragged(prompts, max_length: int):
    """Token ids per prompt as int32 arrays (a pickled array crosses the pipe as one buffer; a list of 512 Python ints is
    512 objects to rebuild on the side that also launches the encoder's kernels)."""
    import numpy as np
    _TOK.enable_truncation(max_length=int(max_length))
    return [np.asarray(e.ids, dtype=np.int32) for e in _TOK.encode_batch(list(prompts))]
