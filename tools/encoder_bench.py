"""Corpus-embed rate and per-stage breakdown of the encoder path (tools/bench_extras.encode_breakdown) for the BERT-base bf16 and
BERT-large fp16 shapes, fused layer stack against the transformers forward.  `python -m tools.encoder_bench [--quick]`."""
import json
import sys

import torch

from tools import bench_extras as bx


def query_latency(kind, dtype, dev, reps=40):
    """Median wall time of batch_encode(one short query) — what get_query_embeddings (ComoRAG.py:909-935) waits for."""
    import time
    import numpy as np
    from comorag_amd.embedding_model.bge import HipBGEEmbeddingModel
    from comorag_amd.utils.config_utils import BaseConfig
    from tools.synthetic import random_bert, synthetic_wordpiece_tokenizer
    tok, words = synthetic_wordpiece_tokenizer()
    out = {}
    for fused in (True, False):
        cfg = BaseConfig(embedding_model_name=f"bge-{kind}-random-init", embedding_model_dtype=dtype, device=dev.index or 0, embedding_fused_encoder=fused)
        em = HipBGEEmbeddingModel(cfg, cfg.embedding_model_name, model=random_bert(kind, vocab_size=len(tok)), tokenizer=tok)
        q = " ".join(words[:12])
        for _ in range(5):
            em.batch_encode(q)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); em.batch_encode(q); ts.append(time.perf_counter() - t0)
        out[em.encoder_path] = float(np.median(ts) * 1e6)
        em.close()
    return out


def main():
    quick = "--quick" in sys.argv
    dev = torch.device("cuda", 0)
    cases = [("base", "bf16", 256, 0), ("base", "bf16", 256, 4), ("large", "fp16", 256, 0)]
    if quick:
        cases = cases[:1]
    for kind, dtype, n, procs in cases:
        res, em = bx.encode_breakdown(torch, dev, kind, dtype, n_chunks=n, batch=32, tok_processes=procs)
        em.close()
        keep = ("model", "value", "forward_only_chunks_per_s", "transformers_forward_only_chunks_per_s", "tokenizer_only_chunks_per_s",
                "forward_TFLOPs", "frac", "attention_us_per_layer", "attention_TFLOPs", "add_layernorm_us", "add_layernorm_GBps",
                "pool_l2norm_us_per_batch", "end_to_end_over_forward_only", "tokenizer_processes", "encoder_path", "host_ms")
        line = {k: res[k] for k in keep if k in res}
        if procs == 0:
            line["single_query_encode_us"] = query_latency(kind, dtype, dev)
        print(json.dumps(line))


if __name__ == "__main__":
    main()
