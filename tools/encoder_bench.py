"""Corpus-embed rate and per-stage breakdown of the encoder path (tools/bench_extras.encode_breakdown) for the BERT-base bf16 and
BERT-large fp16 shapes, fused layer stack against the transformers forward.  `python -m tools.encoder_bench [--quick]`."""
import json
import sys

import torch

from tools import bench_extras as bx


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
                "pool_l2norm_us_per_batch", "end_to_end_over_forward_only", "tokenizer_processes", "encoder_path")
        print(json.dumps({k: res[k] for k in keep if k in res}))


if __name__ == "__main__":
    main()
