"""A single short query through the encoder (what ComoRAG issues per question, ComoRAG.py:937-967 -> embedding_model.batch_encode(query)):
wall time per call, the tokenizer's share, and — under `rocprofv3 --kernel-trace` — kernels and device time per call.
    python tools/query_encode_latency.py [calls=60]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from comorag_amd.embedding_model.bge import HipBGEEmbeddingModel
from comorag_amd.utils.config_utils import BaseConfig
from tools.synthetic import random_bert, synthetic_wordpiece_tokenizer


def med(f, n):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts) * 1e6), float(np.min(ts) * 1e6)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    tok, words = synthetic_wordpiece_tokenizer()
    for kind, dtype in (("base", "bf16"), ("large", "fp16")):
        cfg = BaseConfig(embedding_model_name=f"bge-{kind}-random-init", embedding_model_dtype=dtype, embedding_query_cache=0)      # (every call a forward: the product keeps the last 256 single-string rows)
        em = HipBGEEmbeddingModel(cfg, cfg.embedding_model_name, model=random_bert(kind, vocab_size=len(tok)), tokenizer=tok)
        q = " ".join(words[:12])
        for _ in range(8):
            em.batch_encode(q)
        torch.cuda.synchronize()
        call = med(lambda: em.batch_encode(q), n)
        dev = med(lambda: em.batch_encode_dev(q), n)
        prompt = [em.embedding_config.encode_params.get("passage_instruction", "") + q]
        tk = med(lambda: em._ragged(prompt, 512), n)
        print(f"bge-{kind} {dtype}: batch_encode(one 12-word query) median {call[0]:.1f} us (min {call[1]:.1f}); rows left on the device {dev[0]:.1f}; "
              f"tokenizer alone {tk[0]:.1f}; encoder path {em.encoder_path}", flush=True)
        em.close()


if __name__ == "__main__":
    main()
