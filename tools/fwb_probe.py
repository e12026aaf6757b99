"""End-to-end corpus encode vs embedding_forward_batches (token budget of a forward mini-batch, in reference batches) and bucket window.
    python tools/fwb_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from comorag_amd.embedding_model.bge import HipBGEEmbeddingModel
from comorag_amd.utils.config_utils import BaseConfig
from tools.synthetic import random_bert, synthetic_chunks, synthetic_wordpiece_tokenizer
dev = torch.device("cuda", 0)
tok, words = synthetic_wordpiece_tokenizer()
for kind, dt, n in (("base", "bf16", 2048), ("large", "fp16", 782)):
    chunks = synthetic_chunks(words, n, tokens_per_chunk=560)
    model = random_bert(kind, vocab_size=len(tok))
    for fwb, win in ((1, 4), (2, 4), (4, 4), (2, 8), (4, 8)):
        cfg = BaseConfig(embedding_model_name=f"bge-{kind}-random-init", embedding_batch_size=32, embedding_model_dtype=dt, device=0)
        cfg.embedding_forward_batches = fwb; cfg.embedding_bucket_window = win
        em = HipBGEEmbeddingModel(cfg, cfg.embedding_model_name, model=model, tokenizer=tok)
        em.batch_encode(chunks[: 2 * 32 * max(fwb, win)]); em.batch_encode(chunks[: 2 * 32 * max(fwb, win)])
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); out = em.batch_encode(chunks); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"{kind} {dt} forward_batches {fwb} window {win}: {n / np.median(ts):.0f} chunks/s (min {n / max(ts):.0f}, max {n / min(ts):.0f})", flush=True)
        em.close(); del em
    del model
