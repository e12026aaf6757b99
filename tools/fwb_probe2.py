"""One forward mini-batch of b x 512-token chunks through HipBGEEmbeddingModel._forward_ragged: wall time per call (synchronised) for b = 32, 64, 128,
eager and as a replayed hipGraph, and the kernel list of one call.  python tools/fwb_probe2.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from comorag_amd.embedding_model.bge import HipBGEEmbeddingModel
from comorag_amd.utils.config_utils import BaseConfig
from tools.synthetic import random_bert, synthetic_chunks, synthetic_wordpiece_tokenizer
tok, words = synthetic_wordpiece_tokenizer()
chunks = synthetic_chunks(words, 128, tokens_per_chunk=560)
model = random_bert("base", vocab_size=len(tok))
for graphs in (0, 24):
    cfg = BaseConfig(embedding_model_name="bge-base-random-init", embedding_batch_size=32, embedding_model_dtype="bf16", device=0)
    cfg.embedding_hip_graphs = graphs
    em = HipBGEEmbeddingModel(cfg, cfg.embedding_model_name, model=model, tokenizer=tok)
    ids = em._ragged(chunks, 512)
    for b in (32, 64, 128):
        for _ in range(4): em._forward_ragged(ids[:b], True)
        torch.cuda.synchronize()
        ts = []
        for _ in range(8):
            t0 = time.perf_counter(); em._forward_ragged(ids[:b], True); t1 = time.perf_counter(); torch.cuda.synchronize(); ts.append((t1 - t0, time.perf_counter() - t0))
        h = np.median([a for a, _ in ts]) * 1e3; w = np.median([c for _, c in ts]) * 1e3
        print(f"graphs {graphs} b {b}: host call {h:.2f} ms, wall incl. sync {w:.2f} ms = {b / w * 1e3:.0f} chunks/s", flush=True)
    if graphs == 0:
        from torch.profiler import profile, ProfilerActivity
        for b in (32, 64):
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                em._forward_ragged(ids[:b], True); torch.cuda.synchronize()
            rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:9]
            print(f"  b {b} top kernels:")
            for e in rows: print(f"    {e.key[:90]:90s} x{e.count:3d} {e.device_time_total / 1e3:8.2f} ms")
    em.close()
