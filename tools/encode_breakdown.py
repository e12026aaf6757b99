"""Where an encode step's time goes: tokenisation (host) vs encoder forward + pool (GPU), and end-to-end rates."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from comorag_amd.embedding_model.bge import HipBGEEmbeddingModel, tokenize_batch
from comorag_amd.utils.config_utils import BaseConfig
from tools.synthetic import random_bert, synthetic_chunks, synthetic_wordpiece_tokenizer
tok, words = synthetic_wordpiece_tokenizer()
chunks = synthetic_chunks(words, 256)
t0 = time.perf_counter(); [tok(chunks[:32], padding=True, truncation=True, max_length=512, return_tensors="pt") for _ in range(4)]; t1 = time.perf_counter()
[tokenize_batch(tok, chunks[:32], 512) for _ in range(4)]; t2 = time.perf_counter()
print(f"tokenise 32 chunks: HF pt {(t1-t0)/4*1e3:.1f} ms, tokenize_batch {(t2-t1)/4*1e3:.1f} ms; cpus {os.cpu_count()}", flush=True)
for dtype in ("bf16", "auto"):
    for threads in (1, 8):
        cfg = BaseConfig(embedding_model_name="bge-base-random-init", embedding_batch_size=32, embedding_model_dtype=dtype, device=0)
        cfg.embedding_tokenizer_threads = threads
        em = HipBGEEmbeddingModel(cfg, cfg.embedding_model_name, model=random_bert("base", vocab_size=len(tok)), tokenizer=tok)
        em.batch_encode(chunks[:64]); torch.cuda.synchronize()
        inp = em._tokenize(chunks[:32], 512)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(4): em._forward_pool(inp, True)
        torch.cuda.synchronize(); fwd = (time.perf_counter() - t0) / 4
        t0 = time.perf_counter(); em.batch_encode(chunks); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{dtype} tokenizer_threads {threads}: forward+pool {fwd*1e3:.2f} ms/batch; end-to-end {256/dt:.0f} chunks/s", flush=True)
