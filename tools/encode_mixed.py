"""Corpus-embed rate on chunks of MIXED length (40..480 tokens), length-bucketed mini-batches vs arrival order.
    python tools/encode_mixed.py [n_chunks=512] [dtype=bf16]"""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from comorag_amd.embedding_model.bge import HipBGEEmbeddingModel
from comorag_amd.utils.config_utils import BaseConfig
from tools.synthetic import random_bert, synthetic_wordpiece_tokenizer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
tok, words = synthetic_wordpiece_tokenizer()
rnd = random.Random(5)
chunks = [" ".join(rnd.choice(words) for _ in range(rnd.choice([40, 80, 160, 320, 480]))) for _ in range(n)]
model = random_bert("base", vocab_size=len(tok))
for bucket in (True, False):
    cfg = BaseConfig(embedding_model_name="bge-base-random-init", embedding_batch_size=32, embedding_model_dtype=dtype, embedding_length_bucketing=bucket)
    em = HipBGEEmbeddingModel(cfg, cfg.embedding_model_name, model=model, tokenizer=tok)
    em.batch_encode(chunks[:64]); torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter(); out = em.batch_encode(chunks); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"mixed-length chunks, {dtype}, bucketing {bucket}, pass {rep}: {n / dt:.0f} chunks/s", flush=True)
