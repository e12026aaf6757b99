"""Encoder parity probe: fused 16-bit stack vs the fp32 oracle (and the transformers forward in the same dtype) at BGE shapes,
for several sharpening factors of the random-init attention.  python tools/encoder_parity_probe.py [base|large] [bf16|fp16] [scales]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from comorag_amd.embedding_model.bge import HipBGEEmbeddingModel
from comorag_amd.utils.config_utils import BaseConfig
from tools.bench_extras import encoder_parity
from tools.synthetic import random_bert, synthetic_chunks, synthetic_wordpiece_tokenizer
kind = sys.argv[1] if len(sys.argv) > 1 else "base"
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
scales = [float(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "1,2,3").split(",")]
tok, words = synthetic_wordpiece_tokenizer()
chunks = synthetic_chunks(words, 32, tokens_per_chunk=560)
for sc in scales:
    model = random_bert(kind, vocab_size=len(tok))
    with torch.no_grad():
        for lyr in model.encoder.layer:
            lyr.attention.self.query.weight.mul_(sc)
            lyr.attention.self.key.weight.mul_(sc)
    cfg = BaseConfig(embedding_model_name=f"bge-{kind}-random-init", embedding_batch_size=8, embedding_model_dtype=dtype)
    em = HipBGEEmbeddingModel(cfg, cfg.embedding_model_name, model=model, tokenizer=tok)
    r = encoder_parity(torch, em, chunks)
    print(kind, dtype, "scale", sc, em.encoder_path, json.dumps({k: v for k, v in r.items() if k not in ("bar", "oracle")}), flush=True)
    em.close()
