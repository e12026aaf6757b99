"""cProfile of batch_encode(one short query): where the host side of a single-query encode goes (python tools/query_encode_profile.py)."""
import os, sys, cProfile, pstats, io
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from comorag_amd.embedding_model.bge import HipBGEEmbeddingModel
from comorag_amd.utils.config_utils import BaseConfig
from tools.synthetic import random_bert, synthetic_wordpiece_tokenizer
tok, words = synthetic_wordpiece_tokenizer()
cfg = BaseConfig(embedding_model_name="bge-base-random-init", embedding_model_dtype="bf16", embedding_query_cache=0)
em = HipBGEEmbeddingModel(cfg, cfg.embedding_model_name, model=random_bert("base", vocab_size=len(tok)), tokenizer=tok)
q = " ".join(words[:12])
for _ in range(10): em.batch_encode(q)
pr = cProfile.Profile(); pr.enable()
for _ in range(300): em.batch_encode(q)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
