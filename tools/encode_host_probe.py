"""Host-side behaviour of the threaded corpus encode on this box: end-to-end rate and where the launching thread's time goes, for a
few tokenizer-thread / rayon settings (each variant in a fresh interpreter: rayon's pool size is fixed at first use).
`python -m tools.encode_host_probe`"""
import json
import os
import subprocess
import sys

CHILD = r'''
import json, os, sys, time, torch
from tools import bench_extras as bx
from comorag_amd.embedding_model.bge import HipBGEEmbeddingModel
from comorag_amd.utils.config_utils import BaseConfig
from tools.synthetic import random_bert, synthetic_chunks, synthetic_wordpiece_tokenizer
threads, n = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0)
tok, words = synthetic_wordpiece_tokenizer()
cfg = BaseConfig(embedding_model_name="bge-base-random-init", embedding_batch_size=32, embedding_model_dtype="bf16", embedding_tokenizer_threads=threads)
em = HipBGEEmbeddingModel(cfg, cfg.embedding_model_name, model=random_bert("base", vocab_size=len(tok)), tokenizer=tok)
chunks = synthetic_chunks(words, n, tokens_per_chunk=560)
em.batch_encode(chunks[:96]); torch.cuda.synchronize()
em._trace = []
t0 = time.perf_counter(); em.batch_encode(chunks); torch.cuda.synchronize(); dt = time.perf_counter() - t0
tr = em._trace
print(json.dumps({"chunks_per_s": n / dt, "ms": dt * 1e3, "wait_tokens_ms": sum(t[0] for t in tr) * 1e3, "pad_launch_ms": sum(t[1] for t in tr) * 1e3,
                  "windows": len(tr), "affinity": len(os.sched_getaffinity(0)), "cpu_count": os.cpu_count()}))
em.close()
'''


def main():
    try:
        print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
    except OSError as e:
        print("cgroup cpu.max: n/a", e)
    for label, env, threads, n in (("default (2 tokenizer threads)", {}, 2, 1024), ("RAYON_NUM_THREADS=8", {"RAYON_NUM_THREADS": "8"}, 2, 1024),
                                   ("RAYON_NUM_THREADS=16, 4 threads", {"RAYON_NUM_THREADS": "16"}, 4, 1024),
                                   ("TOKENIZERS_PARALLELISM=false, 8 threads", {"TOKENIZERS_PARALLELISM": "false"}, 8, 1024),
                                   ("1 tokenizer thread", {}, 1, 1024)):
        r = subprocess.run([sys.executable, "-c", CHILD, str(threads), str(n)], env={**os.environ, **env}, capture_output=True, text=True, timeout=300)
        print(label, "->", (r.stdout.strip().splitlines() or [r.stderr[-400:]])[-1])


if __name__ == "__main__":
    main()
