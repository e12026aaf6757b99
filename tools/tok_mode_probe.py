"""End-to-end corpus encode (BERT-base bf16, 1024 chunks of 512 tokens) per tokenizer mode on THIS box: threads only, worker processes
started lazily (the default), worker processes with a bounded rayon pool.  python tools/tok_mode_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools import bench_extras as bx
dev = torch.device("cuda", 0)
for rep in range(2):
    for name, tp, env in (("threads", 0, None), ("procs-auto", -1, None), ("procs-rayon8", -1, "8"), ("procs-rayon2", -1, "2"), ("procs8-rayon4", -8, "4")):
        if env is None: os.environ.pop("CMR_TOKWORKER_RAYON", None)
        else: os.environ["CMR_TOKWORKER_RAYON"] = env
        res, em = bx.encode_breakdown(torch, dev, "base", "bf16", 1024, tok_processes=tp, parity=False)
        print(f"rep {rep} {name:14s}: e2e {res['value']:.0f} chunks/s, forward-only {res['forward_only_chunks_per_s']:.0f}, ratio {res['end_to_end_over_forward_only']:.3f}, "
              f"tokenizer-only {res['tokenizer_only_chunks_per_s']:.0f}, host {res['host_ms']}", flush=True)
        em.close()
