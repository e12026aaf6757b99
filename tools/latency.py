"""Single-call latency of the host-buffer API (what ComoRAG's per-question threads see): python tools/latency.py [rows ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from comorag_amd.index import DenseIndex
from tools import env_options
dim, k = 768, int(os.environ.get("LAT_K", "20"))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
for rows in [int(x) for x in (sys.argv[1:] or ["6", "1000", "100000", "1000000", "2000000"])]:
    idx = DenseIndex(dim, os.environ.get("LAT_DTYPE", "bf16"), capacity_hint=rows, options=env_options())
    for b in range(0, rows, 250_000):
        x = torch.randn((min(250_000, rows - b), dim), generator=g, device=dev)
        idx.append_dev((x / x.norm(dim=1, keepdim=True)).contiguous())
    torch.cuda.synchronize()
    for B in [int(b) for b in os.environ.get("LAT_BATCHES", "1,8,64").split(",")]:
        q = np.random.default_rng(B).standard_normal((B, dim)).astype(np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
        for _ in range(5): idx.search(q, min(k, rows))
        t = []
        for _ in range(50):
            t0 = time.perf_counter(); idx.search(q, min(k, rows)); t.append(time.perf_counter() - t0)
        for _ in range(3): idx.scores(q[:1])
        ts = []
        for _ in range(30):
            t0 = time.perf_counter(); s = idx.scores(q[:1]); ts.append(time.perf_counter() - t0)
        print(f"rows {rows:>9} B {B:>3} k {min(k, rows):>3}: search median {np.median(t)*1e6:8.1f} us  min {np.min(t)*1e6:8.1f} us   | full scores(B=1) median {np.median(ts)*1e6:8.1f} us", flush=True)
    idx.close()
