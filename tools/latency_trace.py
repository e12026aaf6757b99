"""A few synchronous single-query searches per corpus size, for a rocprofv3 kernel trace of the dependent launch chain
(tools/rocpd_timeline.py prints it):  rocprofv3 --kernel-trace -d out -o w -- python tools/latency_trace.py [rows ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from comorag_amd.index import DenseIndex
from tools import env_options
dim, k = 768, 20
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
for rows in [int(x) for x in (sys.argv[1:] or ["100000", "1000000"])]:
    idx = DenseIndex(dim, "bf16", capacity_hint=rows, options=env_options())
    for b in range(0, rows, 250_000):
        x = torch.randn((min(250_000, rows - b), dim), generator=g, device=dev)
        idx.append_dev((x / x.norm(dim=1, keepdim=True)).contiguous())
    torch.cuda.synchronize()
    B = int(os.environ.get("LAT_B", "1"))
    q = np.random.default_rng(1).standard_normal((B, dim)).astype(np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
    for _ in range(5): idx.search(q, k)
    t = []
    for _ in range(12):
        t0 = time.perf_counter(); idx.search(q, k); t.append(time.perf_counter() - t0)
    print(f"rows {rows}: median {np.median(t)*1e6:.1f} us (under the tracer)", flush=True)
    idx.close()
