"""Pipelined search loop only (for rocprofv3 kernel traces): python tools/pipe_only.py [rows] [batch] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from comorag_amd.index import DenseIndex
from tools import env_options
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
dim, k = 768, 20
dev = torch.device("cuda", 0); g = torch.Generator(device=dev); g.manual_seed(7)
idx = DenseIndex(dim, "bf16", capacity_hint=rows, options=env_options())
for b in range(0, rows, 250_000):
    x = torch.randn((min(250_000, rows - b), dim), generator=g, device=dev); idx.append_dev((x / x.norm(dim=1, keepdim=True)).contiguous())
q = torch.randn((B, dim), generator=g, device=dev); q = (q / q.norm(dim=1, keepdim=True)).contiguous()
outs = [(torch.empty((B, k), dtype=torch.int64, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev)) for _ in range(2)]
torch.cuda.synchronize()
for i in range(4): h = idx.search_pipelined(q, k, outs[i & 1][0], outs[i & 1][1])
idx.sync(h); torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps): h = idx.search_pipelined(q, k, outs[i & 1][0], outs[i & 1][1])
t1 = time.perf_counter()
idx.sync(h); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
# the last batch against a synchronous search of the same queries (bit for bit)
import numpy as np
si, ss = idx.search_dev(q, k); torch.cuda.synchronize()
same = bool(torch.equal(si, outs[(steps - 1) & 1][0]) and torch.equal(ss, outs[(steps - 1) & 1][1]))
print(f"options {env_options()} pipelined == synchronous: {same}; ", end="")
print(f"rows {rows} B {B}: host enqueue {(t1 - t0) / steps * 1e6:.1f} us/step; pipelined step {dt*1e3:.3f} ms = {B/dt:.0f} q/s", flush=True)
idx.close()
