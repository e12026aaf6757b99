"""Round-2 check of the experimental wide-kernel variants (batch 256 in one corpus pass):
default (8-block groups x 12 stages) vs CMR_WIDE_GROUP=16 vs CMR_WIDE_STAGGER=1 vs CMR_WIDE_BURST=1.
For each: ids/scores must equal the default variant bit for bit; prints step and kernel time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from comorag_amd.index import DenseIndex
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim, B, k = 768, 256, 20
dev = torch.device("cuda", 0); g = torch.Generator(device=dev); g.manual_seed(7)
blocks = []
for b in range(0, rows, 250_000):
    x = torch.randn((min(250_000, rows - b), dim), generator=g, device=dev); blocks.append((x / x.norm(dim=1, keepdim=True)).to(torch.float32).contiguous())
q = torch.randn((B, dim), generator=g, device=dev); q = (q / q.norm(dim=1, keepdim=True)).contiguous()
outs = [(torch.empty((B, k), dtype=torch.int64, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev)) for _ in range(2)]
ref = None
for name, env in [("default 8x12", {}), ("group 16x6", {"CMR_WIDE_GROUP": "16"}), ("staggered DMA", {"CMR_WIDE_STAGGER": "1"}),
                  ("paired MFMAs", {"CMR_WIDE_BURST": "1"})]:
    for kk in ("CMR_WIDE_GROUP", "CMR_WIDE_STAGGER", "CMR_WIDE_BURST"): os.environ.pop(kk, None)
    os.environ.update(env)
    idx = DenseIndex(dim, "bf16", capacity_hint=rows)
    for x in blocks: idx.append_dev(x)
    torch.cuda.synchronize()
    for i in range(5): h = idx.search_pipelined(q, k, outs[i & 1][0], outs[i & 1][1])
    idx.sync(h); torch.cuda.synchronize()
    got = (outs[0][0].cpu().numpy().copy(), outs[0][1].cpu().numpy().copy())
    if ref is None: ref = got
    same = np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1])
    idx.profile(True); n = 20
    t0 = time.perf_counter()
    for i in range(n): h = idx.search_pipelined(q, k, outs[i & 1][0], outs[i & 1][1])
    idx.sync(h); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    pr = idx.profile_collect(); kms = pr["total_ms"] / max(pr["launches"], 1)
    print(f"{name:14s}: equal to default {same}; step {dt*1e3:.3f} ms, kernel {kms:.3f} ms, {B/dt:.0f} q/s, "
          f"{2.0*B*rows*dim/(kms*1e-3)/1e12:.0f} TFLOP/s", flush=True)
    idx.close()
