"""The finishing stage at full size: same index, scan_fin toggled; results bit for bit, time per call.  python tools/fin_large.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from comorag_amd.index import DenseIndex
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim, k = 768, 20
g = torch.Generator(device="cuda"); g.manual_seed(3)
idx = DenseIndex(dim, "bf16", capacity_hint=rows)
for b in range(0, rows, 250_000):
    x = torch.randn((min(250_000, rows - b), dim), generator=g, device="cuda")
    idx.append_dev((x / x.norm(dim=1, keepdim=True)).contiguous())
torch.cuda.synchronize()
for B in (1, 8):
    q = np.random.default_rng(B).standard_normal((B, dim)).astype(np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
    out = {}
    for fin in (0, 1):
        idx.set_option("scan_fin", fin)
        for _ in range(3): r = idx.search(q, k)
        t = []
        for _ in range(12):
            t0 = time.perf_counter(); r = idx.search(q, k); t.append(time.perf_counter() - t0)
        out[fin] = (r, np.median(t) * 1e6)
    same = all(np.array_equal(a, b) for a, b in zip(out[0][0], out[1][0]))
    print(f"rows {rows} B {B}: chain {out[0][1]:.1f} us, finishing stage {out[1][1]:.1f} us, results identical: {same}", flush=True)
idx.close()
