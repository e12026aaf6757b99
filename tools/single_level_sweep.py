"""Synchronous search latency (median us, Python wrapper included) against the single-sampling-level limit (queries x panels):
python tools/single_level_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from comorag_amd.index import DenseIndex
dev = torch.device("cuda", 0); dim, k = 768, 20
g = torch.Generator(device=dev); g.manual_seed(11)
rng = np.random.default_rng(12)
for rows in (1_000_000, 2_000_000, 4_000_000):
    idx = DenseIndex(dim, "bf16", capacity_hint=rows)
    for b in range(0, rows, 250_000):
        x = torch.randn((250_000, dim), generator=g, device=dev); idx.append_dev((x / x.norm(dim=1, keepdim=True)).contiguous())
    torch.cuda.synchronize()
    for nq in (1, 2, 4, 8):
        q = rng.standard_normal((nq, dim)).astype(np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
        ref, row = None, []
        for lim in (0, 320_000, 640_000, 1_280_000, 10_000_000):
            idx.set_option("sample_single_max", lim)
            for _ in range(4): out = idx.search(q, k)
            t = []
            for _ in range(40):
                t0 = time.perf_counter(); out = idx.search(q, k); t.append(time.perf_counter() - t0)
            if ref is None: ref = out
            same = all(np.array_equal(a, b) for a, b in zip(ref, out))
            row.append(f"{lim}: {np.median(t) * 1e6:.0f}{'' if same else ' MISMATCH'}")
        print(f"rows {rows} nq {nq} (q x panels {nq * rows // 32}): " + " | ".join(row), flush=True)
    idx.close()
