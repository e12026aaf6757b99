"""Same-box A/B of wide-kernel builds (B = 256 in one corpus pass): run once per library,
    COMORAG_HIP_LIB=build_exp/lib_<variant>.so python tools/wide_ab.py <tag> [rows=10000000] [launches=30] [ref.npz]
prints the HIP-event time of the main scan per launch (synchronous search_dev, every launch timed) and checks the ids / scores against
ref.npz (written by the first variant that runs: every variant must return the SAME bits)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from comorag_amd.index import DenseIndex
tag = sys.argv[1]
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
ref = sys.argv[4] if len(sys.argv) > 4 else None
B, dim, k = 256, 768, 20
dev = torch.device("cuda", 0); g = torch.Generator(device=dev); g.manual_seed(7)
idx = DenseIndex(dim, "bf16", capacity_hint=rows)
for b in range(0, rows, 250_000):
    x = torch.randn((min(250_000, rows - b), dim), generator=g, device=dev)
    idx.append_dev((x / x.norm(dim=1, keepdim=True)).contiguous())
q = torch.randn((B, dim), generator=g, device=dev); q = (q / q.norm(dim=1, keepdim=True)).contiguous()
torch.cuda.synchronize()
for _ in range(3): ids, sc = idx.search_dev(q, k)
torch.cuda.synchronize()
got_i, got_s = ids.cpu().numpy().copy(), sc.cpu().numpy().copy()
same = None
if ref:
    if os.path.exists(ref):
        r = np.load(ref); same = bool(np.array_equal(r["ids"], got_i) and np.array_equal(r["sc"], got_s))
    else:
        np.savez(ref, ids=got_i, sc=got_s)
res = []
for rep in range(3):
    idx.profile(1)
    t0 = time.perf_counter()
    for _ in range(n): idx.search_dev(q, k)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    pr = idx.profile_collect()
    res.append((pr["total_ms"] / max(pr["launches"], 1), dt * 1e3))
kms = [r[0] for r in res]
print(f"{tag:10s} rows {rows} B {B}: wide kernel {min(kms):.3f} / {sorted(kms)[1]:.3f} / {max(kms):.3f} ms per launch (min / median / max of 3 x {n}), sync step {sorted(r[1] for r in res)[1]:.3f} ms, "
      f"{2.0 * B * rows * dim / (sorted(kms)[1] * 1e-3) / 1e12:.0f} TFLOP/s, {pr['bytes_per_launch'] / (sorted(kms)[1] * 1e-3) / 1e9:.0f} GB/s; equals reference bits: {same}", flush=True)
idx.close()
