"""Synonymy self-join at entity scale: all-pairs neighbours with score >= 0.8 over E x E entity vectors.
threshold-filter path (fused kernel started at the threshold) vs the materialise-and-select path (k = 2047).
    python tools/selfjoin_bench.py [entities=200000] [dtype=bf16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from comorag_amd.index import DenseIndex
E = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dim, thr = 768, 0.8
dev = torch.device("cuda", 0); g = torch.Generator(device=dev); g.manual_seed(3)
x = torch.randn((E, dim), generator=g, device=dev)
dup = torch.randint(0, E, (E // 10,), generator=g, device=dev)                 # 10 % of the entities get a near-duplicate
x[:E // 10] = x[dup] + 0.15 * x[:E // 10]
x = (x / x.norm(dim=1, keepdim=True)).contiguous()
idx = DenseIndex(dim, dtype, capacity_hint=E); idx.append_dev(x); torch.cuda.synchronize()
xh = x.cpu().numpy()
B = 1024
def run(fn, nb):
    fn(xh[:B]); t0 = time.perf_counter()
    for b in range(nb): fn(xh[b * B:(b + 1) * B])
    return (time.perf_counter() - t0) / nb
nb = min(8, E // B)
t_thr = run(lambda q: idx.search_min_score(q, 128, thr), nb)
t_mat = run(lambda q: idx.search(q, 2047, with_minmax=False), max(1, nb // 4))
a = idx.search_min_score(xh[:B], 128, thr); b = idx.search(xh[:B], 2047, with_minmax=False)
same = all(np.array_equal(a[0][i][a[0][i] >= 0], b[0][i][b[1][i] >= thr][:128]) for i in range(B))
print(f"entities {E} {dtype}: threshold-filter {t_thr*1e3:.2f} ms per {B} queries ({E / B * t_thr:.2f} s for the whole self-join), "
      f"materialise + select k=2047 {t_mat*1e3:.2f} ms per {B} queries ({E / B * t_mat:.2f} s); {t_mat / t_thr:.1f}x; same neighbours >= {thr}: {same}", flush=True)
idx.close()
