import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import retrieval_np as orc
from comorag_amd.index import DenseIndex
from tools import env_options
n0, d, k = 2_000_000, 768, 20
g = torch.Generator(device="cuda"); g.manual_seed(99)
idx = DenseIndex(d, "bf16", capacity_hint=n0, options=env_options())
host = []
for _ in range(8):
    x = torch.randn((n0 // 8, d), generator=g, device="cuda"); x = (x / x.norm(dim=1, keepdim=True)).contiguous()
    idx.append_dev(x); host.append(x.cpu().numpy())
X = np.concatenate(host); del host
rng = np.random.default_rng(5)
n = len(X)
for cycle in range(5):
    probes = orc.synthetic_queries(8, d, seed=1000 + cycle, planted=X[rng.integers(0, n0, 4)])
    ids, sc, mn, mx = idx.search(probes, k)
    n_new = 65_536 if cycle == 2 else 25
    new = orc.synthetic_corpus(n_new, d, seed=2000 + cycle)
    new[0] = probes[0]
    idx.append(new)
    n += n_new
    for rep in range(3):
        hit, hsc, _, hmx = idx.search(probes[:1], 1)
        print("cycle", cycle, "rep", rep, "rows", n, "hit", hit[0, 0], "want", n - n_new, "score", hsc[0, 0], "max", hmx[0], flush=True)
idx.close()
