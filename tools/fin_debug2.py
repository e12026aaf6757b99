import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import retrieval_np as orc
from comorag_amd.index import DenseIndex
n, d = int(sys.argv[1]), int(sys.argv[2])
X = orc.synthetic_corpus(n, d, seed=1)
P = orc.synthetic_queries(8, d, seed=2, planted=X[[5, 77, 1000, 4242]])
for name, opts in [("chain", {"scan_fin": 0}), ("fin", {})]:
    idx = DenseIndex(d, "bf16", capacity_hint=n, options=opts); idx.append(X)
    a = idx.search(P, 20)
    new = orc.synthetic_corpus(25, d, seed=3); new[0] = P[0]
    idx.append(new)
    for k in (1, 2, 20):
        hit, hsc, mn, mx = idx.search(P[:1], k)
        print(name, "k", k, "hit", hit[0][:3], "score", hsc[0][:3], "want row", n, "min/max", mn, mx, flush=True)
    idx.close()
