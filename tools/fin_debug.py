import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import retrieval_np as orc
from comorag_amd.index import DenseIndex
n, d, nq, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 20
X = orc.synthetic_corpus(n, d, seed=1); Q = orc.synthetic_queries(nq, d, seed=2, planted=X)
chain = DenseIndex(d, "bf16", options={"scan_fin": 0}); chain.append(X)
want = chain.search(Q, k)
for name, opts in [("fin", {}), ("fin dense=1 (merge decides)", {"scan_fin_dense": 1}), ("fin no asm ring", {"scan_asm_ring": 0}),
                   ("fin grid 64", {"scan_grid": 64}), ("fin grid 200", {"scan_grid": 200})]:
    idx = DenseIndex(d, "bf16", options=opts); idx.append(X)
    for rep in range(3):
        got = idx.search(Q, k)
        same = [bool(np.array_equal(a, b)) for a, b in zip(got, want)]
        miss = [len(set(want[0][i].tolist()) - set(got[0][i].tolist())) for i in range(nq)]
        print(name, "rep", rep, "ids/sc/min/max equal:", same, "missing per query:", miss, "mx", got[3][:2], want[3][:2], flush=True)
    idx.close()
