"""The finishing stage against the chain, call after call on one workspace, with and without the polled done word."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from comorag_amd.index import DenseIndex
n, d, nq, k = 1_000_003, 64, 8, 20
rng = np.random.default_rng((n + nq + k) % 997)
X = rng.standard_normal((n, d)).astype(np.float32); X /= np.linalg.norm(X, axis=1, keepdims=True)
Q = rng.standard_normal((nq, d)).astype(np.float32); Q /= np.linalg.norm(Q, axis=1, keepdims=True)
chain = DenseIndex(d, "bf16", options={"scan_fin": 0}); chain.append(X)
for poll in (0, 1):
    fin = DenseIndex(d, "bf16", options={"scan_fin_queries": 32, "sync_poll": poll}); fin.append(X)
    bad = 0
    for rep in range(40):
        m = (8, 1, 8, 7)[rep % 4]
        a = fin.search(Q[:m], k); b = chain.search(Q[:m], k)
        for name, x, y in zip(("ids", "scores", "min", "max"), a, b):
            if not np.array_equal(x, y):
                bad += 1
                w = np.argwhere(np.asarray(x) != np.asarray(y))
                print(f"poll {poll} call {rep} m {m}: {name} differ at {w[:6].tolist()} ({len(w)} places): {np.asarray(x)[tuple(w[0])]} vs {np.asarray(y)[tuple(w[0])]}", flush=True)
    print(f"sync_poll {poll}: {bad} mismatching fields in 40 calls", flush=True)
    fin.close()
