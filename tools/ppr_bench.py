"""DPR-seeded PPR per query (cmr_index_ppr) and the PageRank alone (cmr_graph_ppr), median wall time per call.
    python tools/ppr_bench.py [passages=5000] [entities=1500] [threads=1]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from comorag_amd.index import DenseIndex
from comorag_amd.ppr import DeviceGraph, ppr_passage_scores
n_pass = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
n_ent = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
n_thr = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dim = 768
rng = np.random.default_rng(7001)
X = rng.standard_normal((n_pass, dim)).astype(np.float32); X /= np.linalg.norm(X, axis=1, keepdims=True)
idx = DenseIndex(dim, "f32"); idx.append(X)
nv = n_ent + n_pass
pv = (n_ent + np.arange(n_pass)).astype(np.int32)
src = np.concatenate([rng.integers(0, n_ent, 3 * n_pass), rng.integers(0, n_ent, 2 * n_ent)]).astype(np.int32)
dst = np.concatenate([np.repeat(pv, 3), rng.integers(0, n_ent, 2 * n_ent)]).astype(np.int32)
keep = src != dst; src, dst = src[keep], dst[keep]
w = rng.uniform(0.5, 1.5, len(src))
q = rng.standard_normal(dim).astype(np.float32); q /= np.linalg.norm(q)
phrase = np.zeros(nv); phrase[rng.integers(0, n_ent, 6)] = rng.uniform(0.2, 1.0, 6)
g = DeviceGraph(nv, src, dst, w); g.set_passage_vertices(pv)
for _ in range(5): ppr_passage_scores(idx, g, q, phrase, 0.05)
t = []
for _ in range(60):
    t0 = time.perf_counter(); ppr_passage_scores(idx, g, q, phrase, 0.05); t.append(time.perf_counter() - t0)
t2 = []
for _ in range(30):
    t0 = time.perf_counter(); g.ppr(phrase + 1e-3); t2.append(time.perf_counter() - t0)
print(f"{n_pass} passages / {n_ent} entities, {len(src)} edges: cmr_index_ppr {np.median(t)*1e6:.0f} us per query, cmr_graph_ppr alone {np.median(t2)*1e6:.0f} us", flush=True)
if n_thr > 1:
    def work():
        for _ in range(40): ppr_passage_scores(idx, g, q, phrase, 0.05)
    th = [threading.Thread(target=work) for _ in range(n_thr)]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    print(f"{n_thr} threads x 40 queries: {(time.perf_counter() - t0) / 40 / n_thr * 1e6:.0f} us per query (aggregate)", flush=True)
g.close(); idx.close()
