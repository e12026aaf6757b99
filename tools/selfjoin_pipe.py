"""The synonymy self-join by route: one stream (cmr_index_search_min_score_dev) against throughput mode
(cmr_index_search_min_score_pipelined: what retrieval.retrieve_knn(min_score=) runs), by query-block size; passes and us per pass.
    python tools/selfjoin_pipe.py [entities=200000] [dtype=bf16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from comorag_amd.index import DenseIndex
E = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dim, thr, K = 768, 0.8, 128
dev = torch.device("cuda", 0); g = torch.Generator(device=dev); g.manual_seed(3)
x = torch.randn((E, dim), generator=g, device=dev)
dup = torch.randint(0, E, (E // 10,), generator=g, device=dev)
x[:E // 10] = x[dup] + 0.15 * x[:E // 10]
x = (x / x.norm(dim=1, keepdim=True)).contiguous()
opts = {}
for o in os.environ.get("CMR_OPTS", "").split(","):
    if "=" in o: opts[o.split("=")[0]] = int(o.split("=")[1])
idx = DenseIndex(dim, dtype, capacity_hint=E, options=opts); idx.append_dev(x); torch.cuda.synchronize()
ids_t = torch.empty((E, K), dtype=torch.int64, device=dev); sc_t = torch.empty((E, K), dtype=torch.float32, device=dev)
flops = 2.0 * E * E * dim
ref = None
routes = [("one stream", 1024), ("pipelined", 1000), ("pipelined", 1024), ("pipelined", 2048), ("pipelined", 4096), ("pipelined", 8192), ("one stream", 4096)]
if os.environ.get("SJ_ROUTES"):
    routes = [(("pipelined" if r[0] == "p" else "one stream"), int(r[1:])) for r in os.environ["SJ_ROUTES"].split(",")]
for mode, blk in routes:
    best = None
    for rep in range(3):
        ids_t.fill_(-7); torch.cuda.synchronize()
        t0 = time.perf_counter(); done = None
        for b0 in range(0, E, blk):
            b1 = min(b0 + blk, E)
            if mode == "pipelined": done = idx.search_min_score_pipelined(x[b0:b1], K, thr, ids_t[b0:b1], sc_t[b0:b1])
            else: idx.search_min_score_dev(x[b0:b1], K, thr, ids_t[b0:b1], sc_t[b0:b1])
        if done is not None: idx.sync(done)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    h = ids_t.cpu().numpy()
    ref = h if ref is None else ref
    passes = sum((min(b0 + blk, E) - b0 + 255) // 256 for b0 in range(0, E, blk))
    print(f"{E} x {E} x {dim} {dtype}, {mode:10s} blocks of {blk:5d}: {best*1e3:7.2f} ms = {flops / best / 1e12:6.0f} TFLOP/s = {flops / best / 2.5e15:.3f} of 2.5 PF; "
          f"{passes} passes of <= 256 queries, {best / passes * 1e6:6.1f} us per pass; same ids as the first route: {np.array_equal(h, ref)}", flush=True)
idx.close()
