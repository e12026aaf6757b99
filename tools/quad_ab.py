"""A/B of the wide-batch routes on one index: register-resident wide kernel (wide_mode=1) vs the query-split grid of the
narrow kernel (wide_mode=2) with both cache policies.  Synchronous search_dev calls timed with the library's HIP events
(kernel) and wall clock (call); results of every route compared bit for bit with route 1.

    python tools/quad_ab.py [rows] [batches e.g. 256,192,128] [reps]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from comorag_amd.index import DenseIndex

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
batches = [int(b) for b in (sys.argv[2] if len(sys.argv) > 2 else "256").split(",")]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
dim, k = 768, 20
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(7)
idx = DenseIndex(dim, "bf16", capacity_hint=rows)
for b in range(0, rows, 250_000):
    x = torch.randn((min(250_000, rows - b), dim), generator=g, device=dev)
    idx.append_dev((x / x.norm(dim=1, keepdim=True)).contiguous())
torch.cuda.synchronize()
routes = [("wide_kernel", {"wide_mode": 1}), ("quad_default_policy", {"wide_mode": 2, "stream_nt": -1}),
          ("quad_nt", {"wide_mode": 2, "stream_nt": 1}), ("narrow_passes", {"scan_no_wide": 1})]
if os.environ.get("QUAD_AB_ROUTES"):
    routes = [r for r in routes if r[0] in os.environ["QUAD_AB_ROUTES"].split(",")]
for B in batches:
    q = torch.randn((B, dim), generator=g, device=dev)
    q = (q / q.norm(dim=1, keepdim=True)).contiguous()
    ref = None
    for name, opts in routes:
        for o in ("wide_mode", "scan_no_wide"):
            idx.set_option(o, 0)
        idx.set_option("stream_nt", -1)
        for o, v in opts.items():
            idx.set_option(o, v)
        for _ in range(3):
            ids, sc = idx.search_dev(q, k)
        torch.cuda.synchronize()
        idx.profile(1)
        t0 = time.perf_counter()
        for _ in range(reps):
            ids, sc = idx.search_dev(q, k)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps
        idx.profile(False)
        pr = idx.profile_collect()
        kms = pr["total_ms"] / max(pr["launches"], 1)
        per_call = pr["launches"] / reps
        same = None
        if ref is None:
            ref = (ids.clone(), sc.clone())
        else:
            same = bool(torch.equal(ids, ref[0]) and torch.equal(sc, ref[1]))
        by = rows * dim * 2
        print(f"rows {rows} B {B} {name:22s}: main scan {kms:.3f} ms x {per_call:.0f}/call = {by / (kms * 1e-3) / 1e12:.2f} TB/s, "
              f"{2.0 * min(B, 256) * rows * dim / per_call / (kms * 1e-3) / 1e15 if per_call else 0:.3f} PF/s per launch; call {wall * 1e3:.3f} ms; == wide kernel: {same}", flush=True)
idx.close()
