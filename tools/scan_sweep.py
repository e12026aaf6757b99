"""Developer sweep: time the scan under different knobs (env read at index creation).
    python tools/scan_sweep.py [rows] [batch]
Prints kernel ms (HIP events around the main scan), whole-step ms and GB/s."""
import os, sys, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from comorag_amd.index import DenseIndex
from tools import env_options

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dim, k = 768, int(os.environ.get("SWEEP_K", "20"))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
q = torch.randn((batch, dim), generator=g, device=dev); q = (q / q.norm(dim=1, keepdim=True)).contiguous()
blocks = []
def gen():
    for b in range(0, rows, 250_000):
        x = torch.randn((min(250_000, rows - b), dim), generator=g, device=dev)
        yield (x / x.norm(dim=1, keepdim=True)).contiguous()
variants = [dict(), dict(CMR_SCAN_NO_WIDE="1"), dict(CMR_SCAN_RING="8"), dict(CMR_SCAN_ASM_RING="0"), dict(CMR_SCAN_ASM_RING="0", CMR_SCAN_RING="8"),
            dict(CMR_SCAN_NO_SAMPLE="1"), dict(CMR_SCAN_GRID="128"), dict(CMR_SCAN_GRID="512")]
if len(sys.argv) > 3:
    variants = [dict(kv.split("=") for kv in v.split(",") if kv) for v in sys.argv[3:]]
for env in variants:
    for kk in ("CMR_SCAN_RING", "CMR_SCAN_ASM_RING", "CMR_SCAN_NO_SAMPLE", "CMR_SCAN_GRID", "CMR_SCAN_NO_WIDE", "CMR_PIPE_RESERVE_CUS", "CMR_WIDE_ABL"):
        os.environ.pop(kk, None)
    os.environ.update(env)
    idx = DenseIndex(dim, "bf16", capacity_hint=rows, options=env_options())
    g.manual_seed(2)
    for x in gen():
        idx.append_dev(x)
    torch.cuda.synchronize()
    for _ in range(5):
        idx.search_dev(q, k)
    torch.cuda.synchronize()
    steps = 30
    idx.profile(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        idx.search_dev(q, k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    idx.profile(False)
    p = idx.profile_collect()
    kms = p["total_ms"] / p["launches"]
    outs = [(torch.empty((batch, k), dtype=torch.int64, device=dev), torch.empty((batch, k), dtype=torch.float32, device=dev)) for _ in range(2)]
    for i in range(4):
        h = idx.search_pipelined(q, k, outs[i & 1][0], outs[i & 1][1])
    idx.sync(h); torch.cuda.synchronize()
    idx.profile(True)
    t0 = time.perf_counter()
    for i in range(steps):
        h = idx.search_pipelined(q, k, outs[i & 1][0], outs[i & 1][1])
    idx.sync(h); torch.cuda.synchronize()
    dt2 = (time.perf_counter() - t0) / steps * 1e3
    idx.profile(False)
    p2 = idx.profile_collect(); kms2 = p2["total_ms"] / max(p2["launches"], 1)
    print(f"{env!s:50s} kernel {kms:7.3f} ms step {dt:7.3f} ms | pipelined: kernel {kms2:7.3f} ms step {dt2:7.3f} ms  qps {batch/dt*1e3:8.0f} / {batch/dt2*1e3:8.0f}", flush=True)
    idx.close()
