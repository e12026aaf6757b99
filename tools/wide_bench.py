"""Wide-batch scan (B = 256 in one corpus pass) at full size: kernel / step time, and bit-equality with the
multi-pass narrow kernel (CMR_SCAN_NO_WIDE=1) on the same index contents.
    python tools/wide_bench.py [rows=10000000] [batch=256]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from comorag_amd.index import DenseIndex
from tools import env_options
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dim, k = 768, 20
dev = torch.device("cuda", 0); g = torch.Generator(device=dev); g.manual_seed(7)
blocks = []
for b in range(0, rows, 250_000):
    x = torch.randn((min(250_000, rows - b), dim), generator=g, device=dev); blocks.append((x / x.norm(dim=1, keepdim=True)).to(torch.float32).contiguous())
q = torch.randn((B, dim), generator=g, device=dev); q = (q / q.norm(dim=1, keepdim=True)).contiguous()
outs = [(torch.empty((B, k), dtype=torch.int64, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev)) for _ in range(2)]
ref = None
for name, env in [("wide", {}), ("narrow passes", {"CMR_SCAN_NO_WIDE": "1"})]:
    os.environ.pop("CMR_SCAN_NO_WIDE", None)
    os.environ.update(env)
    idx = DenseIndex(dim, "bf16", capacity_hint=rows, options=env_options())
    for x in blocks: idx.append_dev(x)
    torch.cuda.synchronize()
    ids, sc = idx.search_dev(q, k); torch.cuda.synchronize()
    got = (ids.cpu().numpy().copy(), sc.cpu().numpy().copy())
    if ref is None: ref = got
    same = np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1])
    idx.profile(True); n = 10
    t0 = time.perf_counter()
    for _ in range(n): idx.search_dev(q, k)
    torch.cuda.synchronize(); dts = (time.perf_counter() - t0) / n
    pr = idx.profile_collect(); ksync = pr["total_ms"] / max(pr["launches"], 1)
    for i in range(5): h = idx.search_pipelined(q, k, outs[i & 1][0], outs[i & 1][1])
    idx.sync(h); torch.cuda.synchronize()
    psame = np.array_equal(ref[0], outs[0][0].cpu().numpy()) and np.array_equal(ref[1], outs[0][1].cpu().numpy())
    idx.profile(True); n = 20
    t0 = time.perf_counter()
    for i in range(n): h = idx.search_pipelined(q, k, outs[i & 1][0], outs[i & 1][1])
    idx.sync(h); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    pr = idx.profile_collect(); kms = pr["total_ms"] / max(pr["launches"], 1)
    print(f"rows {rows} B {B} {name:14s}: equal to wide {same} (pipelined {psame}); sync step {dts*1e3:.3f} ms kernel {ksync:.3f} ms/launch | "
          f"pipelined step {dt*1e3:.3f} ms, kernel {kms:.3f} ms/launch ({pr['launches']} launches), {B/dt:.0f} q/s, "
          f"{pr['bytes_per_launch']/(kms*1e-3)/1e9:.0f} GB/s, {2.0*min(B,256)*rows*dim/(kms*1e-3)/1e12:.0f} TFLOP/s", flush=True)
    idx.close()
