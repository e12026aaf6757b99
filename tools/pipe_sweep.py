"""Pipelined-search step time vs CMR_PIPE_RESERVE_CUS and shard size (one process, several indices)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from comorag_amd.index import DenseIndex
from tools import env_options
batch, dim, k = 64, 768, 20
rows_list = [int(x) for x in sys.argv[1].split(",")]
reserves = [int(x) for x in sys.argv[2].split(",")]
qglobals = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "32").split(",")]   # CMR_SAMPLE_DIV values
dev = torch.device("cuda", 0); g = torch.Generator(device=dev); g.manual_seed(1)
q = torch.randn((batch, dim), generator=g, device=dev); q = (q / q.norm(dim=1, keepdim=True)).contiguous()
outs = [(torch.empty((batch, k), dtype=torch.int64, device=dev), torch.empty((batch, k), dtype=torch.float32, device=dev)) for _ in range(2)]
for rows in rows_list:
    blocks = []
    for b in range(0, rows, 250_000):
        x = torch.randn((min(250_000, rows - b), dim), generator=g, device=dev); blocks.append((x / x.norm(dim=1, keepdim=True)).contiguous())
    for rsv, qgl in [(r, g_) for g_ in qglobals for r in reserves]:
        os.environ["CMR_PIPE_RESERVE_CUS"] = str(rsv)
        os.environ["CMR_SAMPLE_DIV"] = str(qgl)
        idx = DenseIndex(dim, "bf16", capacity_hint=rows, options=env_options())
        for x in blocks: idx.append_dev(x)
        torch.cuda.synchronize()
        for i in range(20): h = idx.search_pipelined(q, k, outs[i & 1][0], outs[i & 1][1])
        idx.sync(h); torch.cuda.synchronize()
        n = 100
        t0 = time.perf_counter()
        for i in range(n): h = idx.search_pipelined(q, k, outs[i & 1][0], outs[i & 1][1])
        idx.sync(h); torch.cuda.synchronize()
        dt0 = (time.perf_counter() - t0) / n
        idx.profile(True)
        t0 = time.perf_counter()
        for i in range(n): h = idx.search_pipelined(q, k, outs[i & 1][0], outs[i & 1][1])
        idx.sync(h); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        pr = idx.profile_collect()
        kms = pr["total_ms"] / max(pr["launches"], 1)
        print(f"rows {rows:9d} sample_div {qgl:3d} reserve {rsv:3d}: step {dt0*1e6:7.1f} us (profiled {dt*1e6:7.1f})  main-scan kernel {kms*1e3:7.1f} us  ({pr['bytes_per_launch']/kms/1e6:6.0f} GB/s)", flush=True)
        idx.close()
    del blocks
