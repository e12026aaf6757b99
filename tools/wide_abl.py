"""Wide-kernel ablations (CMR_WIDE_ABL): sync kernel time of one B=256 pass.  python tools/wide_abl.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from comorag_amd.index import DenseIndex
from tools import env_options
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim, k, B = 768, 20, 256
dev = torch.device("cuda", 0); g = torch.Generator(device=dev); g.manual_seed(7)
blocks = []
for b in range(0, rows, 250_000):
    x = torch.randn((min(250_000, rows - b), dim), generator=g, device=dev); blocks.append((x / x.norm(dim=1, keepdim=True)).contiguous())
q = torch.randn((B, dim), generator=g, device=dev); q = (q / q.norm(dim=1, keepdim=True)).contiguous()
for abl in [int(a) for a in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,1,2,3,4,5".split(","))]:
    os.environ["CMR_WIDE_ABL"] = str(abl)
    idx = DenseIndex(dim, "bf16", capacity_hint=rows, options=env_options())
    for x in blocks: idx.append_dev(x)
    torch.cuda.synchronize()
    for _ in range(3): idx.search_dev(q, k)
    torch.cuda.synchronize()
    idx.profile(True)
    for _ in range(10): idx.search_dev(q, k)
    torch.cuda.synchronize()
    pr = idx.profile_collect(); kms = pr["total_ms"] / max(pr["launches"], 1)
    print(f"ABL {abl}: main-scan kernel {kms:.3f} ms/launch", flush=True)
    idx.close()
