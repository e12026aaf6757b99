import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from comorag_amd.index import DenseIndex
rows, batch, dim, k = int(sys.argv[1]), 64, 768, 20
dev = torch.device("cuda", 0); g = torch.Generator(device=dev); g.manual_seed(1)
q = torch.randn((batch, dim), generator=g, device=dev); q = (q / q.norm(dim=1, keepdim=True)).contiguous()
idx = DenseIndex(dim, "bf16", capacity_hint=rows)
for b in range(0, rows, 250_000):
    x = torch.randn((min(250_000, rows - b), dim), generator=g, device=dev); idx.append_dev((x / x.norm(dim=1, keepdim=True)).contiguous())
torch.cuda.synchronize()
outs = [(torch.empty((batch, k), dtype=torch.int64, device=dev), torch.empty((batch, k), dtype=torch.float32, device=dev)) for _ in range(2)]
for i in range(40):
    h = idx.search_pipelined(q, k, outs[i & 1][0], outs[i & 1][1])
idx.sync(h); torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(40):
    h = idx.search_pipelined(q, k, outs[i & 1][0], outs[i & 1][1])
t1 = time.perf_counter()
idx.sync(h); torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {((t1-t0)/40)*1e6:.1f} us/step; total {((t2-t0)/40)*1e6:.1f} us/step")
