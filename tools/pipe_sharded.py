import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from comorag_amd.sharded import ShardedIndex
rows, batch, dim, k = int(sys.argv[1]), 64, 768, 20
prof = len(sys.argv) > 2 and sys.argv[2] == "prof"
dev = torch.device("cuda", 0); g = torch.Generator(device=dev); g.manual_seed(1)
q = torch.randn((batch, dim), generator=g, device=dev); q = (q / q.norm(dim=1, keepdim=True)).contiguous()
sh = ShardedIndex(dim, "bf16", capacity_hint=rows)
for b in range(0, rows, 250_000):
    x = torch.randn((min(250_000, rows - b), dim), generator=g, device=dev); sh.local.append_dev((x / x.norm(dim=1, keepdim=True)).contiguous())
torch.cuda.synchronize()
for i in range(10): sh.search_pipelined(q, k, i & 1)
torch.cuda.synchronize()
if prof: sh.local.profile(True)
t0 = time.perf_counter()
for i in range(40): b = sh.search_pipelined(q, k, i & 1)
t1 = time.perf_counter()
b['done'].synchronize(); torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"sharded pipelined prof={prof}: host enqueue {((t1-t0)/40)*1e6:.1f} us/step; total {((t2-t0)/40)*1e6:.1f} us/step")
