"""Timeline of the scan with the finishing stage (development build: CMR_EXTRA_HIPCC_FLAGS=-DCMR_FIN_DEBUG CMR_BUILD_LIB=comorag_amd/lib/libfindbg.so
python -m comorag_amd.build; COMORAG_HIP_LIB=comorag_amd/lib/libfindbg.so python tools/fin_timeline.py [rows ...])"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from comorag_amd.index import DenseIndex
from tools import env_options
dim, k = 768, 20
dev = torch.device("cuda", 0); g = torch.Generator(device=dev); g.manual_seed(1)
for rows in [int(x) for x in (sys.argv[1:] or ["1000000", "2000000"])]:
    idx = DenseIndex(dim, "bf16", capacity_hint=rows, options=env_options())
    for b in range(0, rows, 250_000):
        x = torch.randn((min(250_000, rows - b), dim), generator=g, device=dev)
        idx.append_dev((x / x.norm(dim=1, keepdim=True)).contiguous())
    torch.cuda.synchronize()
    for B in (1, 8):
        q = np.random.default_rng(B).standard_normal((B, dim)).astype(np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
        print(f"== rows {rows} B {B}", flush=True)
        for _ in range(4):
            t0 = time.perf_counter(); idx.search(q, k); dt = time.perf_counter() - t0
            torch.cuda.synchronize(); print(f"   call {dt * 1e6:.1f} us", flush=True)
    idx.close()
