"""Attention kernel alone: us per layer and PFLOP/s at 32 x 512 tokens (BERT-base: 12 heads; BERT-large: 16), full and ragged lengths."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from comorag_amd import _lib as L
b, l = 32, 512
for heads, dt, cdt in ((12, torch.bfloat16, L.CMR_BF16), (16, torch.float16, L.CMR_F16)):
    hidden = heads * 64
    qkv = torch.randn((b * l, 3 * hidden), device="cuda").to(dt)
    out = torch.empty((b * l, hidden), dtype=dt, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for name, lens_h in (("full", np.full(b, l, np.int32)), ("ragged", (np.arange(b) * 16 + 17).astype(np.int32).clip(max=l))):
        lens = torch.from_numpy(lens_h).cuda()
        run = lambda: L.check(L.lib().cmr_encoder_attention(0, C.c_void_p(qkv.data_ptr()), cdt, C.c_void_p(lens.data_ptr()), b, l, heads, 64, C.c_void_p(out.data_ptr()), C.c_void_p(s)))
        for _ in range(5): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): run()
        torch.cuda.synchronize(); dt_ = (time.perf_counter() - t0) / 50
        fl = 4.0 * float((lens_h.astype(np.float64) ** 2).sum()) * hidden
        print(f"heads {heads} {str(dt)[6:]} {name}: {dt_ * 1e6:.1f} us, {fl / dt_ / 1e15:.3f} PFLOP/s = {fl / dt_ / 2.5e15:.3f} of 2.5 PF", flush=True)
