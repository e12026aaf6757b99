"""The encoder attention kernel alone (for rocprofv3 PMC passes): python -m tools.attn_only [b] [l] [heads] [reps]"""
import sys
import numpy as np
import torch
from comorag_amd import _lib as L
import ctypes as C

b, l, heads, reps = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 32), (2, 512), (3, 12), (4, 30)))
hidden = heads * 64
qkv = torch.randn((b * l, 3 * hidden), device="cuda").to(torch.bfloat16)
lens = torch.full((b,), l, dtype=torch.int32, device="cuda")
out = torch.empty((b * l, hidden), dtype=torch.bfloat16, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(reps):
    L.check(L.lib().cmr_encoder_attention(0, C.c_void_p(qkv.data_ptr()), L.CMR_BF16, C.c_void_p(lens.data_ptr()), b, l, heads, 64, C.c_void_p(out.data_ptr()), C.c_void_p(s)))
torch.cuda.synchronize()
print("ok", float(out.float().abs().mean()))
