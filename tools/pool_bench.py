"""Roofline of the encoder-tail kernel pair (masked mean-pool + L2-normalise)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from comorag_amd.embedding_model.bge import pool_l2norm
out = []
for (b, l, d, dt) in [(32, 512, 768, torch.float32), (32, 512, 768, torch.bfloat16), (32, 512, 1024, torch.float16), (256, 512, 768, torch.float32)]:
    h = torch.randn((b, l, d), device="cuda").to(dt)
    lens = torch.randint(l // 2, l + 1, (b,), device="cuda"); lens[0] = l
    m = (torch.arange(l, device="cuda")[None, :] < lens[:, None]).to(torch.int64)
    for _ in range(5): pool_l2norm(h, m)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): pool_l2norm(h, m)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    # algorithmic bytes: masked tokens are skipped, so count the tokens actually read
    tokens = int(m.sum().item())
    nbytes = tokens * d * h.element_size() + b * l * 8 + b * d * 4
    out.append({"b": b, "l": l, "d": d, "dtype": str(dt).replace("torch.", ""), "us": ms * 1e3, "GBps": nbytes / (ms * 1e-3) / 1e9,
                "frac_of_8TBps": nbytes / (ms * 1e-3) / 8e12, "algorithmic_bytes": nbytes})
    print(out[-1], flush=True)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "pool_bench.json"), "w"), indent=1)
