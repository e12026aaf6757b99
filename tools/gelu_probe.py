"""Is `torch._addmm_activation(bias, x, w.T, use_gelu=True)` ONE hipBLASLt launch on this build, what does it compute, and what does it cost
against GEMM + bias followed by PyTorch's GELU kernel?  (fused_bert.FusedBertLayers: the FFN-up projection of every layer.)
    python tools/gelu_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from comorag_amd.embedding_model.fused_bert import gelu_epilogue_available

dev = torch.device("cuda", 0)
for dt, (m, h, f) in ((torch.bfloat16, (16384, 768, 3072)), (torch.float16, (16384, 1024, 4096)), (torch.bfloat16, (2048, 768, 3072))):
    g = torch.Generator(device=dev); g.manual_seed(3)
    x = torch.randn((m, h), generator=g, device=dev).to(dt)
    w = (torch.randn((f, h), generator=g, device=dev) * 0.03).to(dt)
    b = (torch.randn((f,), generator=g, device=dev) * 0.1).to(dt)
    w2 = (torch.randn((h, f), generator=g, device=dev) * 0.02).to(dt)
    lin = F.linear(x.float(), w.float(), b.float())
    y_exact = F.gelu(F.linear(x, w, b))
    y_fused = torch._addmm_activation(b, x, w.t(), use_gelu=True)
    e_t = (y_fused.float() - F.gelu(lin, approximate="tanh")).abs()
    e_e = (y_fused.float() - F.gelu(lin)).abs()
    e_x = (y_exact.float() - F.gelu(lin)).abs()
    print(f"{dt} {m}x{h}x{f}: available {gelu_epilogue_available(dev, dt)}; fused vs tanh-GELU(fp32) max {e_t.max():.3e} mean {e_t.mean():.3e}; "
          f"fused vs erf-GELU(fp32) max {e_e.max():.3e} mean {e_e.mean():.3e}; unfused 16-bit path vs erf-GELU(fp32) max {e_x.max():.3e} mean {e_x.mean():.3e}", flush=True)

    def timeit(fn, n=50):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    t_lin = timeit(lambda: F.linear(x, w, b))
    t_un = timeit(lambda: F.gelu(F.linear(x, w, b)))
    t_fu = timeit(lambda: torch._addmm_activation(b, x, w.t(), use_gelu=True))
    t_ffn_un = timeit(lambda: F.linear(F.gelu(F.linear(x, w, b)), w2))
    t_ffn_fu = timeit(lambda: F.linear(torch._addmm_activation(b, x, w.t(), use_gelu=True), w2))
    fl = 2.0 * m * h * f
    print(f"   us per call: linear {t_lin:.1f} ({fl / t_lin / 1e6:.0f} TF) | linear + gelu kernel {t_un:.1f} | _addmm_activation {t_fu:.1f} ({fl / t_fu / 1e6:.0f} TF) | "
          f"whole FFN unfused {t_ffn_un:.1f} fused {t_ffn_fu:.1f}", flush=True)
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(3): torch._addmm_activation(b, x, w.t(), use_gelu=True)
        torch.cuda.synchronize()
    for e in prof.key_averages():
        print("   fused call kernel:", e.key[:110], "x", e.count, f"{e.device_time_total / max(e.count, 1):.1f} us")
