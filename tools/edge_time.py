"""Time conv_in / conv_out launches (HIP events): edge_time.py B [T]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddim_audio_amd import _lib
B = int(sys.argv[1]); T = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
lib = _lib.load()
dt, tdt = _lib.DDIMX_BF16, torch.bfloat16
H, W, C0 = T, 256, 32
x = torch.randn(B, 2, H, W, device="cuda")
w = torch.randn(C0, 2, 3, 3, device="cuda") * 0.2; bias = torch.randn(C0, device="cuda") * 0.1
y = torch.empty(B, H, W, C0, device="cuda", dtype=tdt)
stats = torch.zeros(int(lib.ddimx_conv_in_stats_floats(B, C0, H, W)), device="cuda")
a = torch.randn(B, H, W, C0, device="cuda").to(tdt); s2 = torch.randn(B, H, W, C0, device="cuda").to(tdt)
wo = torch.randn(9 * 2 * C0, device="cuda") * 0.1; bo = torch.zeros(2, device="cuda")
eps = torch.empty(B, 2, H, W, device="cuda")
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
ci = lambda: _lib.check(lib.ddimx_conv_in_fwd(dt, _lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(y), _lib.ptr(stats), B, 2, C0, H, W, _lib.stream()))
co = lambda: _lib.check(lib.ddimx_conv_out_fwd(dt, _lib.ptr(a), _lib.ptr(s2), _lib.ptr(wo), _lib.ptr(bo), _lib.ptr(eps), B, C0, 2, H, W, _lib.stream()))
print("B", B, "T", T, "conv_in %.1f us (%.2f TB/s)  conv_out %.1f us (%.2f TB/s)" % (
    t(ci), (x.numel() * 4 + y.numel() * 2) / t(ci) / 1e6, t(co), (2 * a.numel() * 2 + eps.numel() * 4) / t(co) / 1e6), os.environ.get("DDIMX_CONV_IN_VALU", ""))
