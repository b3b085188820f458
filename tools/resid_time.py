"""Time one resid launch (y = x + h*scale + shift, + statistics of y): resid_time.py LEVEL B"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddim_audio_amd import _lib
lvl, B = int(sys.argv[1]), int(sys.argv[2])
lib = _lib.load()
dt, tdt = _lib.DDIMX_BF16, torch.bfloat16
C = [32, 64, 96, 128, 192, 256][lvl]; H, W = 1024 >> lvl, 256 >> lvl
x = torch.randn(B, H, W, C, device="cuda").to(tdt); h = torch.randn(B, H, W, C, device="cuda").to(tdt); y = torch.empty_like(x)
scale = torch.rand(B, C, device="cuda") + 0.5; shift = torch.randn(B, C, device="cuda") * 0.1
stats = torch.zeros(int(lib.ddimx_conv3x3_stats_floats(dt, C, B, H, W)) * 4, device="cuda")
def run(n):
    for _ in range(n):
        _lib.check(lib.ddimx_resid_gn_fwd(dt, C, _lib.ptr(x), _lib.ptr(h), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(y), _lib.ptr(stats), B, H, W, _lib.stream()))
run(5); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(50); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 50
print(os.path.basename(os.environ.get("DDIMX_LIB", "default")), "resid level", lvl, "B", B, "us/launch %.1f" % us, "TB/s %.2f" % (3 * x.numel() * 2 / us / 1e6))
