"""Time conv3_wreg_kernel launches (HIP events): wreg_time.py LEVEL B XF [H W]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddim_audio_amd import _lib
lvl = int(sys.argv[1]); B = int(sys.argv[2]); xf = int(sys.argv[3])
lib = _lib.load()
dt, tdt = _lib.DDIMX_BF16, torch.bfloat16
C = [32, 64, 96, 128, 192, 256][lvl]; H, W = 1024 >> lvl, 256 >> lvl
if len(sys.argv) > 5: H, W = int(sys.argv[4]), int(sys.argv[5])
x = torch.randn(B, H, W, C, device="cuda").to(tdt); y = torch.empty_like(x)
wt = torch.randn(C, C, 3, 3, device="cuda") * (1.0 / (9 * C) ** 0.5)
wf = torch.empty(9 * C * C, dtype=tdt, device="cuda")
_lib.check(lib.ddimx_pack_conv_frag(_lib.ptr(wt), _lib.ptr(wf), C, C, _lib.stream()))
temb = torch.randn(B, C, device="cuda") * 0.1
scale = torch.rand(B, C, device="cuda") + 0.5; shift = torch.randn(B, C, device="cuda") * 0.1
stats = torch.zeros(int(lib.ddimx_conv3x3_stats_floats(dt, C, B, H, W)), device="cuda")
def run(n):
    for _ in range(n):
        _lib.check(lib.ddimx_conv3x3_wreg_fwd(C, _lib.ptr(x), _lib.ptr(wf), _lib.ptr(wf), None, _lib.ptr(temb), C, _lib.ptr(scale), _lib.ptr(shift), xf, 1,
                                              _lib.ptr(y), _lib.ptr(stats), B, H, W, _lib.stream()))
run(5); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(50); e1.record(); torch.cuda.synchronize()
print("wreg level", lvl, "C", C, "B", B, "xf", xf, "WPS", os.environ.get("DDIMX_CONV_WPS", "-"), "us/launch %.1f" % (e0.elapsed_time(e1) * 1e3 / 50))
