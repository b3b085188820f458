"""Time the Residual_Block convs through the wave-specialised kernel (conv_ws.h) against the walk's previous kernels:
   python tools/ws_time.py LEVEL [B]   (LEVEL 0 / 1; isolated back-to-back launches, HIP events)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddim_audio_amd import _lib
lvl = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
lib = _lib.load()
dt, tdt = _lib.DDIMX_BF16, torch.bfloat16
dev = torch.device("cuda", 0)
C = [32, 64, 96][lvl]
H, W = 1024 >> lvl, 256 >> lvl
x = torch.randn(B, H, W, C, device=dev).to(tdt)
y = torch.empty_like(x)
wt = torch.randn(C, C, 3, 3, device=dev) * (1.0 / (9 * C) ** 0.5)
wf = torch.empty(9 * C * C, dtype=tdt, device=dev)
_lib.check(lib.ddimx_pack_conv_frag(_lib.ptr(wt), _lib.ptr(wf), C, C, _lib.stream()))
temb = torch.randn(B, C, device=dev) * 0.1
scale = torch.rand(B, C, device=dev) + 0.5
shift = torch.randn(B, C, device=dev) * 0.1
stats = torch.zeros(int(lib.ddimx_conv3x3_stats_floats(dt, C, B, H, W)) + 4096, device=dev)
def run(xf, n):
    for _ in range(n):
        _lib.check(lib.ddimx_conv3x3_ws_fwd(C, _lib.ptr(x), _lib.ptr(wf), None, _lib.ptr(temb), C, _lib.ptr(scale), _lib.ptr(shift), xf, 1,
                                            _lib.ptr(y), _lib.ptr(stats), B, H, W, _lib.stream()))
for xf in (2, 1):
    run(xf, 5); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(xf, 50); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    gb = 2 * x.numel() * 2 / 1e9
    print(f"ws level {lvl} C {C} B {B} xf {xf}: {us:.1f} us/launch  {gb / us * 1e6 / 1e3:.2f} TB/s  {gb / us * 1e6 / 8e3:.2f} of 8 TB/s")
