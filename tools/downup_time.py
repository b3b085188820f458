"""Time one Downsample or Upsample(+skip) launch (HIP events): downup_time.py {down|up} LEVEL B
(LEVEL = the smaller-resolution level, 1..5; knobs: DDIMX_CONV_VAR / DDIMX_CONV_WPS, read once per process)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddim_audio_amd import _lib
kind, lvl, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
lib = _lib.load()
dt, tdt = _lib.DDIMX_BF16, torch.bfloat16
CH = [32, 64, 96, 128, 192, 256]
cp, c = CH[lvl - 1], CH[lvl]
H, W = 1024 >> (lvl - 1), 256 >> (lvl - 1)
big = torch.randn(B, H, W, cp, device="cuda").to(tdt); small = torch.randn(B, H // 2, W // 2, c, device="cuda").to(tdt)
wd = (torch.randn(16 * c * cp, device="cuda") * 0.02).to(tdt); wu = (torch.randn(2 * 6 * 2 * cp * c, device="cuda") * 0.02).to(tdt)
bd = torch.zeros(c, device="cuda"); bu = torch.zeros(2 * cp, device="cuda")
yd, yu = torch.empty_like(small), torch.empty_like(big)
def run(n):
    for _ in range(n):
        if kind == "down":
            _lib.check(lib.ddimx_downsample_fwd(dt, cp, c, _lib.ptr(big), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(yd), B, H, W, _lib.stream()))
        else:
            _lib.check(lib.ddimx_upsample_add_fwd(dt, c, cp, _lib.ptr(small), _lib.ptr(wu), _lib.ptr(bu), _lib.ptr(big), _lib.ptr(yu), B, H // 2, W // 2, _lib.stream()))
run(5); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(50); e1.record(); torch.cuda.synchronize()
print(kind, "level", lvl, "B", B, "us/launch %.1f" % (e0.elapsed_time(e1) * 1e3 / 50))
