"""Run one level's fused 3x3 conv repeatedly (for rocprofv3 --pmc runs).  usage: conv_one.py LEVEL [B] [REPS]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ddim_audio_amd import _lib  # noqa: E402

lvl = int(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
xf = int(os.environ.get("DDIMX_ONE_XF", "2"))  # 2: the block's first conv (affine + SiLU input, + temb); 1: its second (affine input, + bias)
lib = _lib.load()
dt, tdt = _lib.DDIMX_BF16, torch.bfloat16
dev = torch.device("cuda", 0)
C = [32, 64, 96, 128, 192, 256][lvl]
H, W = 1024 >> lvl, 256 >> lvl
x = torch.randn(B, H, W, C, device=dev).to(tdt)
y = torch.empty_like(x)
w = (torch.randn(9 * C * C, device=dev) * (1.0 / (9 * C) ** 0.5)).to(tdt)
temb = torch.randn(B, C, device=dev) * 0.1
scale = torch.rand(B, C, device=dev) + 0.5
shift = torch.randn(B, C, device=dev) * 0.1
stats = torch.zeros(int(lib.ddimx_conv3x3_stats_floats(dt, C, B, H, W)), device=dev)
bias = torch.randn(C, device=dev) * 0.1
pipe = C <= 64 and os.environ.get("DDIMX_ONE_PIPE", "1") != "0"    # what the inference walk launches at C = 32 / 64 (csrc/conv_pipe.h)
wreg = C >= 64 and os.environ.get("DDIMX_ONE_WREG", "1") != "0"  # C >= 96 (and the previous kernel of C = 64): csrc/conv_wreg.h
if wreg or pipe:
    wt = torch.randn(C, C, 3, 3, device=dev) * (1.0 / (9 * C) ** 0.5)
    wf = torch.empty(9 * C * C, dtype=tdt, device=dev)
    _lib.check(lib.ddimx_pack_conv_frag(_lib.ptr(wt), _lib.ptr(wf), C, C, _lib.stream()))
if pipe:
    stats = torch.zeros(int(lib.ddimx_conv3x3_pipe_stats_floats(C, B, H, W)), device=dev)
for _ in range(reps):
    if pipe:
        _lib.check(lib.ddimx_conv3x3_pipe_fwd(C, _lib.ptr(x), _lib.ptr(wf), None if xf == 2 else _lib.ptr(bias), _lib.ptr(temb) if xf == 2 else None, C,
                                              _lib.ptr(scale), _lib.ptr(shift), xf, _lib.ptr(y), _lib.ptr(stats), B, H, W, _lib.stream()))
    elif wreg:
        _lib.check(lib.ddimx_conv3x3_wreg_fwd(C, _lib.ptr(x), _lib.ptr(w), _lib.ptr(wf), None if xf == 2 else _lib.ptr(bias),
                                              _lib.ptr(temb) if xf == 2 else None, C, _lib.ptr(scale), _lib.ptr(shift), xf, 1,
                                              _lib.ptr(y), _lib.ptr(stats), B, H, W, _lib.stream()))
    elif xf == 2:
        _lib.check(lib.ddimx_conv3x3_fwd(dt, C, _lib.ptr(x), _lib.ptr(w), None, _lib.ptr(temb), C, _lib.ptr(scale), _lib.ptr(shift), 2, 1,
                                         _lib.ptr(y), _lib.ptr(stats), B, H, W, _lib.stream()))
    else:
        _lib.check(lib.ddimx_conv3x3_fwd(dt, C, _lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), None, 0, _lib.ptr(scale), _lib.ptr(shift), xf, 1,
                                         _lib.ptr(y), _lib.ptr(stats), B, H, W, _lib.stream()))
torch.cuda.synchronize()
print("done", float(y.float().abs().mean()))
if os.environ.get("DDIMX_ONE_TIME"):  # tools/power_probe.sh: a sustained loop, us per launch
    import time
    n = int(os.environ["DDIMX_ONE_TIME"])
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n):
        if pipe:
            _lib.check(lib.ddimx_conv3x3_pipe_fwd(C, _lib.ptr(x), _lib.ptr(wf), None if xf == 2 else _lib.ptr(bias), _lib.ptr(temb) if xf == 2 else None, C,
                                                  _lib.ptr(scale), _lib.ptr(shift), xf, _lib.ptr(y), _lib.ptr(stats), B, H, W, _lib.stream()))
        elif wreg:
            _lib.check(lib.ddimx_conv3x3_wreg_fwd(C, _lib.ptr(x), _lib.ptr(w), _lib.ptr(wf), None if xf == 2 else _lib.ptr(bias),
                                                  _lib.ptr(temb) if xf == 2 else None, C, _lib.ptr(scale), _lib.ptr(shift), xf, 1,
                                                  _lib.ptr(y), _lib.ptr(stats), B, H, W, _lib.stream()))
        else:
            _lib.check(lib.ddimx_conv3x3_fwd(dt, C, _lib.ptr(x), _lib.ptr(w), None if xf == 2 else _lib.ptr(bias), _lib.ptr(temb) if xf == 2 else None, C,
                                             _lib.ptr(scale), _lib.ptr(shift), xf, 1, _lib.ptr(y), _lib.ptr(stats), B, H, W, _lib.stream()))
    torch.cuda.synchronize()
    print("sustained us/launch %.2f over %d launches" % ((time.time() - t0) * 1e6 / n, n))
