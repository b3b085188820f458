"""A/B timing of conv_mfma tile variants per U-Net level (tuning tool; run on the GPU box).
usage: python tools/conv_tune.py [B] [T]   -- prints us per launch for DDIMX_CONV_VAR in {default,2,3,4,5}"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ddim_audio_amd import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
lib = _lib.load()
dt, tdt, es = _lib.DDIMX_BF16, torch.bfloat16, 2
dev = torch.device("cuda", 0)
CH = [32, 64, 96, 128, 192, 256]


def timed(fn, reps=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for lvl, C in enumerate(CH):
    H, W = T >> lvl, 256 >> lvl
    x = torch.randn(B, H, W, C, device=dev).to(tdt)
    y = torch.empty_like(x)
    ref = None
    w = (torch.randn(9 * C * C, device=dev) * (1.0 / (9 * C) ** 0.5)).to(tdt)
    temb = torch.randn(B, C, device=dev) * 0.1
    scale = torch.rand(B, C, device=dev) + 0.5
    shift = torch.randn(B, C, device=dev) * 0.1
    stats = torch.zeros(int(lib.ddimx_conv3x3_stats_floats(dt, C, B, H, W)), device=dev)
    st = _lib.stream()

    def conv():
        _lib.check(lib.ddimx_conv3x3_fwd(dt, C, _lib.ptr(x), _lib.ptr(w), None, _lib.ptr(temb), C, _lib.ptr(scale), _lib.ptr(shift),
                                         2, 1, _lib.ptr(y), _lib.ptr(stats), B, H, W, st))

    elems = B * H * W * C
    out = []
    for var in (None, 0, 1, 2, 3, 4, 5):
        if var is None:
            os.environ.pop("DDIMX_CONV_VAR", None)
        else:
            os.environ["DDIMX_CONV_VAR"] = str(var)
        us = timed(conv)
        torch.cuda.synchronize()
        if ref is None:
            ref = y.float().clone()
            same = True
        else:
            same = bool(torch.equal(ref, y.float()))
        out.append(f"var={var}: {us:7.1f} us {2 * elems * es / us / 1e3:6.0f} GB/s {2.0 * elems * 9 * C / us / 1e6:5.0f} TF same={same}")
    os.environ.pop("DDIMX_CONV_VAR", None)
    print(f"L{lvl} C={C} [{B},{H},{W}]:\n   " + "\n   ".join(out), flush=True)
