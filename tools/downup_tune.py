"""A/B timing of the Downsample / Upsample conv variants per level (tuning tool; run on the GPU box).
usage: python tools/downup_tune.py [B] [T]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddim_audio_amd import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
lib = _lib.load()
dt, tdt = _lib.DDIMX_BF16, torch.bfloat16
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


for lvl in range(1, 6):
    cp, c = CH[lvl - 1], CH[lvl]
    H, W = T >> (lvl - 1), 256 >> (lvl - 1)          # the larger (level l-1) grid
    big = torch.randn(B, H, W, cp, device="cuda").to(tdt)
    small = torch.randn(B, H // 2, W // 2, c, device="cuda").to(tdt)
    wd = (torch.randn(16 * c * cp, device="cuda") * 0.02).to(tdt)
    wu = (torch.randn(2 * 6 * 2 * cp * c, device="cuda") * 0.02).to(tdt)
    bd = torch.zeros(c, device="cuda")
    bu = torch.zeros(2 * cp, device="cuda")
    yd, yu = torch.empty_like(small), torch.empty_like(big)
    st = _lib.stream()

    def down():
        _lib.check(lib.ddimx_downsample_fwd(dt, cp, c, _lib.ptr(big), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(yd), B, H, W, st))

    def up():
        _lib.check(lib.ddimx_upsample_add_fwd(dt, c, cp, _lib.ptr(small), _lib.ptr(wu), _lib.ptr(bu), _lib.ptr(big), _lib.ptr(yu), B, H // 2,
                                              W // 2, st))

    for name, fn, y in (("down", down, yd), ("up", up, yu)):
        out, ref = [], None
        for var in (None, 0, 1, 2, 3, 4, 5):
            if var is None:
                os.environ.pop("DDIMX_CONV_VAR", None)
            else:
                os.environ["DDIMX_CONV_VAR"] = str(var)
            us = timed(fn)
            if ref is None:
                ref = y.float().clone()
            out.append(f"var={var}: {us:6.1f} us same={bool(torch.equal(ref, y.float()))}")
        os.environ.pop("DDIMX_CONV_VAR", None)
        print(f"L{lvl} {name} {cp}<->{c} [{B},{H},{W}]: " + " | ".join(out), flush=True)
