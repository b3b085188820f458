"""Per-phase cycle breakdown of the fused 3x3 conv (diagnostic build).  On the GPU box:
   python -m ddim_audio_amd.build --stamp && DDIMX_LIB=ddim_audio_amd/libddimx_stamp.so python tools/conv_stamps.py LEVEL [B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ddim_audio_amd import _lib  # noqa: E402

lvl = int(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
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
stamps = torch.zeros(12 * (1 << 16), dtype=torch.int64, device=dev)
for _ in range(3):
    stamps.zero_()
    _lib.check(lib.ddimx_debug_conv3x3_stamps(dt, C, _lib.ptr(x), _lib.ptr(w), _lib.ptr(temb), _lib.ptr(scale), _lib.ptr(shift),
                                              _lib.ptr(y), _lib.ptr(stats), _lib.ptr(stamps), B, H, W, _lib.stream()))
torch.cuda.synchronize()
s = stamps.cpu().reshape(-1, 12)
s = s[s.sum(1) > 0].double()
names = ["issue next halo", "MFMA loop", "barrier A", "epilogue 1 (acc->LDS)", "commit (separate out)", "barrier B", "epilogue 2 (stores+stats)",
         "barrier C", "commit (overlay)", "barrier D", "-", "-"]
tot = s.sum(1).mean()
print(f"L{lvl} C={C} B={B}: {s.shape[0]} waves, mean stamped cycles per wave {tot:.0f}")
for k, n in enumerate(names):
    if s[:, k].sum() > 0:
        print(f"  {n:28s} {s[:, k].mean():10.0f} cycles  {100 * s[:, k].mean() / tot:5.1f}%")
