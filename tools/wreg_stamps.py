"""Per-phase cycle breakdown + timeline of conv3_wreg_kernel (diagnostic build):
   DDIMX_LIB=ddim_audio_amd/libddimx_stamp.so python tools/wreg_stamps.py LEVEL [B] [XF]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ddim_audio_amd import _lib  # noqa: E402

lvl = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 8; xf = int(sys.argv[3]) if len(sys.argv) > 3 else 2
lib = _lib.load()
dt, tdt = _lib.DDIMX_BF16, torch.bfloat16
dev = torch.device("cuda", 0)
C = [32, 64, 96, 128, 192, 256][lvl]
H, W = 1024 >> lvl, 256 >> lvl
x = torch.randn(B, H, W, C, device=dev).to(tdt)
y = torch.empty_like(x)
wt = torch.randn(C, C, 3, 3, device=dev) * (1.0 / (9 * C) ** 0.5)
wf = torch.empty(9 * C * C, dtype=tdt, device=dev)
_lib.check(lib.ddimx_pack_conv_frag(_lib.ptr(wt), _lib.ptr(wf), C, C, _lib.stream()))
temb = torch.randn(B, C, device=dev) * 0.1
scale = torch.rand(B, C, device=dev) + 0.5
shift = torch.randn(B, C, device=dev) * 0.1
stats = torch.zeros(int(lib.ddimx_conv3x3_stats_floats(dt, C, B, H, W)), device=dev)
stamps = torch.zeros(16 * (1 << 17), dtype=torch.int64, device=dev)
lib.ddimx_debug_set_stamps(_lib.ptr(stamps))
for _ in range(3):
    stamps.zero_()
    _lib.check(lib.ddimx_conv3x3_wreg_fwd(C, _lib.ptr(x), _lib.ptr(wf), _lib.ptr(wf), None, _lib.ptr(temb), C, _lib.ptr(scale), _lib.ptr(shift), xf, 1,
                                          _lib.ptr(y), _lib.ptr(stats), B, H, W, _lib.stream()))
torch.cuda.synchronize()
raw = stamps.cpu().reshape(-1, 16)
raw = raw[raw[:, 12] > 0]
s = raw[:, :12].double()
names = ["(loop top)", "MFMA loop", "barrier A", "epilogue 1 (acc->LDS)", "-", "barrier B", "epilogue 2 (stores+stats)", "barrier C",
         "halo load + transform", "barrier D", "(after loop)", "statistics tail"]
tot = s.sum(1).mean()
print(f"wreg L{lvl} C={C} B={B} xf={xf}: {s.shape[0]} waves, mean stamped cycles per wave {tot:.0f}")
for k, n in enumerate(names):
    if s[:, k].sum() > 0:
        print(f"  {n:28s} {s[:, k].mean():10.0f} cycles  {100 * s[:, k].mean() / tot:5.1f}%")
t0, t1, t2 = raw[:, 12].double(), raw[:, 13].double(), raw[:, 14].double()
base = t0.min()
span = float(((t2 - base) / 100.0).max())
print(f"kernel span {span:.1f} us; prologue mean {(t1 - t0).mean() / 100:.2f} us (min {(t1 - t0).min() / 100:.2f} max {(t1 - t0).max() / 100:.2f}); "
      f"loop+tail mean {(t2 - t1).mean() / 100:.2f} us; wave life mean {(t2 - t0).mean() / 100:.2f} us")
ent = (t0 - base) / 100.0
print("  entries per tenth:", [int(v) for v in torch.histc(ent.float(), bins=10, min=0, max=span).tolist()])
print("  exits per tenth:  ", [int(v) for v in torch.histc(((t2 - base) / 100.0).float(), bins=10, min=0, max=span).tolist()])
