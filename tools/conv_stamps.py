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
stamps = torch.zeros(16 * (1 << 16), dtype=torch.int64, device=dev)
for _ in range(3):
    stamps.zero_()
    _lib.check(lib.ddimx_debug_conv3x3_stamps(dt, C, _lib.ptr(x), _lib.ptr(w), _lib.ptr(temb), _lib.ptr(scale), _lib.ptr(shift),
                                              _lib.ptr(y), _lib.ptr(stats), _lib.ptr(stamps), B, H, W, _lib.stream()))
torch.cuda.synchronize()
raw = stamps.cpu().reshape(-1, 16)
raw = raw[raw[:, 12] > 0]
s = raw[:, :12].double()
names = ["issue next halo", "MFMA loop", "barrier A", "epilogue 1 (acc->LDS)", "commit (separate out)", "barrier B", "epilogue 2 (stores+stats)",
         "barrier C", "commit (overlay)", "barrier D", "(after loop: drain)", "statistics tail"]
if os.environ.get("DDIMX_STAMP_XF") == "1" and os.environ.get("DDIMX_CONV_FOLD", "1") != "0" and C == 32:
    names = ["issue next halo DMA", "MFMA loop", "barrier A", "epilogue 1 (acc->LDS)", "-", "barrier B", "epilogue 2 (stores+stats)", "-",
             "top: wait halo + barrier", "(loop overhead)", "(after loop)", "statistics tail"]
tot = s.sum(1).mean()
print(f"L{lvl} C={C} B={B}: {s.shape[0]} waves, mean stamped cycles per wave {tot:.0f}")
for k, n in enumerate(names):
    if s[:, k].sum() > 0:
        print(f"  {n:28s} {s[:, k].mean():10.0f} cycles  {100 * s[:, k].mean() / tot:5.1f}%")
# ---- timeline (s_memrealtime, 100 MHz): when do workgroups start, how long is the prologue, how full are the CU slots
t0, t1, t2 = raw[:, 12].double(), raw[:, 13].double(), raw[:, 14].double()
base = t0.min()
us = lambda v: (v - base) / 100.0  # noqa: E731
span = float(us(t2).max())
print(f"kernel span (first entry -> last exit) {span:.1f} us")
ent, ext = us(t0), us(t2)
print(f"  wave entry times: median {ent.median():.1f}  90% {ent.quantile(0.9):.1f}  max {ent.max():.1f} us")
print(f"  prologue (entry -> tile loop): mean {(t1 - t0).mean() / 100:.2f} us   min {(t1 - t0).min() / 100:.2f}  max {(t1 - t0).max() / 100:.2f}")
print(f"  tile loop + tail (loop start -> exit): mean {(t2 - t1).mean() / 100:.2f} us; wave life mean {(t2 - t0).mean() / 100:.2f} us")
clk = s.sum(1) / ((t2 - t1) / 100.0) / 1e3
print(f"  in-kernel clock over the stamped part: median {clk.median():.2f} GHz")
hw = raw[:, 15]
cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5) | (((hw >> 32) & 0xF) << 8)  # cu, sh, se, xcc
busy = {}
for c, a_, b_ in zip(cu.tolist(), ent.tolist(), ext.tolist()):
    busy.setdefault(c, []).append((a_, b_))
occ = []
for c, iv in busy.items():
    occ.append(sum(b_ - a_ for a_, b_ in iv) / span)
occ = torch.tensor(occ)
print(f"  {len(busy)} distinct CUs seen; resident waves per CU averaged over the span: mean {occ.mean():.2f} min {occ.min():.2f} max {occ.max():.2f}")
hist = torch.histc(ent.float(), bins=10, min=0, max=span)
print("  entries per tenth of the span:", [int(v) for v in hist.tolist()])
hist = torch.histc(ext.float(), bins=10, min=0, max=span)
print("  exits per tenth of the span:  ", [int(v) for v in hist.tolist()])
