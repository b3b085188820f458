"""Per-wave cycle breakdown + timeline of conv3_pipe_kernel (diagnostic build):
   DDIMX_LIB=ddim_audio_amd/libddimx_stamp.so [DDIMX_PIPE_DBG=1|2|4|6|7] python tools/pipe_stamps.py LEVEL [B] [XF]
   DBG bits (timing only, results are garbage): 1 no MFMAs, 2 no XF stage, 4 no EPI stage."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ddim_audio_amd import _lib  # noqa: E402

lvl = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 8; xf = int(sys.argv[3]) if len(sys.argv) > 3 else 2
lib = _lib.load()
tdt = torch.bfloat16
dev = torch.device("cuda", 0)
C = [32, 64][lvl]
H, W = 1024 >> lvl, 256 >> lvl
x = torch.randn(B, H, W, C, device=dev).to(tdt)
y = torch.empty_like(x)
wt = torch.randn(C, C, 3, 3, device=dev) * (1.0 / (9 * C) ** 0.5)
wf = torch.empty(9 * C * C, dtype=tdt, device=dev)
_lib.check(lib.ddimx_pack_conv_frag(_lib.ptr(wt), _lib.ptr(wf), C, C, _lib.stream()))
temb = torch.randn(B, C, device=dev) * 0.1
scale = torch.rand(B, C, device=dev) + 0.5
shift = torch.randn(B, C, device=dev) * 0.1
stats = torch.zeros(int(lib.ddimx_conv3x3_pipe_stats_floats(C, B, H, W)), device=dev)
stamps = torch.zeros(16 * (1 << 16), dtype=torch.int64, device=dev)
lib.ddimx_debug_set_stamps(_lib.ptr(stamps))
def run():
    _lib.check(lib.ddimx_conv3x3_pipe_fwd(C, _lib.ptr(x), _lib.ptr(wf), None, _lib.ptr(temb), C, _lib.ptr(scale), _lib.ptr(shift), xf,
                                          _lib.ptr(y), _lib.ptr(stats), B, H, W, _lib.stream()))
for _ in range(3):
    stamps.zero_()
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
raw = stamps.cpu().reshape(-1, 16)
raw = raw[raw[:, 12] > 0]
s = raw[:, :12].double()
names = ["tile bodies", "halo hand-over (lgkmcnt + barrier)", "drain (last block)", "statistics tail"]
tot = s.sum(1).mean()
tiles = 8 if not os.environ.get("DDIMX_PIPE_TPW") else int(os.environ["DDIMX_PIPE_TPW"])
print(f"pipe L{lvl} C={C} B={B} xf={xf} dbg={os.environ.get('DDIMX_PIPE_DBG', '0')}: {e0.elapsed_time(e1) * 50:.1f} us per launch (stamped build); "
      f"{s.shape[0]} waves, stamped cycles per wave {tot:.0f}")
for k, n in enumerate(names):
    print(f"  {n:36s} {s[:, k].mean():10.0f} cycles  {100 * s[:, k].mean() / tot:5.1f}%   per tile {s[:, k].mean() / tiles:8.0f}")
t0, t1, t2 = raw[:, 12].double(), raw[:, 13].double(), raw[:, 14].double()
base = t0.min()
span = float(((t2 - base) / 100.0).max())
print(f"  kernel span {span:.1f} us; prologue mean {(t1 - t0).mean() / 100:.2f} us (min {(t1 - t0).min() / 100:.2f} max {(t1 - t0).max() / 100:.2f}); "
      f"loop+tail mean {(t2 - t1).mean() / 100:.2f} us; wave life mean {(t2 - t0).mean() / 100:.2f} us; clock {tot / ((t2 - t1).mean() * 10):.2f} GHz")
print("  entries per tenth:", [int(v) for v in torch.histc(((t0 - base) / 100.0).float(), bins=10, min=0, max=span).tolist()])
print("  exits per tenth:  ", [int(v) for v in torch.histc(((t2 - base) / 100.0).float(), bins=10, min=0, max=span).tolist()])
