"""Per-phase cycle breakdown of conv3_ws_kernel per role (diagnostic build):
   DDIMX_LIB=ddim_audio_amd/libddimx_stamp.so python tools/ws_stamps.py LEVEL [B] [XF]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddim_audio_amd import _lib
lvl = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 8; xf = int(sys.argv[3]) if len(sys.argv) > 3 else 2
lib = _lib.load()
dt, tdt = _lib.DDIMX_BF16, torch.bfloat16
dev = torch.device("cuda", 0)
C = [32, 64, 96][lvl]
H, W = 1024 >> lvl, 256 >> lvl
x = torch.randn(B, H, W, C, device=dev).to(tdt); y = torch.empty_like(x)
wt = torch.randn(C, C, 3, 3, device=dev) * (1.0 / (9 * C) ** 0.5)
wf = torch.empty(9 * C * C, dtype=tdt, device=dev)
_lib.check(lib.ddimx_pack_conv_frag(_lib.ptr(wt), _lib.ptr(wf), C, C, _lib.stream()))
temb = torch.randn(B, C, device=dev) * 0.1; scale = torch.rand(B, C, device=dev) + 0.5; shift = torch.randn(B, C, device=dev) * 0.1
stats = torch.zeros(int(lib.ddimx_conv3x3_stats_floats(dt, C, B, H, W)) + 4096, device=dev)
stamps = torch.zeros(16 * (1 << 15), dtype=torch.int64, device=dev)
lib.ddimx_debug_set_stamps(_lib.ptr(stamps))
for _ in range(3):
    stamps.zero_()
    _lib.check(lib.ddimx_conv3x3_ws_fwd(C, _lib.ptr(x), _lib.ptr(wf), None, _lib.ptr(temb), C, _lib.ptr(scale), _lib.ptr(shift), xf, 1,
                                        _lib.ptr(y), _lib.ptr(stats), B, H, W, _lib.stream()))
torch.cuda.synchronize()
raw = stamps.cpu().reshape(-1, 16)
raw = raw[raw[:, 12] > 0]
names = ["(loop top)", "MFMA loop", "wait A", "epilogue 1", "wait B", "(loader top)", "drain", "commit (transform)", "issue", "wait A", "wait B", "wait for the halo loads"]
for role, sel in (("MFMA waves", raw[:, 1] > 0), ("loader waves", raw[:, 6] + raw[:, 7] > 0)):
    r = raw[sel]
    s = r[:, :12].double()
    tot = s.sum(1).mean()
    print(f"ws L{lvl} C={C} B={B} xf={xf} {role}: {s.shape[0]} waves, stamped cycles per wave {tot:.0f}")
    for k, n in enumerate(names):
        if s[:, k].sum() > 0:
            print(f"  {n:24s} {s[:, k].mean():10.0f} cycles  {100 * s[:, k].mean() / tot:5.1f}%")
    t0, t1, t2 = r[:, 12].double(), r[:, 13].double(), r[:, 14].double()
    base = raw[:, 12].double().min()
    print(f"  prologue mean {(t1 - t0).mean() / 100:.2f} us; loop+tail mean {(t2 - t1).mean() / 100:.2f} us; kernel span {float(((raw[:, 14].double() - base) / 100).max()):.1f} us")
