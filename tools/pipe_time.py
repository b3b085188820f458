"""Time the Residual_Block convs of levels 0-1 (HIP events), new kernel beside the one it replaces:
   pipe_time.py LEVEL B [H W]      prints, per input transform (1 affine, 2 affine + SiLU): conv3_pipe_kernel, the previous kernel
   (conv_mfma_kernel at C = 32, conv3_wreg_kernel at C = 64), achieved HBM rate of the algorithmic bytes, max |difference|."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddim_audio_amd import _lib
lvl = int(sys.argv[1]); B = int(sys.argv[2])
lib = _lib.load()
dt, tdt = _lib.DDIMX_BF16, torch.bfloat16
C = [32, 64][lvl]; H, W = 1024 >> lvl, 256 >> lvl
if len(sys.argv) > 4: H, W = int(sys.argv[3]), int(sys.argv[4])
x = torch.randn(B, H, W, C, device="cuda").to(tdt)
y_new, y_old = torch.empty_like(x), torch.empty_like(x)
wt = torch.randn(C, C, 3, 3, device="cuda") * (1.0 / (9 * C) ** 0.5)
wf = torch.empty(9 * C * C, dtype=tdt, device="cuda")
_lib.check(lib.ddimx_pack_conv_frag(_lib.ptr(wt), _lib.ptr(wf), C, C, _lib.stream()))
wp = wt.permute(2, 3, 0, 1).contiguous().to(tdt)   # [tap][O][I]: ddimx_pack_conv's layout
temb = torch.randn(B, C, device="cuda") * 0.1
scale = torch.rand(B, C, device="cuda") + 0.5; shift = torch.randn(B, C, device="cuda") * 0.1
stats = torch.zeros(max(int(lib.ddimx_conv3x3_stats_floats(dt, C, B, H, W)), int(lib.ddimx_conv3x3_pipe_stats_floats(C, B, H, W))), device="cuda")
def new(xf):
    _lib.check(lib.ddimx_conv3x3_pipe_fwd(C, _lib.ptr(x), _lib.ptr(wf), None, _lib.ptr(temb), C, _lib.ptr(scale), _lib.ptr(shift), xf,
                                          _lib.ptr(y_new), _lib.ptr(stats), B, H, W, _lib.stream()))
def old(xf):
    if C == 32:
        _lib.check(lib.ddimx_conv3x3_fwd(dt, C, _lib.ptr(x), _lib.ptr(wp), None, _lib.ptr(temb), C, _lib.ptr(scale), _lib.ptr(shift), xf, 1,
                                         _lib.ptr(y_old), _lib.ptr(stats), B, H, W, _lib.stream()))
    else:
        _lib.check(lib.ddimx_conv3x3_wreg_fwd(C, _lib.ptr(x), _lib.ptr(wp), _lib.ptr(wf), None, _lib.ptr(temb), C, _lib.ptr(scale), _lib.ptr(shift), xf, 1,
                                              _lib.ptr(y_old), _lib.ptr(stats), B, H, W, _lib.stream()))
def t(fn, xf, n=50):
    for _ in range(5): fn(xf)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn(xf)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
mb = 2 * B * H * W * C * 2 / 1e6
for xf in (1, 2):
    tn, to = t(new, xf), t(old, xf)
    d = float((y_new.float() - y_old.float()).abs().max())
    print(f"level {lvl} C {C} B {B} {H}x{W} xf {xf} TPW {os.environ.get('DDIMX_PIPE_TPW', '-')}: pipe {tn:.1f} us ({mb / tn * 1e3:.0f} GB/s = {mb / tn / 8:.3f} of 8 TB/s) | previous {to:.1f} us | max diff {d:.4f}")
