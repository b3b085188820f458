"""Does a power-capped kernel lose time on fewer CUs?  The level-0 conv looped on streams created with hipExtStreamCreateWithCUMask
(256 / 224 / 192 / 128 CUs enabled, spread evenly over the 8 XCDs), and two such streams side by side (heavy conv on 192 CUs, a chain of
small kernels on the other 64).   cumask_test.py"""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddim_audio_amd import _lib

hip = ctypes.CDLL("libamdhip64.so")
lib = _lib.load()
tdt = torch.bfloat16
B, C, H, W = 8, 32, 1024, 256
x = torch.randn(B, H, W, C, device="cuda").to(tdt); y = torch.empty_like(x)
wt = torch.randn(C, C, 3, 3, device="cuda") * (1.0 / (9 * C) ** 0.5)
wf = torch.empty(9 * C * C, dtype=tdt, device="cuda")
_lib.check(lib.ddimx_pack_conv_frag(_lib.ptr(wt), _lib.ptr(wf), C, C, _lib.stream()))
bias = torch.randn(C, device="cuda") * 0.1
scale = torch.rand(B, C, device="cuda") + 0.5; shift = torch.randn(B, C, device="cuda") * 0.1
stats = torch.zeros(int(lib.ddimx_conv3x3_pipe_stats_floats(C, B, H, W)), device="cuda")
# a light chain: level-4-sized resid launches (latency-bound)
xs = torch.randn(B, 64, 16, 192, device="cuda").to(tdt); hs = torch.randn_like(xs); ys = torch.empty_like(xs)
sc4 = torch.rand(B, 192, device="cuda") + 0.5; sh4 = torch.randn(B, 192, device="cuda") * 0.1
st4 = torch.zeros(1 << 20, device="cuda")

def masked_stream(keep_per_xcd):
    """CU mask with `keep_per_xcd` of each XCD's 32 CUs enabled (bit i = CU i; XCD-major numbering assumed: CU = xcd * 32 + k)."""
    words = (ctypes.c_uint32 * 8)()
    for xcd in range(8):
        words[xcd] = (1 << keep_per_xcd) - 1 if keep_per_xcd < 32 else 0xFFFFFFFF
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)

def inv_masked_stream(skip_per_xcd):
    words = (ctypes.c_uint32 * 8)()
    for xcd in range(8):
        words[xcd] = 0xFFFFFFFF & ~((1 << skip_per_xcd) - 1)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)

def conv():
    _lib.check(lib.ddimx_conv3x3_pipe_fwd(C, _lib.ptr(x), _lib.ptr(wf), _lib.ptr(bias), None, 0, _lib.ptr(scale), _lib.ptr(shift), 1,
                                          _lib.ptr(y), _lib.ptr(stats), B, H, W, _lib.stream()))
def light():
    _lib.check(lib.ddimx_resid_gn_fwd(_lib.DDIMX_BF16, 192, _lib.ptr(xs), _lib.ptr(hs), _lib.ptr(sc4), _lib.ptr(sh4), _lib.ptr(ys), _lib.ptr(st4), B, 64, 16, _lib.stream()))

def loop(fn, stream, n):
    with torch.cuda.stream(stream):
        for _ in range(200): fn()
    torch.cuda.synchronize(); t0 = time.time()
    with torch.cuda.stream(stream):
        for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.time() - t0) * 1e6 / n

print("conv level 0 B=8 (second conv of the block), sustained us per launch by enabled CUs:")
for keep in (32, 28, 24, 16):
    s = masked_stream(keep)
    print(f"  {keep * 8:3d} CUs: {loop(conv, s, 20000):.1f} us")
print("light chain alone (level-4 resid launches), us per launch:")
for keep in (32, 8, 4):
    s = masked_stream(keep)
    print(f"  {keep * 8:3d} CUs: {loop(light, s, 20000):.2f} us")
# side by side: heavy on 24 CUs per XCD, light on the other 8; and both unmasked
for name, sh_, sl_ in (("masked 192 | 64", inv_masked_stream(8), masked_stream(8)), ("both unmasked", torch.cuda.Stream(), torch.cuda.Stream())):
    n = 15000
    for st_, fn in ((sh_, conv), (sl_, light)):
        with torch.cuda.stream(st_):
            for _ in range(100): fn()
    torch.cuda.synchronize(); t0 = time.time()
    eh, el = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(n):
        with torch.cuda.stream(sh_): conv()
        with torch.cuda.stream(sl_):
            for _ in range(8): light()
    torch.cuda.synchronize()
    dt = (time.time() - t0) * 1e6 / n
    print(f"side by side ({name}): {dt:.1f} us per {{1 conv + 8 light launches}} (alone: conv ~65, 8 light ~{8 * 4.6:.0f})")
