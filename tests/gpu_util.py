"""Helpers for the GPU parity tests: thin ctypes calls into libddimx per op (C ABI of include/ddimx.h)."""
import torch

from ddim_audio_amd import _lib

F32, BF16 = _lib.DDIMX_F32, _lib.DDIMX_BF16
TORCH_DT = {F32: torch.float32, BF16: torch.bfloat16}
# parity gates (SURVEY section 8c yardsticks): relative to the std of the expected tensor
TOL = {F32: dict(mx=1e-4, rms=2e-5), BF16: dict(mx=1.5e-1, rms=2e-2)}


def dev():
    return torch.device("cuda", 0)


def to_nhwc(x, dt):
    lib = _lib.load()
    x = x.to(dev(), torch.float32).contiguous()
    b, c, h, w = x.shape
    out = torch.empty((b, h, w, c), dtype=TORCH_DT[dt], device=dev())
    _lib.check(lib.ddimx_to_nhwc(dt, _lib.ptr(x), _lib.ptr(out), b, c, h, w, _lib.stream()))
    return out


def from_nhwc(y, dt):
    lib = _lib.load()
    b, h, w, c = y.shape
    out = torch.empty((b, c, h, w), dtype=torch.float32, device=dev())
    _lib.check(lib.ddimx_from_nhwc(dt, _lib.ptr(y), _lib.ptr(out), b, c, h, w, _lib.stream()))
    return out.cpu()


def pack_conv(w, dt):
    lib = _lib.load()
    w = w.to(dev(), torch.float32).contiguous()
    o, i, kh, kw = w.shape
    dst = torch.empty(kh * kw * o * i, dtype=TORCH_DT[dt], device=dev())
    _lib.check(lib.ddimx_pack_conv(dt, _lib.ptr(w), _lib.ptr(dst), o, i, kh, kw, _lib.stream()))
    return dst


def pack_convT(w, dt):
    lib = _lib.load()
    w = w.to(dev(), torch.float32).contiguous()
    i, o = w.shape[0], w.shape[1]
    dst = torch.empty(2 * 6 * 2 * o * i, dtype=TORCH_DT[dt], device=dev())
    _lib.check(lib.ddimx_pack_convT(dt, _lib.ptr(w), _lib.ptr(dst), i, o, _lib.stream()))
    return dst


def g(t):
    return t.to(dev(), torch.float32).contiguous()


def resblock(sd, p, x, temb, dt):
    lib = _lib.load()
    b, c, h, w = x.shape
    xn = to_nhwc(x, dt)
    yn = torch.empty_like(xn)
    ws = torch.empty(int(lib.ddimx_op_workspace_bytes(dt, b, c, h, w)), dtype=torch.uint8, device=dev())
    keep = [g(sd[p + k]) for k in ("norm.0.weight", "norm.0.bias", "norm.1.weight", "norm.1.bias", "norm.2.weight", "conv.1.bias")]
    w0, w1 = pack_conv(sd[p + "conv.0.weight"], dt), pack_conv(sd[p + "conv.1.weight"], dt)
    tg = g(temb)
    _lib.check(lib.ddimx_resblock_fwd(dt, c, _lib.ptr(xn), _lib.ptr(yn), _lib.ptr(tg), tg.shape[1], _lib.ptr(keep[0]),
                                      _lib.ptr(keep[1]), _lib.ptr(w0), _lib.ptr(keep[2]), _lib.ptr(keep[3]), _lib.ptr(w1),
                                      _lib.ptr(keep[5]), _lib.ptr(keep[4]), _lib.ptr(ws), b, h, w, _lib.stream()))
    return from_nhwc(yn, dt)


def downsample(wt, bias, x, dt):
    lib = _lib.load()
    b, cin, h, w = x.shape
    cout = wt.shape[0]
    xn = to_nhwc(x, dt)
    yn = torch.empty((b, h // 2, w // 2, cout), dtype=TORCH_DT[dt], device=dev())
    wp, bg = pack_conv(wt, dt), g(bias)
    _lib.check(lib.ddimx_downsample_fwd(dt, cin, cout, _lib.ptr(xn), _lib.ptr(wp), _lib.ptr(bg), _lib.ptr(yn), b, h, w, _lib.stream()))
    return from_nhwc(yn, dt)


def upsample_add(wt, bias, x, skip, dt):
    lib = _lib.load()
    b, cin, h, w = x.shape
    cout = wt.shape[1]
    xn, sn = to_nhwc(x, dt), to_nhwc(skip, dt)
    yn = torch.empty_like(sn)
    wp, b2 = pack_convT(wt, dt), g(torch.cat([bias, bias]))
    _lib.check(lib.ddimx_upsample_add_fwd(dt, cin, cout, _lib.ptr(xn), _lib.ptr(wp), _lib.ptr(b2), _lib.ptr(sn), _lib.ptr(yn), b, h, w,
                                          _lib.stream()))
    return from_nhwc(yn, dt)


def check_close(got, want, dt, what="", scale=1.0):
    got = torch.as_tensor(got, dtype=torch.float64).reshape(-1)
    want = torch.as_tensor(want, dtype=torch.float64).reshape(-1)
    assert got.shape == want.shape, (got.shape, want.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    s = float(want.std()) + 1e-30
    d = got - want
    mx, rms = float(d.abs().max()) / s, float(d.square().mean().sqrt()) / s
    tol = TOL[dt]
    assert mx <= tol["mx"] * scale and rms <= tol["rms"] * scale, f"{what}: max {mx:.3e} rms {rms:.3e} (rel. to std) exceeds {tol}"
    return mx, rms


# ---- training ops ------------------------------------------------------------------------------------
def pack_conv_dgrad(w, dt):
    lib = _lib.load()
    w = w.to(dev(), torch.float32).contiguous()
    o, i = w.shape[0], w.shape[1]
    dst = torch.empty(9 * o * i, dtype=TORCH_DT[dt], device=dev())
    _lib.check(lib.ddimx_pack_conv_dgrad(dt, _lib.ptr(w), _lib.ptr(dst), o, i, _lib.stream()))
    return dst


def resblock_train(sd, p, x, temb, dy, dt):
    """Training forward + backward of one Residual_Block: returns (y, dx, {param: grad}, dtemb), all on the CPU."""
    lib = _lib.load()
    b, c, h, w = x.shape
    xn, dyn = to_nhwc(x, dt), to_nhwc(dy, dt)
    yn, u1, u2, dxn = (torch.empty_like(xn) for _ in range(4))
    small = torch.empty(int(lib.ddimx_rb_tape_floats(b, c)), dtype=torch.float32, device=dev())
    ws = torch.empty(int(lib.ddimx_op_workspace_bytes(dt, b, c, h, w)), dtype=torch.uint8, device=dev())
    names = ("norm.0.weight", "norm.0.bias", "norm.1.weight", "norm.1.bias", "norm.2.weight", "conv.1.bias")
    k = {n: g(sd[p + n]) for n in names}
    w0, w1 = pack_conv(sd[p + "conv.0.weight"], dt), pack_conv(sd[p + "conv.1.weight"], dt)
    tg = g(temb)
    _lib.check(lib.ddimx_resblock_fwd_train(dt, c, _lib.ptr(xn), _lib.ptr(yn), _lib.ptr(tg), tg.shape[1], _lib.ptr(k["norm.0.weight"]),
                                            _lib.ptr(k["norm.0.bias"]), _lib.ptr(w0), _lib.ptr(k["norm.1.weight"]),
                                            _lib.ptr(k["norm.1.bias"]), _lib.ptr(w1), _lib.ptr(k["conv.1.bias"]),
                                            _lib.ptr(k["norm.2.weight"]), _lib.ptr(u1), _lib.ptr(u2), _lib.ptr(small), _lib.ptr(ws),
                                            b, h, w, _lib.stream()))
    wd0, wd1 = pack_conv_dgrad(sd[p + "conv.0.weight"], dt), pack_conv_dgrad(sd[p + "conv.1.weight"], dt)
    grads = {n: torch.full(tuple(sd[p + n].shape), float("nan"), device=dev()) for n in names + ("conv.0.weight", "conv.1.weight")}
    dtemb = torch.full((b, c), float("nan"), device=dev())
    bws = torch.empty(int(lib.ddimx_resblock_bwd_workspace_bytes(dt, b, c, h, w)), dtype=torch.uint8, device=dev())
    _lib.check(lib.ddimx_resblock_bwd(dt, c, _lib.ptr(xn), _lib.ptr(u1), _lib.ptr(u2), _lib.ptr(small), _lib.ptr(dyn), _lib.ptr(dxn),
                                      _lib.ptr(k["norm.0.weight"]), _lib.ptr(k["norm.1.weight"]), _lib.ptr(k["norm.2.weight"]),
                                      _lib.ptr(wd0), _lib.ptr(wd1), _lib.ptr(grads["norm.0.weight"]), _lib.ptr(grads["norm.0.bias"]),
                                      _lib.ptr(grads["conv.0.weight"]), _lib.ptr(grads["norm.1.weight"]), _lib.ptr(grads["norm.1.bias"]),
                                      _lib.ptr(grads["conv.1.weight"]), _lib.ptr(grads["conv.1.bias"]), _lib.ptr(grads["norm.2.weight"]),
                                      _lib.ptr(dtemb), c, _lib.ptr(bws), b, h, w, _lib.stream()))
    torch.cuda.synchronize()
    return from_nhwc(yn, dt), from_nhwc(dxn, dt), {n: v.cpu() for n, v in grads.items()}, dtemb.cpu()
