"""Time the Residual_Block backward pieces at one level (tuning tool): python tools/wgrad_one.py LEVEL B [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from ddim_audio_amd import _lib, synth  # noqa: E402
import gpu_util as G  # noqa: E402

lvl, b = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
c = [32, 64, 96, 128, 192, 256][lvl]
h, w = 1024 >> lvl, 256 >> lvl
dt = G.BF16
lib = _lib.load()
x = torch.randn(b, h, w, c, device="cuda").to(torch.bfloat16)
dy = torch.randn_like(x)
y, u1, u2, dx = (torch.empty_like(x) for _ in range(4))
p = "t."
shapes = {p + "norm.0.weight": (c,), p + "norm.0.bias": (c,), p + "norm.1.weight": (c,), p + "norm.1.bias": (c,), p + "norm.2.weight": (c,),
          p + "conv.0.weight": (c, c, 3, 3), p + "conv.1.weight": (c, c, 3, 3), p + "conv.1.bias": (c,)}
sd = synth.fill_state_dict({k: torch.empty(s) for k, s in shapes.items()})
names = ("norm.0.weight", "norm.0.bias", "norm.1.weight", "norm.1.bias", "norm.2.weight", "conv.1.bias")
k = {n: G.g(sd[p + n]) for n in names}
w0, w1 = G.pack_conv(sd[p + "conv.0.weight"], dt), G.pack_conv(sd[p + "conv.1.weight"], dt)
wd0, wd1 = G.pack_conv_dgrad(sd[p + "conv.0.weight"], dt), G.pack_conv_dgrad(sd[p + "conv.1.weight"], dt)
small = torch.empty(int(lib.ddimx_rb_tape_floats(b, c)), dtype=torch.float32, device="cuda")
ws = torch.empty(int(lib.ddimx_op_workspace_bytes(dt, b, c, h, w)), dtype=torch.uint8, device="cuda")
bws = torch.empty(int(lib.ddimx_resblock_bwd_workspace_bytes(dt, b, c, h, w)), dtype=torch.uint8, device="cuda")
tg = torch.randn(b, c, device="cuda")
grads = {n: torch.empty(tuple(sd[p + n].shape), device="cuda") for n in names + ("conv.0.weight", "conv.1.weight")}
dtemb = torch.empty(b, c, device="cuda")


def fwd():
    _lib.check(lib.ddimx_resblock_fwd_train(dt, c, _lib.ptr(x), _lib.ptr(y), _lib.ptr(tg), c, _lib.ptr(k["norm.0.weight"]),
                                            _lib.ptr(k["norm.0.bias"]), _lib.ptr(w0), _lib.ptr(k["norm.1.weight"]), _lib.ptr(k["norm.1.bias"]),
                                            _lib.ptr(w1), _lib.ptr(k["conv.1.bias"]), _lib.ptr(k["norm.2.weight"]), _lib.ptr(u1), _lib.ptr(u2),
                                            _lib.ptr(small), _lib.ptr(ws), b, h, w, _lib.stream()))


def bwd():
    _lib.check(lib.ddimx_resblock_bwd(dt, c, _lib.ptr(x), _lib.ptr(u1), _lib.ptr(u2), _lib.ptr(small), _lib.ptr(dy), _lib.ptr(dx),
                                      _lib.ptr(k["norm.0.weight"]), _lib.ptr(k["norm.1.weight"]), _lib.ptr(k["norm.2.weight"]), _lib.ptr(wd0),
                                      _lib.ptr(wd1), _lib.ptr(grads["norm.0.weight"]), _lib.ptr(grads["norm.0.bias"]),
                                      _lib.ptr(grads["conv.0.weight"]), _lib.ptr(grads["norm.1.weight"]), _lib.ptr(grads["norm.1.bias"]),
                                      _lib.ptr(grads["conv.1.weight"]), _lib.ptr(grads["conv.1.bias"]), _lib.ptr(grads["norm.2.weight"]),
                                      _lib.ptr(dtemb), c, _lib.ptr(bws), b, h, w, _lib.stream()))


fwd()
for name, fn in (("fwd", fwd), ("bwd", bwd)):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print(f"level {lvl} C={c} B={b} resblock {name}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms")
