"""GPU parity tests of the training path: HIP forward-with-tape and backward kernels (through the C ABI) against
autograd over the CPU oracle (itself pinned to the reference's gradients by tests/golden/train.npz).
Gates: fp32 mode max|d| <= 1e-4 * std (x4 for parameter gradients, which are long fp32 sums); bf16 mode
rms <= 2e-2 * std, max <= 1.5e-1 * std -- the same yardsticks as the forward."""
import numpy as np
import pytest
import torch

from ddim_audio_amd import _lib, synth
from oracle import ref_cpu
import gpu_util as G
from test_gpu_ops import _rb_sd

pytestmark = pytest.mark.gpu
DTS = [G.F32, G.BF16]


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("c,hw,b", [(32, (40, 72), 3), (64, (24, 40), 2), (96, (17, 33), 2), (128, (16, 24), 2),
                                    (192, (9, 20), 1), (256, (20, 9), 2)])
def test_resblock_backward(dt, c, hw, b):
    p = f"rbt{c}."
    sd = _rb_sd(p, c)
    x = synth.gaussian(p + "x", (b, c, *hw)) * 1.5 + 0.3
    temb = synth.gaussian(p + "temb", (b, c)) * 0.5
    dy = synth.gaussian(p + "dy", (b, c, *hw))
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr, tr = x.clone().requires_grad_(True), temb.clone().requires_grad_(True)
    yr = ref_cpu.residual_block(leaf, p, xr, tr)
    yr.backward(dy)
    y, dx, grads, dtemb = G.resblock_train(sd, p, x, temb, dy, dt)
    G.check_close(y, yr.detach(), dt, f"train fwd C={c}")
    G.check_close(dx, xr.grad, dt, f"dx C={c}")
    G.check_close(dtemb, tr.grad, dt, f"dtemb C={c}", scale=4.0)
    for n, gv in grads.items():
        G.check_close(gv, leaf[p + n].grad, dt, f"d {n} C={c}", scale=4.0)
