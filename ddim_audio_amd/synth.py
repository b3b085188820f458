"""Deterministic synthetic weights and inputs (torch-RNG independent).

SURVEY §8(c)/(d): torch's default init zeroes every ``Residual_Block.norm[2].weight``
(reference ``models/diffusion.py:25``), which makes every block the identity, so parity vectors
and benchmarks fill all 388 parameters from a counter-based hash of (parameter name, element
index) instead.  The same fill is applied to the reference ``Model`` when golden vectors are made
(``oracle/make_golden.py``), to the CPU oracle and to the HIP-backed ``Model``; only seeds and
outputs are stored, never the 189 MB of weights.
"""
import math
import zlib

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def uniform_pm1(tag, n, seed=0):
    """n values in [-1, 1), each an exact multiple of 2^-23 (so exact in fp32), from hash(tag, i)."""
    base = np.uint64((zlib.crc32(tag.encode()) << 20) ^ (seed * 0x632BE5AB))
    with np.errstate(over="ignore"):
        h = _splitmix64(np.arange(n, dtype=np.uint64) + _splitmix64(np.array([base], dtype=np.uint64))[0])
    u = (h >> np.uint64(40)).astype(np.float64) * (2.0 ** -23) - 1.0
    return u.astype(np.float32)


def _rule(name, shape):
    """(offset, amplitude) of the fill for one state_dict entry."""
    last = name.rsplit(".", 1)[-1]
    is_norm = ".norm." in name or "LayerNorm" in name
    if is_norm and last == "weight":
        if ".norm.2." in name:
            return 0.4, 0.3  # residual-branch gain: non-zero so blocks are not identities
        return 1.0, 0.3
    if last == "bias":
        return 0.0, 0.1
    # conv / conv-transpose / linear weight: variance 1/fan_in
    if len(shape) == 4:
        if ".conv.weight" in name and name.startswith("up_modules"):
            fan_in = shape[0] * 4  # ConvTranspose2d(k4,s2): [Cin, Cout, 4, 4], 2x2 taps reach one output
        else:
            fan_in = shape[1] * shape[2] * shape[3]
    else:
        fan_in = shape[-1]
    return 0.0, math.sqrt(3.0 / fan_in)


def fill_state_dict(sd, seed=0, skip=("temb.te",)):
    """Overwrite every tensor of ``sd`` (a name -> tensor mapping) in place; returns ``sd``."""
    with torch.no_grad():
        for name, t in sd.items():
            if name in skip:
                continue
            off, amp = _rule(name, tuple(t.shape))
            v = uniform_pm1(name, t.numel(), seed).astype(np.float64) * amp + off
            t.copy_(torch.from_numpy(v.astype(np.float32)).reshape(t.shape).to(t.dtype))
    return sd


def fill_module(module, seed=0):
    """Fill a module's parameters (the ``temb.te`` buffer keeps its sinusoid table)."""
    fill_state_dict(dict(module.named_parameters()), seed)
    return module


def gaussian(tag, shape, seed=0):
    """Standard-normal fp32 tensor from the hash stream (Box-Muller), for inputs and noise."""
    n = int(np.prod(shape))
    m = (n + 1) // 2
    u1 = (uniform_pm1(tag + "#a", m, seed).astype(np.float64) + 1.0) * 0.5
    u2 = (uniform_pm1(tag + "#b", m, seed).astype(np.float64) + 1.0) * 0.5
    r = np.sqrt(-2.0 * np.log(np.maximum(u1, 2.0 ** -24)))
    z = np.concatenate([r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)])[:n]
    return torch.from_numpy(z.astype(np.float32)).reshape(shape)
