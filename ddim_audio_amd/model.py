"""``Model(config)`` -- the reference's U-Net epsilon-predictor interface on the HIP library.

Mirrors reference ``models/diffusion.py:170-294``: same constructor argument (the full config
Namespace), same 388 parameter + 1 buffer names and shapes (so reference checkpoints load with
``strict=True`` and ``EMAHelper`` / ``classify_group`` see the names they expect), same
``forward(input[B,C,T,F], t[B] int64) -> [B,C,T,F]``.  The module holds parameters only; every
arithmetic op of ``forward`` runs in ``libddimx.so`` (``csrc/``) through the C ABI of
``include/ddimx.h``.  There is no CPU or eager-PyTorch fallback: a missing library, a CPU tensor or
an unsupported configuration raises.
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn

_POS_CH = 128  # reference models/diffusion.py:98
_EMB_CH = 512  # reference models/diffusion.py:99
_GROUPS = 8  # reference models/diffusion.py:19


def embedding_sizes(mcfg):
    """Per-block timestep-embedding widths, down blocks then up blocks (models/diffusion.py:178-184)."""
    e = [c for r, c in zip(mcfg.res, mcfg.ch) for _ in range(r)]
    return e + e[::-1]


def _rb(inv, p, c, k):
    for i in range(3):
        inv[f"{p}norm.{i}.weight"] = (c,)
        if i < 2:
            inv[f"{p}norm.{i}.bias"] = (c,)
    inv[f"{p}conv.0.weight"] = (c, c, k, k)
    inv[f"{p}conv.1.weight"] = (c, c, k, k)
    inv[f"{p}conv.1.bias"] = (c,)


def state_inventory(config):
    """name -> shape of the state_dict, in the reference's registration order."""
    m = config.model
    assert len(m.ch) == len(m.krn) == len(m.res)
    nlev = len(m.ch)
    inv = OrderedDict()
    emb = sum(embedding_sizes(m))
    inv["temb.te"] = (config.diffusion.num_diffusion_timesteps, _POS_CH)
    for i, (o, k) in enumerate(((_EMB_CH, _POS_CH), (_EMB_CH, _EMB_CH), (emb, _EMB_CH))):
        inv[f"temb.weight.{i}.weight"] = (o, k)
        inv[f"temb.weight.{i}.bias"] = (o,)
    inv["down_modules.0.weight"] = (m.ch[0], m.channels, 3, 3)
    inv["down_modules.0.bias"] = (m.ch[0],)
    for lvl in range(nlev):
        base = f"down_modules.{lvl + 1}."
        j = 0
        if lvl > 0:
            inv[base + "0.conv.weight"] = (m.ch[lvl], m.ch[lvl - 1], 4, 4)
            inv[base + "0.conv.bias"] = (m.ch[lvl],)
            j = 1
        for r in range(m.res[lvl]):
            _rb(inv, f"{base}{j + r}.", m.ch[lvl], m.krn[lvl])
    for k in range(nlev):
        lvl = nlev - 1 - k
        base = f"up_modules.{k}."
        for r in range(m.res[lvl]):
            _rb(inv, f"{base}{r}.", m.ch[lvl], m.krn[lvl])
        if lvl > 0:  # ConvTranspose2d weight layout [Cin, Cout, 4, 4]
            inv[f"{base}{m.res[lvl]}.conv.weight"] = (m.ch[lvl], m.ch[lvl - 1], 4, 4)
            inv[f"{base}{m.res[lvl]}.conv.bias"] = (m.ch[lvl - 1],)
    inv[f"up_modules.{nlev}.weight"] = (m.channels, m.ch[0], 3, 3)
    inv[f"up_modules.{nlev}.bias"] = (m.channels,)
    tr = m.transformers
    width = m.ch[-1] * (m.f_size // (2 ** (nlev - 1)))
    hid, inter = tr.kwargs.hidden_size, tr.kwargs.intermediate_size
    assert tr.channels == hid, "transformers.channels must equal kwargs.hidden_size"
    inv["transformer.embedding.LayerNorm.weight"] = (width,)
    inv["transformer.embedding.LayerNorm.bias"] = (width,)
    inv["transformer.embedding.projection.weight"] = (hid, width)
    inv["transformer.embedding.projection.bias"] = (hid,)
    for i in range(tr.kwargs.num_hidden_layers):
        p = f"transformer.encoder.layer.{i}."
        inv[p + "fourier.output.LayerNorm.weight"] = (hid,)
        inv[p + "fourier.output.LayerNorm.bias"] = (hid,)
        inv[p + "intermediate.dense.weight"] = (inter, hid)
        inv[p + "intermediate.dense.bias"] = (inter,)
        inv[p + "output.dense.weight"] = (hid, inter)
        inv[p + "output.dense.bias"] = (hid,)
        inv[p + "output.LayerNorm.weight"] = (hid,)
        inv[p + "output.LayerNorm.bias"] = (hid,)
    inv["transformer.compute_out.weight"] = (width, hid)
    inv["transformer.compute_out.bias"] = (width,)
    return inv


def timestep_table(n, ch=_POS_CH):
    """``temb.te``: interleaved sin/cos table (reference models/diffusion.py:81-102), built in fp32
    with the same op order as the reference so the buffer matches it to the last bit."""
    pos = torch.arange(n, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, ch, 2, dtype=torch.float32) * (-math.log(10000.0) / ch))
    te = torch.zeros(n, ch)
    te[:, 0::2] += torch.sin(pos * div)
    te[:, 1::2] += torch.cos(pos * div)
    return te


class _Node(nn.Module):
    """Parameter container; the tree of these reproduces the reference's dotted names."""


def _default_init_(name, p):
    last = name.rsplit(".", 1)[-1]
    is_norm = ".norm." in name or "LayerNorm" in name
    with torch.no_grad():
        if is_norm:
            if last == "bias" or ".norm.2." in name:
                p.zero_()  # norm[2].weight = 0: every block starts as the identity (models/diffusion.py:25)
            else:
                p.fill_(1.0)
            return
        if p.dim() == 4:
            transposed = name.startswith("up_modules") and ".conv.weight" in name
            fan_in = (p.shape[0] if not transposed else p.shape[1]) if False else None
            # torch's fan_in convention is shape[1]*k*k for both Conv2d and ConvTranspose2d weights
            fan_in = p.shape[1] * p.shape[2] * p.shape[3]
        elif p.dim() == 2:
            fan_in = p.shape[1]
        else:
            fan_in = None
        if fan_in is not None:
            p.uniform_(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in))
            p._ddimx_fan_in = fan_in


def parse_tensor_type(s):
    """Legacy tensor-type string (configs/audio.yml:26,42) -> (device or None, dtype)."""
    if not s:
        return None, torch.float32
    table = {"FloatTensor": torch.float32, "BFloat16Tensor": torch.bfloat16}
    kind = s.rsplit(".", 1)[-1]
    if kind not in table:
        raise NotImplementedError(f"model dtype {s!r}: the HIP path implements FloatTensor and BFloat16Tensor")
    return ("cuda" if ".cuda." in s else "cpu"), table[kind]


def build_parameters(root, config):
    inv = state_inventory(config)
    fan = {}
    for name, shape in inv.items():
        parts = name.split(".")
        node = root
        for part in parts[:-1]:
            if part not in node._modules:
                node.add_module(part, _Node())
            node = node._modules[part]
        if name == "temb.te":
            node.register_buffer("te", timestep_table(*shape))
            continue
        p = nn.Parameter(torch.empty(shape, dtype=torch.float32))
        _default_init_(name, p)
        if parts[-1] == "weight" and hasattr(p, "_ddimx_fan_in"):
            fan[".".join(parts[:-1])] = p._ddimx_fan_in
        node.register_parameter(parts[-1], p)
    # biases of conv / linear layers: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) like torch's defaults
    with torch.no_grad():
        for name, p in root.named_parameters():
            owner, last = name.rsplit(".", 1)
            if last == "bias" and owner in fan:
                b = 1.0 / math.sqrt(fan[owner])
                p.uniform_(-b, b)
    return inv
