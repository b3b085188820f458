"""Checkpoint wire format of the reference runner (``runners/diffusion.py:185-199,293-313``).

``torch.save`` of the list ``[model.state_dict(), optimizer.state_dict(), epoch, step, ema_shadow]`` (the EMA entry
only when ``config.model.ema``); sampling loads ``states[0]`` with ``strict=True`` and, with EMA enabled, swaps
``states[-1]`` into the parameters.  Because ``Model`` keeps the reference's 389 state_dict keys, files written by the
reference load here and vice versa.  Host-side plumbing only (no tensor arithmetic of the hot path).
"""
import os

import torch

from .ema import EMAHelper


def save_checkpoint(log_path, model, optimizer, epoch, step, ema_helper=None):
    """Write ``ckpt_{step}.pth`` and ``ckpt.pth`` like ``train_step`` does (runners/diffusion.py:185-199)."""
    states = [model.state_dict(), optimizer.state_dict() if optimizer is not None else {}, epoch, step]
    if ema_helper is not None:
        states.append(ema_helper.state_dict())
    os.makedirs(log_path, exist_ok=True)
    torch.save(states, os.path.join(log_path, "ckpt_{}.pth".format(step)))
    torch.save(states, os.path.join(log_path, "ckpt.pth"))
    return states


def load_for_sampling(log_path, model, use_ema=True, ema_rate=0.9999, ckpt_id=None, map_location=None):
    """``Diffusion.sample`` up to ``model.eval()`` (runners/diffusion.py:293-313,331): load states[0] strictly,
    optionally swap in the EMA shadow (states[-1]), return (model in eval mode, ema_helper or None)."""
    name = "ckpt.pth" if ckpt_id is None else f"ckpt_{ckpt_id}.pth"
    states = torch.load(os.path.join(log_path, name), map_location=map_location, weights_only=False)
    model.load_state_dict(states[0], strict=True)
    ema_helper = None
    if use_ema:
        ema_helper = EMAHelper(mu=ema_rate)
        ema_helper.register(model)
        dev = next(model.parameters()).device
        ema_helper.load_state_dict({k: v.to(dev) for k, v in states[-1].items()})
        ema_helper.ema(model)
    model.eval()
    return model, ema_helper
