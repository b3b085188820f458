"""Checkpoint wire format of the reference runner (``runners/diffusion.py:185-199,239-254,293-313``).

``torch.save`` of the list ``[model.state_dict(), optimizer.state_dict(), epoch, step, ema_shadow]`` (the EMA entry
only when ``config.model.ema``); sampling loads ``states[0]`` with ``strict=True`` and, with EMA enabled, swaps
``states[-1]`` into the parameters.  Because ``Model`` keeps the reference's 389 state_dict keys and ``FusedAdam`` keeps
torch's optimizer layout, files written by the reference load here (``tests/golden/ckpt_micro.pth`` was written by the
reference's own ``train_step``) and files written here load in the reference.

Resume is fixed forward.  The reference saves only the LAST optimizer of its dict (the loop variable leaked from
``:162``) and its resume branch cannot run (``self.config.optim.eps`` does not exist, ``:248``; ``optimizer`` is whatever
the construction loop left behind, ``:249``).  Here ``states[1]`` still IS the last optimizer's state_dict -- so the list
stays readable by the reference -- and carries one extra key, ``"ddimx_resume"``, that ``torch.optim.Optimizer
.load_state_dict`` ignores: every optimizer's and scheduler's state by group name and the dropout call counter.
``resume_training`` restores all of it; from a reference-written file (no such key) it restores what that file holds: the
model, the last group's optimizer, epoch / step, the EMA shadow, and re-derives the LambdaLR factors from ``step``.
Host-side plumbing only (no tensor arithmetic of the hot path).
"""
import os

import torch

from .ema import EMAHelper

RESUME_KEY = "ddimx_resume"


def _as_dict(optimizers):
    if optimizers is None:
        return {}
    if isinstance(optimizers, dict):
        return optimizers
    return {"default": optimizers}


def save_checkpoint(log_path, model, optimizer, epoch, step, ema_helper=None, schedulers=None):
    """Write ``ckpt_{step}.pth`` and ``ckpt.pth`` like ``train_step`` does (runners/diffusion.py:185-199).
    ``optimizer``: one optimizer or the ``{group: optimizer}`` dict of the training step (``schedulers`` likewise)."""
    opts = _as_dict(optimizer)
    last = list(opts.values())[-1] if opts else None
    osd = dict(last.state_dict()) if last is not None else {}
    if opts:
        osd[RESUME_KEY] = {
            "optimizers": {k: o.state_dict() for k, o in opts.items()},
            "schedulers": {k: s.state_dict() for k, s in (schedulers or {}).items()},
            "dropout_calls": int(getattr(model, "_dropout_calls", 0)),
        }
    states = [model.state_dict(), osd, epoch, step]
    if ema_helper is not None:
        states.append(ema_helper.state_dict())
    os.makedirs(log_path, exist_ok=True)
    torch.save(states, os.path.join(log_path, "ckpt_{}.pth".format(step)))
    torch.save(states, os.path.join(log_path, "ckpt.pth"))
    return states


def _load(log_path, ckpt_id, map_location):
    name = "ckpt.pth" if ckpt_id is None else f"ckpt_{ckpt_id}.pth"
    return torch.load(os.path.join(log_path, name), map_location=map_location, weights_only=False)


def load_for_sampling(log_path, model, use_ema=True, ema_rate=0.9999, ckpt_id=None, map_location=None):
    """``Diffusion.sample`` up to ``model.eval()`` (runners/diffusion.py:293-313,331): load states[0] strictly,
    optionally swap in the EMA shadow (states[-1]), return (model in eval mode, ema_helper or None)."""
    states = _load(log_path, ckpt_id, map_location)
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


def resume_training(log_path, model, optimizers, schedulers=None, ema_helper=None, ckpt_id=None, map_location=None):
    """The working counterpart of the reference's resume branch (runners/diffusion.py:239-254): returns (epoch, step).
    ``optimizers`` / ``schedulers``: the ``{group: ...}`` dicts of the training step (``train.TrainingState``)."""
    states = dict(zip(["model", "optimizer", "epoch", "step", "ema_helper"], _load(log_path, ckpt_id, map_location)))
    model.load_state_dict(states["model"], strict=True)
    if hasattr(model, "invalidate"):
        model.invalidate()
    opts, schs = _as_dict(optimizers), dict(schedulers or {})
    osd = dict(states["optimizer"])
    extra = osd.pop(RESUME_KEY, None)
    step = int(states["step"])
    if extra is not None:
        missing = set(opts) - set(extra["optimizers"])
        if missing:
            raise RuntimeError(f"checkpoint has no optimizer state for group(s) {sorted(missing)}")
        for k, o in opts.items():
            o.load_state_dict(extra["optimizers"][k])
        for k, s in schs.items():
            if k in extra["schedulers"]:
                s.load_state_dict(extra["schedulers"][k])
        model._dropout_calls = int(extra.get("dropout_calls", 0))
    elif opts:
        # a file written by the reference: it holds the state of the last optimizer only
        list(opts.values())[-1].load_state_dict(osd)
        for k, s in schs.items():  # LambdaLR: the factor is a pure function of the number of scheduler steps taken
            s.last_epoch = step
            for g, base, lam in zip(s.optimizer.param_groups, s.base_lrs, s.lr_lambdas):
                g["lr"] = base * lam(step)
            s._last_lr = [g["lr"] for g in s.optimizer.param_groups]
    if ema_helper is not None and states.get("ema_helper") is not None:
        dev = next(model.parameters()).device
        ema_helper.load_state_dict({k: v.to(dev) for k, v in states["ema_helper"].items()})
    return int(states["epoch"]), step
