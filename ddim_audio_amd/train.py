"""The training step of the reference runner (``runners/diffusion.py:64-87,130-173,213-238``) on libddimx kernels.

Only the step itself is mirrored -- parameter grouping, noise / antithetic timestep sampling, loss, backward, per-group
gradient clipping, optimizer + LR-scheduler steps, EMA.  Dataset iteration, logging and checkpoint cadence stay with the
caller (out of scope: SURVEY section 8).  Every tensor operation is a HIP kernel of libddimx; nothing syncs with the host
(the reference calls ``loss.item()`` twice per step, ``:146-150``): the loss is returned as a device scalar.
"""
import torch

from . import losses, optim
from .configs import dict2namespace
from .ema import EMAHelper


class parameter_option:
    """Same shape as the reference's helper (``runners/diffusion.py:64-67``)."""

    def __init__(self):
        self.config = {}
        self.params = []


def classify_group(config, model):
    """Route parameters to config groups by top-level module name (``runners/diffusion.py:71-87``); unlike the
    reference this does not pop ``top_level_name`` out of the caller's config."""
    top, groups = {}, {}
    for gname, sub in vars(config).items():
        sub = dict(vars(sub))
        for n in sub.pop("top_level_name"):
            top[n] = gname
        groups[gname] = parameter_option()
        groups[gname].config = dict2namespace(sub)
    for name, p in model.named_parameters():
        groups[top.get(name.split(".")[0], "default")].params.append(p)
    return {k: v for k, v in groups.items() if v.params}


class TrainingState:
    """Optimizers, schedulers, clip groups and EMA for one model (``runners/diffusion.py:217-238``)."""

    def __init__(self, config, model):
        self.config = config
        self.optimizers, self.schedulers = {}, {}
        for name, g in classify_group(config.optimization.optimizer, model).items():
            self.optimizers[name] = optim.get_optimizer(g.config, g.params)
            sch = optim.get_scheduler(g.config, self.optimizers[name])
            if sch:
                self.schedulers[name] = sch
        self.grad_group = classify_group(config.optimization.grad_norm, model)
        self.ema_helper = None
        if config.model.ema:
            self.ema_helper = EMAHelper(mu=config.model.ema_rate)
            self.ema_helper.register(model)


def antithetic_timesteps(n, num_timesteps, generator=None):
    """``runners/diffusion.py:141-142``: CPU draw of ceil(n/2) timesteps, mirrored, truncated to n."""
    t = torch.randint(low=0, high=num_timesteps, size=((n + 1) // 2,), generator=generator)
    return torch.cat([t, num_timesteps - t - 1], dim=0)[:n]


def train_step(model, x, state, alphas, e=None, t=None):
    """One optimisation step (``Diffusion.train_step``, ``runners/diffusion.py:130-173``).  ``x``: [B, C, T, F] on the GPU;
    ``e`` / ``t`` default to fresh noise / antithetic timesteps.  Returns (loss, {clip group: total grad norm}) as device
    tensors."""
    model.train()
    n = x.size(0)
    if e is None:
        e = torch.randn_like(x)
    if t is None:
        t = antithetic_timesteps(n, alphas.numel())
    t = t.to(x.device)
    loss = losses.loss_registry[state.config.model.type](model, x, t, e, alphas)
    for o in state.optimizers.values():
        o.zero_grad()
    loss.backward()
    norms = {}
    for name, g in state.grad_group.items():
        if g.config.grad_clip is not None:
            norms[name] = optim.clip_grad_norm_(g.params, g.config.grad_clip)
    for o in state.optimizers.values():
        o.step()
    for s in state.schedulers.values():
        s.step()
    if hasattr(model, "invalidate"):
        model.invalidate()  # the fused optimizers write through raw pointers: packed weights must be rebuilt
    if state.ema_helper is not None:
        state.ema_helper.update(model)
    return loss.detach(), norms
