"""Optimizer / LR-schedule factories of the training-step tail (reference ``functions/__init__.py:5-60``).

Round-1 status: thin pass-through to ``torch.optim`` so the reference runner's call sites resolve;
the fused multi-tensor HIP optimizer (and AdaBelief, whose source is an un-vendored submodule of the
reference) is SURVEY section 8f row 1 and not built yet.
"""
import torch.optim as optim
from torch.optim.lr_scheduler import LambdaLR


def lr_factor(step, warmup):
    """``min(((1+s)/w)^-0.5, (1+s)/w)`` -- reference ``functions/__init__.py:55-59``."""
    return min(((1 + step) / warmup) ** -0.5, (1 + step) / warmup)


def get_optimizer(config, parameters):
    kw = dict(lr=config.lr, weight_decay=config.weight_decay)
    if config.optimizer in ("Adam", "AdamW"):
        cls = optim.Adam if config.optimizer == "Adam" else optim.AdamW
        return cls(parameters, betas=tuple(config.beta), amsgrad=config.amsgrad, eps=config.eps, **kw)
    if config.optimizer == "RMSProp":
        return optim.RMSprop(parameters, **kw)
    if config.optimizer == "SGD":
        return optim.SGD(parameters, lr=config.lr, momentum=0.9)
    raise NotImplementedError("Optimizer {} not understood.".format(config.optimizer))


def get_scheduler(config, optimizer):
    if config.warmup:
        return LambdaLR(optimizer, lambda step: lr_factor(step, config.warmup))
