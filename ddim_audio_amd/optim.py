"""Optimizer tail of the training step (reference ``functions/__init__.py:5-60``, ``runners/diffusion.py:155-173``).

``get_optimizer`` returns fused multi-tensor optimizers for ``Adam`` / ``AdamW`` (one libddimx launch per group
instead of ~10 tiny launches per tensor); ``clip_grad_norm_`` is the fused counterpart of
``torch.nn.utils.clip_grad_norm_`` (global L2 norm + in-place scaling, the coefficient never leaves the device).
``AdaBelief`` lives in an un-vendored submodule of the reference (``External/step-clip-optimizer``, source absent):
it is implemented from the published algorithm (Zhuang et al. 2020) with the flags the reference passes
(``weight_decouple=True, fixed_decay=False, rectify=False``); ``clip_step`` other than ``None`` raises.  PARITY UNPINNED
for this optimizer: it is checked against a restatement of the paper's update only.  RMSProp / SGD pass through to
``torch.optim``.  CPU tensors are rejected: like the rest of the package there is no CPU fallback.
"""
import torch
import torch.optim as optim
from torch.optim.lr_scheduler import LambdaLR

from . import _lib


def lr_factor(step, warmup):
    """``min(((1+s)/w)^-0.5, (1+s)/w)`` -- reference ``functions/__init__.py:55-59``."""
    return min(((1 + step) / warmup) ** -0.5, (1 + step) / warmup)


class _Tables:
    """Device pointer / size / block tables over a list of equally-shaped tensor lists (cached per pointer set)."""

    def __init__(self, lists, device):
        blk = _lib.load().ddimx_ema_block_elems()
        self.key = tuple(t.data_ptr() for ts in lists for t in ts)
        sizes, bt, bo = [], [], []
        for i, t in enumerate(lists[0]):
            sizes.append(t.numel())
            for off in range(0, t.numel(), blk):
                bt.append(i); bo.append(off)
        mk = lambda v, dt: torch.tensor(v, dtype=dt, device=device)  # noqa: E731
        self.ptrs = [mk([t.data_ptr() for t in ts], torch.int64) for ts in lists]
        self.sizes, self.bt, self.bo, self.nblk = mk(sizes, torch.int64), mk(bt, torch.int32), mk(bo, torch.int64), len(bt)


def _check(ts, what):
    for t in ts:
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError(f"{what}: tensors must be contiguous fp32 on the GPU (no CPU fallback)")


_clip_cache = {}


def clip_grad_norm_(parameters, max_norm, norm_type=2.0):
    """Fused ``torch.nn.utils.clip_grad_norm_`` (L2 only): returns the total norm as a 0-dim device tensor."""
    if float(norm_type) != 2.0:
        raise NotImplementedError("only the L2 norm is implemented")
    grads = [p.grad for p in ([parameters] if isinstance(parameters, torch.Tensor) else parameters) if p.grad is not None]
    if not grads:
        return torch.tensor(0.0)
    _check(grads, "clip_grad_norm_")
    dev = grads[0].device
    key = tuple(g.data_ptr() for g in grads)
    tb = _clip_cache.get(key)
    if tb is None:
        _clip_cache.clear()
        tb = _clip_cache[key] = _Tables([grads], dev)
        tb.partial = torch.empty(tb.nblk, dtype=torch.float32, device=dev)
        tb.out = torch.empty(2, dtype=torch.float32, device=dev)
    lib = _lib.load()
    with torch.cuda.device(dev):
        _lib.check(lib.ddimx_grad_norm_multi(_lib.ptr(tb.ptrs[0]), _lib.ptr(tb.sizes), _lib.ptr(tb.bt), _lib.ptr(tb.bo), tb.nblk,
                                             float(max_norm), _lib.ptr(tb.partial), _lib.ptr(tb.out), _lib.stream()))
        _lib.check(lib.ddimx_scale_multi(_lib.ptr(tb.ptrs[0]), _lib.ptr(tb.sizes), _lib.ptr(tb.bt), _lib.ptr(tb.bo), tb.nblk,
                                         _lib.ptr(tb.out[1:]), _lib.stream()))
    return tb.out[0]


class FusedAdam(optim.Optimizer):
    """``torch.optim.Adam`` (``decoupled=False``) / ``AdamW`` (``decoupled=True``) semantics, amsgrad off, one
    multi-tensor HIP launch per parameter group.  ``param_groups[i]['lr']`` is honoured, so ``LambdaLR`` works."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, decoupled=True):
        """decoupled: False = Adam (L2 in the gradient), True = AdamW, 2 = AdaBelief (decoupled decay)."""
        if amsgrad:
            raise NotImplementedError("amsgrad is not implemented in the fused optimizer")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, decoupled=decoupled))
        self._tables = {}

    # -- checkpoint wire format: interchangeable with torch.optim.Adam / AdamW (runners/diffusion.py:187,245-249) ----------
    def state_dict(self):
        """Same layout as ``torch.optim.AdamW.state_dict()``: per-parameter ``step`` as a 0-dim fp32 tensor, no
        library-private keys in ``param_groups`` (``decoupled`` is a constructor property, like torch's class choice)."""
        sd = super().state_dict()
        sd["state"] = {k: {n: (torch.tensor(float(v)) if n == "step" else v) for n, v in st.items()} for k, st in sd["state"].items()}
        sd["param_groups"] = [{k: v for k, v in g.items() if k != "decoupled"} for g in sd["param_groups"]]
        return sd

    def load_state_dict(self, state_dict):
        """Accepts states written by this class and by ``torch.optim.Adam`` / ``AdamW`` (tensor ``step``, no ``decoupled`` key,
        extra torch keys such as ``foreach`` / ``capturable`` are carried along untouched)."""
        super().load_state_dict(state_dict)
        for g in self.param_groups:
            g.setdefault("decoupled", self.defaults["decoupled"])
            if g.get("amsgrad"):
                raise NotImplementedError("amsgrad is not implemented in the fused optimizer")
            g["betas"] = tuple(g["betas"])
        for st in self.state.values():
            if "step" in st:
                st["step"] = int(round(float(st["step"])))
            for n in ("exp_avg", "exp_avg_sq"):
                if n in st:
                    st[n] = st[n].float().contiguous()
        self._tables = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
            steps = {self.state[p]["step"] for p in ps}
            if len(steps) != 1:
                raise RuntimeError("FusedAdam: parameters of one group must share the step count")
            lists = [[p.data for p in ps], [p.grad for p in ps], [self.state[p]["exp_avg"] for p in ps],
                     [self.state[p]["exp_avg_sq"] for p in ps]]
            for ts in lists:
                _check(ts, "FusedAdam")
            dev = ps[0].device
            key = tuple(t.data_ptr() for ts in lists for t in ts)
            tb = self._tables.get(gi)
            if tb is None or tb.key != key:
                tb = self._tables[gi] = _Tables(lists, dev)
            b1, b2 = group["betas"]
            dyn = getattr(self, "dyn", None)
            if dyn is not None:
                # per-step scalars in device memory (train.GraphedTrainStep writes them before every replay)
                steps.pop()
                with torch.cuda.device(dev):
                    _lib.check(lib.ddimx_adam_multi_dyn(_lib.ptr(tb.ptrs[0]), _lib.ptr(tb.ptrs[1]), _lib.ptr(tb.ptrs[2]),
                                                        _lib.ptr(tb.ptrs[3]), _lib.ptr(tb.sizes), _lib.ptr(tb.bt), _lib.ptr(tb.bo), tb.nblk,
                                                        None, _lib.ptr(dyn[gi]), float(b1), float(b2), float(group["eps"]),
                                                        float(group["weight_decay"]), int(group["decoupled"]), _lib.stream()))
                torch.autograd.graph.increment_version(ps)
                continue
            with torch.cuda.device(dev):
                _lib.check(lib.ddimx_adam_multi(_lib.ptr(tb.ptrs[0]), _lib.ptr(tb.ptrs[1]), _lib.ptr(tb.ptrs[2]), _lib.ptr(tb.ptrs[3]),
                                                _lib.ptr(tb.sizes), _lib.ptr(tb.bt), _lib.ptr(tb.bo), tb.nblk, None, float(group["lr"]),
                                                float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                                                int(steps.pop()), int(group["decoupled"]), _lib.stream()))
            # the kernel wrote through raw pointers: bump the version counters so that caches keyed on them (the model's
            # packed weights) notice, exactly as an in-place torch op would
            torch.autograd.graph.increment_version(ps)
        return loss


def adam_step_scalars(group, step):
    """(lr, 1 - beta1^step, sqrt(1 - beta2^step)) as the fp32 values ``ddimx_adam_multi`` derives from its by-value arguments
    (betas pass through C floats before the double-precision ``pow``): what ``ddimx_adam_multi_dyn`` reads from device memory."""
    import numpy as np
    b1, b2 = (float(np.float32(b)) for b in group["betas"])
    return (float(np.float32(group["lr"])), float(np.float32(1.0 - b1 ** step)), float(np.float32((1.0 - b2 ** step) ** 0.5)))


def get_optimizer(config, parameters):
    if config.optimizer in ("Adam", "AdamW"):
        return FusedAdam(parameters, lr=config.lr, weight_decay=config.weight_decay, betas=config.beta, amsgrad=config.amsgrad,
                         eps=config.eps, decoupled=config.optimizer == "AdamW")
    if config.optimizer == "AdaBelief":
        if getattr(config, "clip_step", None) is not None:
            raise NotImplementedError("AdaBelief clip_step belongs to the reference's un-vendored step-clip-optimizer fork "
                                      "(source absent); only clip_step: null is implemented")
        return FusedAdam(parameters, lr=config.lr, weight_decay=config.weight_decay, betas=config.beta, amsgrad=config.amsgrad,
                         eps=config.eps, decoupled=2)
    if config.optimizer == "RMSProp":
        return optim.RMSprop(parameters, lr=config.lr, weight_decay=config.weight_decay)
    if config.optimizer == "SGD":
        return optim.SGD(parameters, lr=config.lr, momentum=0.9)
    raise NotImplementedError("Optimizer {} not understood.".format(config.optimizer))


def get_scheduler(config, optimizer):
    if config.warmup:
        return LambdaLR(optimizer, lambda step: lr_factor(step, config.warmup))
