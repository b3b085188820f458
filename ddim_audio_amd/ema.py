"""``EMAHelper`` (reference ``models/ema.py:4-51``) with a single multi-tensor HIP launch per update."""
import torch
import torch.nn as nn

from . import _lib


class EMAHelper(object):
    def __init__(self, mu=0.999):
        self.mu = mu
        self.shadow = {}
        self._tables = None

    @staticmethod
    def _unwrap(module):
        return module.module if isinstance(module, nn.DataParallel) else module

    def register(self, module):
        module = self._unwrap(module)
        for name, param in module.named_parameters():
            if param.requires_grad:
                self.shadow[name] = param.data.clone()
        self._tables = None

    def _build_tables(self, pairs, device):
        blk = _lib.load().ddimx_ema_block_elems()
        sh_ptrs, p_ptrs, sizes, blk_t, blk_o = [], [], [], [], []
        for i, (s, p) in enumerate(pairs):
            sh_ptrs.append(s.data_ptr()); p_ptrs.append(p.data_ptr()); sizes.append(p.numel())
            for off in range(0, p.numel(), blk):
                blk_t.append(i); blk_o.append(off)
        mk = lambda v, dt: torch.tensor(v, dtype=dt, device=device)  # noqa: E731
        return {"key": tuple(sh_ptrs + p_ptrs), "sh": mk(sh_ptrs, torch.int64), "p": mk(p_ptrs, torch.int64),
                "n": mk(sizes, torch.int64), "bt": mk(blk_t, torch.int32), "bo": mk(blk_o, torch.int64), "nblk": len(blk_t)}

    def update(self, module):
        """shadow = (1 - mu) * param + mu * shadow for every trainable parameter (models/ema.py:16-23)."""
        module = self._unwrap(module)
        pairs = []
        for name, param in module.named_parameters():
            if param.requires_grad:
                s, p = self.shadow[name], param.data
                if not (p.is_cuda and s.is_cuda and p.dtype == s.dtype == torch.float32 and p.is_contiguous() and s.is_contiguous()):
                    raise RuntimeError(f"EMAHelper.update: {name} must be contiguous fp32 on the GPU (no CPU fallback)")
                pairs.append((s, p))
        if not pairs:
            return
        dev = pairs[0][1].device
        key = tuple([s.data_ptr() for s, _ in pairs] + [p.data_ptr() for _, p in pairs])
        if self._tables is None or self._tables["key"] != key:
            self._tables = self._build_tables(pairs, dev)
        tb = self._tables
        with torch.cuda.device(dev):
            _lib.check(_lib.load().ddimx_ema_update_multi(_lib.ptr(tb["sh"]), _lib.ptr(tb["p"]), _lib.ptr(tb["n"]), _lib.ptr(tb["bt"]),
                                                          _lib.ptr(tb["bo"]), tb["nblk"], float(self.mu), _lib.stream()))

    def ema(self, module):
        module = self._unwrap(module)
        for name, param in module.named_parameters():
            if param.requires_grad:
                param.data.copy_(self.shadow[name].data)
        if hasattr(module, "invalidate"):
            module.invalidate()  # packed weights must be rebuilt from the swapped-in values

    def ema_copy(self, module):
        inner = self._unwrap(module)
        """A second model instance holding the EMA weights (reference models/ema.py:32-45; unused by its runner)."""
        copy = type(inner)(inner._full_config) if hasattr(inner, "_full_config") else None
        if copy is None:
            raise NotImplementedError("ema_copy needs the full config; it is unused by the reference runner")
        copy.to(next(inner.parameters()).device)
        copy.load_state_dict(inner.state_dict())
        self.ema(copy)
        return copy

    def state_dict(self):
        return self.shadow

    def load_state_dict(self, state_dict):
        self.shadow = state_dict
        self._tables = None
