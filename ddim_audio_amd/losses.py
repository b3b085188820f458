"""``noise_estimation_loss`` (reference ``functions/losses.py:4-18``) on libddimx kernels.

Forward value only: q-sample, model forward and the squared-error reduction run in HIP; the
backward pass (training) is the next scope row and is not built yet, so the returned tensor carries
no autograd graph.
"""
import torch

from . import _lib


def noise_estimation_loss(model, x0, t, e, a, keepdim=False):
    lib = _lib.load()
    if not x0.is_cuda:
        raise RuntimeError("noise_estimation_loss runs only on a ROCm GPU (no CPU fallback)")
    with torch.no_grad(), torch.cuda.device(x0.device):
        x0c, ec = x0.float().contiguous(), e.float().contiguous()
        ac = a.to(x0.device, torch.float32).contiguous()
        tc = t.to(x0.device, torch.int64).contiguous()
        b = x0c.size(0)
        per = x0c.numel() // b
        x = torch.empty_like(x0c)
        _lib.check(lib.ddimx_qsample(_lib.ptr(x0c), _lib.ptr(ec), _lib.ptr(ac), _lib.ptr(tc), _lib.ptr(x), b, per, _lib.stream()))
        out = model(x, tc)
        partial = torch.empty(b * 64, dtype=torch.float32, device=x0.device)
        loss = torch.empty(b + 1, dtype=torch.float32, device=x0.device)
        _lib.check(lib.ddimx_sqerr_loss(_lib.ptr(ec), _lib.ptr(out.contiguous()), _lib.ptr(partial), _lib.ptr(loss), b, per,
                                        _lib.stream()))
    return loss[:b] if keepdim else loss[b]


loss_registry = {"simple": noise_estimation_loss}
