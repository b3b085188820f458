"""``noise_estimation_loss`` (reference ``functions/losses.py:4-18``) on libddimx kernels.

q-sample, model forward and the squared-error reduction run in HIP.  In training mode with autograd enabled the model
call is an autograd node (``ddim_audio_amd.model._UNetTrainFn``) and the reduction gets its hand-written backward
(``ddimx_sqerr_loss_bwd``), so ``loss.backward()`` works as in the reference runner (``runners/diffusion.py:143-150``).
"""
import torch

from . import _lib


class _SqErrFn(torch.autograd.Function):
    """per-sample sum of (e - out)^2 over (1, 2, 3): returns [B + 1] = per-sample losses and their batch mean."""

    @staticmethod
    def forward(ctx, out, e):
        lib = _lib.load()
        b = out.size(0)
        per = out.numel() // b
        partial = torch.empty(b * 64, dtype=torch.float32, device=out.device)
        loss = torch.empty(b + 1, dtype=torch.float32, device=out.device)
        _lib.check(lib.ddimx_sqerr_loss(_lib.ptr(e), _lib.ptr(out), _lib.ptr(partial), _lib.ptr(loss), b, per, _lib.stream()))
        ctx.save_for_backward(out, e)
        return loss

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        out, e = ctx.saved_tensors
        b = out.size(0)
        per = out.numel() // b
        # the [B] entry is the batch mean (functions/losses.py:18): the kernel folds its upstream gradient into the per-sample ones
        gc = g.contiguous()  # (what autograd hands over already is: no launch)
        d = torch.empty_like(out)
        with torch.cuda.device(out.device):
            _lib.check(lib.ddimx_sqerr_loss_bwd_mean(_lib.ptr(e), _lib.ptr(out), _lib.ptr(gc), _lib.ptr(d), b, per, _lib.stream()))
        return d, None


def noise_estimation_loss(model, x0, t, e, a, keepdim=False):
    lib = _lib.load()
    if not x0.is_cuda:
        raise RuntimeError("noise_estimation_loss runs only on a ROCm GPU (no CPU fallback)")
    with torch.cuda.device(x0.device):
        with torch.no_grad():
            x0c, ec = x0.float().contiguous(), e.float().contiguous()
            ac = a.to(x0.device, torch.float32).contiguous()
            tc = t.to(x0.device, torch.int64).contiguous()
            b = x0c.size(0)
            per = x0c.numel() // b
            x = torch.empty_like(x0c)
            _lib.check(lib.ddimx_qsample(_lib.ptr(x0c), _lib.ptr(ec), _lib.ptr(ac), _lib.ptr(tc), _lib.ptr(x), b, per, _lib.stream()))
        out = model(x, tc).contiguous()
        loss = _SqErrFn.apply(out, ec)
    return loss[:b] if keepdim else loss[b]


loss_registry = {"simple": noise_estimation_loss}
