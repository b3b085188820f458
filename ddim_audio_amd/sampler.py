"""``generalized_steps`` -- the reference's DDIM / eta-generalised reverse loop on the HIP library.

Mirrors reference ``functions/denoising.py:10-52`` (same signature, same return value, same
in-place semantics: ``xs[0]`` is the caller's tensor and ``x`` is updated in place when it already
is a GPU float tensor).  Differences that are fixes, not behaviour changes:
the device follows the model / ``x`` instead of hard-coded ``torch.cuda.*Tensor`` strings; the
per-step scalars live in a device table indexed by a device step counter so the whole step
(timestep fill, U-Net, fused x0-prediction + x_{t-1} update) replays as one hipGraph; ``randn_like``
is only drawn when eta > 0.  All tensor arithmetic runs in libddimx kernels.
"""
import os

import numpy as np
import torch

from . import _lib
from .schedule import ddim_coefficients


def _selected(select_index, index, n):
    return select_index is None or index in select_index or index - n in select_index


class _StepState:
    """Device-side state of one sampling run: coefficient table, step counter, timestep vector."""

    def __init__(self, coef64, batch, device):
        self.coef = torch.from_numpy(coef64.astype(np.float32)).to(device).contiguous()
        self.step = torch.zeros(1, dtype=torch.int32, device=device)
        self.t = torch.zeros(batch, dtype=torch.int64, device=device)


def _one_step(lib, model, xt, x0buf, st, noise):
    _lib.check(lib.ddimx_step_begin(_lib.ptr(st.coef), _lib.ptr(st.step), _lib.ptr(st.t), st.t.numel(), _lib.stream()))
    et = model(xt, st.t)
    if et.dtype != torch.float32 or not et.is_contiguous():
        et = et.float().contiguous()
    _lib.check(lib.ddimx_ddim_update(_lib.ptr(xt), _lib.ptr(et), _lib.ptr(noise), _lib.ptr(x0buf), _lib.ptr(st.coef),
                                     _lib.ptr(st.step), xt.numel(), _lib.stream()))
    _lib.check(lib.ddimx_step_end(_lib.ptr(st.step), _lib.stream()))


def generalized_steps(x, seq, model, alpha, select_index, **kwargs):
    """x [B,C,T,F]; seq: increasing timesteps; alpha: fp32 alphas-cumprod table; returns (xs, x0_preds)
    as lists of CPU tensors for the selected iterations (``select_index`` semantics of the reference:
    ``None`` = all, else iteration indices, negative allowed)."""
    lib = _lib.load()
    eta = float(kwargs.get("eta", 0))
    seq = list(seq)
    n_iter = len(seq)
    device = None
    if isinstance(model, torch.nn.Module):
        p = next(model.parameters(), None)
        if p is not None and p.is_cuda:
            device = p.device
    if device is None:
        device = x.device if x.is_cuda else torch.device("cuda", torch.cuda.current_device())
    with torch.no_grad(), torch.cuda.device(device):
        xs = [x]
        x0_preds = []
        # reference :18  xt = x.type("torch.cuda.FloatTensor"): no copy when x already is one
        xt = x if (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()) else x.to(device, torch.float32).contiguous()
        if xt.numel() % 4:
            raise RuntimeError("sample tensor size must be a multiple of 4 elements")
        coef = ddim_coefficients(seq, alpha, eta)
        st = _StepState(coef, xt.size(0), device)
        x0buf = torch.empty_like(xt)
        use_graph = (eta == 0.0 and n_iter >= 4 and os.environ.get("DDIMX_GRAPH", "1") != "0"
                     and not torch.cuda.is_current_stream_capturing())
        graph = None
        for index in range(n_iter):
            noise = torch.randn_like(xt) if eta != 0.0 else None
            if graph is not None:
                graph.replay()
            else:
                _one_step(lib, model, xt, x0buf, st, noise)
                if use_graph and index == 0:
                    # step 0 ran eagerly (it also sized the model's workspace); capture one generic step
                    torch.cuda.synchronize(device)
                    graph = torch.cuda.CUDAGraph()
                    try:
                        with torch.cuda.graph(graph):
                            _one_step(lib, model, xt, x0buf, st, None)
                    except Exception:
                        graph = None
                        use_graph = False
                        raise
            if _selected(select_index, index, n_iter):
                x0_preds.append(x0buf.to("cpu"))
                xs.append(xt.to("cpu"))
    return xs, x0_preds
