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
from .schedule import ddim_coefficients, ddpm_coefficients


def _selected(select_index, index, n):
    return select_index is None or index in select_index or index - n in select_index


class DDIMStepper:
    """One sampling run's device state and its step function.

    ``step()`` = reference ``functions/denoising.py:22-43`` for one iteration: timestep fill, model
    forward, fused x0-prediction + x_{t-1} update, counter advance.  The scalars come from a device
    table indexed by a device counter, so after the first (eager) step the same launch sequence is
    captured once into a hipGraph and replayed for every later step.  (The two batch shards on two streams live inside the
    library call, ``ddimx_unet_fwd_forked``: the captured graph has two parallel branches.)

    Ownership.  The captured graph holds raw pointers into the model's packed weights, embedding table,
    DFT / positional tables and workspaces, into this object's ``xt`` / ``x0`` / ``eps`` / ``t`` / ``coef`` / ``counter``,
    and its capture recorded the fork / join events of ``ForkContext``.  All of these are referenced from HERE for as long as
    the graph exists (``_refs``, ``_ctx``), the graph is destroyed FIRST (``close``), and a replay is refused -- the step
    falls back to eager launches and re-captures -- when the model has re-allocated any of those buffers since the capture
    (``Model._gen``); a repack (new parameter values) is carried out in place before the replay.  Nothing is allocated on a
    side stream or inside the capture: the workspace is reserved and the eps buffer allocated on the launch stream before.
    """

    def __init__(self, model, xt, coef64, use_graph=True, noise_fn=None, slot=0, fork=True):
        self.graph = None          # first attribute: close() / __del__ must find it whatever else failed
        self._ctx = self._refs = None
        self.lib = _lib.load()
        self.model, self.xt = model, xt
        dev = xt.device
        self.coef = torch.from_numpy(np.ascontiguousarray(coef64, dtype=np.float32)).to(dev).contiguous()
        self.n_iter = self.coef.size(0)
        self.counter = torch.zeros(1, dtype=torch.int32, device=dev)
        self.t = torch.zeros(xt.size(0), dtype=torch.int64, device=dev)
        self.x0 = torch.empty_like(xt)
        self.noise_fn = noise_fn
        self.use_graph = (use_graph and noise_fn is None and os.environ.get("DDIMX_GRAPH", "1") != "0"
                          and not torch.cuda.is_current_stream_capturing())
        self.done = 0
        self.captures = 0
        self._capture_pending = self.use_graph
        self._gen = None
        self.native = hasattr(model, "forward_slot")  # ddim_audio_amd.Model; anything else is called as model(x, t)
        # workspace slot of the model this stepper computes in (steppers that run concurrently on different streams must not share
        # scratch memory) and whether its forward may fork into two batch shards itself
        self.slot, self.fork = slot, fork
        self.eps = torch.empty_like(xt) if self.native else None  # the forward writes here: no allocation per step

    def _prepare(self):
        """On the launch stream: weight packing (a no-op unless a parameter changed), tables, the workspace."""
        if self.native:
            dev, t_len = self.xt.device, self.xt.size(2)
            self.model.prepare(dev, t_len)
            self.model.reserve(dev, self.xt.size(0), t_len, self.slot)

    def _launch(self, noise):
        lib, st = self.lib, _lib.stream()
        xt, t, x0 = self.xt, self.t, self.x0
        _lib.check(lib.ddimx_step_begin(_lib.ptr(self.coef), _lib.ptr(self.counter), _lib.ptr(t), t.numel(), st))
        if self.native:
            # a graph stepper's EAGER launches (the sizing step in front of a capture, the fallback after its graph went stale) stay on
            # one stream: the two-shard fork is for the captured step (and for eager-only steppers) -- same bits either way, and no eager
            # two-stream launch right after an executable graph with parallel branches was destroyed
            fork = self.fork and (not self.use_graph or torch.cuda.is_current_stream_capturing())
            et = self.model(xt, t, _slot=self.slot, _fork=fork, _ctx=self._ctx, _out=self.eps)
        else:
            et = self.model(xt, t)
            if et.dtype != torch.float32 or not et.is_contiguous():
                et = et.float().contiguous()
        _lib.check(lib.ddimx_ddim_update(_lib.ptr(xt), _lib.ptr(et), _lib.ptr(noise), _lib.ptr(x0), _lib.ptr(self.coef),
                                         _lib.ptr(self.counter), xt.numel(), st))
        _lib.check(lib.ddimx_step_end(_lib.ptr(self.counter), st))

    def rewind(self):
        """Restart the coefficient table (benchmark loops longer than the schedule)."""
        self.counter.zero_()

    def _stale(self):
        """Before a replay.  ``_prepare`` runs the model's own staleness test -- (data_ptr, version) of every parameter, the
        same key the eager forward uses -- so an optimizer step, ``load_state_dict`` (also the plain nn.Module one), an in-place
        ``p.copy_()``, an EMA swap-in or ``invalidate()`` repack the weights and the embedding table IN PLACE on the launch
        stream: the graph stays valid and the replay sees the new values.  (Host work of a replay loop that is otherwise idle:
        the GPU step takes milliseconds.)  The graph is stale only if a buffer it points at was re-allocated since the capture
        (``Model._gen``: .to() / .type(), another T, a larger batch) or the model left eval mode."""
        if not self.native:
            return False
        m = self.model
        if m.training:
            return True
        self._prepare()
        return m._gen != self._gen

    def _drop_graph(self):
        """Destroy the graph, THEN release what its capture referenced (events, buffers)."""
        g, self.graph = self.graph, None
        if g is not None:
            torch.cuda.synchronize(self.xt.device)  # no replay in flight when the executable graph goes away
            del g
            torch.cuda.synchronize(self.xt.device)  # ... and the runtime has finished with it before its events / buffers go
        self._ctx = self._refs = None

    def close(self):
        self._drop_graph()

    def __del__(self):
        # a stepper that is simply dropped may still have its last replay in flight: the same order as close(), with the same
        # synchronisation (an executable graph destroyed under a running replay, then the events and buffers it references
        # freed, is a use-after-free inside the runtime's completion thread)
        try:
            self._drop_graph()
        except Exception:
            try:
                g, self.graph = self.graph, None
                del g                     # hipGraphExecDestroy first ...
                self._ctx = self._refs = None  # ... then the events its capture recorded and the buffers it points at
            except Exception:
                pass

    def _capture(self):
        dev = self.xt.device
        if self.native and self.fork and self.model.fork_mask and self.xt.size(0) >= 4:
            self._ctx = self.model.new_fork_context(dev)  # created (and first recorded) eagerly, owned here
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        # thread_local: only THIS thread's calls are checked against the capture -- other threads of the process (a collective
        # library's proxy / watchdog threads, a data loader pinning memory) may allocate or free while we capture
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            self._launch(None)
        self.graph = g
        self.captures += 1
        if self.native:
            self._refs = self.model.captured_refs()
            self._gen = self.model._gen

    def step(self):
        if self.graph is not None and self._stale():
            self._drop_graph()
            self._capture_pending = self.use_graph
        if self.graph is not None:
            self.graph.replay()
        else:
            self._prepare()
            self._launch(self.noise_fn(self.xt) if self.noise_fn is not None else None)
            if self._capture_pending and not (self.native and self.model.training):
                # this step ran eagerly (the first one also sized the model's workspaces); capture one generic step
                self._capture_pending = False
                self._capture()
        self.done += 1


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
        noise_fn = (lambda ref: torch.randn_like(ref)) if eta != 0.0 else None  # reference :42 draws it every step
        stepper = DDIMStepper(model, xt, coef, use_graph=(n_iter >= 4), noise_fn=noise_fn)
        try:
            for index in range(n_iter):
                stepper.step()
                if _selected(select_index, index, n_iter):
                    x0_preds.append(stepper.x0.to("cpu"))
                    xs.append(xt.to("cpu"))
        finally:
            stepper.close()  # graph first, then the events / buffers it referenced
    return xs, x0_preds


def ddpm_steps(x, seq, model, b, select_index, **kwargs):
    """Ancestral sampler of the reference (``functions/denoising.py:55-92``): same signature and return value
    (every iteration appends the clamped x0 prediction and the new sample as CPU tensors; ``select_index`` must be
    None like upstream).  ``b`` is the fp32 beta table.  The per-step update is one libddimx pass
    (``ddimx_ddpm_update``); the noise is drawn with ``torch.randn_like`` like the reference (``noise_fn`` kwarg:
    test hook returning the noise tensor for iteration k)."""
    if select_index is not None:
        raise NotImplementedError("Specifying select_index is not implemented in ddpm_steps.")
    lib = _lib.load()
    noise_fn = kwargs.get("noise_fn")
    seq = list(seq)
    device = None
    if isinstance(model, torch.nn.Module):
        p0 = next(model.parameters(), None)
        if p0 is not None and p0.is_cuda:
            device = p0.device
    if device is None:
        device = x.device if x.is_cuda else torch.device("cuda", torch.cuda.current_device())
    with torch.no_grad(), torch.cuda.device(device):
        xs, x0_preds = [x], []
        cur = x.to(device, torch.float32).contiguous().clone()
        nxt, x0buf = torch.empty_like(cur), torch.empty_like(cur)
        coef = torch.from_numpy(ddpm_coefficients(seq, b)).to(device).contiguous()
        counter = torch.zeros(1, dtype=torch.int32, device=device)
        t = torch.zeros(cur.size(0), dtype=torch.int64, device=device)
        for k in range(len(seq)):
            st = _lib.stream()
            _lib.check(lib.ddimx_step_begin_ex(_lib.ptr(coef), 7, _lib.ptr(counter), _lib.ptr(t), t.numel(), st))
            e = model(cur, t)
            if e.dtype != torch.float32 or not e.is_contiguous():
                e = e.float().contiguous()
            noise = (noise_fn(k, cur) if noise_fn is not None else torch.randn_like(cur)).to(device, torch.float32).contiguous()
            _lib.check(lib.ddimx_ddpm_update(_lib.ptr(cur), _lib.ptr(e), _lib.ptr(noise), _lib.ptr(x0buf), _lib.ptr(nxt),
                                             _lib.ptr(coef), _lib.ptr(counter), cur.numel(), st))
            _lib.check(lib.ddimx_step_end(_lib.ptr(counter), st))
            x0_preds.append(x0buf.to("cpu"))
            xs.append(nxt.to("cpu"))
            cur, nxt = nxt, cur
    return xs, x0_preds
