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


def train_step(model, x, state, alphas, e=None, t=None, _assign_grads=False):
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
    if _assign_grads:
        # (GraphedTrainStep) the same gradients, taken with respect to the fresh leaf aliases the forward differentiated
        # (model._alias_leaves) and assigned to the parameters: no AccumulateGrad node of an earlier step is involved
        params = [p for _, p in model.named_parameters()]
        with torch.autograd.set_multithreading_enabled(False):  # (the capture stays in the capturing thread)
            grads = torch.autograd.grad(loss, model._leaf_aliases, allow_unused=True)
        for p, g in zip(params, grads):
            p.grad = g
        model._leaf_aliases = None
    else:
        # the backward is ONE autograd node (model._UNetTrainFn) that launches on two streams and records events: it runs in the
        # calling thread, not in the engine's device thread (nothing to parallelise, and no cross-thread stream state)
        with torch.autograd.set_multithreading_enabled(False):
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


class GraphedTrainStep:
    """``train_step`` replayed from one hipGraph (single rank): forward, loss, backward, clipping, optimizer, EMA -- about
    1 500 launches -- are captured once and replayed per step, which removes the host's launch cost where it matters (small
    batches; at 32 samples per GPU the step is GPU-bound either way).

    What changes from step to step is kept out of the graph's frozen kernel arguments: the batch, the noise and the timesteps
    live in static buffers; each Adam group's (lr, 1 - beta1^step, sqrt(1 - beta2^step)) and the dropout call counter are
    device scalars written before every replay (``ddimx_adam_multi_dyn``, ``ddimx_set_dropout_counter``), computed on the host
    exactly as the eager step computes them -- a replayed step is bit-identical to an eager one
    (tests/test_gpu_train.py::test_graphed_train_step_is_bit_identical_to_eager).  The Python-side state (optimizer step
    counts, LambdaLR, ``model._dropout_calls``) is advanced per replay, so checkpoints and a later switch back to eager
    stepping see what an eager run would have left.

    The first ``warmup`` calls run eagerly (they size every workspace and build the optimizer's pointer tables); the next call
    captures.  Returns ``(loss, norms)`` as device tensors that the NEXT call overwrites.  Only FusedAdam groups (Adam / AdamW /
    AdaBelief) and a fixed batch shape are supported; data-parallel runs keep the eager step (its all-reduce is staged on
    events of a side stream)."""

    def __init__(self, model, state, alphas, warmup=2):
        self.model, self.state, self.alphas = model, state, alphas
        self.graph = self._refs = None
        self.warmup, self.calls = max(1, int(warmup)), 0
        self._side = torch.cuda.Stream()
        for o in state.optimizers.values():
            if not isinstance(o, optim.FusedAdam):
                raise NotImplementedError("GraphedTrainStep needs FusedAdam optimizer groups (Adam / AdamW / AdaBelief)")
        import torch.distributed as tdist
        if tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1:
            raise NotImplementedError("GraphedTrainStep is single-rank (the data-parallel step overlaps its all-reduce eagerly)")

    # ---- per-step device scalars ------------------------------------------------------------------------------
    def _groups(self):
        return [(o, gi, g) for o in self.state.optimizers.values() for gi, g in enumerate(o.param_groups)]

    def _write_scalars(self):
        """Values of the step about to run: every group's Adam scalars (step count + 1) and the dropout call counter + 1.
        Staged through a RING of pinned host slots, each guarded by a HIP event recorded behind its upload: a slot is
        overwritten only after the upload that last read it has executed.  (One pinned buffer re-used every call would let
        step k's asynchronous upload carry step k+1's values -- the host runs ahead of a ~50 ms replay)"""
        vals = []
        for o, gi, g in self._groups():
            ps = [p for p in g["params"] if p.grad is not None or p in o.state]
            step = (o.state[ps[0]]["step"] if ps and o.state[ps[0]] else 0) + 1
            vals += list(optim.adam_step_scalars(g, step)) + [0.0]
        k = self._slot = (self._slot + 1) % len(self._h_dyn)
        self._h_ev[k].synchronize()  # the upload that used this slot _RING calls ago (no-op when never recorded)
        self._h_dyn[k].copy_(torch.tensor(vals, dtype=torch.float32))
        self._h_ctr[k][0] = int(getattr(self.model, "_dropout_calls", 0)) + 1
        self._d_dyn.copy_(self._h_dyn[k], non_blocking=True)
        self._d_ctr.copy_(self._h_ctr[k], non_blocking=True)
        self._h_ev[k].record()

    def _advance_host_state(self):
        """What the captured Python code did to host-side state during capture, once per replay."""
        for o in self.state.optimizers.values():
            for g in o.param_groups:
                for p in g["params"]:
                    if o.state[p]:
                        o.state[p]["step"] += 1
            o._step_count = getattr(o, "_step_count", 0) + 1  # what LambdaLR's call-order check looks at
        for s in self.state.schedulers.values():
            s.step()
        self.model._dropout_calls = int(getattr(self.model, "_dropout_calls", 0)) + 1
        if hasattr(self.model, "invalidate"):
            self.model.invalidate()  # the replay updated the parameters through raw pointers

    def _snapshot(self):
        snap = {"steps": [{p: o.state[p]["step"] for g in o.param_groups for p in g["params"] if o.state[p]}
                          for o in self.state.optimizers.values()],
                "lrs": [[g["lr"] for g in o.param_groups] for o in self.state.optimizers.values()],
                "sched": {k: s.state_dict() for k, s in self.state.schedulers.items()},
                "calls": int(getattr(self.model, "_dropout_calls", 0))}
        return snap

    def _restore(self, snap):
        for o, steps, lrs in zip(self.state.optimizers.values(), snap["steps"], snap["lrs"]):
            for p, v in steps.items():
                o.state[p]["step"] = v
            for g, lr in zip(o.param_groups, lrs):
                g["lr"] = lr
        for k, s in self.state.schedulers.items():
            s.load_state_dict(snap["sched"][k])
        self.model._dropout_calls = snap["calls"]

    # ---- the step ------------------------------------------------------------------------------------------------
    _RING = 4

    def _capture(self, x):
        from . import _lib
        dev = x.device
        ng = len(self._groups())
        self._h_dyn = [torch.empty(4 * ng, dtype=torch.float32).pin_memory() for _ in range(self._RING)]
        self._h_ctr = [torch.zeros(1, dtype=torch.int64).pin_memory() for _ in range(self._RING)]
        self._h_ev = [torch.cuda.Event() for _ in range(self._RING)]
        self._slot = -1
        self._d_dyn = torch.zeros(4 * ng, dtype=torch.float32, device=dev)
        self._d_ctr = torch.zeros(1, dtype=torch.int64, device=dev)
        self.x = torch.empty_like(x)
        self.e = torch.empty_like(x)
        self.t = torch.zeros(x.size(0), dtype=torch.int64, device=dev)
        self.alphas = self.alphas.to(dev)
        snap = self._snapshot()
        if hasattr(self.model, "_ensure_tables"):
            # host-built tables (posenc, DFT) must exist before the capture: building them is a host-to-device copy.  (Nothing else is
            # prepared here: the weight re-pack must stay INSIDE the captured forward, every replay follows an optimizer step.)
            self.model._ensure_tables(x.size(2), dev)
        # the captured backward forks its weight gradients onto a second stream (ddimx_unet_bwd_forked): this capture's own event set
        self._bwd_ctx = None
        # -- only on request (DDIMX_CAPTURE_FORK=1): measured on this ROCm, a process that DESTROYS a captured training graph with the
        # second-stream branch in it and then goes on launching eagerly on two streams is killed by an abort / segfault inside a runtime
        # thread in 5 of 12 runs of tests/test_gpu_configs.py, against 0 of 8 with the captured backward on one stream
        # (tools/dbg/bisect_graphed.sh); the eager step keeps its branch, the replayed step gives up 1.2 ms of 49
        import os
        if (getattr(self.model, "bwd_fork", False) and hasattr(self.model, "new_bwd_fork_context")
                and os.environ.get("DDIMX_CAPTURE_FORK", "0") == "1"):
            self._bwd_ctx = self.model.new_bwd_fork_context(dev)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        # The device-scalar hooks exist only while the capture runs: the optimizers read (lr, bias corrections) from _d_dyn and
        # the forward adds the device call counter to its dropout seed.  They are cleared again before anything else can run,
        # so an eager train_step / optimizer.step() on the same model and state between replays takes the by-value path with
        # the host-side counts that every replay keeps advancing (_advance_host_state) -- never a stale device scalar.
        k = 0
        for o in self.state.optimizers.values():
            o.dyn = [self._d_dyn[4 * (k + gi):4 * (k + gi) + 3] for gi in range(len(o.param_groups))]
            k += len(o.param_groups)
        self.model._dropout_ctr_dev = self._d_ctr
        _lib.check(_lib.load().ddimx_set_dropout_counter(self.model._handle, _lib.ptr(self._d_ctr)))
        self.model._alias_leaves = True
        self.model._capture_bwd_ctx = self._bwd_ctx
        try:
            with torch.cuda.graph(g, stream=self._side):
                self.loss, self.norms = train_step(self.model, self.x, self.state, self.alphas, e=self.e, t=self.t, _assign_grads=True)
        finally:
            self.model._alias_leaves = False
            self.model._capture_bwd_ctx = None
            for o in self.state.optimizers.values():
                o.dyn = None
            self.model._dropout_ctr_dev = None
            _lib.check(_lib.load().ddimx_set_dropout_counter(self.model._handle, None))
        self._restore(snap)  # capturing ran the host side of one step without executing it
        self.graph = g
        # what the graph points at, kept next to it: the flat gradient buffer and the training workspace are re-used by eager
        # steps, but must not be FREED (a batch-shape change re-allocates them) while this graph can still be replayed
        # ... and so are the posenc / DFT tables (Model._tables keeps only the latest T: an eval forward at another length would free
        # them), the eval workspaces and the embedding table (Model.captured_refs).  The model's buffer generation is recorded: a
        # replay after .to() / .type() / a re-allocation would run on pointers of an earlier generation
        self._refs = [getattr(self.model, n, None) for n in ("_flat_grad", "_train_ws", "_packed", "_packed_bwd")]
        if hasattr(self.model, "captured_refs"):
            self._refs += self.model.captured_refs()
        self._gen = getattr(self.model, "_gen", None)

    def close(self):
        """Back to eager stepping.  The graph goes first, then the buffers it points at."""
        g, self.graph = self.graph, None
        if g is not None:
            torch.cuda.synchronize()
            del g
            torch.cuda.synchronize()  # the runtime has finished with the graph before its events / buffers go
        self._refs = None
        self._bwd_ctx = None

    def __del__(self):
        # dropped without close(): the last replay may still be in flight -- same order, same synchronisation (sampler.DDIMStepper)
        try:
            self.close()
        except Exception:
            try:
                g, self.graph = self.graph, None
                del g
                self._refs = None
                self._bwd_ctx = None
            except Exception:
                pass

    def __call__(self, x, e=None, t=None):
        n = x.size(0)
        if e is None:
            e = torch.randn_like(x)
        if t is None:
            t = antithetic_timesteps(n, self.alphas.numel())
        self.calls += 1
        if self.graph is None and self.calls <= self.warmup:
            # warm-up on the stream the capture will use: autograd's AccumulateGrad nodes remember the stream they were created
            # on, and a capture must not be made to wait for another (non-capturing) stream
            cur = torch.cuda.current_stream(x.device)
            self._side.wait_stream(cur)
            with torch.cuda.stream(self._side):
                self.model._alias_leaves = True
                try:
                    out = train_step(self.model, x, self.state, self.alphas, e=e, t=t, _assign_grads=True)
                finally:
                    self.model._alias_leaves = False
            cur.wait_stream(self._side)
            return out
        if self.graph is not None and getattr(self.model, "_gen", None) != self._gen:
            # the model re-allocated a buffer the graph points at (another T through an eval forward, .to() / .type(), a repack into
            # a new buffer): drop the graph (its buffers stay alive in _refs until it is gone) and capture again on this call
            self.close()
        if self.graph is None:
            self._capture(x)
        if x.shape != self.x.shape:
            raise RuntimeError(f"GraphedTrainStep was captured for batches of shape {tuple(self.x.shape)}, got {tuple(x.shape)}")
        self.x.copy_(x, non_blocking=True)
        self.e.copy_(e, non_blocking=True)
        self.t.copy_(t.to(torch.int64), non_blocking=True)
        self._write_scalars()
        self.graph.replay()
        self._advance_host_state()
        return self.loss, self.norms
