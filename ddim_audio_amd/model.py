"""``Model(config)`` -- the reference's U-Net epsilon-predictor interface on the HIP library.

Mirrors reference ``models/diffusion.py:170-294``: same constructor argument (the full config
Namespace), same 388 parameter + 1 buffer names and shapes (so reference checkpoints load with
``strict=True`` and ``EMAHelper`` / ``classify_group`` see the names they expect), same
``forward(input[B,C,T,F], t[B] int64) -> [B,C,T,F]``.  The module holds parameters only; every
arithmetic op of ``forward`` runs in ``libddimx.so`` (``csrc/``) through the C ABI of
``include/ddimx.h``.  There is no CPU or eager-PyTorch fallback: a missing library, a CPU tensor or
an unsupported configuration raises.
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn

_POS_CH = 128  # reference models/diffusion.py:98
_EMB_CH = 512  # reference models/diffusion.py:99
_GROUPS = 8  # reference models/diffusion.py:19


def embedding_sizes(mcfg):
    """Per-block timestep-embedding widths, down blocks then up blocks (models/diffusion.py:178-184)."""
    e = [c for r, c in zip(mcfg.res, mcfg.ch) for _ in range(r)]
    return e + e[::-1]


def _rb(inv, p, c, k):
    for i in range(3):
        inv[f"{p}norm.{i}.weight"] = (c,)
        if i < 2:
            inv[f"{p}norm.{i}.bias"] = (c,)
    inv[f"{p}conv.0.weight"] = (c, c, k, k)
    inv[f"{p}conv.1.weight"] = (c, c, k, k)
    inv[f"{p}conv.1.bias"] = (c,)


def state_inventory(config):
    """name -> shape of the state_dict, in the reference's registration order."""
    m = config.model
    assert len(m.ch) == len(m.krn) == len(m.res)
    nlev = len(m.ch)
    inv = OrderedDict()
    emb = sum(embedding_sizes(m))
    inv["temb.te"] = (config.diffusion.num_diffusion_timesteps, _POS_CH)
    for i, (o, k) in enumerate(((_EMB_CH, _POS_CH), (_EMB_CH, _EMB_CH), (emb, _EMB_CH))):
        inv[f"temb.weight.{i}.weight"] = (o, k)
        inv[f"temb.weight.{i}.bias"] = (o,)
    inv["down_modules.0.weight"] = (m.ch[0], m.channels, 3, 3)
    inv["down_modules.0.bias"] = (m.ch[0],)
    for lvl in range(nlev):
        base = f"down_modules.{lvl + 1}."
        j = 0
        if lvl > 0:
            inv[base + "0.conv.weight"] = (m.ch[lvl], m.ch[lvl - 1], 4, 4)
            inv[base + "0.conv.bias"] = (m.ch[lvl],)
            j = 1
        for r in range(m.res[lvl]):
            _rb(inv, f"{base}{j + r}.", m.ch[lvl], m.krn[lvl])
    for k in range(nlev):
        lvl = nlev - 1 - k
        base = f"up_modules.{k}."
        for r in range(m.res[lvl]):
            _rb(inv, f"{base}{r}.", m.ch[lvl], m.krn[lvl])
        if lvl > 0:  # ConvTranspose2d weight layout [Cin, Cout, 4, 4]
            inv[f"{base}{m.res[lvl]}.conv.weight"] = (m.ch[lvl], m.ch[lvl - 1], 4, 4)
            inv[f"{base}{m.res[lvl]}.conv.bias"] = (m.ch[lvl - 1],)
    inv[f"up_modules.{nlev}.weight"] = (m.channels, m.ch[0], 3, 3)
    inv[f"up_modules.{nlev}.bias"] = (m.channels,)
    tr = m.transformers
    width = m.ch[-1] * (m.f_size // (2 ** (nlev - 1)))
    hid, inter = tr.kwargs.hidden_size, tr.kwargs.intermediate_size
    assert tr.channels == hid, "transformers.channels must equal kwargs.hidden_size"
    inv["transformer.embedding.LayerNorm.weight"] = (width,)
    inv["transformer.embedding.LayerNorm.bias"] = (width,)
    inv["transformer.embedding.projection.weight"] = (hid, width)
    inv["transformer.embedding.projection.bias"] = (hid,)
    for i in range(tr.kwargs.num_hidden_layers):
        p = f"transformer.encoder.layer.{i}."
        inv[p + "fourier.output.LayerNorm.weight"] = (hid,)
        inv[p + "fourier.output.LayerNorm.bias"] = (hid,)
        inv[p + "intermediate.dense.weight"] = (inter, hid)
        inv[p + "intermediate.dense.bias"] = (inter,)
        inv[p + "output.dense.weight"] = (hid, inter)
        inv[p + "output.dense.bias"] = (hid,)
        inv[p + "output.LayerNorm.weight"] = (hid,)
        inv[p + "output.LayerNorm.bias"] = (hid,)
    inv["transformer.compute_out.weight"] = (width, hid)
    inv["transformer.compute_out.bias"] = (width,)
    return inv


def timestep_table(n, ch=_POS_CH):
    """``temb.te``: interleaved sin/cos table (reference models/diffusion.py:81-102), built in fp32
    with the same op order as the reference so the buffer matches it to the last bit."""
    pos = torch.arange(n, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, ch, 2, dtype=torch.float32) * (-math.log(10000.0) / ch))
    te = torch.zeros(n, ch)
    te[:, 0::2] += torch.sin(pos * div)
    te[:, 1::2] += torch.cos(pos * div)
    return te


class _Node(nn.Module):
    """Parameter container; the tree of these reproduces the reference's dotted names."""


def _default_init_(name, p):
    last = name.rsplit(".", 1)[-1]
    is_norm = ".norm." in name or "LayerNorm" in name
    with torch.no_grad():
        if is_norm:
            if last == "bias" or ".norm.2." in name:
                p.zero_()  # norm[2].weight = 0: every block starts as the identity (models/diffusion.py:25)
            else:
                p.fill_(1.0)
            return
        if p.dim() == 4:
            # torch's fan_in convention is shape[1]*k*k for both Conv2d and ConvTranspose2d weights
            fan_in = p.shape[1] * p.shape[2] * p.shape[3]
        elif p.dim() == 2:
            fan_in = p.shape[1]
        else:
            fan_in = None
        if fan_in is not None:
            p.uniform_(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in))
            p._ddimx_fan_in = fan_in


def parse_tensor_type(s):
    """Legacy tensor-type string (configs/audio.yml:26,42) -> (device or None, dtype)."""
    if not s:
        return None, torch.float32
    table = {"FloatTensor": torch.float32, "BFloat16Tensor": torch.bfloat16}
    kind = s.rsplit(".", 1)[-1]
    if kind not in table:
        raise NotImplementedError(f"model dtype {s!r}: the HIP path implements FloatTensor and BFloat16Tensor")
    return ("cuda" if ".cuda." in s else "cpu"), table[kind]


def build_parameters(root, config):
    inv = state_inventory(config)
    fan = {}
    for name, shape in inv.items():
        parts = name.split(".")
        node = root
        for part in parts[:-1]:
            if part not in node._modules:
                node.add_module(part, _Node())
            node = node._modules[part]
        if name == "temb.te":
            node.register_buffer("te", timestep_table(*shape))
            continue
        p = nn.Parameter(torch.empty(shape, dtype=torch.float32))
        _default_init_(name, p)
        if parts[-1] == "weight" and hasattr(p, "_ddimx_fan_in"):
            fan[".".join(parts[:-1])] = p._ddimx_fan_in
        node.register_parameter(parts[-1], p)
    # biases of conv / linear layers: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) like torch's defaults
    with torch.no_grad():
        for name, p in root.named_parameters():
            owner, last = name.rsplit(".", 1)
            if last == "bias" and owner in fan:
                b = 1.0 / math.sqrt(fan[owner])
                p.uniform_(-b, b)
    return inv


# ---------------------------------------------------------------------------------------------------
def _posenc_table(s, width, c_last, fr):
    """Add_Encoding table of TransformerEmbedding (reference models/diffusion.py:81-92,131-140): built
    on the host with the reference's op order for a power-of-two-rounded length, sliced to S (any
    order of S works; the reference's inverted cache test does not -- SURVEY section 5), then permuted
    from the reference token order (c*Fr + f) to the library's NHWC order (f*C + c)."""
    size = 2 ** math.ceil(math.log2(s)) if s > 1 else 1
    pos = torch.arange(size, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, width, 2, dtype=torch.float32) * (-math.log(10000.0) / width))
    pe = torch.zeros(size, width)
    pe[:, 0::2] += torch.sin(pos * div)
    pe[:, 1::2] += torch.cos(pos * div)
    return pe[:s].reshape(s, c_last, fr).permute(0, 2, 1).reshape(s, width).contiguous()


def _dft_tables(n):
    """cos / sin of 2*pi*k*m/n in float64, rounded once to fp32 (argument reduced exactly mod n)."""
    import numpy as np
    k = np.arange(n, dtype=np.int64)
    ang = 2.0 * np.pi * ((k[:, None] * k[None, :]) % n).astype(np.float64) / n
    return np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)


class ForkContext:
    """Second stream + fork / join events of ``ddimx_unet_fwd_forked`` / ``ddimx_unet_bwd_forked`` for ONE owner: either a
    model's eager calls or one graph capture.  An event set is never shared between eager launches and a capture, nor between
    two captures (a HIP event last recorded inside a capture must not be re-recorded eagerly), and it must outlive every graph
    whose capture recorded it and die after that graph: so whoever captures owns its context and drops it after the graph
    (``DDIMStepper.close``).  The same holds for the STREAM: a capture's context (``private=True``) gets a HIP stream of its own,
    created through the runtime and never handed to torch's stream pool -- a stream that has been forked into a capture is not
    launched on eagerly afterwards (round 4: eager launches on such a stream, shared with the model's eager context, ended in
    aborts inside a runtime thread).  The events are created here -- eagerly, by a first record on the current stream; torch
    creates the hipEvent lazily and that must not happen inside a capture."""

    def __init__(self, device, n_events, private=False):
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("a ForkContext must be created before the capture that uses it")
        self.device = device
        self._owned = None
        with torch.cuda.device(device):
            if private:
                self.aux = None
                self._owned = self.aux_handle = _hip_stream_create()
            else:
                self.aux = torch.cuda.Stream(device=device)
                self.aux_handle = self.aux.cuda_stream
            self.events = [torch.cuda.Event() for _ in range(n_events)]
            for e in self.events:
                e.record()

    def event_array(self):
        import ctypes
        return (ctypes.c_void_p * len(self.events))(*[e.cuda_event for e in self.events])

    def __del__(self):
        h, self._owned = getattr(self, "_owned", None), None
        if h:
            try:
                _hip_stream_destroy(h)
            except Exception:
                pass


def _hip():
    import ctypes
    lib = getattr(_hip, "_lib", None)
    if lib is None:
        lib = _hip._lib = ctypes.CDLL("libamdhip64.so")  # already mapped by torch / libddimx.so
        lib.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
        lib.hipStreamDestroy.argtypes = [ctypes.c_void_p]
    return lib


def _hip_stream_create():
    import ctypes
    h = ctypes.c_void_p()
    err = _hip().hipStreamCreateWithFlags(ctypes.byref(h), 1)  # hipStreamNonBlocking
    if err != 0 or not h.value:
        raise RuntimeError("hipStreamCreateWithFlags failed: %d" % err)
    return h.value


def _hip_stream_destroy(h):
    import ctypes
    _hip().hipStreamDestroy(ctypes.c_void_p(h))


class _UNetTrainFn(torch.autograd.Function):
    """Autograd node of the training-mode forward: ddimx_unet_fwd_train keeps a tape, backward = ddimx_unet_bwd.
    Gradients of all parameters land in ONE fresh fp32 buffer (views are handed to autograd), which is also the buffer a
    data-parallel run all-reduces (``model.grad_sync``, see ddim_audio_amd/dist.py)."""

    @staticmethod
    def forward(ctx, model, x, t, tables, *params):
        from . import _lib
        import ctypes
        lib = _lib.load()
        b, t_len = x.size(0), x.size(2)
        dev = x.device
        need = int(lib.ddimx_train_workspace_bytes(model._handle, b, t_len))
        ws = getattr(model, "_train_ws", None)
        if ws is None or ws.numel() < need or ws.device != dev:
            model._train_ws = None
            ws = model._train_ws = torch.empty(need, dtype=torch.uint8, device=dev)
        tape = torch.empty(int(lib.ddimx_train_tape_bytes(model._handle, b, t_len)), dtype=torch.uint8, device=dev)
        out = torch.empty_like(x)
        p = float(getattr(model.config.transformers.kwargs, "hidden_dropout_prob", 0.0))
        model._dropout_calls = getattr(model, "_dropout_calls", 0) + 1
        # one mask stream per (process seed, data-parallel rank, forward call): ranks with the same manual seed must not
        # draw the same masks; ``_dropout_calls`` travels with resume checkpoints (checkpoint.save_checkpoint)
        import torch.distributed as tdist
        rank = tdist.get_rank() if tdist.is_available() and tdist.is_initialized() else 0
        # (under train.GraphedTrainStep the call counter is added on the device instead: ddimx_set_dropout_counter)
        calls = 0 if getattr(model, "_dropout_ctr_dev", None) is not None else model._dropout_calls
        seed = (torch.initial_seed() * 0x9E3779B1 + (rank * 0xC2B2AE3D27D4EB4F) + calls) & 0xFFFFFFFFFFFFFFFF
        tb = _lib.DdimxTables(tables[0].data_ptr(), tables[1].data_ptr(), tables[2].data_ptr())
        _lib.check(lib.ddimx_unet_fwd_train(model._handle, _lib.ptr(model._packed), ctypes.byref(tb), _lib.ptr(ws), ws.numel(),
                                            _lib.ptr(tape), tape.numel(), _lib.ptr(x), _lib.ptr(t), _lib.ptr(out), b, t_len, p, seed,
                                            _lib.stream()))
        ctx.model, ctx.tape, ctx.x, ctx.t, ctx.tables, ctx.p, ctx.seed = model, tape, x, t, tables, p, seed
        ctx.params = params
        ctx.packed, ctx.packed_bwd = model._packed, model._packed_bwd  # keep the buffers this forward used alive
        return out

    @staticmethod
    def backward(ctx, d_eps):
        from . import _lib
        import ctypes
        lib = _lib.load()
        model, x = ctx.model, ctx.x
        if ctx.tape is None:
            raise RuntimeError("the tape of this forward was already consumed: backward through the same Model.forward twice "
                               "(retain_graph) is not supported")
        b, t_len = x.size(0), x.size(2)
        total, layout = model._grad_layout(lib)
        with torch.cuda.device(x.device):
            # One flat gradient buffer, kept across steps while no parameter holds a gradient (the usual
            # zero_grad(set_to_none=True) loop): the .grad tensors are views of it, like DDP's gradient_as_bucket_view,
            # so the fused optimizer's pointer tables stay valid.  If gradients are being accumulated a fresh buffer is used.
            flat = getattr(model, "_flat_grad", None)
            if flat is None or flat.numel() != total or flat.device != x.device or any(q.grad is not None for q in ctx.params):
                flat = torch.empty(total, dtype=torch.float32, device=x.device)
                if all(q.grad is None for q in ctx.params):
                    model._flat_grad = flat
            ws = model._train_ws
            tb = _lib.DdimxTables(ctx.tables[0].data_ptr(), ctx.tables[1].data_ptr(), ctx.tables[2].data_ptr())
            g = d_eps.contiguous()
            sync = getattr(model, "grad_sync", None)
            staged = getattr(sync, "staged", None) if sync is not None else None
            # the weight gradients run on a second stream beside the data-gradient chain (ddimx_unet_bwd_forked; bit-identical
            # results).  Eager calls use the model's own event set; a capture uses the set of whoever captures (never shared)
            fc = None
            if model.bwd_fork:
                fc = model._capture_bwd_ctx if torch.cuda.is_current_stream_capturing() else model._eager_bwd_context(x.device)
            side = (ctypes.c_void_p(fc.aux_handle), fc.event_array(), len(fc.events)) if fc is not None else (None, None, 0)
            if staged is not None and sync.active():
                # data parallel with overlap: the backward records an event per gradient bucket (up path, bottleneck, the rest)
                # and each bucket's all-reduce is issued on a side stream as soon as its event has fired, while the remaining
                # backward kernels keep running on this stream
                evs = getattr(model, "_bucket_events", None)
                if evs is None:
                    evs = [torch.cuda.Event() for _ in range(3)]
                    for e in evs:
                        e.record()  # torch creates the hipEvent lazily: force it, the library re-records it
                    model._bucket_events = evs
                rng = (ctypes.c_longlong * 6)()
                _lib.check(lib.ddimx_grad_buckets(model._handle, rng))
                arr = (ctypes.c_void_p * 3)(*[e.cuda_event for e in evs])
                _lib.check(lib.ddimx_unet_bwd_forked(model._handle, _lib.ptr(ctx.packed), _lib.ptr(ctx.packed_bwd), ctypes.byref(tb),
                                                     _lib.ptr(ws), ws.numel(), _lib.ptr(ctx.tape), ctx.tape.numel(), _lib.ptr(x),
                                                     _lib.ptr(ctx.t), _lib.ptr(g), _lib.ptr(flat), b, t_len, ctx.p, ctx.seed, arr, 3,
                                                     _lib.stream(), *side))
                staged(flat, [(rng[2 * i], rng[2 * i + 1]) for i in range(3)], evs)
            else:
                _lib.check(lib.ddimx_unet_bwd_forked(model._handle, _lib.ptr(ctx.packed), _lib.ptr(ctx.packed_bwd), ctypes.byref(tb),
                                                     _lib.ptr(ws), ws.numel(), _lib.ptr(ctx.tape), ctx.tape.numel(), _lib.ptr(x),
                                                     _lib.ptr(ctx.t), _lib.ptr(g), _lib.ptr(flat), b, t_len, ctx.p, ctx.seed, None, 0,
                                                     _lib.stream(), *side))
                if sync is not None:
                    sync(flat)  # data parallel: average the whole gradient buffer over ranks
        ctx.tape = None
        return (None, None, None, None) + tuple(flat[o:o + n].view(shape) for o, n, shape in layout)


class Model(_Node):
    """Drop-in for reference ``models.diffusion.Model`` (``models/diffusion.py:170-294``)."""

    def __init__(self, config):
        super().__init__()
        self.config = config.model  # same attribute as the reference (:174)
        self._full_config = config  # what EMAHelper.ema_copy needs to build a second instance (models/ema.py:32-45)
        self._inventory = build_parameters(self, config)
        self.embedding_size = embedding_sizes(config.model)
        self._n_timesteps = config.diffusion.num_diffusion_timesteps
        dev, act = parse_tensor_type(getattr(config.model, "dtype", None))
        # transformers.dtype (reference :242-246,267-279: the transformer is re-cast to its own dtype, which lets the convs run
        # half while the FNet stays fp32 -- the reference's only workable half setup, fftn has no bf16).  It selects the operand
        # type of the FNet's dense GEMMs; absent = fp32.
        tr_dtype = getattr(config.model.transformers, "dtype", None)
        self._fnet_dtype = parse_tensor_type(tr_dtype)[1] if tr_dtype else torch.float32
        if self._fnet_dtype == torch.bfloat16 and act != torch.bfloat16:
            raise NotImplementedError("transformers.dtype BFloat16Tensor needs model.dtype BFloat16Tensor")
        self._act_dtype = act
        self._handle = None
        self._packed = None
        self._packed_key = None
        self._dirty = True
        self._tables = {}
        self._workspace = None
        self._temb_buf = None   # [n_timesteps, E] eval-mode BetaEmbedding table: allocated once, rebuilt in place on a repack
        self._temb_table = None  # = _temb_buf while it is valid for the packed weights (eval mode), else None
        self._eager_fork = None  # ForkContext of the eager (non-captured) forked forwards
        self._eager_bwd_fork = None   # ... and of the eager backwards (weight gradients on a second stream: ddimx_unet_bwd_forked)
        self._capture_bwd_ctx = None  # set by train.GraphedTrainStep around its capture: the captured backward's own ForkContext
        # generation of the device buffers a captured graph holds raw pointers to (packed weights, embedding table, tables,
        # workspaces): bumped whenever one of them is re-allocated; a capturer compares it before every replay
        # (sampler.DDIMStepper) and re-captures instead of replaying pointers of an earlier generation
        self._gen = 0
        # which parts of the eval forward run as two batch shards on two streams (ddimx_unet_fwd_forked): bit l = level l,
        # bit 16 = the FNet; 0 turns it off.  Default: everything -- measured: the gain needs the two shards to
        # run independently from the input conv to the output conv; forking only some levels (every join is a rendezvous)
        # gives nothing, although per-op microbenchmarks say the deep levels alone prefer whole-batch launches.
        import os
        self.fork_mask = int(os.environ.get("DDIMX_FORK_MASK", hex(((1 << len(config.model.ch)) - 1) | 0x10000)), 0)
        self.bwd_fork = True  # False: the backward stays on one stream
        if dev == "cuda":
            self.to("cuda")  # like nn.Module.type("torch.cuda.FloatTensor") in the reference (:234-235)

    # -- lifecycle -----------------------------------------------------------------------------------
    def train(self, mode=True):
        self._dirty = True  # EMAHelper.ema()/checkpoint loads precede .eval()/.train() in the reference runner
        return super().train(mode)

    def invalidate(self):
        """Force re-packing of the weights on the next forward (after out-of-band ``.data`` writes)."""
        self._dirty = True

    def _apply(self, fn, *a, **k):
        # .to() / .type(): every derived device buffer is rebuilt on the next forward.  Dropping the references here frees
        # nothing a live graph still points at: a capturer holds its own references (captured_refs) and sees the new
        # generation before its next replay.
        self._dirty = True
        self._eager_fork = self._eager_bwd_fork = None
        self._tables = {}
        self._workspace = None
        self._packed = None
        self._temb_buf = self._temb_table = None
        self._gen = getattr(self, "_gen", 0) + 1
        return super()._apply(fn, *a, **k)

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None:
            try:
                from . import _lib
                _lib.load().ddimx_destroy(h)
            except Exception:
                pass

    def _ensure_handle(self):
        from . import _lib
        if self._handle is not None:
            return _lib.load()
        lib = _lib.load()
        m = self.config
        cfg = _lib.DdimxConfig()
        cfg.in_channels, cfg.f_size, cfg.n_levels = m.channels, m.f_size, len(m.ch)
        if len(m.ch) > _lib.MAX_LEVELS:
            raise NotImplementedError("more than 8 U-Net levels")
        for i, (c, r, k) in enumerate(zip(m.ch, m.res, m.krn)):
            cfg.ch[i], cfg.res[i], cfg.krn[i] = c, r, k
        kw = m.transformers.kwargs
        if getattr(m.transformers, "module", "FNetEncoder") != "FNetEncoder":
            raise NotImplementedError("only transformers.module == FNetEncoder is implemented")
        if getattr(kw, "hidden_act", "gelu_new") != "gelu_new":
            raise NotImplementedError("only hidden_act == gelu_new is implemented")
        cfg.n_timesteps = self._n_timesteps
        cfg.fnet_hidden, cfg.fnet_layers, cfg.fnet_inter = kw.hidden_size, kw.num_hidden_layers, kw.intermediate_size
        cfg.fnet_ln_eps = kw.layer_norm_eps
        cfg.act_dtype = _lib.DDIMX_BF16 if self._act_dtype == torch.bfloat16 else _lib.DDIMX_F32
        cfg.fnet_dtype = _lib.DDIMX_BF16 if self._fnet_dtype == torch.bfloat16 else _lib.DDIMX_F32
        import ctypes
        h = ctypes.c_void_p()
        _lib.check(lib.ddimx_create(ctypes.byref(cfg), ctypes.byref(h)))
        # cross-check the host mirror against the library's plan (names, sizes, order)
        n = lib.ddimx_num_params(h)
        names = list(self._inventory.keys())
        if n != len(names):
            raise RuntimeError(f"libddimx plan has {n} tensors, host mirror {len(names)}")
        nm, ne = ctypes.c_char_p(), ctypes.c_longlong()
        for i, (name, shape) in enumerate(self._inventory.items()):
            _lib.check(lib.ddimx_param_info(h, i, ctypes.byref(nm), ctypes.byref(ne)))
            numel = 1
            for d in shape:
                numel *= d
            if nm.value.decode() != name or ne.value != numel:
                raise RuntimeError(f"plan mismatch at {i}: {nm.value.decode()}[{ne.value}] vs {name}[{numel}]")
        self._handle = h
        return lib

    def _state_tensors(self):
        sd = dict(self.named_parameters())
        sd["temb.te"] = self.temb.te
        return [sd[k] for k in self._inventory]

    def _ensure_packed(self, lib, device):
        from . import _lib
        tensors = self._state_tensors()
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        if not self._dirty and self._packed is not None and key == self._packed_key:
            return
        for name, t in zip(self._inventory, tensors):
            if t.device != device or t.dtype != torch.float32 or not t.is_contiguous():
                raise RuntimeError(f"parameter {name} must be a contiguous fp32 tensor on {device} (is {t.dtype} on {t.device})")
        if self._packed is None or self._packed.device != device:
            self._packed = torch.empty(int(lib.ddimx_packed_bytes(self._handle)), dtype=torch.uint8, device=device)
            self._gen += 1
        import ctypes
        arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
        _lib.check(lib.ddimx_pack_weights(self._handle, arr, len(tensors), _lib.ptr(self._packed), _lib.stream()))
        self._packed_key = key
        self._pack_gen = getattr(self, "_pack_gen", 0) + 1  # every real repack; the backward packings follow this counter
        self._dirty = False
        self._temb_table = None
        if not self.training:
            # eval mode only (a training step re-packs after every optimizer step and reads neither): the FNet's fragment-order /
            # LayerNorm-folded copies for the launch-lean bottleneck path, and the BetaEmbedding table
            _lib.check(lib.ddimx_pack_fnet_inference(self._handle, _lib.ptr(self._packed), _lib.stream()))
            self._build_temb_table(lib, device, tensors)

    def _build_temb_table(self, lib, device, tensors):
        """Eval mode: BetaEmbedding (reference :110-120) is a pure function of t, so its [n_timesteps, E] table is computed once
        per weight set with the same kernels (``ddimx_temb_fwd`` over t = 0..n-1; rows are computed independently, so they are
        bit-identical to a per-call evaluation) and the forward copies row t[b]."""
        from . import _lib
        names = list(self._inventory)
        get = lambda n: tensors[names.index(n)]  # noqa: E731
        n, e = self._n_timesteps, sum(self.embedding_size)
        tt = torch.arange(n, dtype=torch.int64, device=device)
        h1 = torch.empty(n, _EMB_CH, dtype=torch.float32, device=device)
        h2 = torch.empty_like(h1)
        out = self._temb_buf  # rebuilt IN PLACE: graphs captured against an earlier weight set keep a valid pointer
        if out is None or out.device != device or tuple(out.shape) != (n, e):
            out = self._temb_buf = torch.empty(n, e, dtype=torch.float32, device=device)
            self._gen += 1
        w = [get(f"temb.weight.{i}.{k}") for i in range(3) for k in ("weight", "bias")]
        _lib.check(lib.ddimx_temb_fwd(_lib.ptr(get("temb.te")), _lib.ptr(tt), *[_lib.ptr(v) for v in w], _lib.ptr(h1), _lib.ptr(h2),
                                      _lib.ptr(out), n, _POS_CH, _EMB_CH, e, _lib.stream()))
        self._temb_table = out

    def prepare(self, device, t_len):
        """Pack weights / build tables for a [*, C, t_len, F] forward on the CURRENT stream.  The sampler calls this before it
        forks its batch shards onto side streams, so that no shard races the packing launches of another."""
        lib = self._ensure_handle()
        with torch.cuda.device(device):
            self._ensure_packed(lib, device)
            self._ensure_tables(t_len, device)

    def reserve(self, device, batch, t_len, slot=0):
        """Allocate (or grow) workspace ``slot`` for a [batch, C, t_len, F] forward on the CURRENT stream.  The sampler calls
        this for every shard from its launch stream before it forks, so that no workspace is ever allocated on a side stream
        or inside a capture."""
        lib = self._ensure_handle()
        need = int(lib.ddimx_workspace_bytes(self._handle, batch, t_len))
        if need <= 0:
            raise RuntimeError("libddimx: bad workspace size for B=%d T=%d" % (batch, t_len))
        if self._workspace is None:
            self._workspace = {}
        wsp = self._workspace.get(slot)
        if wsp is None or wsp.numel() < need or wsp.device != device:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("workspace slot %d must be reserved before the capture (Model.reserve)" % slot)
            with torch.cuda.device(device):
                wsp = self._workspace[slot] = torch.empty(need, dtype=torch.uint8, device=device)
            self._gen += 1
        return wsp

    def captured_refs(self):
        """Every device buffer a graph captured over this model's forward holds a raw pointer to.  The capturer keeps the
        list next to its graph, so the buffers live exactly as long as the graph, whatever happens to the model."""
        refs = [self._packed, self._temb_buf] + list((self._workspace or {}).values())
        for v in self._tables.values():
            refs += list(v)
        return [r for r in refs if r is not None]

    def new_fork_context(self, device):
        """A ForkContext for ONE graph capture (owned by the capturer), with a stream of its own."""
        ef = self._eager_context(device)
        return ForkContext(device, len(ef.events), private=True)

    def new_bwd_fork_context(self, device):
        """A ForkContext for the backward inside ONE graph capture (``train.GraphedTrainStep`` owns it and sets
        ``_capture_bwd_ctx`` around its capture), with a stream of its own."""
        ef = self._eager_bwd_context(device)
        return ForkContext(device, len(ef.events), private=True)

    def _eager_bwd_context(self, device):
        ef = self._eager_bwd_fork
        if ef is None or ef.device != device:
            from . import _lib
            with torch.cuda.device(device):
                ef = self._eager_bwd_fork = ForkContext(device, int(_lib.load().ddimx_bwd_side_events(self._handle)))
        return ef

    def _eager_context(self, device):
        ef = self._eager_fork
        if ef is None or ef.device != device:
            ef = self._eager_fork = ForkContext(device, 2 * len(self.config.ch) + 4)  # one event per fork / join of a call
        return ef

    def _ensure_packed_bwd(self, lib, device):
        """Backward-only weight packings (data-gradient conv layouts, transposed FNet matrices); follows _ensure_packed."""
        from . import _lib
        # keyed on the repack generation, not on (data_ptr, _version): writes through ``p.data`` / raw pointers followed by
        # invalidate() repack the forward weights without changing that tuple, and the backward packings must follow
        if getattr(self, "_packed_bwd_gen", None) == self._pack_gen and getattr(self, "_packed_bwd", None) is not None:
            return
        tensors = self._state_tensors()
        if getattr(self, "_packed_bwd", None) is None or self._packed_bwd.device != device:
            self._packed_bwd = torch.empty(int(lib.ddimx_packed_bwd_bytes(self._handle)), dtype=torch.uint8, device=device)
        import ctypes
        arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
        _lib.check(lib.ddimx_pack_weights_bwd(self._handle, arr, len(tensors), _lib.ptr(self._packed), _lib.ptr(self._packed_bwd),
                                              _lib.stream()))
        self._packed_bwd_gen = self._pack_gen

    def _grad_layout(self, lib):
        """(total floats, [(offset, numel, shape)] per parameter in named_parameters() order) of the flat gradient buffer."""
        if getattr(self, "_glayout", None) is None:
            offs = []
            for i, (name, shape) in enumerate(self._inventory.items()):
                if name == "temb.te":
                    continue
                numel = 1
                for d in shape:
                    numel *= d
                offs.append((int(lib.ddimx_grad_offset(self._handle, i)), numel, tuple(shape)))
            self._glayout = (int(lib.ddimx_grad_floats(self._handle)), offs)
        return self._glayout

    def _ensure_tables(self, t_len, device):
        key = (t_len, str(device))
        if key not in self._tables:
            nlev = len(self.config.ch)
            s = t_len >> (nlev - 1)
            fr = self.config.f_size >> (nlev - 1)
            c_last = self.config.ch[-1]
            hid = self.config.transformers.kwargs.hidden_size
            pe = _posenc_table(s, c_last * fr, c_last, fr).to(device)
            ch, sh = _dft_tables(hid)
            cs, ss = _dft_tables(s)
            import numpy as np
            dh = torch.from_numpy(np.stack([ch, sh], axis=1).reshape(2 * hid, hid)).to(device)  # rows 2k: cos_k, 2k+1: sin_k
            ds = torch.from_numpy(np.concatenate([cs, -ss], axis=1)).contiguous().to(device)      # [S][2S] = [cos | -sin]
            self._tables = {key: (pe, dh, ds)}  # keep only the latest T
            self._gen += 1
        return self._tables[key]

    # -- forward -------------------------------------------------------------------------------------
    def forward_slot(self, input, t, slot, out=None):
        """``forward`` over workspace ``slot``: concurrent calls on different HIP streams (the sampler's batch shards) must
        not share scratch memory.  Inference only; the in-library fork is off (the caller already runs shards in parallel)."""
        return self.forward(input, t, _slot=slot, _fork=False, _out=out)

    def forward(self, input, t, _slot=0, _fork=True, _ctx=None, _out=None):
        """input [B, C, T, F] fp32 on the GPU, t [B] int64 -> eps [B, C, T, F] fp32 (reference :237-294).
        eval mode or no_grad: ``ddimx_unet_fwd``.  train mode with grad enabled: ``ddimx_unet_fwd_train`` (dropout active,
        tape kept) as an autograd node whose backward fills every parameter's gradient (``ddimx_unet_bwd``).
        Internal keywords (sampler): ``_slot`` workspace slot, ``_fork`` allow the two-shard forward, ``_ctx`` the capturer's
        ForkContext (a capture without one runs the unforked forward: same result, bit for bit), ``_out`` preallocated eps."""
        from . import _lib
        if not input.is_cuda:
            raise RuntimeError("ddim_audio_amd.Model computes only through libddimx on a ROCm GPU; got a CPU tensor "
                               "(there is no CPU fallback)")
        if input.dtype != torch.float32:
            raise RuntimeError("the network boundary is fp32 [B,C,T,F] (activations inside use config.model.dtype)")
        b, c, t_len, f = input.shape
        if c != self.config.channels or f != self.config.f_size:
            raise RuntimeError(f"expected [B,{self.config.channels},T,{self.config.f_size}], got {tuple(input.shape)}")
        lib = self._ensure_handle()
        dev = input.device
        with torch.cuda.device(dev):
            self._ensure_packed(lib, dev)
            pe, dh, ds = self._ensure_tables(t_len, dev)
            wsp = self.reserve(dev, b, t_len, _slot)
            x = input.contiguous()
            tt = t.to(device=dev, dtype=torch.int64).contiguous()
            if self.training and torch.is_grad_enabled():
                # training step (reference runners/diffusion.py:134,143-150): tape-keeping forward, autograd node whose
                # backward is ddimx_unet_bwd; dropout as in the reference's train mode (transformers hidden_dropout_prob)
                self._ensure_packed_bwd(lib, dev)
                params = [p for _, p in self.named_parameters()]
                if getattr(self, "_alias_leaves", False):
                    # train.GraphedTrainStep: differentiate with respect to fresh leaf aliases of the parameters.  autograd keeps
                    # one AccumulateGrad node per leaf, tied to the stream it was created on and alive as long as any old graph is;
                    # a stale one from an eager step would make the capture stream hand its gradients to that other stream.
                    params = self._leaf_aliases = [p.detach().requires_grad_(True) for p in params]
                return _UNetTrainFn.apply(self, x, tt, (pe, dh, ds), *params)
            if _out is not None:
                if _out.shape != x.shape or _out.dtype != torch.float32 or not _out.is_contiguous() or _out.device != dev:
                    raise RuntimeError("_out must be a contiguous fp32 tensor of the input's shape on its device")
                out = _out
            else:
                out = torch.empty_like(x)
            tt_ptr = self._temb_table.data_ptr() if (not self.training and getattr(self, "_temb_table", None) is not None) else None
            tables = _lib.DdimxTables(pe.data_ptr(), dh.data_ptr(), ds.data_ptr(), tt_ptr)
            import ctypes
            mask = self.fork_mask if (_fork and b >= 4) else 0
            fc = None
            if mask:
                # eager calls use the model's own event set; a capture uses the set of whoever captures (never shared)
                fc = _ctx if torch.cuda.is_current_stream_capturing() else self._eager_context(dev)
            if fc is not None:
                # two batch shards on two streams (bit-identical results); everything is joined back into
                # the current stream before the call returns
                _lib.check(lib.ddimx_unet_fwd_forked(self._handle, _lib.ptr(self._packed), ctypes.byref(tables), _lib.ptr(wsp),
                                                     wsp.numel(), _lib.ptr(x), _lib.ptr(tt), _lib.ptr(out), b, t_len, _lib.stream(),
                                                     ctypes.c_void_p(fc.aux_handle), fc.event_array(), len(fc.events), mask))
            else:
                _lib.check(lib.ddimx_unet_fwd(self._handle, _lib.ptr(self._packed), ctypes.byref(tables),
                                              _lib.ptr(wsp), wsp.numel(), _lib.ptr(x), _lib.ptr(tt),
                                              _lib.ptr(out), b, t_len, _lib.stream()))
        return out
