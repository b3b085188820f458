"""CPU ORACLE -- test infrastructure, not product code.

A plain-PyTorch fp32 restatement of the reference's DDIM hot path, written as pure functions over a
``state_dict`` (name -> tensor) so it shares no module code with the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; the product
package never does (it raises if the HIP library is missing).

Pinning: every function here is checked against golden vectors produced by importing the real
reference in the authoring container (``oracle/make_golden.py`` -> ``tests/golden/*.npz``; run by
``tests/test_oracle_golden.py``).  The reference has no tests or fixtures of its own, so
those vectors are the only pin.

Each function cites the reference lines it follows (paths relative to the reference repo).
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- encodings
def add_encoding(data):
    """models/diffusion.py:81-92 -- interleaved sin/cos added in place over the last two dims."""
    length, channel = data.shape[-2], data.shape[-1]
    pos = torch.arange(length, dtype=data.dtype).unsqueeze(1)
    div = torch.exp(torch.arange(0, channel, 2, dtype=data.dtype) * (-math.log(10000.0) / channel))
    ang = pos * div
    data[..., 0::2] += torch.sin(ang)
    data[..., 1::2] += torch.cos(ang)
    return data


def timestep_table(num_timesteps, pos_ch=128):
    """models/diffusion.py:98-102 -- the ``temb.te`` buffer."""
    return add_encoding(torch.zeros(num_timesteps, pos_ch))


def beta_embedding(sd, t, prefix="temb."):
    """models/diffusion.py:110-120 -- table lookup then 3 Linear layers with SiLU between."""
    x = sd[prefix + "te"].index_select(0, t)
    x = F.silu(F.linear(x, sd[prefix + "weight.0.weight"], sd[prefix + "weight.0.bias"]))
    x = F.silu(F.linear(x, sd[prefix + "weight.1.weight"], sd[prefix + "weight.1.bias"]))
    return F.linear(x, sd[prefix + "weight.2.weight"], sd[prefix + "weight.2.bias"])


# ----------------------------------------------------------------------------- conv blocks
def residual_block(sd, prefix, x, temb):
    """models/diffusion.py:42-56 -- GN0,SiLU,conv0(+temb),SiLU,GN1,conv1,SiLU,GN2(no bias), +input."""
    eps = 1e-6
    h = F.group_norm(x, 8, sd[prefix + "norm.0.weight"], sd[prefix + "norm.0.bias"], eps)
    h = F.silu(h)
    k = sd[prefix + "conv.0.weight"].shape[-1]
    h = F.conv2d(h, sd[prefix + "conv.0.weight"], None, stride=1, padding=k // 2) + temb[..., None, None]
    h = F.silu(h)
    h = F.group_norm(h, 8, sd[prefix + "norm.1.weight"], sd[prefix + "norm.1.bias"], eps)
    h = F.conv2d(h, sd[prefix + "conv.1.weight"], sd[prefix + "conv.1.bias"], stride=1, padding=k // 2)
    h = F.silu(h)
    h = F.group_norm(h, 8, sd[prefix + "norm.2.weight"], None, eps)
    return x + h


def downsample(sd, prefix, x):
    """models/diffusion.py:70-78 -- Conv2d(k4, s2, p1)."""
    return F.conv2d(x, sd[prefix + "conv.weight"], sd[prefix + "conv.bias"], stride=2, padding=1)


def upsample(sd, prefix, x):
    """models/diffusion.py:59-67 -- ConvTranspose2d(k4, s2, p1), weight [Cin, Cout, 4, 4]."""
    return F.conv_transpose2d(x, sd[prefix + "conv.weight"], sd[prefix + "conv.bias"], stride=2, padding=1)


# ----------------------------------------------------------------------------- FNet bottleneck
def gelu_new(v):
    """transformers activations.py:59-66 (NewGELUActivation)."""
    return 0.5 * v * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (v + 0.044715 * torch.pow(v, 3.0))))


def fnet_layer(sd, prefix, x, eps):
    """transformers modeling_fnet.py:138-252 -- Re(FFT2) mixing + LN, FFN + LN (eval: no dropout)."""
    h = x.shape[-1]
    mix = torch.fft.fftn(x, dim=(1, 2)).real
    y = F.layer_norm(x + mix, (h,), sd[prefix + "fourier.output.LayerNorm.weight"],
                     sd[prefix + "fourier.output.LayerNorm.bias"], eps)
    z = gelu_new(F.linear(y, sd[prefix + "intermediate.dense.weight"], sd[prefix + "intermediate.dense.bias"]))
    z = F.linear(z, sd[prefix + "output.dense.weight"], sd[prefix + "output.dense.bias"])
    return F.layer_norm(z + y, (h,), sd[prefix + "output.LayerNorm.weight"], sd[prefix + "output.LayerNorm.bias"], eps)


def transformer_module(sd, x, n_layers, eps, prefix="transformer."):
    """models/diffusion.py:131-145,158-167 -- +posenc (table length rounded up to a power of two,
    sliced to S), LN, projection, FNet encoder, output Linear.  Eval mode (dropout inactive)."""
    s, width = x.shape[1], x.shape[2]
    size = 2 ** math.ceil(math.log2(s)) if s > 1 else 1
    pe = add_encoding(torch.zeros(size, width, dtype=x.dtype))
    h = x + pe[:s]
    h = F.layer_norm(h, (width,), sd[prefix + "embedding.LayerNorm.weight"],
                     sd[prefix + "embedding.LayerNorm.bias"], eps)
    h = F.linear(h, sd[prefix + "embedding.projection.weight"], sd[prefix + "embedding.projection.bias"])
    for i in range(n_layers):
        h = fnet_layer(sd, f"{prefix}encoder.layer.{i}.", h, eps)
    return F.linear(h, sd[prefix + "compute_out.weight"], sd[prefix + "compute_out.bias"])


# ----------------------------------------------------------------------------- whole network
def embedding_sizes(mcfg):
    """models/diffusion.py:178-184."""
    e = [c for r, c in zip(mcfg.res, mcfg.ch) for _ in range(r)]
    return e + e[::-1]


def model_forward(sd, cfg, x, t):
    """models/diffusion.py:237-294 (eval mode).  ``cfg`` is the full config Namespace."""
    m = cfg.model
    nlev = len(m.ch)
    temb = iter(torch.split(beta_embedding(sd, t), embedding_sizes(m), dim=-1))
    hidden = []
    # down path: entry 0 is the input conv, entry l+1 = [Downsample?] + res[l] blocks
    x = F.conv2d(x, sd["down_modules.0.weight"], sd["down_modules.0.bias"], padding=1)
    hidden.append(x)
    for lvl in range(nlev):
        base = f"down_modules.{lvl + 1}."
        j = 0
        if lvl > 0:
            x = downsample(sd, base + "0.", x)
            j = 1
        for r in range(m.res[lvl]):
            x = residual_block(sd, f"{base}{j + r}.", x, next(temb))
        hidden.append(x)
    # bottleneck: [B,C,S,Fr] -> [B,S,C*Fr]
    b, c, s, fr = x.shape
    tok = x.permute(0, 2, 1, 3).reshape(b, s, c * fr)
    kw = m.transformers.kwargs
    tok = transformer_module(sd, tok, kw.num_hidden_layers, kw.layer_norm_eps)
    x = tok.reshape(b, s, c, fr).permute(0, 2, 1, 3)
    # up path: entry k = res blocks of level nlev-1-k then Upsample; last entry is the output conv
    for k in range(nlev):
        lvl = nlev - 1 - k
        x = x + hidden.pop()
        base = f"up_modules.{k}."
        for r in range(m.res[lvl]):
            x = residual_block(sd, f"{base}{r}.", x, next(temb))
        if lvl > 0:
            x = upsample(sd, f"{base}{m.res[lvl]}.", x)
    x = x + hidden.pop()
    return F.conv2d(x, sd[f"up_modules.{nlev}.weight"], sd[f"up_modules.{nlev}.bias"], padding=1)


# ----------------------------------------------------------------------------- sampler / loss / EMA
def generalized_steps(x, seq, model_fn, alpha, select_index=None, eta=0.0, noise_fn=None):
    """functions/denoising.py:10-52 with the intended (GPU) list semantics: every selected step
    appends an independent copy.  ``model_fn(xt, t)`` returns eps.  Arithmetic order follows the
    reference's in-place fp32 chain with Python-double scalars."""
    a = [1.0] + alpha.to("cpu", torch.float32).numpy().tolist()
    n = x.size(0)
    seq = list(seq)
    seq_next = [-1] + seq[:-1]
    x0_preds, xs = [], [x]  # xs[0] IS the caller's tensor; like the reference it ends up holding the
    xt = x if x.dtype == torch.float32 else x.float()  # final sample because xt aliases it (:17-18)
    t = torch.zeros(n, dtype=torch.long)
    for index, (i, j) in enumerate(zip(reversed(seq), reversed(seq_next))):
        t[...] = i
        at, at_next = a[int(i) + 1], a[int(j) + 1]
        et = model_fn(xt, t)
        xt.add_(et, alpha=-((1 - at) ** 0.5)).div_(at ** 0.5)
        sel = select_index is None or index in select_index or index - len(seq) in select_index
        if sel:
            x0_preds.append(xt.clone())
        c1 = eta * ((1 - at / at_next) * (1 - at_next) / (1 - at)) ** 0.5
        c2 = ((1 - at_next) - c1 ** 2) ** 0.5
        noise = noise_fn(index, xt) if noise_fn is not None else torch.randn_like(xt)
        xt.mul_(at_next ** 0.5).add_(et, alpha=c2).add_(noise, alpha=c1)
        if sel:
            xs.append(xt.clone())
    return xs, x0_preds


def compute_alpha(beta, t):
    """functions/denoising.py:4-7."""
    beta = torch.cat([torch.zeros(1), beta], dim=0)
    return (1 - beta).cumprod(dim=0).index_select(0, t + 1).view(-1, 1, 1, 1)


def ddpm_steps(x, seq, model_fn, betas, noise_fn):
    """functions/denoising.py:55-92 (select_index=None); ``noise_fn(k, x)`` supplies the Gaussian noise."""
    n = x.size(0)
    seq = list(seq)
    seq_next = [-1] + seq[:-1]
    xs, x0_preds = [x], []
    for k, (i, j) in enumerate(zip(reversed(seq), reversed(seq_next))):
        t = torch.ones(n) * i
        next_t = torch.ones(n) * j
        at = compute_alpha(betas, t.long())
        atm1 = compute_alpha(betas, next_t.long())
        beta_t = 1 - at / atm1
        xc = xs[-1]
        e = model_fn(xc, t.long())
        x0 = torch.clamp((1.0 / at).sqrt() * xc - (1.0 / at - 1).sqrt() * e, -1, 1)
        x0_preds.append(x0)
        mean = ((atm1.sqrt() * beta_t) * x0 + ((1 - beta_t).sqrt() * (1 - atm1)) * xc) / (1.0 - at)
        mask = (1 - (t == 0).float()).view(-1, 1, 1, 1)
        xs.append(mean + mask * torch.exp(0.5 * beta_t.log()) * noise_fn(k, xc))
    return xs, x0_preds


def noise_estimation_loss(model_fn, x0, t, e, a, keepdim=False):
    """functions/losses.py:4-18."""
    at = a.index_select(0, t).view(-1, 1, 1, 1)
    x = x0 * at.sqrt() + e * (1.0 - at).sqrt()
    out = model_fn(x, t)
    per = (e - out).square().sum(dim=(1, 2, 3))
    return per if keepdim else per.mean(dim=0)


def ema_update(shadow, params, mu):
    """models/ema.py:16-23 -- shadow = (1-mu)*p + mu*shadow, per tensor."""
    return {k: (1.0 - mu) * params[k] + mu * shadow[k] for k in shadow}


def lr_factor(step, warmup):
    """functions/__init__.py:53-60."""
    return min(((1 + step) / warmup) ** -0.5, (1 + step) / warmup)


# ----------------------------------------------------------------------------- training step
def classify_group(names, group_cfg):
    """runners/diffusion.py:71-87 -- route parameter names by their top-level module name; groups without
    parameters are dropped.  ``group_cfg``: Namespace {group: Namespace(top_level_name=[...], ...)}."""
    top = {}
    for gname, sub in vars(group_cfg).items():
        for n in sub.top_level_name:
            top[n] = gname
    groups = {gname: [] for gname in vars(group_cfg)}
    for n in names:
        groups[top.get(n.split(".")[0], "default")].append(n)
    return {k: v for k, v in groups.items() if v}


def make_optimizer(ocfg, params):
    """functions/__init__.py:5-23 (Adam / AdamW rows; AdaBelief's source is absent from the reference tree)."""
    cls = {"Adam": torch.optim.Adam, "AdamW": torch.optim.AdamW}[ocfg.optimizer]
    return cls(params, lr=ocfg.lr, weight_decay=ocfg.weight_decay, betas=tuple(ocfg.beta), amsgrad=ocfg.amsgrad, eps=ocfg.eps)


class TrainState:
    """Parameters (leaf tensors), optimizers, LambdaLR schedulers and the EMA shadow of one training run."""

    def __init__(self, sd, cfg):
        self.cfg = cfg
        self.buffers = {"temb.te": sd["temb.te"]}
        self.params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != "temb.te"}
        ocfg = cfg.optimization.optimizer
        self.groups = classify_group(self.params.keys(), ocfg)
        self.optimizers = {g: make_optimizer(getattr(ocfg, g), [self.params[n] for n in names])
                           for g, names in self.groups.items()}
        self.schedulers = {g: torch.optim.lr_scheduler.LambdaLR(o, lambda s, w=getattr(ocfg, g).warmup: lr_factor(s, w))
                           for g, o in self.optimizers.items() if getattr(ocfg, g).warmup}
        self.clip_groups = classify_group(self.params.keys(), cfg.optimization.grad_norm)
        self.shadow = {k: v.detach().clone() for k, v in self.params.items()}


def train_step(st, x0, e, t, alphas, mu=0.9999):
    """runners/diffusion.py:130-173 with (e, t) supplied by the caller: loss, backward, per-group clip, optimizer and
    scheduler steps, EMA.  Returns (loss, {clip group: total norm before clipping})."""
    live = dict(st.params, **st.buffers)
    loss = noise_estimation_loss(lambda a, b: model_forward(live, st.cfg, a, b), x0, t, e, alphas)
    for o in st.optimizers.values():
        o.zero_grad()
    loss.backward()
    norms = {}
    for g, names in st.clip_groups.items():
        clip = getattr(st.cfg.optimization.grad_norm, g).grad_clip
        if clip is not None:
            norms[g] = float(torch.nn.utils.clip_grad_norm_([st.params[n] for n in names], clip))
    for o in st.optimizers.values():
        o.step()
    for s in st.schedulers.values():
        s.step()
    with torch.no_grad():
        st.shadow = ema_update(st.shadow, {k: v.detach() for k, v in st.params.items()}, mu)
    return float(loss), norms


def adabelief_step(p, g, m, v, step, lr, betas, eps, weight_decay):
    """PARITY UNPINNED.  The reference's default optimizer (functions/__init__.py:24-42) is `clip_opt.AdaBelief` from the
    un-vendored submodule External/step-clip-optimizer (no source in the reference tree, no pinned revision).  This
    restates the published AdaBelief update (Zhuang et al., NeurIPS 2020, Algorithm 2) with the flags the reference passes:
    weight_decouple=True, fixed_decay=False, rectify=False, amsgrad=False, clip_step=None.  In place on (p, m, v)."""
    b1, b2 = betas
    p.mul_(1.0 - lr * weight_decay)
    m.mul_(b1).add_(g, alpha=1 - b1)
    r = g - m
    v.mul_(b2).addcmul_(r, r, value=1 - b2).add_(eps)
    denom = (v.sqrt() / math.sqrt(1 - b2 ** step)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / (1 - b1 ** step))
