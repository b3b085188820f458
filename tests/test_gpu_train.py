"""GPU parity tests of the training path: HIP forward-with-tape and backward kernels (through the C ABI) against
autograd over the CPU oracle (itself pinned to the reference's gradients by tests/golden/train.npz).
Gates: fp32 mode max|d| <= 1e-4 * std (x4 for parameter gradients, which are long fp32 sums); bf16 mode
rms <= 2e-2 * std, max <= 1.5e-1 * std -- the same yardsticks as the forward."""
import numpy as np
import pytest
import torch

from ddim_audio_amd import _lib, synth
from oracle import ref_cpu
import gpu_util as G
from test_gpu_ops import _rb_sd

pytestmark = pytest.mark.gpu
DTS = [G.F32, G.BF16]


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("c,hw,b", [(32, (40, 72), 3), (64, (24, 40), 2), (96, (17, 33), 2), (128, (16, 24), 2),
                                    (192, (9, 20), 1), (256, (20, 9), 2),
                                    (32, (200, 48), 7)])  # 75 tiles per sample, 525 in all: workgroups walk 2 tiles and straddle samples
def test_resblock_backward(dt, c, hw, b):
    p = f"rbt{c}."
    sd = _rb_sd(p, c)
    x = synth.gaussian(p + "x", (b, c, *hw)) * 1.5 + 0.3
    temb = synth.gaussian(p + "temb", (b, c)) * 0.5
    dy = synth.gaussian(p + "dy", (b, c, *hw))
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr, tr = x.clone().requires_grad_(True), temb.clone().requires_grad_(True)
    yr = ref_cpu.residual_block(leaf, p, xr, tr)
    yr.backward(dy)
    y, dx, grads, dtemb = G.resblock_train(sd, p, x, temb, dy, dt)
    G.check_close(y, yr.detach(), dt, f"train fwd C={c}")
    G.check_close(dx, xr.grad, dt, f"dx C={c}")
    G.check_close(dtemb, tr.grad, dt, f"dtemb C={c}", scale=4.0)
    for n, gv in grads.items():
        G.check_close(gv, leaf[p + n].grad, dt, f"d {n} C={c}", scale=4.0)


# ---- whole network: loss.backward() through the reference-shaped Python API vs the reference's own gradients ----
import ddim_audio_amd as D
from ddim_audio_amd import configs, losses
from ddim_audio_amd.schedule import make_schedule

MODES = [("torch.cuda.FloatTensor", G.F32), ("torch.cuda.BFloat16Tensor", G.BF16)]


def _train_model(name, dtype_str, seed, dropout=0.0):
    d = configs.tiny_dict(dtype_str) if name == "tiny" else configs.audio_dict(dtype_str)
    d["model"]["transformers"]["kwargs"]["hidden_dropout_prob"] = dropout
    d["optimization"]["optimizer"]["default"]["optimizer"] = "Adam"
    cfg = configs.dict2namespace(d)
    m = D.Model(cfg)
    synth.fill_module(m, seed)
    return cfg, m.train()


@pytest.mark.parametrize("mode", MODES, ids=["f32", "bf16"])
@pytest.mark.parametrize("name,shape,seed", [("tiny", (2, 2, 16, 32), 3), ("audio", (2, 2, 32, 256), 0)])
def test_model_backward_golden(golden, mode, name, shape, seed):
    """loss + all 388 parameter gradients against the digest the real reference produced (tests/golden/train.npz)."""
    dtype_str, dt = mode
    g = golden("train")
    cfg, m = _train_model(name, dtype_str, seed)
    _, alphas = make_schedule(cfg.diffusion)
    x0, e = synth.gaussian(f"train.{name}.x0", shape).cuda(), synth.gaussian(f"train.{name}.e", shape).cuda()
    t = torch.from_numpy(g[f"{name}_t"]).cuda()
    loss = losses.noise_estimation_loss(m, x0, t, e, alphas.cuda())
    loss.backward()
    ref_loss = float(g[f"{name}_loss"])
    assert abs(float(loss) - ref_loss) <= (1e-5 if dt == G.F32 else 2e-3) * ref_loss
    names, norms = [str(n) for n in g[f"{name}_names"]], g[f"{name}_gnorms"]
    total = float(np.sqrt((norms ** 2).sum()))
    params = dict(m.named_parameters())
    assert list(params.keys()) == names
    worst = 0.0
    got_total = 0.0
    for pname, ref_norm in zip(names, norms):
        grad = params[pname].grad
        assert grad is not None and torch.isfinite(grad).all(), pname
        flat = grad.detach().cpu().reshape(-1)
        got_total += float(flat.double().square().sum())
        ref = g[f"{name}_g::{pname}"]
        stride = max(1, flat.numel() // 256)
        got = flat[::stride][:256].numpy()
        # per-tensor gate: relative to the tensor's own RMS gradient (floor: a 1e-4 share of the global norm)
        rms_ref = ref_norm / np.sqrt(flat.numel())
        scale = max(rms_ref, 1e-4 * total / np.sqrt(flat.numel()))
        err = float(np.abs(got - ref).max()) / scale
        worst = max(worst, err)
        tol = 2e-3 if dt == G.F32 else 0.5
        assert err <= tol, f"{pname}: max err {err:.3e} x rms (tol {tol})"
        nerr = abs(float(flat.double().square().sum()) ** 0.5 - ref_norm) / max(ref_norm, 1e-4 * total)
        assert nerr <= (5e-4 if dt == G.F32 else 6e-2), f"{pname}: norm err {nerr:.3e}"
    print(f"[backward golden {name} {'f32' if dt == G.F32 else 'bf16'}] worst element {worst:.3e} x rms, global norm rel err "
          f"{abs(got_total ** 0.5 - total) / total:.2e}")
    assert abs(got_total ** 0.5 - total) <= (1e-4 if dt == G.F32 else 2e-2) * total


def test_reference_runner_sequence_golden(golden):
    """The literal statement sequence of the reference's Diffusion.train_step (runners/diffusion.py:130-173) over the drop-in
    symbols -- loss_registry, zero_grad, backward, torch.nn.utils.clip_grad_norm_, optimizer / scheduler steps, EMAHelper --
    reproduces the reference's first optimisation step."""
    from ddim_audio_amd import train
    from ddim_audio_amd.dropin.functions import get_optimizer, get_scheduler
    from ddim_audio_amd.dropin.functions.losses import loss_registry
    from ddim_audio_amd.dropin.models.ema import EMAHelper
    g = golden("train")
    cfg, model = _train_model("tiny", "torch.cuda.FloatTensor", 3)
    optimizers, schedulers = {}, {}
    for name, p_opt in train.classify_group(cfg.optimization.optimizer, model).items():
        optimizers[name] = get_optimizer(p_opt.config, p_opt.params)
        schedulers[name] = get_scheduler(p_opt.config, optimizers[name])
    grad_group = train.classify_group(cfg.optimization.grad_norm, model)
    ema_helper = EMAHelper(mu=cfg.model.ema_rate)
    ema_helper.register(model)
    _, alphas = make_schedule(cfg.diffusion)
    x = synth.gaussian("train.tiny.x0", (2, 2, 16, 32)).cuda()
    e = synth.gaussian("train.tiny.e", (2, 2, 16, 32)).cuda()
    t = torch.from_numpy(g["tiny_t"]).cuda()
    model.train()
    loss = loss_registry[cfg.model.type](model, x, t, e, alphas.cuda())
    assert abs(loss.item() - float(g["step0_loss"])) < 1e-4 * float(g["step0_loss"])
    for optimizer in optimizers.values():
        optimizer.zero_grad()
    loss.backward()
    for name, p_opt in grad_group.items():
        norm = torch.nn.utils.clip_grad_norm_(p_opt.params, p_opt.config.grad_clip)
        assert abs(float(norm) - float(g[f"step0_norm_{name}"])) < 1e-3 * float(norm)
    for optimizer in optimizers.values():
        optimizer.step()
    for scheduler in schedulers.values():
        scheduler.step()
    ema_helper.update(model)
    states = [model.state_dict(), optimizer.state_dict(), 0, 1, ema_helper.state_dict()]
    assert len(states[0]) == 1 + len(list(model.parameters())) and "state" in states[1]
    for n, p in model.named_parameters():
        stride = max(1, p.numel() // 64)
        ref = g[f"step0_p::{n}"]
        lr = 5e-4 if n.startswith("transformer.") else 3e-4
        assert np.abs(p.detach().cpu().reshape(-1)[::stride][:64].numpy() - ref).max() <= 0.02 * lr + 1e-6 * np.abs(ref).max(), n
        sh = ema_helper.shadow[n].cpu().reshape(-1)[::stride][:64].numpy()
        assert np.abs(sh - g[f"step0_ema::{n}"]).max() <= 1e-5 * np.abs(ref).max() + 1e-7, n
    # the next forward must see the updated weights without any explicit invalidation
    with torch.no_grad():
        y1 = model.eval()(x, t)
        fresh = D.Model(cfg)
        fresh.load_state_dict(model.state_dict())
        y2 = fresh.eval()(x, t)
    assert torch.equal(y1, y2)


def test_training_steps_golden(golden):
    """Two full optimisation steps (loss, backward, clip, fused Adam/AdamW, LambdaLR, EMA) against the reference's own
    train_step tail (fp32 mode).  Adam divides by sqrt(v) + eps, which amplifies gradient noise where |g| ~ eps; the gate
    is therefore on the parameter UPDATE relative to lr."""
    from ddim_audio_amd import train
    g = golden("train")
    cfg, m = _train_model("tiny", "torch.cuda.FloatTensor", 3)
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    state = train.TrainingState(cfg, m)
    assert list(state.optimizers.keys()) == [str(s) for s in g["step_groups"]]
    _, alphas = make_schedule(cfg.diffusion)
    alphas = alphas.cuda()
    shape = (2, 2, 16, 32)
    for it in range(2):
        sfx = f".{it}" if it else ""
        x0, e = synth.gaussian(f"train.tiny.x0{sfx}", shape).cuda(), synth.gaussian(f"train.tiny.e{sfx}", shape).cuda()
        t = torch.tensor([5, 994]) if it else torch.from_numpy(g["tiny_t"])
        loss, norms = train.train_step(m, x0, state, alphas, e=e, t=t)
        assert abs(float(loss) - float(g[f"step{it}_loss"])) < 1e-4 * float(g[f"step{it}_loss"])
        for k, v in norms.items():
            assert abs(float(v) - float(g[f"step{it}_norm_{k}"])) < 1e-3 * float(v)
        bad = []
        for n, p in m.named_parameters():
            stride = max(1, p.numel() // 64)
            got = p.detach().cpu().reshape(-1)[::stride][:64].numpy()
            ref = g[f"step{it}_p::{n}"]
            lr = 5e-4 if n.startswith("transformer.") else 3e-4
            # updates are at most ~lr per step (times the warm-up factor); allow 2 % of lr per step taken
            if np.abs(got - ref).max() > 0.02 * lr * (it + 1) + 1e-6 * np.abs(ref).max():
                bad.append((n, float(np.abs(got - ref).max())))
            sh = state.ema_helper.shadow[n].cpu().reshape(-1)[::stride][:64].numpy()
            assert np.abs(sh - g[f"step{it}_ema::{n}"]).max() <= 1e-5 * np.abs(ref).max() + 1e-7, n
        assert not bad, bad[:5]
    assert np.allclose([o.param_groups[0]["lr"] for o in state.optimizers.values()], g["step_lrs"], rtol=1e-6)
    moved = sum(int((p.detach() != before[n]).any()) for n, p in m.named_parameters())
    assert moved == len(before)


def test_dropout_training_mode():
    """hidden_dropout_prob > 0 (the reference's train mode): masks are a function of (seed, call counter), the backward
    regenerates them -- checked with a directional finite difference of the loss at a fixed mask (fp32 mode)."""
    cfg, m = _train_model("tiny", "torch.cuda.FloatTensor", 3, dropout=0.1)
    _, alphas = make_schedule(cfg.diffusion)
    alphas = alphas.cuda()
    shape = (2, 2, 16, 32)
    x0, e = synth.gaussian("train.tiny.x0", shape).cuda(), synth.gaussian("train.tiny.e", shape).cuda()
    t = torch.tensor([123, 876]).cuda()

    def loss_at(call):
        m._dropout_calls = call
        return losses.noise_estimation_loss(m, x0, t, e, alphas)

    l1 = loss_at(7)
    l1.backward()
    g1 = {n: p.grad.clone() for n, p in m.named_parameters()}
    m.zero_grad()
    l1b = loss_at(7)
    l2 = loss_at(8)
    assert float(l1) == float(l1b) and float(l1) != float(l2)       # same seed -> same mask; next call -> another mask
    m.eval()
    with torch.no_grad():
        l_eval = losses.noise_estimation_loss(m, x0, t, e, alphas)
    m.train()
    assert abs(float(l1) - float(l_eval)) > 1e-6 * float(l_eval)     # dropout is really active in train mode
    # directional derivative along a random direction restricted to the transformer (where dropout acts) + one conv
    names = [n for n in g1 if n.startswith("transformer.")] + ["down_modules.1.0.conv.0.weight"]
    params = dict(m.named_parameters())
    dirs = {n: synth.gaussian("dd." + n, tuple(params[n].shape)).cuda() for n in names}
    want = sum(float((g1[n].double() * dirs[n].double()).sum()) for n in names)
    h = 1e-3
    vals = []
    for sgn in (+1, -1):
        with torch.no_grad():
            for n in names:
                params[n].add_(dirs[n], alpha=sgn * h)
        m.invalidate()
        vals.append(float(loss_at(7).detach()))
        with torch.no_grad():
            for n in names:
                params[n].add_(dirs[n], alpha=-sgn * h)
    m.invalidate()
    fd = (vals[0] - vals[1]) / (2 * h)
    assert abs(fd - want) <= 2e-2 * abs(want) + 1e-3, (fd, want)


@pytest.mark.parametrize("mode", MODES, ids=["f32", "bf16"])
def test_model_backward_ragged_shape_vs_oracle(mode):
    """Odd batch and a T whose bottleneck length is not a power of two (ragged conv tiles, ragged weight-gradient tiles,
    non-power-of-two DFT): loss and gradients against autograd through the CPU oracle."""
    dtype_str, dt = mode
    _ragged_case(mode, (3, 2, 24, 32), [0, 999, 412])


@pytest.mark.parametrize("mode", MODES, ids=["f32", "bf16"])
def test_model_backward_tall_shape_vs_oracle(mode):
    """T = 256: several row chunks / strips in the edge-conv gradient kernels, multi-tile persistent conv workgroups."""
    _ragged_case(mode, (2, 2, 256, 32), [77, 940])


def _ragged_case(mode, shape, tt):
    dtype_str, dt = mode
    cfg, m = _train_model("tiny", dtype_str, 5)
    _, alphas = make_schedule(cfg.diffusion)
    x0, e = synth.gaussian("ragged.x0", shape), synth.gaussian("ragged.e", shape)
    t = torch.tensor(tt)
    loss = losses.noise_estimation_loss(m, x0.cuda(), t.cuda(), e.cuda(), alphas.cuda())
    loss.backward()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != "temb.te"}
    live = dict(params, **{"temb.te": sd["temb.te"]})
    ocfg = configs.dict2namespace(configs.tiny_dict("torch.FloatTensor"))
    want = ref_cpu.noise_estimation_loss(lambda a, b: ref_cpu.model_forward(live, ocfg, a, b), x0, t, e, alphas)
    want.backward()
    assert abs(float(loss) - float(want)) <= (1e-5 if dt == G.F32 else 2e-3) * float(want)
    total = sum(float(p.grad.double().square().sum()) for p in params.values()) ** 0.5
    worst = 0.0
    for name, p in m.named_parameters():
        ref = params[name].grad
        got = p.grad.detach().cpu()
        scale = max(float(ref.double().square().mean().sqrt()), 1e-4 * total / ref.numel() ** 0.5)
        err = float((got - ref).abs().max()) / scale
        worst = max(worst, err)
        assert err <= (2e-3 if dt == G.F32 else 0.6), f"{name}: {err:.3e} x rms"
    print(f"[backward ragged {shape} {'f32' if dt == G.F32 else 'bf16'}] worst element {worst:.3e} x rms")


def test_training_reduces_the_loss_and_state_dicts_roundtrip():
    """A few optimisation steps on one fixed batch lower the loss (bf16 mode, dropout on); optimizer / EMA state survives a
    state_dict round trip in the reference's checkpoint layout."""
    from ddim_audio_amd import train
    cfg, m = _train_model("tiny", "torch.cuda.BFloat16Tensor", 11, dropout=0.1)
    cfg.optimization.optimizer.default.warmup = 2     # reach the full learning rate within the test
    cfg.optimization.optimizer.transformer.warmup = 2
    state = train.TrainingState(cfg, m)
    _, alphas = make_schedule(cfg.diffusion)
    alphas = alphas.cuda()
    x = synth.gaussian("fit.x", (4, 2, 32, 32)).cuda()
    e = synth.gaussian("fit.e", (4, 2, 32, 32)).cuda()
    t = torch.tensor([10, 500, 989, 250])
    losses_seen = [float(train.train_step(m, x, state, alphas, e=e, t=t)[0]) for _ in range(12)]
    assert all(np.isfinite(losses_seen)) and losses_seen[-1] < 0.9 * losses_seen[0], losses_seen
    sd = {k: o.state_dict() for k, o in state.optimizers.items()}
    for k, o in state.optimizers.items():
        o.load_state_dict(sd[k])
    loss_after, _ = train.train_step(m, x, state, alphas, e=e, t=t)
    assert np.isfinite(float(loss_after))


# ---- backward twins of the remaining per-op forwards ("_bwd twin of each forward op") ---------------------------
@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("cin,cout,hw,b", [(32, 64, (36, 44), 2), (96, 128, (18, 22), 2), (192, 256, (12, 20), 1)])
def test_downsample_upsample_backward_vs_oracle(dt, cin, cout, hw, b):
    """ddimx_downsample_bwd / ddimx_upsample_add_bwd against autograd through the oracle's Downsample / Upsample
    (models/diffusion.py:59-78): input gradients (incl. the accumulated skip gradient), weight and bias gradients."""
    lib = _lib.load()
    tdt = G.TORCH_DT[dt]
    sd = synth.fill_state_dict({"d.conv.weight": torch.empty(cout, cin, 4, 4), "d.conv.bias": torch.empty(cout),
                                "u.conv.weight": torch.empty(cout, cin, 4, 4), "u.conv.bias": torch.empty(cin)})
    h, w = hw
    # ---- Downsample
    x = synth.gaussian(f"dbw{cin}.x", (b, cin, h, w))
    dy = synth.gaussian(f"dbw{cin}.dy", (b, cout, h // 2, w // 2))
    extra = synth.gaussian(f"dbw{cin}.extra", (b, cin, h, w)) * 0.5
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    ref_cpu.downsample(leaf, "d.", xr).backward(dy)
    xn, dyn, exn = G.to_nhwc(x, dt), G.to_nhwc(dy, dt), G.to_nhwc(extra, dt)
    dxn = torch.empty_like(xn)
    wdg = G.g(sd["d.conv.weight"])
    wd_dgrad = torch.empty(2 * 6 * 2 * cin * cout, dtype=tdt, device=G.dev())
    _lib.check(lib.ddimx_pack_convT(dt, _lib.ptr(wdg), _lib.ptr(wd_dgrad), cout, cin, _lib.stream()))
    d_w, d_b = torch.full((cout, cin, 4, 4), float("nan"), device=G.dev()), torch.full((cout,), float("nan"), device=G.dev())
    ws = torch.empty(int(lib.ddimx_downup_bwd_workspace_bytes(dt, cout, cin, b, h // 2, w // 2)), dtype=torch.uint8, device=G.dev())
    _lib.check(lib.ddimx_downsample_bwd(dt, cin, cout, _lib.ptr(xn), _lib.ptr(dyn), _lib.ptr(wd_dgrad), _lib.ptr(exn), _lib.ptr(dxn),
                                        _lib.ptr(d_w), _lib.ptr(d_b), _lib.ptr(ws), b, h, w, _lib.stream()))
    ex_r = G.from_nhwc(exn, dt)
    G.check_close(G.from_nhwc(dxn, dt), xr.grad + ex_r, dt, f"downsample dx {cin}->{cout}")
    G.check_close(d_w.cpu(), leaf["d.conv.weight"].grad, dt, "downsample dW", scale=4.0)
    G.check_close(d_b.cpu(), leaf["d.conv.bias"].grad, dt, "downsample db", scale=4.0)
    # ---- Upsample (+ skip add: the skip's gradient is dy itself)
    xu = synth.gaussian(f"ubw{cout}.x", (b, cout, h // 2, w // 2))
    dyu = synth.gaussian(f"ubw{cout}.dy", (b, cin, h, w))
    xur = xu.clone().requires_grad_(True)
    ref_cpu.upsample(leaf, "u.", xur).backward(dyu)
    xun, dyun = G.to_nhwc(xu, dt), G.to_nhwc(dyu, dt)
    dxun = torch.empty_like(xun)
    wu_dgrad = G.pack_conv(sd["u.conv.weight"], dt)  # [O = cout(Cin of the upsample)][I = cin][4][4] read as a Conv2d weight
    du_w, du_b = torch.full((cout, cin, 4, 4), float("nan"), device=G.dev()), torch.full((cin,), float("nan"), device=G.dev())
    ws = torch.empty(int(lib.ddimx_downup_bwd_workspace_bytes(dt, cout, cin, b, h // 2, w // 2)), dtype=torch.uint8, device=G.dev())
    _lib.check(lib.ddimx_upsample_add_bwd(dt, cout, cin, _lib.ptr(xun), _lib.ptr(dyun), _lib.ptr(wu_dgrad), _lib.ptr(dxun), _lib.ptr(du_w),
                                          _lib.ptr(du_b), _lib.ptr(ws), b, h // 2, w // 2, _lib.stream()))
    G.check_close(G.from_nhwc(dxun, dt), xur.grad, dt, f"upsample dx {cout}->{cin}")
    G.check_close(du_w.cpu(), leaf["u.conv.weight"].grad, dt, "upsample dW", scale=4.0)
    G.check_close(du_b.cpu(), leaf["u.conv.bias"].grad, dt, "upsample db", scale=4.0)


@pytest.mark.parametrize("dt", DTS)
def test_edge_conv_backward_vs_oracle(dt):
    """ddimx_conv_in_bwd / ddimx_conv_out_bwd against autograd of Conv2d(2->32) and Conv2d(32->2) on `x + hidden[0]`
    (models/diffusion.py:189-208,283-292)."""
    lib = _lib.load()
    b, h, w, c0, cio = 2, 40, 256, 32, 2
    sd = synth.fill_state_dict({"ci.weight": torch.empty(c0, cio, 3, 3), "ci.bias": torch.empty(c0),
                                "co.weight": torch.empty(cio, c0, 3, 3), "co.bias": torch.empty(cio)})
    x = synth.gaussian("ebw.x", (b, cio, h, w))
    dy = synth.gaussian("ebw.dy", (b, c0, h, w))
    wr, br = sd["ci.weight"].clone().requires_grad_(True), sd["ci.bias"].clone().requires_grad_(True)
    dyn = G.to_nhwc(dy, dt)
    torch.nn.functional.conv2d(x, wr, br, padding=1).backward(G.from_nhwc(dyn, dt))
    part = torch.empty(int(lib.ddimx_edge_bwd_workspace_floats(dt, b, c0, cio, h, w)), device=G.dev())
    d_w, d_b = torch.full((c0, cio, 3, 3), float("nan"), device=G.dev()), torch.full((c0,), float("nan"), device=G.dev())
    xg = G.g(x)
    _lib.check(lib.ddimx_conv_in_bwd(dt, _lib.ptr(dyn), _lib.ptr(xg), _lib.ptr(part), _lib.ptr(d_w), _lib.ptr(d_b), b, cio, c0, h, w, _lib.stream()))
    G.check_close(d_w.cpu(), wr.grad, G.F32, "conv_in dW", scale=4.0)
    G.check_close(d_b.cpu(), br.grad, G.F32, "conv_in db", scale=4.0)
    # output conv
    a, s2 = synth.gaussian("ebw.a", (b, c0, h, w)), synth.gaussian("ebw.s", (b, c0, h, w)) * 0.7
    d_eps = synth.gaussian("ebw.de", (b, cio, h, w))
    an, sn = G.to_nhwc(a, dt), G.to_nhwc(s2, dt)
    ssum = (G.from_nhwc(an, dt) + G.from_nhwc(sn, dt)).requires_grad_(True)
    wo, bo = sd["co.weight"].clone().requires_grad_(True), sd["co.bias"].clone().requires_grad_(True)
    torch.nn.functional.conv2d(ssum, wo, bo, padding=1).backward(d_eps)
    wp = G.pack_conv(sd["co.weight"], G.F32)
    dsn = torch.empty_like(an)
    d_wo, d_bo = torch.full((cio, c0, 3, 3), float("nan"), device=G.dev()), torch.full((cio,), float("nan"), device=G.dev())
    deg = G.g(d_eps)
    _lib.check(lib.ddimx_conv_out_bwd(dt, _lib.ptr(deg), _lib.ptr(an), _lib.ptr(sn), _lib.ptr(wp), _lib.ptr(dsn), _lib.ptr(part), _lib.ptr(d_wo),
                                      _lib.ptr(d_bo), b, c0, cio, h, w, _lib.stream()))
    G.check_close(G.from_nhwc(dsn, dt), ssum.grad, dt, "conv_out d(a+b)")
    G.check_close(d_wo.cpu(), wo.grad, G.F32 if dt == G.F32 else G.BF16, "conv_out dW", scale=4.0)
    G.check_close(d_bo.cpu(), bo.grad, G.F32, "conv_out db", scale=4.0)


def test_temb_backward_vs_oracle():
    """ddimx_temb_fwd_train / ddimx_temb_bwd against autograd of BetaEmbedding (models/diffusion.py:110-120)."""
    lib = _lib.load()
    shapes = {"temb.weight.0.weight": (512, 128), "temb.weight.0.bias": (512,), "temb.weight.1.weight": (512, 512),
              "temb.weight.1.bias": (512,), "temb.weight.2.weight": (4416, 512), "temb.weight.2.bias": (4416,)}
    sd = synth.fill_state_dict({k: torch.empty(s) for k, s in shapes.items()})
    sd["temb.te"] = ref_cpu.timestep_table(1000)
    t = torch.tensor([0, 999, 123, 500, 7])
    d_out = synth.gaussian("tbw.dout", (5, 4416))
    leaf = {k: (v.clone().requires_grad_(True) if k != "temb.te" else v) for k, v in sd.items()}
    out_ref = ref_cpu.beta_embedding(leaf, t)
    out_ref.backward(d_out)
    te, tg = G.g(sd["temb.te"]), t.to(G.dev())
    ws = [G.g(sd[f"temb.weight.{i}.{k}"]) for i in range(3) for k in ("weight", "bias")]
    h1, h2 = torch.empty(5, 512, device=G.dev()), torch.empty(5, 512, device=G.dev())
    out = torch.empty(5, 4416, device=G.dev())
    _lib.check(lib.ddimx_temb_fwd_train(_lib.ptr(te), _lib.ptr(tg), *[_lib.ptr(v) for v in ws], _lib.ptr(h1), _lib.ptr(h2), _lib.ptr(out),
                                        5, 128, 512, 4416, _lib.stream()))
    G.check_close(out.cpu(), out_ref.detach(), G.F32, "temb forward (training)")
    grads = [torch.full_like(v, float("nan")) for v in ws]
    dh2, dh1 = torch.empty(5, 512, device=G.dev()), torch.empty(5, 512, device=G.dev())
    dg = G.g(d_out)
    _lib.check(lib.ddimx_temb_bwd(_lib.ptr(dg), _lib.ptr(te), _lib.ptr(tg), _lib.ptr(ws[2]), _lib.ptr(ws[4]), _lib.ptr(h1), _lib.ptr(h2),
                                  _lib.ptr(dh2), _lib.ptr(dh1), *[_lib.ptr(v) for v in grads], 5, 128, 512, 4416, _lib.stream()))
    for i in range(3):
        for j, k in enumerate(("weight", "bias")):
            G.check_close(grads[2 * i + j].cpu(), leaf[f"temb.weight.{i}.{k}"].grad, G.F32, f"temb d{k}{i}", scale=4.0)


# ---- Transformer_Module alone: the `_bwd` twin of ddimx_fnet_fwd ----------------------------------------
@pytest.mark.parametrize("s_len", [4, 32, 96])
def test_fnet_bwd_per_op_vs_autograd_through_the_oracle(s_len):
    """ddimx_fnet_fwd_train + ddimx_fnet_bwd (models/diffusion.py:148-167 and autograd through it) against
    ``ref_cpu.transformer_module`` differentiated by torch autograd on the CPU, fp32 mode, dropout 0: the module output,
    the gradient w.r.t. the input tokens and EVERY transformer.* parameter gradient (102 tensors; S = 96 takes the two-GEMM
    Fourier path and the rounded-up positional table, S <= 32 the fused mixing kernel)."""
    import ctypes
    import ddim_audio_amd as D
    from ddim_audio_amd import configs
    from oracle import ref_cpu
    lib = _lib.load()
    cfg = configs.audio_config("torch.cuda.FloatTensor")
    m = synth.fill_module(D.Model(cfg)).train()
    dev = G.dev()
    t_len, b = s_len * 32, 2
    lib2 = m._ensure_handle()
    with torch.cuda.device(dev):
        m._ensure_packed(lib2, dev)
        m._ensure_packed_bwd(lib2, dev)
        pe, dh, ds = m._ensure_tables(t_len, dev)
    tok = synth.gaussian(f"fnetbwd.x{s_len}", (b, s_len, 2048))
    dout = synth.gaussian(f"fnetbwd.dy{s_len}", (b, s_len, 2048))
    to_lib = lambda v: v.view(b, s_len, 256, 8).permute(0, 1, 3, 2).contiguous()      # reference token order c*8+f -> f*256+c
    from_lib = lambda v: v.view(b, s_len, 8, 256).permute(0, 1, 3, 2).reshape(b, s_len, 2048)
    x = to_lib(tok).to(dev)
    d_out = to_lib(dout).reshape(b * s_len, 2048).to(dev)
    ws = torch.empty(int(lib.ddimx_train_workspace_bytes(m._handle, b, t_len)), dtype=torch.uint8, device=dev)
    tape = torch.empty(int(lib.ddimx_train_tape_bytes(m._handle, b, t_len)), dtype=torch.uint8, device=dev)
    out = torch.full((b * s_len, 2048), float("nan"), device=dev)
    d_x = torch.full((b * s_len, 2048), float("nan"), device=dev)
    total, layout = m._grad_layout(lib)
    grads = torch.full((total,), float("nan"), device=dev)
    tb = _lib.DdimxTables(pe.data_ptr(), dh.data_ptr(), ds.data_ptr())
    st = _lib.stream()
    _lib.check(lib.ddimx_fnet_fwd_train(m._handle, _lib.ptr(m._packed), ctypes.byref(tb), _lib.ptr(ws), ws.numel(), _lib.ptr(tape),
                                        tape.numel(), _lib.ptr(x), _lib.ptr(out), b, t_len, 0.0, 1234, st))
    _lib.check(lib.ddimx_fnet_bwd(m._handle, _lib.ptr(m._packed), _lib.ptr(m._packed_bwd), ctypes.byref(tb), _lib.ptr(ws), ws.numel(),
                                  _lib.ptr(tape), tape.numel(), _lib.ptr(x), _lib.ptr(d_out), _lib.ptr(d_x), _lib.ptr(grads), b, t_len,
                                  0.0, 1234, st))
    torch.cuda.synchronize()
    # oracle: autograd through the CPU restatement
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items() if k.startswith("transformer.")}
    for v in sd.values():
        v.requires_grad_(True)
    xin = tok.clone().requires_grad_(True)
    kw = cfg.model.transformers.kwargs
    y = ref_cpu.transformer_module(sd, xin, kw.num_hidden_layers, kw.layer_norm_eps)
    y.backward(dout)
    G.check_close(from_lib(out.cpu()), y.detach(), G.F32, f"fnet train fwd S={s_len}")
    G.check_close(from_lib(d_x.cpu()), xin.grad, G.F32, f"fnet d_tokens S={s_len}", scale=5.0)
    checked = 0
    for (name, _), (off, numel, shape) in zip(m.named_parameters(), layout):
        g = grads[off:off + numel].cpu().view(shape)
        if not name.startswith("transformer."):
            assert torch.isnan(g).all(), f"{name}: ddimx_fnet_bwd must leave other gradients untouched"
            continue
        want = sd[name].grad
        rms = float(want.double().square().mean().sqrt()) + 1e-30
        err = float((g.double() - want.double()).abs().max()) / rms
        assert torch.isfinite(g).all() and err <= 2e-3, (name, err)
        checked += 1
    assert checked == 4 + 8 * kw.num_hidden_layers + 2
