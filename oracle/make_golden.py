"""Generate tests/golden/*.npz by importing the REAL reference (authoring container only).

Test infrastructure.  Run:  python oracle/make_golden.py   (needs /root/reference; CPU only).
The reference never travels: only the inputs' hash tags and the outputs it produced are stored.
Harness-side shims (none touches the reference's files):
  * empty stand-in modules for the un-vendored git submodules (UPU / SST) that the reference imports
    but does not use on this path;
  * dtype strings overridden to CPU tensor types;
  * ``torch.Tensor.type`` mapped from ``torch.cuda.*`` strings to the CPU types and ``.to("cpu")``
    made to return a copy while ``generalized_steps`` runs, which reproduces the list semantics the
    sampler has on a GPU.
"""
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("DDIMX_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

from ddim_audio_amd import configs, synth  # noqa: E402


def _stub(name, **attrs):
    parts = name.split(".")
    for i in range(1, len(parts) + 1):
        n = ".".join(parts[:i])
        if n not in sys.modules:
            sys.modules[n] = types.ModuleType(n)
    for k, v in attrs.items():
        setattr(sys.modules[name], k, v)


_stub("UPU.layers.normalize.groupnorm", GroupNorm1D=object)
_stub("UPU.signal.denoise", denoise_2d=None)
_stub("SST.utils", AudioDataset=object)
_stub("SST.utils.wav2img", limit_length_img=None, pfft2img=None, pfft2wav=None)

import models.diffusion as ref_model  # noqa: E402  (the reference)
import functions.denoising as ref_denoise  # noqa: E402
import functions.losses as ref_losses  # noqa: E402
import models.ema as ref_ema  # noqa: E402
import functions as ref_functions  # noqa: E402
import runners.diffusion as ref_runner  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
CPU = "torch.FloatTensor"


def np32(t):
    return t.detach().to(torch.float32).cpu().numpy().copy()


def filled(module, seed=0, prefix=""):
    sd = {prefix + k: v for k, v in module.named_parameters()}
    synth.fill_state_dict(sd, seed)
    return module.eval()


def g1_schedule(out):
    cfg = configs.audio_config(CPU)
    args = types.SimpleNamespace()
    d = ref_runner.Diffusion(args, cfg, device=torch.device("cpu"))
    out["betas"] = np32(d.betas)
    out["alphas"] = np32(d.alphas)
    for name in ("quad", "const", "jsd", "sigmoid"):
        out["betas64_" + name] = ref_runner.get_beta_schedule(
            name, beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000)
    # seq construction of sample_image for uniform / quad skip types
    for skip_type, steps in (("uniform", 100), ("uniform", 50), ("quad", 20)):
        d.args = types.SimpleNamespace(sample_type="generalized", skip_type=skip_type, timesteps=steps, eta=0.0)
        seen = {}
        orig = ref_denoise.generalized_steps
        ref_denoise.generalized_steps = lambda x, seq, model, a, **kw: seen.update(seq=list(seq)) or ([x], [])
        try:
            d.sample_image(torch.zeros(1), None)
        finally:
            ref_denoise.generalized_steps = orig
        out[f"seq_{skip_type}_{steps}"] = np.asarray(seen["seq"], dtype=np.int64)
    out["lr_steps"] = np.asarray([0, 1, 999, 1000, 10 ** 4, 10 ** 5], dtype=np.int64)
    for warm in (1000, 10000):
        opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
        sch = ref_functions.get_scheduler(types.SimpleNamespace(warmup=warm), opt)
        out[f"lr_factor_{warm}"] = np.asarray([sch.lr_lambdas[0](int(s)) for s in out["lr_steps"]])


def g2_blocks(out):
    for c, (h, w) in ((32, (16, 8)), (64, (5, 7)), (96, (8, 8)), (128, (4, 8)), (192, (3, 5)), (256, (2, 8))):
        rb = filled(ref_model.Residual_Block(channels=c, kernel_size=3), prefix=f"rb{c}.")
        x = synth.gaussian(f"rb{c}.x", (2, c, h, w))
        temb = synth.gaussian(f"rb{c}.temb", (2, c)) * 0.5
        with torch.no_grad():
            out[f"rb{c}_y"] = np32(rb(x, temb))
    for cin, cout, (h, w) in ((32, 64, (8, 16)), (96, 128, (6, 10)), (192, 256, (4, 8))):
        dn = filled(ref_model.Downsample(in_channels=cin, out_channels=cout), prefix=f"down{cin}.")
        up = filled(ref_model.Upsample(in_channels=cout, out_channels=cin), prefix=f"up{cout}.")
        with torch.no_grad():
            out[f"down{cin}_y"] = np32(dn(synth.gaussian(f"down{cin}.x", (2, cin, h, w))))
            out[f"up{cout}_y"] = np32(up(synth.gaussian(f"up{cout}.x", (2, cout, h // 2, w // 2))))


def g456_model(out):
    cfg = configs.audio_config(CPU)
    model = filled(ref_model.Model(cfg))
    out["n_state_keys"] = np.asarray(len(model.state_dict()))
    out["state_keys"] = np.asarray(list(model.state_dict().keys()))
    out["state_shapes"] = np.asarray([",".join(map(str, v.shape)) for v in model.state_dict().values()])
    te = model.temb.te
    out["te_rows"] = np32(te[[0, 1, 2, 499, 999]])
    with torch.no_grad():
        out["temb_t"] = np.asarray([0, 1, 499, 999], dtype=np.int64)
        out["temb_y"] = np32(model.temb(torch.tensor([0, 1, 499, 999])))
        for s in (4, 32, 96):
            # fresh embedding cache per S: the reference's pos-enc cache test is inverted
            model.transformer.embedding.te = None
            tok = synth.gaussian(f"fnet.x{s}", (1, s, 2048))
            out[f"fnet_s{s}_y"] = np32(model.transformer(tok))
        for tlen, tt in ((32, [0, 999]), (64, [500, 37])):
            model.transformer.embedding.te = None
            x = synth.gaussian(f"model.x{tlen}", (2, 2, tlen, 256))
            out[f"model_T{tlen}_t"] = np.asarray(tt, dtype=np.int64)
            out[f"model_T{tlen}_y"] = np32(model(x, torch.tensor(tt)))
    del model


class _GpuSemantics:
    """Make the reference sampler runnable on CPU with the list semantics it has on a GPU."""

    def __enter__(self):
        self._type, self._to = torch.Tensor.type, torch.Tensor.to
        orig_type, orig_to = self._type, self._to

        def type_(t, dtype=None, *a, **k):
            if isinstance(dtype, str) and dtype.startswith("torch.cuda."):
                dtype = dtype.replace("torch.cuda.", "torch.")
            return orig_type(t, dtype, *a, **k)

        def to_(t, *a, **k):
            if a == ("cpu",) and not k:
                return t.clone()
            if a == ("cuda",) and not k:  # ddpm_steps re-uploads xs[-1] every iteration (:72)
                return t
            return orig_to(t, *a, **k)

        torch.Tensor.type, torch.Tensor.to = type_, to_
        return self

    def __exit__(self, *exc):
        torch.Tensor.type, torch.Tensor.to = self._type, self._to


def g7_sampler(out):
    _, alphas = None, torch.from_numpy(out["alphas"])
    fake = lambda x, t: 0.1 * x + 0.01 * t.float().view(-1, 1, 1, 1)  # noqa: E731
    x = synth.gaussian("sampler.fake.x", (2, 2, 8, 16))
    cases = {"u10": list(range(0, 1000, 100)), "quad8": [int(s) for s in np.linspace(0, np.sqrt(800), 8) ** 2]}
    for name, seq in cases.items():
        for sel_name, sel in (("all", None), ("last", [-1]), ("mix", [0, 3, -2])):
            with _GpuSemantics():
                xs, x0 = ref_denoise.generalized_steps(x.clone(), seq, fake, alphas, sel, eta=0.0)
            out[f"samp_{name}_{sel_name}_xs"] = np.stack([np32(v) for v in xs])
            out[f"samp_{name}_{sel_name}_x0"] = np.stack([np32(v) for v in x0])
        out[f"samp_{name}_seq"] = np.asarray(seq, dtype=np.int64)
    # ddpm_steps (ancestral sampler) with the analytic model and a deterministic noise sequence
    betas = torch.from_numpy(np.load(os.path.join(OUT, "schedule.npz"))["betas"])
    for name, seq in cases.items():
        calls = {"k": 0}

        def det_noise(ref):
            calls["k"] += 1
            return synth.gaussian(f"ddpm.noise.{name}.{calls['k'] - 1}", tuple(ref.shape))

        orig_randn = torch.randn_like
        torch.randn_like = det_noise
        try:
            with _GpuSemantics():
                xs, x0 = ref_denoise.ddpm_steps(x.clone(), seq, fake, betas, None)
        finally:
            torch.randn_like = orig_randn
        out[f"ddpm_{name}_xs"] = np.stack([np32(v) for v in xs])
        out[f"ddpm_{name}_x0"] = np.stack([np32(v) for v in x0])
    # tiny real model, 10 steps: record every model input and the final sample
    cfg = configs.tiny_config(CPU)
    model = filled(ref_model.Model(cfg), seed=3)
    calls = []

    def traced(xt, t):
        calls.append(np32(xt))
        return model(xt, t)

    x = synth.gaussian("sampler.tiny.x", (2, 2, 16, 32))
    seq = list(range(0, 1000, 100))
    with _GpuSemantics(), torch.no_grad():
        xs, x0 = ref_denoise.generalized_steps(x.clone(), seq, traced, alphas, None, eta=0.0)
    out["samp_tiny_inputs"] = np.stack(calls)
    out["samp_tiny_xs"] = np.stack([np32(v) for v in xs])
    out["samp_tiny_x0"] = np.stack([np32(v) for v in x0])
    with torch.no_grad():
        model.transformer.embedding.te = None
        out["tiny_model_y"] = np32(model(x, torch.tensor([7, 901])))
    # G8: loss, gradients, EMA on the tiny model (dropout off: eval mode)
    model.transformer.embedding.te = None
    e = synth.gaussian("train.tiny.e", (2, 2, 16, 32))
    t = torch.tensor([123, 876])
    loss = ref_losses.noise_estimation_loss(model, x, t, e, alphas)
    out["train_loss"] = np32(loss)
    with torch.no_grad():
        out["train_loss_keepdim"] = np32(ref_losses.noise_estimation_loss(model, x, t, e, alphas, keepdim=True))
    loss.backward()
    gsq = sum(float(p.grad.double().square().sum()) for p in model.parameters())
    out["train_grad_norm"] = np.asarray(gsq ** 0.5)
    for name in ("down_modules.0.weight", "down_modules.1.0.conv.0.weight", "up_modules.0.0.norm.2.weight",
                 "transformer.encoder.layer.0.intermediate.dense.bias", "temb.weight.2.bias"):
        out["train_grad::" + name] = np32(dict(model.named_parameters())[name].grad).reshape(-1)[:64]
    ema = ref_ema.EMAHelper(mu=0.9999)
    ema.register(model)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.01 * p.grad)
    ema.update(model)
    out["ema_shadow::down_modules.0.weight"] = np32(ema.shadow["down_modules.0.weight"]).reshape(-1)[:64]
    out["ema_param::down_modules.0.weight"] = np32(dict(model.named_parameters())["down_modules.0.weight"]).reshape(-1)[:64]


def _grad_digest(out, tag, model):
    """Per-parameter gradient digest: L2 norm (float64) + a strided sample of <= 256 elements."""
    names, norms = [], []
    for name, p in model.named_parameters():
        g = p.grad.detach().reshape(-1)
        names.append(name)
        norms.append(float(g.double().square().sum()) ** 0.5)
        stride = max(1, g.numel() // 256)
        out[f"{tag}_g::{name}"] = np32(g[::stride][:256])
    out[f"{tag}_names"] = np.asarray(names)
    out[f"{tag}_gnorms"] = np.asarray(norms, dtype=np.float64)


def g8_train(out):
    """Training-mode forward/backward of the real reference (dropout probability 0 so that it is deterministic),
    then two optimisation steps through the reference's own get_optimizer / get_scheduler / EMAHelper and
    torch's clip_grad_norm_ -- the tail of Diffusion.train_step (runners/diffusion.py:130-173) with fixed (x0, e, t).
    AdaBelief's source is absent, so the default group uses Adam with the default group's hyper-parameters."""
    alphas = torch.from_numpy(np.load(os.path.join(OUT, "schedule.npz"))["alphas"])
    for tag, cfgd, shape, tt, seed in (("tiny", configs.tiny_dict(CPU), (2, 2, 16, 32), [123, 876], 3),
                                       ("audio", configs.audio_dict(CPU), (2, 2, 32, 256), [37, 911], 0)):
        cfgd["model"]["transformers"]["kwargs"]["hidden_dropout_prob"] = 0.0
        cfg = configs.dict2namespace(cfgd)
        model = filled(ref_model.Model(cfg), seed=seed)
        model.train()
        x0 = synth.gaussian(f"train.{tag}.x0", shape)
        e = synth.gaussian(f"train.{tag}.e", shape)
        t = torch.tensor(tt)
        out[f"{tag}_t"] = np.asarray(tt, dtype=np.int64)
        loss = ref_losses.noise_estimation_loss(model, x0, t, e, alphas)
        out[f"{tag}_loss"] = np.asarray(float(loss), dtype=np.float64)
        loss.backward()
        _grad_digest(out, tag, model)
        if tag != "tiny":
            continue
        # ---- two full optimisation steps on the tiny model
        opt_cfg = cfg.optimization.optimizer
        opt_cfg.default.optimizer = "Adam"
        groups = ref_runner.classify_group(opt_cfg, model)
        optimizers = {k: ref_functions.get_optimizer(v.config, v.params) for k, v in groups.items()}
        schedulers = {k: ref_functions.get_scheduler(groups[k].config, o) for k, o in optimizers.items()}
        clip_groups = ref_runner.classify_group(cfg.optimization.grad_norm, model)
        ema = ref_ema.EMAHelper(mu=0.9999)
        ema.register(model)
        for it in range(2):
            if it:
                model.transformer.embedding.te = None
                x0 = synth.gaussian(f"train.{tag}.x0.{it}", shape)
                e = synth.gaussian(f"train.{tag}.e.{it}", shape)
                loss = ref_losses.noise_estimation_loss(model, x0, torch.tensor([5, 994]), e, alphas)
                for o in optimizers.values():
                    o.zero_grad()
                loss.backward()
            out[f"step{it}_loss"] = np.asarray(float(loss), dtype=np.float64)
            for name, g in clip_groups.items():
                out[f"step{it}_norm_{name}"] = np.asarray(float(torch.nn.utils.clip_grad_norm_(g.params, g.config.grad_clip)))
            for o in optimizers.values():
                o.step()
            for s in schedulers.values():
                s.step()
            ema.update(model)
            for name, p in model.named_parameters():
                stride = max(1, p.numel() // 64)
                out[f"step{it}_p::{name}"] = np32(p.reshape(-1)[::stride][:64])
                out[f"step{it}_ema::{name}"] = np32(ema.shadow[name].reshape(-1)[::stride][:64])
        out["step_lrs"] = np.asarray([o.param_groups[0]["lr"] for o in optimizers.values()], dtype=np.float64)
        out["step_groups"] = np.asarray(list(optimizers.keys()))
        out["step_group_sizes"] = np.asarray([len(g.params) for g in groups.values()])


def g9_checkpoint(out):
    """A checkpoint WRITTEN BY THE REFERENCE (Diffusion.train_step, runners/diffusion.py:130-199, saves at step 1) for the
    micro configuration, copied to tests/golden/ckpt_micro.pth, and what the reference's own sampling path makes of it
    (Diffusion.sample :293-313,331: strict load, EMA swap-in, eval; then generalized_steps)."""
    import logging
    import shutil
    import tempfile
    cfgd = configs.micro_dict(CPU)
    cfgd["model"]["transformers"]["kwargs"]["hidden_dropout_prob"] = 0.0
    cfg = configs.dict2namespace(cfgd)
    cfg.tb_logger = types.SimpleNamespace(add_scalar=lambda *a, **k: None)
    tmp = tempfile.mkdtemp()
    args = types.SimpleNamespace(log_path=tmp)
    runner = ref_runner.Diffusion(args, cfg, device=torch.device("cpu"))
    model = filled(ref_model.Model(cfg), seed=11)
    optimizers, schedulers = {}, {}
    for name, p_opt in ref_runner.classify_group(cfg.optimization.optimizer, model).items():
        optimizers[name] = optimizer = ref_functions.get_optimizer(p_opt.config, p_opt.params)
        scheduler = ref_functions.get_scheduler(p_opt.config, optimizer)
        if scheduler:
            schedulers[name] = scheduler
    grad_group = dict(ref_runner.classify_group(cfg.optimization.grad_norm, model))
    ema_helper = ref_ema.EMAHelper(mu=cfg.model.ema_rate)
    ema_helper.register(model)
    x = synth.gaussian("ckpt.x", (2, 2, 8, 8))
    torch.manual_seed(99)
    logging.disable(logging.CRITICAL)
    runner.train_step(model, x, optimizers, schedulers, grad_group, ema_helper, 1, 0)   # step 1 -> writes ckpt_1.pth, ckpt.pth
    logging.disable(logging.NOTSET)
    dst = os.path.join(OUT, "ckpt_micro.pth")
    shutil.copyfile(os.path.join(tmp, "ckpt.pth"), dst)
    states = torch.load(dst, weights_only=False)
    out["n_states"] = np.asarray(len(states))
    out["epoch"], out["step"] = np.asarray(states[2]), np.asarray(states[3])
    out["optim_groups"] = np.asarray(list(optimizers.keys()))
    last = list(optimizers.values())[-1]
    out["optim_last_n_state"] = np.asarray(len(states[1]["state"]))
    out["optim_last_lr"] = np.asarray(states[1]["param_groups"][0]["lr"], dtype=np.float64)
    k0 = sorted(states[1]["state"].keys())[0]
    out["optim_last_exp_avg0"] = np32(states[1]["state"][k0]["exp_avg"]).reshape(-1)[:64]
    out["optim_last_step0"] = np.asarray(float(states[1]["state"][k0]["step"]))
    # --- the reference's sampling path on that file
    m2 = ref_model.Model(cfg)
    m2.load_state_dict(states[0], strict=True)
    eh = ref_ema.EMAHelper(mu=cfg.model.ema_rate)
    eh.register(m2)
    eh.load_state_dict(states[-1])
    eh.ema(m2)
    m2.eval()
    xs0 = synth.gaussian("ckpt.sample.x", (2, 2, 8, 8))
    seq = list(range(0, 1000, 100))
    with _GpuSemantics():
        xs, x0 = ref_denoise.generalized_steps(xs0.clone(), seq, m2, runner.alphas, None, eta=0.0)
    out["sample_final"] = np32(xs[-1])
    out["sample_x0_last"] = np32(x0[-1])
    with torch.no_grad():
        out["ema_model_y"] = np32(m2(synth.gaussian("ckpt.fwd.x", (2, 2, 8, 8)), torch.tensor([3, 777])))
    shutil.rmtree(tmp)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    if sys.argv[1:] == ["ckpt"]:  # only the checkpoint fixture
        ck = {}
        g9_checkpoint(ck)
        np.savez_compressed(os.path.join(OUT, "ckpt.npz"), **ck)
        for f in ("ckpt.npz", "ckpt_micro.pth"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
        return
    if sys.argv[1:] == ["train"]:  # only the training fixtures (the others are unchanged)
        train = {}
        g8_train(train)
        np.savez_compressed(os.path.join(OUT, "train.npz"), **train)
        print("train.npz", os.path.getsize(os.path.join(OUT, "train.npz")))
        return
    sched, blocks, model, sampler = {}, {}, {}, {}
    g1_schedule(sched)
    np.savez_compressed(os.path.join(OUT, "schedule.npz"), **sched)
    g2_blocks(blocks)
    np.savez_compressed(os.path.join(OUT, "blocks.npz"), **blocks)
    g456_model(model)
    np.savez_compressed(os.path.join(OUT, "model.npz"), **model)
    sampler["alphas"] = sched["alphas"]
    g7_sampler(sampler)
    sampler.pop("alphas")
    np.savez_compressed(os.path.join(OUT, "sampler.npz"), **sampler)
    train = {}
    g8_train(train)
    np.savez_compressed(os.path.join(OUT, "train.npz"), **train)
    ck = {}
    g9_checkpoint(ck)
    np.savez_compressed(os.path.join(OUT, "ckpt.npz"), **ck)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
