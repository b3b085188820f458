"""CPU-only tests of the host logic and of the C-ABI surface (no compute calls: there is no GPU here)."""
import os
import re

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ddim_audio_amd import _lib, configs, dist as ddist, schedule

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(REPO, "include", "ddimx.h")).read()
    declared = set(re.findall(r"\b(ddimx_[A-Za-z0-9_]+)\s*\(", hdr)) - {"ddimx_ctx"}
    assert len(declared) >= 25
    lib = _lib.load()  # raises if libddimx.so was not built
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ddimx.h but not exported"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert lib.ddimx_abi_version() == 2


def test_plan_matches_host_inventory_and_sizes():
    """ddimx_create builds the packing plan without a GPU; names/sizes must mirror the state_dict."""
    from ddim_audio_amd.model import Model
    for cfg in (configs.audio_config("torch.FloatTensor"), configs.tiny_config("torch.BFloat16Tensor")):
        m = Model(cfg)
        lib = m._ensure_handle()  # cross-checks every (name, numel) against the library's plan
        assert lib.ddimx_num_params(m._handle) == len(m.state_dict())
        assert lib.ddimx_packed_bytes(m._handle) > 0
        need = lib.ddimx_workspace_bytes(m._handle, 2, 64)
        assert 0 < need < 2 ** 34
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 2, 16, 32), torch.zeros(1, dtype=torch.long))


def test_unsupported_configs_fail_loudly():
    from ddim_audio_amd.model import Model
    cfg = configs.tiny_config("torch.FloatTensor")
    cfg.model.ch = [32, 48, 96]
    with pytest.raises(RuntimeError, match="multiple of 32"):
        Model(cfg)._ensure_handle()
    cfg = configs.tiny_config("torch.cuda.HalfTensor")
    with pytest.raises(NotImplementedError):
        Model(cfg)


def test_ddim_coefficients_reproduce_golden_trajectories(golden):
    """Host scalar logic (seq_next, alpha lookup, c1/c2) drives the analytic-model trajectory of the
    golden set; the per-element update is redone here in numpy fp32 with the same op order."""
    g, gs = golden("sampler"), golden("schedule")
    alphas = torch.from_numpy(gs["alphas"])
    from ddim_audio_amd import synth
    x0 = synth.gaussian("sampler.fake.x", (2, 2, 8, 16)).numpy()
    for name in ("u10", "quad8"):
        seq = g[f"samp_{name}_seq"].tolist()
        coef = schedule.ddim_coefficients(seq, alphas, 0.0).astype(np.float32)
        assert coef.shape == (len(seq), 6) and coef[:, 0].tolist() == [float(s) for s in reversed(seq)]
        assert coef[-1, 3] == 1.0 and coef[-1, 4] == 0.0  # last step: at_next = 1 -> result is the x0 prediction
        x = x0.copy()
        exs, ex0 = g[f"samp_{name}_all_xs"], g[f"samp_{name}_all_x0"]
        for k, (t, s1, s2, s3, c2, c1) in enumerate(coef):
            e = (np.float32(0.1) * x + np.float32(0.01) * np.float32(t)).astype(np.float32)
            p0 = ((x + (-s1) * e) / s2).astype(np.float32)
            x = (p0 * s3 + c2 * e).astype(np.float32)
            assert np.allclose(p0, ex0[k], rtol=2e-6, atol=2e-6 * np.abs(ex0[k]).max())
            assert np.allclose(x, exs[k + 1], rtol=2e-6, atol=2e-6 * np.abs(exs[k + 1]).max())
    c = schedule.ddim_coefficients(list(range(0, 1000, 100)), alphas, 1.0)
    assert np.all(c[:-1, 5] > 0) and np.all(np.isfinite(c))


def test_shard_bounds_cover_batch():
    for n in (1, 2, 7, 8, 32, 33):
        for world in (1, 2, 3, 8):
            spans = [ddist.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.arange(n * 6, dtype=torch.float32).reshape(n, 2, 3)  # every rank builds the same noise tensor
        fake_sampler = lambda shard: shard * 2.0 + 1.0  # stand-in for generalized_steps on this rank's GPU  # noqa: E731
        out = ddist.sample_sharded(full, fake_sampler, gather=True)
        lo, hi = ddist.shard_bounds(n, rank, world)
        q.put((rank, torch.equal(out, full * 2.0 + 1.0), hi - lo))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [5, 8])
def test_sharded_sampling_world2_gloo(n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000 + n
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert sum(k for _, _, k in res) == n


def test_checkpoint_wire_format_roundtrip(tmp_path, golden):
    """[model_sd, optimizer_sd, epoch, step, ema_shadow] as the reference writes it; strict reload + EMA swap."""
    from ddim_audio_amd import synth
    from ddim_audio_amd.checkpoint import load_for_sampling, save_checkpoint
    from ddim_audio_amd.ema import EMAHelper
    from ddim_audio_amd.model import Model
    cfg = configs.tiny_config("torch.FloatTensor")
    m = synth.fill_module(Model(cfg), seed=1)
    ema = EMAHelper(mu=0.9999)
    ema.register(m)
    for k in ema.shadow:
        ema.shadow[k] = ema.shadow[k] * 0.5  # make the shadow differ from the parameters
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    states = save_checkpoint(str(tmp_path), m, opt, epoch=3, step=5000, ema_helper=ema)
    assert len(states) == 5 and states[2] == 3 and states[3] == 5000
    assert os.path.exists(tmp_path / "ckpt_5000.pth") and os.path.exists(tmp_path / "ckpt.pth")
    raw = torch.load(tmp_path / "ckpt.pth", weights_only=False)
    assert list(raw[0].keys()) == list(m.state_dict().keys()) and set(raw[-1].keys()) == {k for k, _ in m.named_parameters()}
    m2, e2 = load_for_sampling(str(tmp_path), Model(cfg), use_ema=True)
    assert not m2.training and m2._dirty  # weights swapped in: the packed copy must be rebuilt
    for (k, p), (_, q) in zip(m2.named_parameters(), m.named_parameters()):
        assert torch.equal(p, q * 0.5), k
    m3, e3 = load_for_sampling(str(tmp_path), Model(cfg), use_ema=False, ckpt_id=5000)
    assert e3 is None and all(torch.equal(p, q) for p, q in zip(m3.parameters(), m.parameters()))


def test_ddpm_coefficients_reproduce_golden(golden):
    """Host table of the ancestral sampler: numpy fp32 replay of the analytic-model trajectory (same op order)."""
    from ddim_audio_amd import synth
    g, gs = golden("sampler"), golden("schedule")
    betas = torch.from_numpy(gs["betas"])
    x0 = synth.gaussian("sampler.fake.x", (2, 2, 8, 16)).numpy()
    for name in ("u10", "quad8"):
        seq = g[f"samp_{name}_seq"].tolist()
        coef = schedule.ddpm_coefficients(seq, betas)
        assert coef.shape == (len(seq), 7) and coef.dtype == np.float32
        x = x0.copy()
        for k, (t, a0, a1, m1, m2, den, sig) in enumerate(coef):
            e = (np.float32(0.1) * x + np.float32(0.01) * np.float32(t)).astype(np.float32)
            p0 = np.clip((a0 * x).astype(np.float32) - (a1 * e).astype(np.float32), -1, 1).astype(np.float32)
            mean = (((m1 * p0).astype(np.float32) + (m2 * x).astype(np.float32)).astype(np.float32) / den).astype(np.float32)
            nz = synth.gaussian(f"ddpm.noise.{name}.{k}", x.shape).numpy()
            x = (mean + (sig * nz).astype(np.float32)).astype(np.float32)
            assert np.array_equal(p0, g[f"ddpm_{name}_x0"][k]), (name, k)
            assert np.array_equal(x, g[f"ddpm_{name}_xs"][k + 1]), (name, k)


def test_product_package_never_touches_the_oracle_or_reference():
    """The oracle is test infrastructure: nothing under ddim_audio_amd/ (or bench.py outside cpu_baseline) may import it."""
    pkg = os.path.join(REPO, "ddim_audio_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(root, f)).read()
                code = "\n".join(ln for ln in src.split("\n") if "import" in ln or "open(" in ln or "exec" in ln)
                assert "oracle" not in code, os.path.join(root, f)  # docstrings may mention it; code may not use it
                assert "/root/reference" not in src, os.path.join(root, f)
    bench = open(os.path.join(REPO, "bench.py")).read()
    assert bench.count("from oracle import") == 1 and "def cpu_baseline" in bench


def _grad_sync_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)  # stand-in for the backward's flat gradient buffer
        ddist.make_grad_sync(bucket_mb=1)(flat)                      # 1 MiB buckets -> several async all-reduces? (small: 1)
        tiny = torch.arange(300001, dtype=torch.float32) + rank      # > 1 bucket at 1 MiB (262144 floats)
        ddist.make_grad_sync(bucket_mb=1)(tiny)
        want = torch.arange(1000, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
        want2 = torch.arange(300001, dtype=torch.float32) + (world - 1) / 2
        q.put((rank, torch.allclose(flat, want), torch.allclose(tiny, want2)))
    finally:
        dist.destroy_process_group()


def test_training_grad_sync_world2_gloo():
    """The data-parallel collective of the training step: bucketed all-reduce + 1/world of the flat gradient buffer."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_grad_sync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(a and b for _, a, b in res)


def _staged_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctypes
        from ddim_audio_amd import _lib, configs
        import ddim_audio_amd as D
        m = D.Model(configs.tiny_config("torch.FloatTensor"))
        lib = m._ensure_handle()
        rng = (ctypes.c_longlong * 6)()
        _lib.check(lib.ddimx_grad_buckets(m._handle, rng))           # the real bucket ranges of the plan-ordered buffer
        ranges = [(rng[2 * i], rng[2 * i + 1]) for i in range(3)]
        total = int(lib.ddimx_grad_floats(m._handle))
        base = torch.arange(total, dtype=torch.float32) % 1013
        flat = base * (rank + 1)
        sync = ddist.make_grad_sync(bucket_mb=1)
        out = sync.staged(flat, ranges, None)                        # CPU rehearsal of the overlapped path: no events
        want = base * (sum(range(1, world + 1)) / world)
        tiles = sorted(ranges)
        covers = tiles[0][0] == 0 and tiles[-1][1] == total and all(a[1] == b[0] for a, b in zip(tiles, tiles[1:]))
        q.put((rank, bool(torch.allclose(out, want)), covers, ranges))
    finally:
        dist.destroy_process_group()


def test_staged_grad_sync_world2_gloo_with_the_plans_bucket_ranges():
    """The bucket plumbing of the overlapped data-parallel path (``make_grad_sync(...).staged``) over the REAL
    ``ddimx_grad_buckets`` ranges of a model plan, two gloo ranks: every bucket is sliced, all-reduced and averaged, the three
    ranges tile the flat buffer, both ranks end with the rank mean.  (The RCCL form of the same function runs on the GPU
    box with a 1-rank group: tests/test_gpu_configs.py::test_staged_grad_sync_runs_on_rccl.)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_staged_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok and covers for _, ok, covers, _ in res), res
    assert res[0][3] == res[1][3]


def test_classify_group_and_antithetic_timesteps():
    """Parameter routing by top-level module name (runners/diffusion.py:71-87) and the antithetic draw (:141-142)."""
    from ddim_audio_amd import train
    from ddim_audio_amd.model import Model
    cfg = configs.tiny_config("torch.FloatTensor")
    m = Model(cfg)
    groups = train.classify_group(cfg.optimization.optimizer, m)
    names = {id(p): n for n, p in m.named_parameters()}
    assert set(groups) == {"transformer", "default"}
    assert all(names[id(p)].startswith("transformer.") for p in groups["transformer"].params)
    assert not any(names[id(p)].startswith("transformer.") for p in groups["default"].params)
    assert len(groups["transformer"].params) + len(groups["default"].params) == len(names)
    assert groups["default"].config.lr == 0.0003 and not hasattr(groups["default"].config, "top_level_name")
    assert hasattr(cfg.optimization.optimizer.default, "top_level_name")  # the caller's config is left intact
    clip = train.classify_group(cfg.optimization.grad_norm, m)
    assert list(clip) == ["default"] and len(clip["default"].params) == len(names)
    g = torch.Generator().manual_seed(5)
    t = train.antithetic_timesteps(5, 1000, g)
    assert t.shape == (5,) and int(t[0] + t[3]) == 999 and int(t[1] + t[4]) == 999


# ---- checkpoint wire format: a file written by the reference's own train_step (tests/golden/ckpt_micro.pth) ----------------
def _micro_cpu():
    from ddim_audio_amd.model import Model
    d = configs.micro_dict("torch.FloatTensor")
    d["model"]["transformers"]["kwargs"]["hidden_dropout_prob"] = 0.0
    cfg = configs.dict2namespace(d)
    return cfg, Model(cfg)


def test_reference_written_checkpoint_loads_strictly_and_oracle_reproduces_its_sampling(golden, tmp_path):
    """runners/diffusion.py:293-313,331 on the reference-written file: strict load into this package's Model, EMA swap-in
    (states[-1]), then the CPU oracle over the loaded parameters reproduces what the reference's own Model /
    generalized_steps produced from the same file (tests/golden/ckpt.npz)."""
    import shutil
    from ddim_audio_amd import checkpoint, synth
    from oracle import ref_cpu
    g = golden("ckpt")
    shutil.copyfile(os.path.join(REPO, "tests", "golden", "ckpt_micro.pth"), tmp_path / "ckpt.pth")
    cfg, m = _micro_cpu()
    states = torch.load(tmp_path / "ckpt.pth", weights_only=True)
    assert len(states) == int(g["n_states"]) == 5 and states[2] == int(g["epoch"]) and states[3] == int(g["step"])
    assert list(states[0].keys()) == list(m.state_dict().keys())
    before = {k: v.clone() for k, v in m.state_dict().items()}
    m, ema = checkpoint.load_for_sampling(str(tmp_path), m, use_ema=True, ema_rate=cfg.model.ema_rate)
    assert not m.training and ema is not None
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    assert any(not torch.equal(sd[k], before[k]) for k in sd)
    for k, v in states[-1].items():  # the EMA shadow is what sits in the parameters now
        assert torch.equal(sd[k], v)
    with torch.no_grad():
        y = ref_cpu.model_forward(sd, cfg, synth.gaussian("ckpt.fwd.x", (2, 2, 8, 8)), torch.tensor([3, 777]))
    want = torch.from_numpy(g["ema_model_y"])
    assert float((y - want).abs().max()) <= 2e-5 * float(want.std())
    alphas = schedule.make_schedule(cfg.diffusion)[1]
    xs, x0 = ref_cpu.generalized_steps(synth.gaussian("ckpt.sample.x", (2, 2, 8, 8)), list(range(0, 1000, 100)),
                                       lambda a, b: ref_cpu.model_forward(sd, cfg, a, b), alphas, None, eta=0.0)
    for got, name in ((xs[-1], "sample_final"), (x0[-1], "sample_x0_last")):
        want = torch.from_numpy(g[name])
        assert float((got - want).abs().max()) <= 2e-4 * float(want.std()), name


def test_resume_training_from_reference_and_own_checkpoints(golden, tmp_path):
    """The fixed-forward resume (runners/diffusion.py:239-254 cannot run upstream): from the reference-written file it
    restores the model, the LAST optimizer group (the only one that file holds), epoch / step, the EMA shadow and the
    LambdaLR factors; from a file written here it restores every optimizer and scheduler, and that file still loads in
    torch.optim.AdamW (the reference's optimizer class) unchanged."""
    import shutil
    from ddim_audio_amd import checkpoint, train
    g = golden("ckpt")
    shutil.copyfile(os.path.join(REPO, "tests", "golden", "ckpt_micro.pth"), tmp_path / "ckpt.pth")
    cfg, m = _micro_cpu()
    state = train.TrainingState(cfg, m)
    assert list(state.optimizers.keys()) == [str(s) for s in g["optim_groups"]]
    epoch, step = checkpoint.resume_training(str(tmp_path), m, state.optimizers, state.schedulers, state.ema_helper)
    assert (epoch, step) == (int(g["epoch"]), int(g["step"])) == (0, 1)
    last = list(state.optimizers.values())[-1]
    assert len(last.state) == int(g["optim_last_n_state"])
    first = last.state[last.param_groups[0]["params"][0]]
    assert first["step"] == int(g["optim_last_step0"]) == 1 and isinstance(first["step"], int)
    assert np.allclose(first["exp_avg"].reshape(-1)[:64].numpy(), g["optim_last_exp_avg0"], rtol=0, atol=0)
    assert abs(last.param_groups[0]["lr"] - float(g["optim_last_lr"])) <= 1e-12
    first_group = list(state.optimizers.values())[0]
    assert len(first_group.state) == 0  # the reference's file has nothing for it
    for name, sch in state.schedulers.items():  # LR factors re-derived from the step count
        w = getattr(cfg.optimization.optimizer, name).warmup
        base = getattr(cfg.optimization.optimizer, name).lr
        assert abs(sch.optimizer.param_groups[0]["lr"] - base * min(((1 + step) / w) ** -0.5, (1 + step) / w)) <= 1e-15
    states = torch.load(tmp_path / "ckpt.pth", weights_only=True)
    assert all(torch.equal(state.ema_helper.shadow[k], v) for k, v in states[-1].items())
    # ---- a checkpoint written here: every group travels, and torch's own optimizer reads states[1]
    for o in state.optimizers.values():
        for p in o.param_groups[0]["params"]:
            o.state[p] = {"step": 7, "exp_avg": torch.full_like(p, 0.25), "exp_avg_sq": torch.full_like(p, 0.5)}
    for s in state.schedulers.values():
        s.last_epoch = 7
    m._dropout_calls = 42
    out = tmp_path / "own"
    checkpoint.save_checkpoint(str(out), m, state.optimizers, 3, 7, state.ema_helper, state.schedulers)
    saved = torch.load(out / "ckpt.pth", weights_only=False)
    assert len(saved) == 5 and saved[2:4] == [3, 7] and os.path.exists(out / "ckpt_7.pth")
    ref_opt = torch.optim.AdamW(list(state.optimizers.values())[-1].param_groups[0]["params"], lr=1.0)
    ref_opt.load_state_dict(saved[1])  # the extra resume key is ignored by torch
    st0 = ref_opt.state[ref_opt.param_groups[0]["params"][0]]
    assert float(st0["step"]) == 7.0 and float(st0["exp_avg"].reshape(-1)[0]) == 0.25
    cfg2, m2 = _micro_cpu()
    state2 = train.TrainingState(cfg2, m2)
    assert checkpoint.resume_training(str(out), m2, state2.optimizers, state2.schedulers, state2.ema_helper) == (3, 7)
    assert m2._dropout_calls == 42
    for k, o in state2.optimizers.items():
        assert len(o.state) == len(state.optimizers[k].state) > 0
        assert all(s["step"] == 7 and float(s["exp_avg_sq"].reshape(-1)[0]) == 0.5 for s in o.state.values())
    assert all(s.last_epoch == 7 for s in state2.schedulers.values())
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
