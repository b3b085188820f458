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
    assert lib.ddimx_abi_version() == 1


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
