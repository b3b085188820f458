"""The one GPU test that brings up a collective library (RCCL, a 1-rank group) -- in a file of its own that sorts LAST among the GPU
tests: the library tears its proxy / watchdog threads down asynchronously after ``destroy_process_group``, and one suite run of round 4
died of a SIGSEGV in a non-Python thread while the NEXT test was capturing a graph.  Nothing captures after this."""
import os

import pytest
import torch

import ddim_audio_amd as D
from ddim_audio_amd import configs, losses, synth
from ddim_audio_amd.schedule import make_schedule

pytestmark = pytest.mark.gpu


def _train_model(dtype_str, fnet=None, seed=0):
    d = configs.audio_dict(dtype_str, fnet)
    d["model"]["transformers"]["kwargs"]["hidden_dropout_prob"] = 0.0   # deterministic function
    d["optimization"]["optimizer"]["default"]["optimizer"] = "AdamW"
    cfg = configs.dict2namespace(d)
    return cfg, synth.fill_module(D.Model(cfg), seed).train()


def test_staged_grad_sync_runs_on_rccl():
    """The REAL ``make_grad_sync(...).staged`` on RCCL (it had never executed -- the two-rank test needs two
    devices and the bucket test replaced it with a stand-in): a 1-rank ``nccl`` process group on the one GPU of the test
    box, ``min_world=1`` so that the data-parallel backward takes the staged path -- the library records the three bucket
    events, the side stream waits on them, the asynchronous RCCL all-reduces (ReduceOp.AVG) are issued under it and the main
    stream waits for them.  With one rank the average is the identity: the gradients must equal the plain backward's bit for
    bit, twice in a row (the events are re-recorded by the second call)."""
    import torch.distributed as dist
    from ddim_audio_amd import dist as ddist
    cfg, m = _train_model("torch.cuda.BFloat16Tensor")
    alphas = make_schedule(cfg.diffusion)[1].cuda()
    shape = (4, 2, 256, 256)
    x0, e = synth.gaussian("rccl.x0", shape).cuda(), synth.gaussian("rccl.e", shape).cuda()
    t = torch.tensor([5, 994, 300, 650]).cuda()
    losses.noise_estimation_loss(m, x0, t, e, alphas).backward()
    plain = m._flat_grad.clone()
    m.zero_grad(set_to_none=True)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))
    own = not dist.is_initialized()
    if own:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()))
    try:
        ddist.attach_grad_sync(m, bucket_mb=16, overlap=True, min_world=1)
        assert m.grad_sync.active()
        # the three buckets the staged backward hands over tile the flat gradient buffer exactly: no float is reduced twice or never
        import ctypes
        from ddim_audio_amd import _lib
        lib = _lib.load()
        rng = (ctypes.c_longlong * 6)()
        _lib.check(lib.ddimx_grad_buckets(m._handle, rng))
        spans = sorted((rng[2 * i], rng[2 * i + 1]) for i in range(3))
        assert spans[0][0] == 0 and spans[-1][1] == int(lib.ddimx_grad_floats(m._handle)) == plain.numel()
        assert all(lo < hi for lo, hi in spans) and all(spans[i][1] == spans[i + 1][0] for i in range(2))
        for _ in range(2):
            losses.noise_estimation_loss(m, x0, t, e, alphas).backward()
            torch.cuda.synchronize()
            # bitwise: the words no parameter owns (64-float alignment padding, the temb.te slot) are torch.empty garbage that the
            # backward never writes -- after the NaN-poisoning tests of this suite they may hold NaN patterns, which == calls unequal
            assert torch.equal(m._flat_grad.view(torch.int32), plain.view(torch.int32)), "the staged backward must compute the same gradients"
            m.zero_grad(set_to_none=True)
        # the un-overlapped form of the same collective
        flat = plain.clone()
        ddist.make_grad_sync(bucket_mb=16, overlap=False)(flat)
        torch.cuda.synchronize()
        assert torch.equal(flat.view(torch.int32), plain.view(torch.int32))
    finally:
        m.grad_sync = None
        if own:
            torch.cuda.synchronize()
            dist.destroy_process_group()
            # the collective library tears its proxy / watchdog threads down asynchronously: let them finish before the next test
            # starts a capture (one suite run of round 4 died of a SIGSEGV in a non-Python thread right after this test)
            import time
            time.sleep(2.0)
