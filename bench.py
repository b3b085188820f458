#!/usr/bin/env python
"""Benchmark of the DDIM denoising hot path (BASELINE.json metric:
"DDIM denoising steps/sec (U-Net fwd/s), 1000-step sample, spectrogram batch").

A *step* is one iteration of ``generalized_steps`` over one batch: timestep fill, U-Net forward
(``Model.forward``), fused x0-prediction + x_{t-1} update -- exactly the product path
(``ddim_audio_amd.sampler.DDIMStepper``, hipGraph replay of libddimx launches).
Workload at N=1 = BASELINE.json configs[1]: batch 8 spectrograms [8,2,1024,256], audio.yml U-Net
(47.2 M parameters, deterministic synthetic weights), 1000-step eta=0 schedule, bf16 activations.
``value`` = sample-forwards per second over all ranks = N * B * K / t  (one unit = one spectrogram
through one U-Net evaluation + one DDIM update); iterations/s is reported beside it.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (see the contract in the task statement) with two extra objects:
``roofline`` (dominant kernel, algorithmic bytes / measured launch time vs the HBM peak) and
``cpu_baseline`` (the CPU oracle timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA peak
MFMA_F32_PEAK_TF = 157.3    # fp32-input MFMA = the fp32 vector rate


def per_kernel_times(model, B, T, reps=20):
    """Time the hot kernels one by one with HIP events on the launch stream (torch's current stream is
    the stream libddimx launches on).  Returns a list of dicts, one per kernel family and level."""
    from ddim_audio_amd import _lib
    lib = _lib.load()
    m = model.config
    bf16 = model._act_dtype == torch.bfloat16
    dt = _lib.DDIMX_BF16 if bf16 else _lib.DDIMX_F32
    tdt = torch.bfloat16 if bf16 else torch.float32
    es = 2 if bf16 else 4
    dev = torch.device("cuda", torch.cuda.current_device())
    rows = []

    def timed(fn):
        fn(); fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3  # seconds per launch

    for lvl, (C, res) in enumerate(zip(m.ch, m.res)):
        H, W = T >> lvl, m.f_size >> lvl
        x = torch.randn(B, H, W, C, device=dev).to(tdt)
        y = torch.empty_like(x)
        h = torch.randn(B, H, W, C, device=dev).to(tdt)
        w = (torch.randn(9 * C * C, device=dev) * (1.0 / (9 * C) ** 0.5)).to(tdt)
        bias = torch.randn(C, device=dev) * 0.1
        temb = torch.randn(B, C, device=dev) * 0.1
        scale = torch.rand(B, C, device=dev) + 0.5
        shift = torch.randn(B, C, device=dev) * 0.1
        stats = torch.empty(int(lib.ddimx_conv3x3_stats_floats(dt, C, B, H, W)) + 2 * B * C * 4096, device=dev)
        st = _lib.stream()

        pipe = bf16 and C == 32  # level 0: the software-pipelined kernel (csrc/conv_pipe.h) -- what the inference walk launches
        wreg = bf16 and C >= 64  # the inference walk's kernel from C = 64 up: weights streamed to registers (csrc/conv_wreg.h)
        if pipe:
            stats = torch.empty(max(stats.numel(), int(lib.ddimx_conv3x3_pipe_stats_floats(C, B, H, W))), device=dev)
        if wreg or pipe:
            w_f32 = torch.randn(C, C, 3, 3, device=dev) * (1.0 / (9 * C) ** 0.5)
            wfr = torch.empty(9 * C * C, dtype=tdt, device=dev)
            _lib.check(lib.ddimx_pack_conv_frag(_lib.ptr(w_f32), _lib.ptr(wfr), C, C, st))

        def conv():  # K1 of the block: GroupNorm affine + SiLU prologue, + timestep embedding, SiLU
            if pipe:
                _lib.check(lib.ddimx_conv3x3_pipe_fwd(C, _lib.ptr(x), _lib.ptr(wfr), None, _lib.ptr(temb), C, _lib.ptr(scale), _lib.ptr(shift), 2,
                                                      _lib.ptr(y), _lib.ptr(stats), B, H, W, st))
            elif wreg:
                _lib.check(lib.ddimx_conv3x3_wreg_fwd(C, _lib.ptr(x), _lib.ptr(w), _lib.ptr(wfr), None, _lib.ptr(temb), C, _lib.ptr(scale),
                                                      _lib.ptr(shift), 2, 1, _lib.ptr(y), _lib.ptr(stats), B, H, W, st))
            else:
                _lib.check(lib.ddimx_conv3x3_fwd(dt, C, _lib.ptr(x), _lib.ptr(w), None, _lib.ptr(temb), C, _lib.ptr(scale),
                                                 _lib.ptr(shift), 2, 1, _lib.ptr(y), _lib.ptr(stats), B, H, W, st))

        def conv2():  # K2 of the block: GroupNorm affine prologue, + bias, SiLU
            if pipe:
                _lib.check(lib.ddimx_conv3x3_pipe_fwd(C, _lib.ptr(x), _lib.ptr(wfr), _lib.ptr(bias), None, 0, _lib.ptr(scale), _lib.ptr(shift), 1,
                                                      _lib.ptr(y), _lib.ptr(stats), B, H, W, st))
            elif wreg:
                _lib.check(lib.ddimx_conv3x3_wreg_fwd(C, _lib.ptr(x), _lib.ptr(w), _lib.ptr(wfr), _lib.ptr(bias), None, 0, _lib.ptr(scale),
                                                      _lib.ptr(shift), 1, 1, _lib.ptr(y), _lib.ptr(stats), B, H, W, st))
            else:
                _lib.check(lib.ddimx_conv3x3_fwd(dt, C, _lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), None, 0, _lib.ptr(scale),
                                                 _lib.ptr(shift), 1, 1, _lib.ptr(y), _lib.ptr(stats), B, H, W, st))

        def resid():
            _lib.check(lib.ddimx_resid_gn_fwd(dt, C, _lib.ptr(x), _lib.ptr(h), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(y),
                                              _lib.ptr(stats), B, H, W, st))

        elems = B * H * W * C
        tc1, tc2, tr = timed(conv), timed(conv2), timed(resid)
        tc = 0.5 * (tc1 + tc2)
        kname = (f"conv3_pipe_kernel<bf16,C={C},3x3>" if pipe else f"conv3_wreg_kernel<bf16,C={C},3x3>" if wreg
                 else f"conv_mfma_kernel<{'bf16' if bf16 else 'f32'},C={C},3x3>")
        rows.append(dict(kernel=kname, level=lvl, launches_per_fwd=4 * res,
                         seconds=tc, seconds_k1=tc1, seconds_k2=tc2, alg_bytes=2 * elems * es + 9 * C * C * es, flops=2.0 * elems * 9 * C))
        rows.append(dict(kernel=f"resid_kernel<{'bf16' if bf16 else 'f32'},C={C}>", level=lvl, launches_per_fwd=2 * res,
                         seconds=tr, alg_bytes=3 * elems * es, flops=3.0 * elems))
        # the whole Residual_Block (models/diffusion.py:42-56) = K1 + K2 + resid: 7 activation passes + both weight sets
        rows.append(dict(kernel=f"Residual_Block<C={C}> (K1 + K2 + resid)", level=lvl, launches_per_fwd=2 * res, block=True,
                         seconds=tc1 + tc2 + tr, alg_bytes=7 * elems * es + 2 * 9 * C * C * es, flops=2 * 2.0 * elems * 9 * C + 3.0 * elems))
        if lvl > 0:  # Downsample (level lvl-1 -> lvl) and Upsample + skip add (lvl -> lvl-1): models/diffusion.py:59-78,284
            Cp, Hp, Wp = m.ch[lvl - 1], H * 2, W * 2
            xp = torch.randn(B, Hp, Wp, Cp, device=dev).to(tdt)
            wd = (torch.randn(16 * C * Cp, device=dev) * (1.0 / (16 * Cp) ** 0.5)).to(tdt)
            wu = (torch.randn(2 * 6 * 2 * Cp * C, device=dev) * (1.0 / (16 * C) ** 0.5)).to(tdt)
            bu = torch.randn(2 * Cp, device=dev) * 0.1
            zp = torch.empty_like(xp)

            if bf16:  # fragment-order copies for the register-streamed kernels (what the walk launches)
                wd_f32 = torch.randn(C, Cp, 4, 4, device=dev) * (1.0 / (16 * Cp) ** 0.5)
                wdf = torch.empty(16 * C * Cp, dtype=tdt, device=dev)
                _lib.check(lib.ddimx_pack_conv_frag_k(_lib.ptr(wd_f32), _lib.ptr(wdf), C, Cp, 16, st))
                wuf = torch.empty_like(wu)
                per = 6 * 2 * Cp * C
                for a_ in range(2):
                    _lib.check(lib.ddimx_pack_frag_from_taps(_lib.ptr(wu[a_ * per:]), _lib.ptr(wuf[a_ * per:]), 6, 2 * Cp, C, st))
                sdn = torch.empty(B * H * W * C * 2 // 8 + 8192, device=dev)
                sup = torch.empty(B * Hp * Wp * Cp * 2 // 8 + 8192, device=dev)

            def down():
                if bf16:
                    _lib.check(lib.ddimx_downsample_wreg_fwd(Cp, C, _lib.ptr(xp), _lib.ptr(wdf), _lib.ptr(bias), _lib.ptr(y), _lib.ptr(sdn), B, Hp, Wp, st))
                else:
                    _lib.check(lib.ddimx_downsample_fwd(dt, Cp, C, _lib.ptr(xp), _lib.ptr(wd), _lib.ptr(bias), _lib.ptr(y), B, Hp, Wp, st))

            def up():
                if bf16:
                    _lib.check(lib.ddimx_upsample_add_wreg_fwd(C, Cp, _lib.ptr(x), _lib.ptr(wuf), _lib.ptr(bu), _lib.ptr(xp), _lib.ptr(zp), _lib.ptr(sup), B, H, W, st))
                else:
                    _lib.check(lib.ddimx_upsample_add_fwd(dt, C, Cp, _lib.ptr(x), _lib.ptr(wu), _lib.ptr(bu), _lib.ptr(xp), _lib.ptr(zp), B, H, W, st))

            ep = B * Hp * Wp * Cp
            fam = "conv3_wreg_kernel" if bf16 else "conv_mfma_kernel"
            rows.append(dict(kernel=f"{fam}<down4 {Cp}->{C}>", level=lvl, launches_per_fwd=1, seconds=timed(down),
                             alg_bytes=(ep + elems) * es + 16 * C * Cp * es, flops=2.0 * elems * 16 * Cp))
            rows.append(dict(kernel=f"{fam}<up4 {C}->{Cp} + skip>", level=lvl, launches_per_fwd=1, seconds=timed(up),
                             alg_bytes=(elems + 2 * ep) * es + 16 * C * Cp * es, flops=2.0 * ep * 4 * C))
            del xp, zp
        del x, y, h, stats
    return rows


def cpu_baseline(T, B, iters=3):
    """The CPU oracle (plain PyTorch fp32 restatement, pinned to the reference by golden vectors) on the host cores: one
    warm-up iteration of the SAME shape, then ``iters`` timed DDIM iterations of the same workload (batch B, length T) --
    BASELINE.md section 3: >= 3 iterations after one warm-up -- plus, as ``cfg1``, one warm-up and one timed iteration at BASELINE
    configs[0]'s shape (num_samples = 1, sampling.t_size = 8192: the reference's own CPU-runnable case; its 100 steps are
    extrapolated from that iteration, every iteration costs the same)."""
    from ddim_audio_amd import configs, schedule, synth
    from ddim_audio_amd.model import state_inventory
    from oracle import ref_cpu
    cfg = configs.audio_config("torch.FloatTensor")
    sd = {k: torch.empty(s) for k, s in state_inventory(cfg).items()}
    synth.fill_state_dict(sd)
    sd["temb.te"] = ref_cpu.timestep_table(cfg.diffusion.num_diffusion_timesteps)
    alphas = schedule.make_schedule(cfg.diffusion)[1]
    cores = min(16, os.cpu_count() or 1)  # the GPU box's CPU share for one GPU
    torch.set_num_threads(cores)
    fn = lambda a, b: ref_cpu.model_forward(sd, cfg, a, b)  # noqa: E731

    def leg(batch, t_len, n):
        x = torch.randn(batch, 2, t_len, 256)
        seq = [int(v) for v in torch.linspace(0, 999, n).tolist()]
        t0 = time.perf_counter()
        ref_cpu.generalized_steps(x, seq, fn, alphas, [-1], eta=0.0)
        return time.perf_counter() - t0

    with torch.no_grad():
        leg(B, T, 1)  # warm-up at the timed shape (allocator, oneDNN primitive caches, thread pool)
        dt = leg(B, T, iters)
        leg(1, 8192, 1)
        dt1 = leg(1, 8192, 1)
    return dict(value=B * iters / dt, unit="sample-fwd/s", cores=cores, kind="port",
                sample=f"{iters} DDIM iterations x batch {B} at T={T} after one same-shape warm-up iteration (oracle/ref_cpu.py, fp32, "
                       f"{cores} threads), {dt:.1f} s",
                cfg1=dict(value=1.0 / dt1, unit="sample-fwd/s", t1024_equivalents_per_s=8.0 / dt1, seconds_for_100_steps=100.0 * dt1,
                          sample=f"1 DDIM iteration x batch 1 at T=8192 after one warm-up (BASELINE configs[0] shape), {dt1:.1f} s"))


def sampler_leg(cfg_strs, B, T, steps, seed=4321):
    """A secondary measurement of the same loop on another configuration (never the headline ``value``): a fresh model, the
    1000-step eta = 0 schedule, one hipGraph per step; 3 warm-up steps, ``steps`` timed ones.  Returns sample-fwd/s."""
    import ddim_audio_amd as D
    from ddim_audio_amd import configs, schedule, synth
    from ddim_audio_amd.sampler import DDIMStepper
    try:
        cfg = configs.audio_config(*cfg_strs)
        model = synth.fill_module(D.Model(cfg)).eval()
        alphas = schedule.make_schedule(cfg.diffusion)[1]
        n_sched = cfg.diffusion.num_diffusion_timesteps
        coef = schedule.ddim_coefficients(schedule.make_seq(n_sched, n_sched), alphas, 0.0)
        g = torch.Generator(device="cuda")
        g.manual_seed(seed)
        x = torch.randn(B, 2, T, cfg.model.f_size, device="cuda", generator=g)
        with torch.no_grad():
            st = DDIMStepper(model, x, coef)
            for _ in range(3):
                st.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                st.step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ok = bool(torch.isfinite(x).all().item())
            st.close()
        del st, model, x
        torch.cuda.empty_cache()
        return {"value": B * steps / dt, "unit": "sample-fwd/s", "batch": B, "t_size": T, "steps": steps, "ms_per_step": dt / steps * 1e3,
                "output_finite": ok}
    except Exception as ex:  # the headline measurement must survive a failure here
        return {"error": f"{type(ex).__name__}: {ex}"[:300]}


def training_leg(args, cfg, dev, rank, world, backend, steps=10):
    """Secondary measurement (never the headline `value`): BASELINE config 4's training step -- noise + antithetic t,
    loss, backward of the whole network, clip, fused AdamW (both groups), EMA -- at `--train-batch` samples per GPU, data
    parallel over the ranks (one all-reduce of the flat gradient buffer per step).  Failures are reported, not raised."""
    import torch.distributed as dist
    try:
        import copy
        import ddim_audio_amd as D
        from ddim_audio_amd import configs, dist as ddist, schedule, synth, train
        tcfg = copy.deepcopy(cfg)
        tcfg.optimization.optimizer.default.optimizer = "AdamW"  # reference-pinned optimizer for both groups
        m = synth.fill_module(D.Model(tcfg))
        if world > 1:
            ddist.attach_grad_sync(m)
        state = train.TrainingState(tcfg, m)
        alphas = schedule.make_schedule(tcfg.diffusion)[1].to(dev)
        b = args.train_batch
        x = torch.randn(b, 2, args.t_size, tcfg.model.f_size, device=dev)
        for _ in range(3):
            train.train_step(m, x, state, alphas)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss, _ = train.train_step(m, x, state, alphas)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dt = float(el.item()) / steps
        no_overlap_ms = None
        if world > 1:  # the same step with the gradient all-reduce issued AFTER the backward (no overlap): the staged path's gain
            from ddim_audio_amd import dist as ddist2
            ddist2.attach_grad_sync(m, overlap=False)
            train.train_step(m, x, state, alphas)
            torch.cuda.synchronize(); dist.barrier()
            tn = time.perf_counter()
            for _ in range(steps):
                train.train_step(m, x, state, alphas)
            torch.cuda.synchronize(); dist.barrier()
            no_overlap_ms = (time.perf_counter() - tn) / steps * 1e3
            ddist2.attach_grad_sync(m)
        ar_ms = None
        if world > 1 and getattr(m, "_flat_grad", None) is not None:  # the step's only collective, timed alone
            torch.cuda.synchronize()
            dist.barrier()
            ta = time.perf_counter()
            for _ in range(3):
                m.grad_sync(m._flat_grad)
            torch.cuda.synchronize()
            ar_ms = (time.perf_counter() - ta) / 3 * 1e3
        graphed_ms = None
        if world == 1:  # the same step replayed from one hipGraph (bit-identical to the eager step, tests/test_gpu_configs.py)
            try:
                gstep = train.GraphedTrainStep(m, state, alphas, warmup=1)
                for _ in range(3):
                    gstep(x)
                torch.cuda.synchronize()
                tg = time.perf_counter()
                for _ in range(steps):
                    loss, _ = gstep(x)
                torch.cuda.synchronize()
                graphed_ms = (time.perf_counter() - tg) / steps * 1e3
                gstep.close()
            except Exception as ex:
                graphed_ms = f"{type(ex).__name__}: {ex}"[:200]
        res = {"value": world * b / dt, "graphed_ms_per_step": graphed_ms, "grad_allreduce_ms": ar_ms, "ms_per_step_no_overlap": no_overlap_ms, "grad_mb": (m._flat_grad.numel() * 4 / 1e6 if getattr(m, "_flat_grad", None) is not None else None), "unit": "train samples/s", "ms_per_step": dt * 1e3, "batch_per_gpu": b, "t_size": args.t_size,
               "steps": steps, "optimizer": "fused AdamW (both groups), clip 1.0, EMA 0.9999", "loss_finite": bool(torch.isfinite(loss)),
               "model_tflops": world * b * 3 * 159.22e9 * args.t_size / 1024.0 / dt / 1e12,
               "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30}
        del m, state, x
        torch.cuda.empty_cache()
        return res
    except Exception as ex:  # the headline measurement above must survive a failure here
        return {"error": f"{type(ex).__name__}: {ex}"[:300]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000, help="timed iterations (default: the whole 1000-step sample, ~4 s)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="spectrograms per GPU (weak scaling)")
    ap.add_argument("--t-size", type=int, default=1024)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--fnet-dtype", default=None, choices=["bf16", "f32"],
                    help="model.transformers.dtype: operand type of the FNet's dense GEMMs (default: same as --dtype; f32 with "
                         "--dtype bf16 is the reference's mixed mode, models/diffusion.py:242-246)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-train-leg", action="store_true", help="skip the secondary training-step measurement")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the secondary sampler legs (batch 32; mixed fp32-FNet mode)")
    ap.add_argument("--train-batch", type=int, default=32, help="samples per GPU of the training leg (BASELINE config 4)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start N rank processes (one per GPU, RCCL world of N) through
        # torch.distributed.run BEFORE anything in this process touches the GPU, wait, and pass their exit code on.  The parent
        # never initialises HIP and never re-execs itself (a child process, not os.exec*).
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and rank == 0:
        print(f"[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: reporting n_gpus={world}", file=sys.stderr)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the hot path has no CPU fallback")
    ndev = torch.cuda.device_count()
    dev_idx = local_rank % max(ndev, 1)  # one rank per GPU; (rehearsals on a 1-GPU box put every rank on cuda:0)
    torch.cuda.set_device(dev_idx)
    dev = torch.device("cuda", dev_idx)
    import torch.distributed as dist
    backend = os.environ.get("DDIMX_BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm; "gloo" for rehearsals
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import numpy as np
    import ddim_audio_amd as D
    from ddim_audio_amd import configs, schedule, synth
    from ddim_audio_amd.sampler import DDIMStepper

    B, T = args.batch, args.t_size
    tstr = "torch.cuda.BFloat16Tensor" if args.dtype == "bf16" else "torch.cuda.FloatTensor"
    fstr = None if args.fnet_dtype is None else ("torch.cuda.BFloat16Tensor" if args.fnet_dtype == "bf16" else "torch.cuda.FloatTensor")
    cfg = configs.audio_config(tstr, fstr)
    model = synth.fill_module(D.Model(cfg)).eval()
    alphas = schedule.make_schedule(cfg.diffusion)[1]
    n_sched = cfg.diffusion.num_diffusion_timesteps
    coef = schedule.ddim_coefficients(schedule.make_seq(n_sched, n_sched), alphas, 0.0)  # the 1000-step sample
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    x = torch.randn(B, 2, T, cfg.model.f_size, device=dev, generator=g)

    with torch.no_grad():
        stepper = DDIMStepper(model, x, coef)

        def run(n):
            for _ in range(n):
                if stepper.done and stepper.done % n_sched == 0:  # schedule exhausted: restart from fresh noise
                    stepper.rewind()
                    x.normal_(generator=g)
                stepper.step()

        run(max(args.warmup, 2))  # >= 2: step 0 is eager and sizes the workspace, step 1 replays the captured graph
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(args.steps)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    et = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(et, op=dist.ReduceOp.MAX)
    elapsed = float(et.item())
    finite = bool(torch.isfinite(x).all().item())
    gather_ms = None
    if world > 1:  # outside the timed region: the optional final all-gather of the x0 predictions, RCCL over xGMI
        from ddim_audio_amd import dist as ddist
        loc = stepper.x0 if backend == "nccl" else stepper.x0.cpu()
        torch.cuda.synchronize()
        dist.barrier()
        tg = time.perf_counter()
        full = ddist.gather_batch(loc, world * B)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - tg) * 1e3
        assert full.size(0) == world * B
        del full

    ranks_seen = None
    if world > 1:  # an independent count of the ranks the collective backend really connects: all-reduce of ones
        one = torch.ones(1, dtype=torch.float32, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        ranks_seen = int(round(float(one.item())))

    extra = {}
    if world == 1 and not args.no_extra_legs and os.environ.get("DDIMX_BENCH_EXTRA", "1") != "0":
        stepper.close()
        # north_star's own batch ("... steps/sec at batch 32 on 1x MI355X"), same dtype mode as the headline
        extra["b32"] = sampler_leg((tstr, fstr), 32, T, 60)
        # the reference's own half-precision mode: bf16 convolutions, fp32 transformer (models/diffusion.py:242-246; fftn has no
        # bf16, so this is its only workable half setup) at the headline batch
        if args.dtype == "bf16":
            extra["mixed"] = sampler_leg((tstr, "torch.cuda.FloatTensor"), B, T, 200)
            extra["mixed"]["fnet_dtype"] = "f32"
            # the reference's DEFAULT dtype (configs/audio.yml:26,42 torch.cuda.FloatTensor): what a user who follows INTEGRATION.md A
            # with the unchanged audio.yml runs -- the exact-fp32 parity kernels (v_mfma_f32_32x32x2_f32), never the headline value
            extra["f32"] = sampler_leg(("torch.cuda.FloatTensor", None), B, T, 50)
            extra["f32"]["dtype"] = "f32"

    train_leg = None
    if not args.no_train_leg and os.environ.get("DDIMX_BENCH_TRAIN", "1") != "0":
        train_leg = training_leg(args, cfg, dev, rank, world, backend)

    if rank == 0:
        iters_per_s = args.steps / elapsed
        out = {
            "metric": "DDIM denoising steps/sec (U-Net fwd/s), 1000-step sample, spectrogram batch",
            "value": world * B * iters_per_s,
            "unit": "sample-fwd/s",
            "n_gpus": world,
            "ranks": world,  # torch.distributed world size (RCCL ranks, one per GPU); 1 = no process group
            "rccl_ranks_seen": ranks_seen,  # sum over ranks of 1 through the collective backend (None at N = 1)
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: batch {B}/GPU x [2,{T},256] spectrograms, audio.yml U-Net (47.2M params, "
                                   f"hash-filled weights; FNet GEMM operands {args.fnet_dtype or args.dtype}), generalized_steps eta=0 over the 1000-step "
                                   f"schedule, one hipGraph per step (inside the forward two batch shards on two "
                                   f"streams, fork mask {model.fork_mask:#x})",
                       "global_batch": world * B, "t_size": T, "parallelism": f"batch-sharded x{world}, no collective in the loop"},
            "iters_per_s": iters_per_s,
            "output_finite": finite,
        }
        if gather_ms is not None:
            out["final_allgather_ms"] = gather_ms  # [N*B,2,T,256] fp32 x0 predictions, untimed part of the run
        flops = 159.22e9 * T / 1024.0  # algorithmic FLOPs per sample-forward
        out["model_tflops"] = out["value"] * flops / 1e12
        if not args.no_roofline:
            with torch.no_grad():
                rows = per_kernel_times(model, B, T)
            ridge = MFMA_BF16_PEAK_TF * 1e12 / (HBM_PEAK_GBS * 1e9) if args.dtype == "bf16" else MFMA_F32_PEAK_TF * 1e12 / (HBM_PEAK_GBS * 1e9)
            mfma_peak = MFMA_BF16_PEAK_TF if args.dtype == "bf16" else MFMA_F32_PEAK_TF
            for r in rows:
                r["gbps"] = r["alg_bytes"] / r["seconds"] / 1e9
                r["tflops"] = r["flops"] / r["seconds"] / 1e12
                r["ms_per_fwd"] = r["seconds"] * r["launches_per_fwd"] * 1e3
                # which roof bounds this kernel (arithmetic intensity of its ALGORITHMIC bytes vs the ridge) and how close it is
                r["bound"] = "hbm" if r["flops"] / r["alg_bytes"] < ridge else "mfma"
                r["frac"] = r["gbps"] / HBM_PEAK_GBS if r["bound"] == "hbm" else r["tflops"] / mfma_peak
                r["frac_hbm"], r["frac_mfma"] = r["gbps"] / HBM_PEAK_GBS, r["tflops"] / mfma_peak
            dom = max((r for r in rows if not r.get("block")), key=lambda r: r["ms_per_fwd"])
            if dom["bound"] == "hbm":
                out["roofline"] = {"bound": "hbm", "achieved": dom["gbps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": dom["gbps"] / HBM_PEAK_GBS, "traffic": None}
            else:
                out["roofline"] = {"bound": "mfma", "achieved": dom["tflops"], "peak": mfma_peak, "unit": "TFLOP/s",
                                   "frac": dom["tflops"] / mfma_peak, "traffic": None}
            tpath = os.path.join(ROOT, "profiles", "traffic.json")  # HBM bytes per launch from rocprofv3 --pmc passes
            if os.path.exists(tpath):
                tr = json.load(open(tpath))
                per = tr.get("per_kernel", {})
                if tr.get("batch") == B and tr.get("t_size") == T:
                    if tr.get("kernel") == dom["kernel"]:
                        out["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
                    for r in rows:  # PMC traffic of the other profiled kernels, per launch
                        if r["kernel"] in per:
                            r["traffic"] = per[r["kernel"]]
            out["roofline"].update(kernel=dom["kernel"], launch_us=dom["seconds"] * 1e6, alg_bytes_per_launch=dom["alg_bytes"],
                                   launches_per_fwd=dom["launches_per_fwd"],
                                   note="isolated back-to-back launches timed with HIP events on the launch stream; inside the step the "
                                        "same kernel runs ~10-15% faster (its input is still in the 256 MB Infinity Cache)")
            # the per-level table the roofline discussion uses: one entry per kernel family and level
            out["roofline"]["per_level"] = [{"kernel": r["kernel"], "level": r["level"], "bound": r["bound"], "frac": round(r["frac"], 4),
                                             "us": round(r["seconds"] * 1e6, 2)} for r in rows if not r.get("block")]
            # north_star words its 60 % target on the block path: algorithmic bytes of the whole block / (K1 + K2 + resid) time
            out["roofline"]["per_block"] = [{"level": r["level"], "us": round(r["seconds"] * 1e6, 2), "gbps": round(r["gbps"], 1),
                                             "frac_hbm": round(r["frac_hbm"], 4), "frac_mfma": round(r["frac_mfma"], 4)}
                                            for r in rows if r.get("block")]
            # a third roof (round 4, profiles/r04/power_probe_convs_sustained.txt): looping this kernel holds the board at its
            # 1400 W power cap with the shader clock pulled down to 1.9-2.2 GHz -- time per launch = energy per launch / 1400 W
            out["roofline"]["power_note"] = ("level-0/1 convs run at the board's 1400 W cap (rocm-smi, sustained loop): "
                                             "neither the HBM nor the MFMA roof is reachable for this op on this board at its energy per launch")
            for r in rows:
                for k in [k for k in r if k.startswith("seconds")]:
                    r[k.replace("seconds", "us")] = r.pop(k) * 1e6
            out["kernels"] = [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()} for r in rows if not r.get("block")]
        if not args.no_cpu_baseline and world == 1:  # the CPU leg runs at N=1 only (the other ranks would just wait)
            out["cpu_baseline"] = cpu_baseline(T, B)
        for k, v in extra.items():
            out[k] = v
        if train_leg is not None:
            out["train_step"] = train_leg
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
