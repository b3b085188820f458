"""Training-step timing on one MI355X (BASELINE config 4 shape: B per GPU x [2, 1024, 256], bf16 activations).

Not the headline metric (that is bench.py's sampling rate); reports train samples/s = B / step time for
`ddim_audio_amd.train.train_step` (loss, backward, clip, fused Adam/AdamW, EMA) on synthetic data.
usage: python tools/train_bench.py [B] [T] [steps] [dtype: bf16|f32] [graph]   (graph: also time train.GraphedTrainStep)
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ddim_audio_amd as D  # noqa: E402
from ddim_audio_amd import configs, synth, train  # noqa: E402
from ddim_audio_amd.schedule import make_schedule  # noqa: E402


def main():
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    t_len = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    dt = sys.argv[4] if len(sys.argv) > 4 else "bf16"
    d = configs.audio_dict("torch.cuda.BFloat16Tensor" if dt == "bf16" else "torch.cuda.FloatTensor")
    d["optimization"]["optimizer"]["default"]["optimizer"] = "AdamW"  # AdaBelief's source is absent upstream
    cfg = configs.dict2namespace(d)
    m = D.Model(cfg)
    synth.fill_module(m, 0)
    state = train.TrainingState(cfg, m)
    _, alphas = make_schedule(cfg.diffusion)
    alphas = alphas.cuda()
    x = torch.randn(b, 2, t_len, 256, device="cuda")
    torch.manual_seed(1234)
    import contextlib, os
    # A/B: DDIMX_MAIN_PRIORITY=-1 runs the step on a high-priority stream (the weight-gradient branch keeps a normal one)
    main = torch.cuda.Stream(priority=int(os.environ["DDIMX_MAIN_PRIORITY"])) if "DDIMX_MAIN_PRIORITY" in os.environ else None
    with (torch.cuda.stream(main) if main is not None else contextlib.nullcontext()):
        for _ in range(2):
            loss, _ = train.train_step(m, x, state, alphas)
        torch.cuda.synchronize()
        phases = {}
        t0 = time.perf_counter()
        for _ in range(steps):
            loss, norms = train.train_step(m, x, state, alphas)
        torch.cuda.synchronize()
        dtm = (time.perf_counter() - t0) / steps
    # forward-only and forward+backward split
    m.train()
    e = torch.randn_like(x)
    tt = train.antithetic_timesteps(b, 1000).cuda()
    from ddim_audio_amd import losses
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    l = losses.noise_estimation_loss(m, x, tt, e, alphas)
    ev[1].record()
    l.backward()
    ev[2].record()
    torch.cuda.synchronize()
    phases = {"fwd_ms": ev[0].elapsed_time(ev[1]), "bwd_ms": ev[1].elapsed_time(ev[2])}
    if len(sys.argv) > 5 and sys.argv[5] == "graph":  # the same step replayed from one hipGraph
        for p in m.parameters():
            p.grad = None
        gstep = train.GraphedTrainStep(m, state, alphas, warmup=1)
        for _ in range(3):
            gstep(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss, norms = gstep(x)
        torch.cuda.synchronize()
        phases["graphed_ms_per_step"] = (time.perf_counter() - t0) / steps * 1e3
    flops = 3 * 159.22e9 * (t_len / 1024) * b
    print(json.dumps({"metric": "train samples/s (1 GPU)", "value": b / dtm, "ms_per_step": dtm * 1e3, "B": b, "T": t_len,
                      "dtype": dt, "loss": float(loss), "grad_norm": {k: float(v) for k, v in norms.items()},
                      "model_tflops": flops / dtm / 1e12, "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30, **phases}))


if __name__ == "__main__":
    main()
