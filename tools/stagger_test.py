"""Two INDEPENDENT half-batch samplers on two streams, offset by a fraction of a step (no join anywhere): does one shard's
latency-bound middle (levels 3-5, FNet) hide under the other's power-bound levels 0-2?   stagger_test.py [B] [steps]
Prints sample-fwd/s for: the default stepper (two shards in lock step inside the forward), two steppers started together, and two
steppers with the second delayed by 0.25 / 0.5 / 0.75 of a step."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ddim_audio_amd as D
from ddim_audio_amd import configs, schedule, synth
from ddim_audio_amd.sampler import DDIMStepper

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
cfg = configs.audio_config("torch.cuda.BFloat16Tensor")
model = synth.fill_module(D.Model(cfg)).eval()
alphas = schedule.make_schedule(cfg.diffusion)[1]
coef = schedule.ddim_coefficients(schedule.make_seq(1000, 1000), alphas, 0.0)
x = torch.randn(B, 2, 1024, 256, device="cuda")

def bench_default():
    st = DDIMStepper(model, x.clone(), coef)
    for _ in range(4): st.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): st.step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st.close()
    return B * steps / dt, dt / steps * 1e3

def bench_two(delay_frac, step_ms):
    h = B // 2
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    xa, xb = x[:h].clone(), x[h:].clone()
    with torch.cuda.stream(sa):
        a = DDIMStepper(model, xa, coef, slot=2, fork=False)
        for _ in range(3): a.step()
    with torch.cuda.stream(sb):
        b = DDIMStepper(model, xb, coef, slot=3, fork=False)
        for _ in range(3): b.step()
    torch.cuda.synchronize()
    spin = torch.empty(64 << 20, device="cuda")
    t0 = time.perf_counter()
    if delay_frac > 0:  # hold stream b back: a memset loop of about delay_frac x step time (67 us per 256 MB fill at ~4 TB/s)
        with torch.cuda.stream(sb):
            for _ in range(max(1, int(delay_frac * step_ms * 1e3 / 70))): spin.zero_()
    for _ in range(steps):
        with torch.cuda.stream(sa): a.step()
        with torch.cuda.stream(sb): b.step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    a.close(); b.close()
    return B * steps / dt, dt / steps * 1e3

with torch.no_grad():
    v, ms = bench_default()
    print(f"default stepper (lock-step shards inside the forward): {v:.1f} sample-fwd/s, {ms:.3f} ms/step")
    for fr in (0.0, 0.25, 0.5, 0.75):
        v2, ms2 = bench_two(fr, ms)
        print(f"two independent half-batch steppers, second delayed by {fr:.2f} step: {v2:.1f} sample-fwd/s, {ms2:.3f} ms/step")
    v, ms = bench_default()
    print(f"default stepper again: {v:.1f} sample-fwd/s, {ms:.3f} ms/step")
