"""Two (or N) fully independent B/N samplers on their own streams (no per-step join) vs the product stepper."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ddim_audio_amd as D
from ddim_audio_amd import configs, synth, schedule
from ddim_audio_amd.sampler import DDIMStepper
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = configs.audio_config("torch.cuda.BFloat16Tensor")
alphas = schedule.make_schedule(cfg.diffusion)[1]
coef = schedule.ddim_coefficients(schedule.make_seq(1000, 1000), alphas, 0.0)
K = 100
def run_indep(n, fork):
    models = [synth.fill_module(D.Model(cfg)).eval() for _ in range(n)]  # separate workspaces; weights identical
    for m in models: m.fork_mask = fork
    streams = [torch.cuda.Stream() for _ in range(n)]
    xs = [torch.randn(B // n, 2, 1024, 256, device="cuda") for _ in range(n)]
    steppers = []
    with torch.no_grad():
        for m, s, x in zip(models, streams, xs):
            with torch.cuda.stream(s):
                st = DDIMStepper(m, x, coef, branches=1)
                st.step(); st.step(); st.step()
            steppers.append(st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            for st, s in zip(steppers, streams):
                with torch.cuda.stream(s):
                    st.step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"{n} independent samplers of B={B // n} (fork mask {fork:#x}): {B * K / dt:8.1f} sample-fwd/s, {dt / K * 1e3:.3f} ms per step", flush=True)
run_indep(1, 0)
run_indep(1, 0x1003f)
run_indep(2, 0)
run_indep(2, 0x1003f)
run_indep(4, 0)
