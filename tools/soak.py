"""Soak test: a full 1000-step sample (B=8, T=1024) and 60 training steps (B=16), checking finiteness and steady memory."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ddim_audio_amd as D  # noqa: E402
from ddim_audio_amd import configs, schedule, synth, train  # noqa: E402

cfg = configs.audio_config("torch.cuda.BFloat16Tensor")
cfg.optimization.optimizer.default.optimizer = "AdamW"
m = synth.fill_module(D.Model(cfg)).eval()
alphas = schedule.make_schedule(cfg.diffusion)[1]
x = torch.randn(8, 2, 1024, 256, device="cuda")
t0 = time.perf_counter()
xs, x0 = D.generalized_steps(x, list(range(1000)), m, alphas, [-1], eta=0.0)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"1000-step sample of 8 spectrograms: {dt:.2f} s = {8000 / dt:.0f} sample-fwd/s, finite={bool(torch.isfinite(xs[-1]).all())}, "
      f"|x0| mean {float(xs[-1].abs().mean()):.3f}")
m = synth.fill_module(D.Model(cfg))
state = train.TrainingState(cfg, m)
xt = torch.randn(16, 2, 1024, 256, device="cuda")
a = alphas.cuda()
mem = []
for i in range(60):
    loss, norms = train.train_step(m, xt, state, a)
    if i % 10 == 9:
        torch.cuda.synchronize()
        mem.append(torch.cuda.memory_allocated() / 2 ** 30)
        print(f"train step {i + 1}: loss {float(loss):.4g} grad-norm {float(list(norms.values())[0]):.4g} mem {mem[-1]:.2f} GiB", flush=True)
assert all(abs(v - mem[0]) < 0.05 for v in mem), mem
print("ok")
