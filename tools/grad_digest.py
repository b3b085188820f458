"""SHA-256 of every parameter gradient of one training forward + backward on fixed synthetic inputs (audio.yml widths).
Run it under two settings of an A/B environment hook (DDIMX_BWD_STATS_FUSED, DDIMX_WGRAD_SIDE, ...) that must not change a bit:
    python tools/grad_digest.py [bf16|f32] [B] [T]
prints one digest over all gradients and the loss."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ddim_audio_amd as D  # noqa: E402
from ddim_audio_amd import configs, losses  # noqa: E402
from ddim_audio_amd.schedule import make_schedule  # noqa: E402

dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
T = int(sys.argv[3]) if len(sys.argv) > 3 else 256
cfg = configs.audio_config("torch.cuda.BFloat16Tensor" if dt == "bf16" else "torch.cuda.FloatTensor")
torch.manual_seed(3)
m = D.Model(cfg)
m.train()
g = torch.Generator(device="cuda").manual_seed(5)
x0 = torch.randn(B, cfg.model.channels, T, cfg.model.f_size, device="cuda", generator=g)
e = torch.randn(B, cfg.model.channels, T, cfg.model.f_size, device="cuda", generator=g)
t = torch.randint(0, 1000, (B,), device="cuda", generator=g)
alphas = make_schedule(cfg.diffusion)[1].cuda()
loss = losses.noise_estimation_loss(m, x0, t, e, alphas)
loss.backward()
torch.cuda.synchronize()
h = hashlib.sha256()
for n, p in m.named_parameters():
    h.update(n.encode())
    h.update(p.grad.detach().cpu().numpy().tobytes())
print(f"{dt} B={B} T={T} loss={float(loss):.9g} grads sha256={h.hexdigest()}")
