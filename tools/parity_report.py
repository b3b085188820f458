"""Print the parity margins (HIP path vs golden vectors from the reference) for both activation dtypes.
Test infrastructure: run on the GPU box, e.g. `python tools/parity_report.py`."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ddim_audio_amd as D  # noqa: E402
from ddim_audio_amd import configs, synth  # noqa: E402


def rel(a, b):
    a, b = torch.as_tensor(a, dtype=torch.float64).reshape(-1), torch.as_tensor(b, dtype=torch.float64).reshape(-1)
    s = float(b.std())
    d = a - b
    return float(d.abs().max()) / s, float(d.square().mean().sqrt()) / s


gm = np.load(os.path.join(ROOT, "tests", "golden", "model.npz"))
for name in ("torch.cuda.FloatTensor", "torch.cuda.BFloat16Tensor"):
    m = synth.fill_module(D.Model(configs.audio_config(name))).eval()
    for tlen in (32, 64):
        x = synth.gaussian(f"model.x{tlen}", (2, 2, tlen, 256)).cuda()
        t = torch.from_numpy(gm[f"model_T{tlen}_t"]).cuda()
        with torch.no_grad():
            y = m(x, t).cpu()
        mx, rms = rel(y, gm[f"model_T{tlen}_y"])
        print(f"{name:28s} T={tlen:3d}  max {mx:.3e}  rms {rms:.3e}  (x std of expected)")
