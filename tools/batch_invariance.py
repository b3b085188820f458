"""Check that a sample's forward result is bit-identical alone and inside batches of several sizes (bf16 and fp32)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ddim_audio_amd as D  # noqa: E402
from ddim_audio_amd import configs, synth  # noqa: E402

t_len = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for dtype in ("torch.cuda.BFloat16Tensor", "torch.cuda.FloatTensor"):
    m = synth.fill_module(D.Model(configs.audio_config(dtype))).eval()
    x = synth.gaussian("inv.x", (16, 2, t_len, 256)).cuda()
    t = torch.arange(16).cuda() * 60 + 5
    with torch.no_grad():
        solo = m(x[:1], t[:1])
        for b in (2, 3, 8, 16):
            y = m(x[:b], t[:b])
            print(dtype, "T", t_len, "B", b, "sample 0 identical to B=1:", bool(torch.equal(y[:1], solo)),
                  "max diff", float((y[:1] - solo).abs().max()))
