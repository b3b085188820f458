"""Does Model.forward read workspace bytes it never wrote?  Pre-fill the workspace with two different patterns and compare."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ddim_audio_amd as D
from ddim_audio_amd import configs, synth
for dts in ("torch.cuda.BFloat16Tensor", "torch.cuda.FloatTensor"):
    cfg = configs.audio_config(dts)
    m = synth.fill_module(D.Model(cfg)).eval()
    for B in (8, 32, 33):
        g = torch.Generator(device="cuda"); g.manual_seed(1)
        x = torch.randn(B, 2, 1024, 256, device="cuda", generator=g)
        t = torch.randint(0, 1000, (B,), device="cuda")
        outs = []
        with torch.no_grad():
            m(x, t)
            for pat in (0x00, 0xFF, 0x7F, 0x3C):
                m._workspace[0].fill_(pat)
                outs.append(m(x, t).clone())
            ys = m(x[B // 2:], t[B // 2:])
        torch.cuda.synchronize()
        print(dts, "B", B, "equal across workspace fills:", [bool(torch.equal(o, outs[0])) for o in outs], "finite", bool(torch.isfinite(outs[1]).all()),
              "second half alone equal:", bool(torch.equal(ys, outs[0][B // 2:])))
