import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ddim_audio_amd as D
from ddim_audio_amd import configs, synth
for dts in ("torch.cuda.FloatTensor", "torch.cuda.BFloat16Tensor"):
    m = synth.fill_module(D.Model(configs.audio_config(dts))).eval()
    for B, T in ((5, 64), (4, 64), (5, 1024)):
        x = synth.gaussian("fork.x", (B, 2, T, 256)).cuda()
        t = (torch.arange(B) * 177 % 1000).cuda()
        with torch.no_grad():
            for mask in (0, 0x10003, 0x1003F):
                m.fork_mask = mask
                y = m(x, t).clone()
                y2 = m(x.flip(0).contiguous(), t.flip(0).contiguous()).clone()
                d = (y2 != y.flip(0)).flatten(1).any(1).nonzero().flatten().tolist()
                # graph
                xs, ts = x.clone(), t.clone()
                g = torch.cuda.CUDAGraph()
                m(xs, ts); torch.cuda.synchronize()
                with torch.cuda.graph(g):
                    ys = m(xs, ts)
                g.replay(); torch.cuda.synchronize()
                e1 = bool(torch.equal(ys, y))
                xs.copy_(x.flip(0)); ts.copy_(t.flip(0)); g.replay(); torch.cuda.synchronize()
                dg = (ys != y.flip(0)).flatten(1).any(1).nonzero().flatten().tolist()
                mx = float((ys - y.flip(0)).abs().max() / y.abs().max())
                print(f"{dts[11:]:16s} B={B} T={T} mask={mask:#x}: eager flip differing samples {d}; graph replay == eager {e1}; graph flipped differing samples {dg} (max rel {mx:.1e})", flush=True)
