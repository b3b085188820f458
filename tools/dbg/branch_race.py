"""Which of {graph, eager} x {1, 2 branches} runs of the 50-step B=64 bf16 sampler agree bit for bit?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ddim_audio_amd as D
from ddim_audio_amd import configs, synth, schedule
from ddim_audio_amd.sampler import DDIMStepper
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
cfg = configs.audio_config("torch.cuda.BFloat16Tensor")
m = synth.fill_module(D.Model(cfg)).eval()
alphas = schedule.make_schedule(cfg.diffusion)[1]
seq = list(range(0, 1000, 1000 // nsteps))
coef = schedule.ddim_coefficients(seq, alphas, 0.0)
g = torch.Generator(device="cuda"); g.manual_seed(1234)
x = torch.randn(B, 2, 1024, 256, device="cuda", generator=g)
res = {}
with torch.no_grad():
    for rep in range(2):
        for nb in (1, 2):
            for graph in (True, False):
                xt = x.clone()
                st = DDIMStepper(m, xt, coef, use_graph=graph, branches=nb)
                hist = []
                for i in range(len(seq)):
                    st.step()
                    if i in (0, 1, 2, 5, 10, 20, len(seq) - 1):
                        hist.append(xt.clone())
                torch.cuda.synchronize()
                res[(rep, nb, graph)] = hist
keys = list(res)
ref = res[keys[0]]
for k in keys:
    eq = [bool(torch.equal(a, b)) for a, b in zip(res[k], ref)]
    print(k, "equal to", keys[0], "at checkpoints (steps 0,1,2,5,10,20,last):", eq, "max|x| %.3g" % float(res[k][-1].abs().max()))
