"""Variations of the 2-branch eager sampler at B=64 to find what makes it non-reproducible."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ddim_audio_amd as D
from ddim_audio_amd import configs, synth, schedule, sampler
from ddim_audio_amd.sampler import DDIMStepper
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = configs.audio_config("torch.cuda.BFloat16Tensor")
m = synth.fill_module(D.Model(cfg)).eval()
alphas = schedule.make_schedule(cfg.diffusion)[1]
seq = list(range(0, 1000, 100))
coef = schedule.ddim_coefficients(seq, alphas, 0.0)
g = torch.Generator(device="cuda"); g.manual_seed(1234)
x = torch.randn(B, 2, 1024, 256, device="cuda", generator=g)

class Fake:
    def forward_slot(self, x, t, k): return self(x, t)
    def prepare(self, *a): pass
    def __call__(self, x, t): return torch.tanh(x * 0.7) + 0.001 * t.float().view(-1, 1, 1, 1)

def run(model, nb, graph=False, sync=False, nsteps=3):
    xt = x.clone()
    st = DDIMStepper(model, xt, coef, use_graph=graph, branches=nb)
    outs = []
    for i in range(nsteps):
        st.step()
        if sync: torch.cuda.synchronize()
        outs.append(xt.clone())
    torch.cuda.synchronize()
    return outs

def cmp(tag, a, ref):
    msg = []
    for i, (u, v) in enumerate(zip(a, ref)):
        if torch.equal(u, v): msg.append("eq")
        else:
            d = (u != v).flatten(1).any(1).nonzero().flatten().tolist()
            msg.append("DIFF samples %s maxrel %.2e" % (d[:6] + (["..."] if len(d) > 6 else []), float((u - v).abs().max() / v.abs().max())))
    print(tag, msg, flush=True)

with torch.no_grad():
    ref = run(m, 1)
    for r in range(3): cmp(f"V1 model nb=2 eager rep{r}", run(m, 2), ref)
    for r in range(2): cmp(f"V4 model nb=2 eager sync-each-step rep{r}", run(m, 2, sync=True), ref)
    for r in range(2): cmp(f"V1g model nb=2 graph rep{r}", run(m, 2, graph=True), ref)
    keep, m._temb_table = m._temb_table, None
    refm = run(m, 1)
    cmp("V6 ref(MLP temb) vs ref(table)", refm, ref)
    for r in range(2): cmp(f"V6 model nb=2 eager, MLP temb rep{r}", run(m, 2), refm)
    m._temb_table = keep
    f = Fake()
    reff = run(f, 1)
    for r in range(3): cmp(f"V2 fake nb=2 eager rep{r}", run(f, 2), reff)
    for r in range(2): cmp(f"V2 fake nb=4 eager rep{r}", run(f, 4), reff)
    for r in range(2): cmp(f"V1 model nb=4 eager rep{r}", run(m, 4), ref)
    os.environ["HIP_LAUNCH_BLOCKING"] = "0"
