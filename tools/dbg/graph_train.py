import faulthandler, sys, os, torch
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ddim_audio_amd as D
from ddim_audio_amd import configs, synth, train
from ddim_audio_amd.schedule import make_schedule
b = int(sys.argv[1])
d = configs.audio_dict("torch.cuda.BFloat16Tensor")
d["optimization"]["optimizer"]["default"]["optimizer"] = "AdamW"
cfg = configs.dict2namespace(d)
m = synth.fill_module(D.Model(cfg), 0)
st = train.TrainingState(cfg, m)
alphas = make_schedule(cfg.diffusion)[1].cuda()
x = torch.randn(b, 2, 1024, 256, device="cuda")
g = train.GraphedTrainStep(m, st, alphas, warmup=2)
for i in range(6):
    loss, _ = g(x)
    torch.cuda.synchronize()
    print(i, float(loss), flush=True)
