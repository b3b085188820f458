"""Partitioned sampling: two half-batch samplers, each with a HEAVY stream (levels 0..split-1: most of the chip, shared by both) and a
LIGHT stream (deep levels + FNet: a few CUs), created with hipExtStreamCreateWithCUMask, eager launches, the second sampler offset by
part of a step.   partition_split_test.py [B] [steps]"""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ddim_audio_amd as D
from ddim_audio_amd import configs, schedule, synth
from ddim_audio_amd.sampler import DDIMStepper

hip = ctypes.CDLL("libamdhip64.so")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
cfg = configs.audio_config("torch.cuda.BFloat16Tensor")
model = synth.fill_module(D.Model(cfg)).eval()
alphas = schedule.make_schedule(cfg.diffusion)[1]
coef = schedule.ddim_coefficients(schedule.make_seq(1000, 1000), alphas, 0.0)
x = torch.randn(B, 2, 1024, 256, device="cuda")

def masked(word):
    words = (ctypes.c_uint32 * 8)(*([word] * 8))
    s = ctypes.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words) == 0
    return torch.cuda.ExternalStream(s.value)

def run(heavy_word, light_word, level, delay_frac, step_ms, check=None):
    h = B // 2
    xa, xb = x[:h].clone(), x[h:].clone()
    HA, HB = masked(heavy_word), masked(heavy_word)
    LA, LB = masked(light_word), masked(light_word)
    with torch.cuda.stream(HA):
        ea = [torch.cuda.Event() for _ in range(2)]
        for e in ea: e.record()
        a = DDIMStepper(model, xa, coef, slot=2, fork=False, split=(LA, ea, level))
        for _ in range(3): a.step()
    with torch.cuda.stream(HB):
        eb = [torch.cuda.Event() for _ in range(2)]
        for e in eb: e.record()
        b = DDIMStepper(model, xb, coef, slot=3, fork=False, split=(LB, eb, level))
        for _ in range(3): b.step()
    torch.cuda.synchronize()
    spin = torch.empty(64 << 20, device="cuda")
    t0 = time.perf_counter()
    if delay_frac > 0:
        with torch.cuda.stream(HB):
            for _ in range(max(1, int(delay_frac * step_ms * 1e3 / 70))): spin.zero_()
    for _ in range(steps):
        with torch.cuda.stream(HA): a.step()
        with torch.cuda.stream(HB): b.step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out = torch.cat([xa, xb]).clone()
    a.close(); b.close()
    return B * steps / dt, dt / steps * 1e3, out

with torch.no_grad():
    st = DDIMStepper(model, x.clone(), coef)
    for _ in range(3): st.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): st.step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ref = st.xt.clone(); st.close()
    ms = dt / steps * 1e3
    print(f"default stepper (graph, lock-step shards): {B * steps / dt:.1f} sample-fwd/s, {ms:.3f} ms/step")
    # ONE configuration per process (every run creates four hardware queues; leaked queues of earlier runs slow later ones down)
    heavy, light, level, fr = int(sys.argv[3], 0), int(sys.argv[4], 0), int(sys.argv[5]), float(sys.argv[6])
    v, m2, out = run(heavy, light, level, fr, ms)
    print(f"partitioned heavy {heavy:#010x} | light {light:#010x} per XCD, split level {level}, second delayed {fr:.1f} step: {v:.1f} sample-fwd/s, "
          f"{m2:.3f} ms/step  [bit-identical to the default stepper: {bool(torch.equal(out, ref))}]")
