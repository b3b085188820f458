"""Two independent half-batch samplers, each on its own HALF OF THE CHIP (streams created with hipExtStreamCreateWithCUMask: 16 of every
XCD's 32 CUs each), launched eagerly (a captured graph does not keep a stream's CU mask), optionally offset by half a step: does a
shard that owns its CUs keep its latency-bound middle moving while the other half runs power-bound convolutions?
   partition_test.py [B] [steps]"""
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
model.fork_mask = 0
alphas = schedule.make_schedule(cfg.diffusion)[1]
coef = schedule.ddim_coefficients(schedule.make_seq(1000, 1000), alphas, 0.0)
x = torch.randn(B, 2, 1024, 256, device="cuda")

def masked(word):
    words = (ctypes.c_uint32 * 8)(*([word] * 8))
    s = ctypes.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words) == 0
    return torch.cuda.ExternalStream(s.value)

def run(sa, sb, delay_frac, step_ms, graph):
    h = B // 2
    xa, xb = x[:h].clone(), x[h:].clone()
    with torch.cuda.stream(sa):
        a = DDIMStepper(model, xa, coef, use_graph=graph, slot=2, fork=False)
        for _ in range(3): a.step()
    with torch.cuda.stream(sb):
        b = DDIMStepper(model, xb, coef, use_graph=graph, slot=3, fork=False)
        for _ in range(3): b.step()
    torch.cuda.synchronize()
    spin = torch.empty(64 << 20, device="cuda")
    t0 = time.perf_counter()
    if delay_frac > 0:
        with torch.cuda.stream(sb):
            for _ in range(max(1, int(delay_frac * step_ms * 1e3 / 70))): spin.zero_()
    for _ in range(steps):
        with torch.cuda.stream(sa): a.step()
        with torch.cuda.stream(sb): b.step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    a.close(); b.close()
    return B * steps / dt, dt / steps * 1e3

with torch.no_grad():
    model.fork_mask = 0x1003f
    st = DDIMStepper(model, x.clone(), coef)
    for _ in range(4): st.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): st.step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st.close()
    ms = dt / steps * 1e3
    print(f"default stepper (graph, lock-step shards): {B * steps / dt:.1f} sample-fwd/s, {ms:.3f} ms/step")
    model.fork_mask = 0
    for name, sa, sb, graph in (("plain streams, eager", torch.cuda.Stream(), torch.cuda.Stream(), False),
                                ("half-chip streams (CU masks 0x0000FFFF | 0xFFFF0000), eager", masked(0x0000FFFF), masked(0xFFFF0000), False),
                                ("interleaved half-chip streams (0x55555555 | 0xAAAAAAAA), eager", masked(0x55555555), masked(0xAAAAAAAA), False)):
        for fr in (0.0, 0.5):
            v, m2 = run(sa, sb, fr, ms, graph)
            print(f"two half-batch steppers, {name}, second delayed by {fr:.1f} step: {v:.1f} sample-fwd/s, {m2:.3f} ms/step")
