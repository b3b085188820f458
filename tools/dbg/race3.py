"""Which part of the 2-stream step makes results irreproducible?  Variants of one step, 12 repetitions each, B=64 bf16."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ddim_audio_amd as D
from ddim_audio_amd import configs, synth, schedule, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dts = sys.argv[2] if len(sys.argv) > 2 else "torch.cuda.BFloat16Tensor"
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 12
cfg = configs.audio_config(dts)
m = synth.fill_module(D.Model(cfg)).eval()
lib = _lib.load()
alphas = schedule.make_schedule(cfg.diffusion)[1]
seq = list(range(0, 1000, 100))
coef = torch.from_numpy(schedule.ddim_coefficients(seq, alphas, 0.0).astype("float32")).cuda()
g = torch.Generator(device="cuda"); g.manual_seed(1234)
x = torch.randn(B, 2, 1024, 256, device="cuda", generator=g)
h = B // 2
side = torch.cuda.Stream()
counter = torch.zeros(1, dtype=torch.int32, device="cuda")

def sbegin(t): _lib.check(lib.ddimx_step_begin(_lib.ptr(coef), _lib.ptr(counter), _lib.ptr(t), t.numel(), _lib.stream()))
def upd(xt, et, x0): _lib.check(lib.ddimx_ddim_update(_lib.ptr(xt), _lib.ptr(et), None, _lib.ptr(x0), _lib.ptr(coef), _lib.ptr(counter), xt.numel(), _lib.stream()))

def step(variant):
    xt = x.clone(); x0 = torch.empty_like(xt); t = torch.zeros(B, dtype=torch.int64, device="cuda")
    main = torch.cuda.current_stream()
    if variant == "serial":
        sbegin(t); e = m(xt, t); upd(xt, e, x0)
    elif variant == "serial_halves":   # two half-batches one after the other on ONE stream
        for k, (lo, hi) in enumerate(((0, h), (h, B))):
            sbegin(t[lo:hi]); e = m.forward_slot(xt[lo:hi], t[lo:hi], k); upd(xt[lo:hi], e, x0[lo:hi])
    elif variant == "full":            # what DDIMStepper does: begin + forward + update inside each branch
        side.wait_stream(main)
        with torch.cuda.stream(side):
            sbegin(t[h:]); e1 = m.forward_slot(xt[h:], t[h:], 1); upd(xt[h:], e1, x0[h:])
        sbegin(t[:h]); e0 = m.forward_slot(xt[:h], t[:h], 0); upd(xt[:h], e0, x0[:h])
        main.wait_stream(side)
    elif variant == "t_before_fork":   # t filled for the whole batch before the fork
        sbegin(t)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            e1 = m.forward_slot(xt[h:], t[h:], 1); upd(xt[h:], e1, x0[h:])
        e0 = m.forward_slot(xt[:h], t[:h], 0); upd(xt[:h], e0, x0[:h])
        main.wait_stream(side)
    elif variant == "fwd_only_forked":  # only the two forwards are concurrent; begin before, update after the join
        sbegin(t)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            e1 = m.forward_slot(xt[h:], t[h:], 1)
        e0 = m.forward_slot(xt[:h], t[:h], 0)
        main.wait_stream(side)
        upd(xt[:h], e0, x0[:h]); upd(xt[h:], e1, x0[h:])
    elif variant == "fwd_only_forked_keep":  # same, and eps tensors are preallocated (no allocator traffic on the side stream)
        sbegin(t)
        e = torch.empty_like(xt)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            e1 = m.forward_slot(xt[h:], t[h:], 1)
        e0 = m.forward_slot(xt[:h], t[:h], 0)
        main.wait_stream(side)
        upd(xt[:h], e0, x0[:h]); upd(xt[h:], e1, x0[h:])
    torch.cuda.synchronize()
    return xt

with torch.no_grad():
    m.prepare(x.device, 1024)
    ref = step("serial")
    for v in ("serial", "serial_halves", "full", "t_before_fork", "fwd_only_forked", "full", "fwd_only_forked"):
        bad = []
        for r in range(REPS):
            y = step(v)
            if not torch.equal(y, ref):
                d = (y != ref).flatten(1).any(1).nonzero().flatten().tolist()
                bad.append((r, d[:8], "%.1e" % float((y - ref).abs().max() / ref.abs().max())))
        print(f"{v:22s} mismatching runs {len(bad)}/{REPS}: {bad[:6]}", flush=True)
