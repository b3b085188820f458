"""Two forward passes on two streams vs the same passes run one after the other: where do they first disagree?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ddim_audio_amd as D
from ddim_audio_amd import configs, synth

def al(v): return (v + 255) & ~255

def fields(B, T, es):
    ch = [32, 64, 96, 128, 192, 256]
    out, off = [], 0
    def take(name, n):
        nonlocal off
        out.append((name, off, off + n)); off += al(n)
    take("temb_h1", B * 512 * 4); take("temb_h2", B * 512 * 4); take("temb", B * 4416 * 4)
    take("A", B * T * 256 * 32 * es)
    for l, c in enumerate(ch):
        n = B * (T >> l) * (256 >> l) * c * es
        take(f"xd{l}", n); take(f"xu{l}", n)
    take("h1", B * T * 256 * 32 * es); take("h2", B * T * 256 * 32 * es)
    out.append(("rest(stats,scale,shift,fnet)", off, 1 << 62))
    return out

def where(off, fl):
    for name, lo, hi in fl:
        if lo <= off < hi:
            return f"{name}+{off - lo}"
    return "?"

dts = sys.argv[1] if len(sys.argv) > 1 else "torch.cuda.BFloat16Tensor"
Bh = int(sys.argv[2]) if len(sys.argv) > 2 else 32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
es = 2 if "BFloat16" in dts else 4
cfg = configs.audio_config(dts)
m = synth.fill_module(D.Model(cfg)).eval()
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.randn(2 * Bh, 2, 1024, 256, device="cuda", generator=g)
t = torch.randint(0, 1000, (2 * Bh,), device="cuda")
side = torch.cuda.Stream()
fl = fields(Bh, 1024, es)
with torch.no_grad():
    m.prepare(x.device, 1024)
    ya = m.forward_slot(x[:Bh], t[:Bh], 0).clone(); yb = m.forward_slot(x[Bh:], t[Bh:], 1).clone()
    torch.cuda.synchronize()
    wa, wb = m._workspace[0].clone(), m._workspace[1].clone()
    # serial again: byte-identical workspaces?
    m.forward_slot(x[:Bh], t[:Bh], 0); m.forward_slot(x[Bh:], t[Bh:], 1); torch.cuda.synchronize()
    print("serial repeat: workspace identical", bool(torch.equal(wa, m._workspace[0])), bool(torch.equal(wb, m._workspace[1])))
    for rep in range(reps):
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            y1 = m.forward_slot(x[Bh:], t[Bh:], 1)
        y0 = m.forward_slot(x[:Bh], t[:Bh], 0)
        main.wait_stream(side)
        torch.cuda.synchronize()
        ok0, ok1 = bool(torch.equal(y0, ya)), bool(torch.equal(y1, yb))
        msg = f"rep {rep}: out equal {ok0} {ok1}"
        for nm, ws, ref in (("slot0", m._workspace[0], wa), ("slot1", m._workspace[1], wb)):
            n = min(ws.numel(), ref.numel())
            d = (ws[:n] != ref[:n])
            if bool(d.any()):
                idx = d.nonzero().flatten()
                first, last, cnt = int(idx[0]), int(idx[-1]), int(idx.numel())
                # which fields contain differences
                hit = []
                for name, lo, hi in fl:
                    hi2 = min(hi, n)
                    if lo < n and bool(d[lo:hi2].any()):
                        k = int(d[lo:hi2].nonzero().flatten()[0])
                        hit.append(f"{name}(first +{k}, {int(d[lo:hi2].sum())} bytes)")
                msg += f" | {nm}: {cnt} bytes differ, first at {where(first, fl)}; fields: {hit}"
        print(msg, flush=True)
