"""Which op gives different results when another stream keeps the GPU busy?  Victim ops run on a side stream through the
per-op C ABI while the main stream runs whole forwards (the aggressor); outputs are compared with a quiet run."""
import os, sys, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ddim_audio_amd as D
from ddim_audio_amd import configs, synth, _lib
import gpu_util as G
lib = _lib.load()
dt = G.BF16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else G.F32
tdt = G.TORCH_DT[dt]
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = configs.audio_config("torch.cuda.BFloat16Tensor" if dt == G.BF16 else "torch.cuda.FloatTensor")
m = synth.fill_module(D.Model(cfg)).eval()
g = torch.Generator(device="cuda"); g.manual_seed(7)
xa = torch.randn(32, 2, 1024, 256, device="cuda", generator=g); ta = torch.randint(0, 1000, (32,), device="cuda")
side = torch.cuda.Stream()
B = 8
CH = [32, 64, 96, 128, 192, 256]

def mk_resblock(l):
    C, H, W = CH[l], 1024 >> l, 256 >> l
    x = torch.randn(B, H, W, C, device="cuda", generator=g).to(tdt); y = torch.empty_like(x)
    temb = torch.randn(B, C, device="cuda", generator=g) * 0.3
    gw = [torch.rand(C, device="cuda", generator=g) + 0.5 for _ in range(3)]; gb = [torch.randn(C, device="cuda", generator=g) * 0.1 for _ in range(3)]
    w0 = (torch.randn(9 * C * C, device="cuda", generator=g) / (9 * C) ** 0.5).to(tdt); w1 = (torch.randn(9 * C * C, device="cuda", generator=g) / (9 * C) ** 0.5).to(tdt)
    ws = torch.empty(int(lib.ddimx_op_workspace_bytes(dt, B, C, H, W)), dtype=torch.uint8, device="cuda")
    def run():
        _lib.check(lib.ddimx_resblock_fwd(dt, C, _lib.ptr(x), _lib.ptr(y), _lib.ptr(temb), C, _lib.ptr(gw[0]), _lib.ptr(gb[0]), _lib.ptr(w0), _lib.ptr(gw[1]),
                                          _lib.ptr(gb[1]), _lib.ptr(w1), _lib.ptr(gb[2]), _lib.ptr(gw[2]), _lib.ptr(ws), B, H, W, _lib.stream()))
        return y
    return run

def mk_conv(l, xf, act):
    C, H, W = CH[l], 1024 >> l, 256 >> l
    x = torch.randn(B, H, W, C, device="cuda", generator=g).to(tdt); y = torch.empty_like(x)
    w = (torch.randn(9 * C * C, device="cuda", generator=g) / (9 * C) ** 0.5).to(tdt)
    temb = torch.randn(B, C, device="cuda", generator=g) * 0.1
    sc = torch.rand(B, C, device="cuda", generator=g) + 0.5; sh = torch.randn(B, C, device="cuda", generator=g) * 0.1
    stats = torch.zeros(int(lib.ddimx_conv3x3_stats_floats(dt, C, B, H, W)), device="cuda")
    def run():
        _lib.check(lib.ddimx_conv3x3_fwd(dt, C, _lib.ptr(x), _lib.ptr(w), None, _lib.ptr(temb), C, _lib.ptr(sc), _lib.ptr(sh), xf, act, _lib.ptr(y), _lib.ptr(stats), B, H, W, _lib.stream()))
        return torch.cat([y.float().flatten(), stats])
    return run

def mk_downup(l):  # down l-1 -> l and up l -> l-1
    Cp, C, H, W = CH[l - 1], CH[l], 1024 >> (l - 1), 256 >> (l - 1)
    x = torch.randn(B, H, W, Cp, device="cuda", generator=g).to(tdt)
    wd = (torch.randn(16 * C * Cp, device="cuda", generator=g) / (16 * Cp) ** 0.5).to(tdt); bd = torch.randn(C, device="cuda", generator=g) * 0.1
    y = torch.empty(B, H // 2, W // 2, C, device="cuda", dtype=tdt)
    wu = (torch.randn(2 * 6 * 2 * Cp * C, device="cuda", generator=g) / (16 * C) ** 0.5).to(tdt); bu = torch.randn(2 * Cp, device="cuda", generator=g) * 0.1
    z = torch.empty_like(x)
    def run():
        _lib.check(lib.ddimx_downsample_fwd(dt, Cp, C, _lib.ptr(x), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(y), B, H, W, _lib.stream()))
        _lib.check(lib.ddimx_upsample_add_fwd(dt, C, Cp, _lib.ptr(y), _lib.ptr(wu), _lib.ptr(bu), _lib.ptr(x), _lib.ptr(z), B, H // 2, W // 2, _lib.stream()))
        return torch.cat([y.float().flatten(), z.float().flatten()])
    return run

def mk_fnet():
    lib2 = m._ensure_handle(); m.prepare(xa.device, 1024)
    pe, dh, ds = m._ensure_tables(1024, xa.device)
    x = torch.randn(B, 32, 8, 256, device="cuda", generator=g).to(tdt)
    ws = torch.empty(int(lib.ddimx_workspace_bytes(m._handle, B, 1024)), dtype=torch.uint8, device="cuda")
    out = torch.empty(B * 32, 2048, device="cuda")
    tb = _lib.DdimxTables(pe.data_ptr(), dh.data_ptr(), ds.data_ptr())
    def run():
        _lib.check(lib.ddimx_fnet_fwd(m._handle, _lib.ptr(m._packed), ctypes.byref(tb), _lib.ptr(ws), ws.numel(), _lib.ptr(x), _lib.ptr(out), B, 1024, _lib.stream()))
        return out
    return run

def mk_edge():
    x = torch.randn(B, 2, 1024, 256, device="cuda", generator=g)
    w = torch.randn(32, 2, 3, 3, device="cuda", generator=g) * 0.2; b = torch.randn(32, device="cuda", generator=g) * 0.1
    y = torch.empty(B, 1024, 256, 32, device="cuda", dtype=tdt)
    st = torch.zeros(int(lib.ddimx_conv_in_stats_floats(B, 32, 1024, 256)), device="cuda")
    wo = torch.randn(9 * 2 * 32, device="cuda", generator=g) * 0.1; bo = torch.randn(2, device="cuda", generator=g) * 0.1
    eps = torch.empty(B, 2, 1024, 256, device="cuda")
    def run():
        _lib.check(lib.ddimx_conv_in_fwd(dt, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), _lib.ptr(st), B, 2, 32, 1024, 256, _lib.stream()))
        _lib.check(lib.ddimx_conv_out_fwd(dt, _lib.ptr(y), _lib.ptr(y), _lib.ptr(wo), _lib.ptr(bo), _lib.ptr(eps), B, 32, 2, 1024, 256, _lib.stream()))
        return torch.cat([eps.flatten(), st])
    return run

victims = [(f"resblock L{l}", mk_resblock(l)) for l in range(6)]
victims += [(f"conv3x3 L{l} xf=2 act=1", mk_conv(l, 2, 1)) for l in range(6)]
victims += [(f"down/up L{l}", mk_downup(l)) for l in range(1, 6)]
victims += [("fnet_fwd", mk_fnet()), ("conv_in+conv_out", mk_edge())]
with torch.no_grad():
    m(xa, ta); torch.cuda.synchronize()
    for name, fn in victims:
        ref = fn().clone(); torch.cuda.synchronize()
        quiet = sum(0 if torch.equal(fn(), ref) else 1 for _ in range(4)); torch.cuda.synchronize()
        bad = 0
        for r in range(REPS):
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            m(xa, ta)                                   # aggressor on the main stream (about 15 ms of GPU work)
            with torch.cuda.stream(side):
                outs = [fn().clone() for _ in range(3)]  # victim runs on the side stream meanwhile
            main.wait_stream(side)
            torch.cuda.synchronize()
            bad += sum(0 if torch.equal(o, ref) else 1 for o in outs)
        print(f"{name:28s} quiet mismatches {quiet}/4, with aggressor {bad}/{3 * REPS}", flush=True)
