"""Do kernels of two HIP streams overlap on this GPU?  N launches on one stream vs N/2 + N/2 on two streams."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ddim_audio_amd import _lib
lib = _lib.load()
dt, tdt = _lib.DDIMX_BF16, torch.bfloat16
CH = [32, 64, 96, 128, 192, 256]

def mk(l, B):
    C, H, W = CH[l], 1024 >> l, 256 >> l
    x = torch.randn(B, H, W, C, device="cuda").to(tdt); y = torch.empty_like(x)
    temb = torch.randn(B, C, device="cuda") * 0.3
    gw = [torch.rand(C, device="cuda") + 0.5 for _ in range(3)]; gb = [torch.randn(C, device="cuda") * 0.1 for _ in range(3)]
    w0 = (torch.randn(9 * C * C, device="cuda") / (9 * C) ** 0.5).to(tdt); w1 = (torch.randn(9 * C * C, device="cuda") / (9 * C) ** 0.5).to(tdt)
    ws = torch.empty(int(lib.ddimx_op_workspace_bytes(dt, B, C, H, W)), dtype=torch.uint8, device="cuda")
    def run():
        _lib.check(lib.ddimx_resblock_fwd(dt, C, _lib.ptr(x), _lib.ptr(y), _lib.ptr(temb), C, _lib.ptr(gw[0]), _lib.ptr(gb[0]), _lib.ptr(w0), _lib.ptr(gw[1]),
                                          _lib.ptr(gb[1]), _lib.ptr(w1), _lib.ptr(gb[2]), _lib.ptr(gw[2]), _lib.ptr(ws), B, H, W, _lib.stream()))
    return run

def timeit(fn):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e6

def mk_downup(l, B):  # down l-1 -> l, then up l -> l-1 (+skip)
    Cp, C, H, W = CH[l - 1], CH[l], 1024 >> (l - 1), 256 >> (l - 1)
    x = torch.randn(B, H, W, Cp, device="cuda").to(tdt)
    wd = (torch.randn(16 * C * Cp, device="cuda") / (16 * Cp) ** 0.5).to(tdt); bd = torch.randn(C, device="cuda") * 0.1
    y = torch.empty(B, H // 2, W // 2, C, device="cuda", dtype=tdt)
    wu = (torch.randn(2 * 6 * 2 * Cp * C, device="cuda") / (16 * C) ** 0.5).to(tdt); bu = torch.randn(2 * Cp, device="cuda") * 0.1
    z = torch.empty_like(x)
    def run():
        _lib.check(lib.ddimx_downsample_fwd(dt, Cp, C, _lib.ptr(x), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(y), B, H, W, _lib.stream()))
        _lib.check(lib.ddimx_upsample_add_fwd(dt, C, Cp, _lib.ptr(y), _lib.ptr(wu), _lib.ptr(bu), _lib.ptr(x), _lib.ptr(z), B, H // 2, W // 2, _lib.stream()))
    return run

def mk_fnet(B):
    import ctypes
    import ddim_audio_amd as D
    from ddim_audio_amd import configs, synth
    global _m
    if "_m" not in globals():
        _m = synth.fill_module(D.Model(configs.audio_config("torch.cuda.BFloat16Tensor"))).eval()
        _m._ensure_handle(); _m.prepare(torch.device("cuda", 0), 1024)
    m = _m
    pe, dh, ds = m._ensure_tables(1024, torch.device("cuda", 0))
    x = torch.randn(B, 32, 8, 256, device="cuda").to(tdt)
    ws = torch.empty(int(lib.ddimx_workspace_bytes(m._handle, B, 1024)), dtype=torch.uint8, device="cuda")
    out = torch.empty(B * 32, 2048, device="cuda")
    tb = _lib.DdimxTables(pe.data_ptr(), dh.data_ptr(), ds.data_ptr())
    def run():
        _lib.check(lib.ddimx_fnet_fwd(m._handle, _lib.ptr(m._packed), ctypes.byref(tb), _lib.ptr(ws), ws.numel(), _lib.ptr(x), _lib.ptr(out), B, 1024, _lib.stream()))
    return run

side = torch.cuda.Stream()
N = 40
cases = [(f"resblock L{l}", (lambda B, l=l: mk(l, B))) for l in (0, 1, 2, 3, 4, 5)]
cases += [(f"down+up L{l}", (lambda B, l=l: mk_downup(l, B))) for l in (1, 2, 3, 4, 5)]
cases += [("fnet", mk_fnet)]
for name, f in cases:
    l = name
    a, b = f(4), f(4)
    c = f(8)
    def one():
        for _ in range(N): a(); b()
    def two():
        main = torch.cuda.current_stream(); side.wait_stream(main)
        with torch.cuda.stream(side):
            for _ in range(N): b()
        for _ in range(N): a()
        main.wait_stream(side)
    def big():
        for _ in range(N): c()
    g1, g2, g3 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    one(); two(); big(); torch.cuda.synchronize()
    with torch.cuda.graph(g1): one()
    with torch.cuda.graph(g2): two()
    with torch.cuda.graph(g3): big()
    t1, t2, t3 = timeit(g1.replay), timeit(g2.replay), timeit(g3.replay)
    print(f"{l:14s}: B=4 x2 one stream {t1 / N:7.1f} us/pair | two streams {t2 / N:7.1f} us/pair | one B=8 launch chain {t3 / N:7.1f} us", flush=True)
