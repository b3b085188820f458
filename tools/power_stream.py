"""Board power / shader clock while HBM-streaming kernels loop for ~3 s each (what does moving bytes alone cost on this board?):
   resid at level 0 (3 passes: read x, read h, write y + statistics) and a plain device-to-device copy (1 read + 1 write).
   power_stream.py [B]     prints us per launch, TB/s of the algorithmic bytes, watts and MHz sampled in the middle of each loop."""
import os, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddim_audio_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lib = _lib.load()
dt, tdt = _lib.DDIMX_BF16, torch.bfloat16
C, H, W = 32, 1024, 256
x = torch.randn(B, H, W, C, device="cuda").to(tdt); h = torch.randn(B, H, W, C, device="cuda").to(tdt); y = torch.empty_like(x)
scale = torch.rand(B, C, device="cuda") + 0.5; shift = torch.randn(B, C, device="cuda") * 0.1
stats = torch.zeros(int(lib.ddimx_conv3x3_stats_floats(dt, C, B, H, W)) * 4, device="cuda")

def smi():
    out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks", "-d", "0"], capture_output=True, text=True).stdout
    w = [l.split(":")[-1].strip() for l in out.splitlines() if "Power (W)" in l]
    c = [l.split("(")[-1].strip(")") for l in out.splitlines() if "sclk" in l]
    return (w[0] if w else "?"), (c[0] if c else "?")

def measure(name, fn, nbytes, n):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    samples = []
    def sampler():
        time.sleep(1.0)
        for _ in range(3):
            samples.append(smi()); time.sleep(0.3)
    th = threading.Thread(target=sampler); th.start()
    t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    dt_ = time.time() - t0
    th.join()
    us = dt_ * 1e6 / n
    print(f"{name}: {us:.1f} us per launch over {n} launches ({dt_:.1f} s), {nbytes / us / 1e6:.2f} TB/s; power / clock samples: {samples}")

def resid():
    _lib.check(lib.ddimx_resid_gn_fwd(dt, C, _lib.ptr(x), _lib.ptr(h), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(y), _lib.ptr(stats), B, H, W, _lib.stream()))
measure(f"resid_kernel level 0 B={B} (3 passes of {x.numel() * 2 / 1e6:.0f} MB)", resid, 3 * x.numel() * 2, 40000)
measure(f"device-to-device copy of {x.numel() * 2 / 1e6:.0f} MB", lambda: y.copy_(x), 2 * x.numel() * 2, 50000)
