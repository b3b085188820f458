"""Compile the HIP sources with -Rpass-analysis=kernel-resource-usage and print VGPR/AGPR/spill/scratch/LDS
per kernel (build tooling; run in the authoring container)."""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ddim_audio_amd", "csrc")
files = sys.argv[1:] or ["conv_inst_bf16_c3.hip", "conv_inst_bf16_du.hip", "conv_inst_f32_c3.hip", "conv_inst_f32_du.hip"]
for f in files:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c",
                        os.path.join(CSRC, f), "-o", "/tmp/_kr.o", "-Rpass-analysis=kernel-resource-usage"],
                       capture_output=True, text=True)
    cur = None
    for line in r.stderr.splitlines():
        m = re.search(r"remark: (?:\s*)(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
        elif cur is not None:
            cur[k.split(" ")[0] if k != "VGPRs Spill" else "Spill"] = v
            if k.startswith("LDS Size"):
                n = cur["name"]
                mm = re.search(r"ConvCfgI(DF16b|f)Li(\d+)ELi(\d+)ELi(\d+)ELi(\d)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)", n)
                if mm:
                    n = f"conv<{'bf16' if mm.group(1) == 'DF16b' else 'f32'},cin={mm.group(2)},nout={mm.group(3)},nb={mm.group(4)},mode={mm.group(5)},tile={mm.group(6)}x{mm.group(7)},waves={mm.group(8)}x{mm.group(9)}>"
                flag = "  <-- SPILL/SCRATCH" if cur.get("Spill", "0") != "0" or cur.get("ScratchSize", "0") != "0" else ""
                print(f"{f:24s} LDS={cur.get('LDS')} V={cur.get('VGPRs'):>4} A={cur.get('AGPRs'):>4} occ={cur.get('Occupancy')} spill={cur.get('Spill')} "
                      f"scratch={cur.get('ScratchSize')}  {n[:90]}{flag}")
                cur = None
