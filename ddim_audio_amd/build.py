"""Build libddimx.so (hand-written HIP for gfx950) in-tree with hipcc.

``python -m ddim_audio_amd.build`` or ``build()``.  hipcc cross-compiles without a GPU.  Objects go
to ``csrc/build/`` (git-ignored); the shared library lands next to the sources so it travels with
the tree to the GPU box.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libddimx.so")
SOURCES = ["api.cpp", "kernels.hip", "gemm.hip", "fnet_dense.hip", "conv_inst_bf16_c3.hip", "conv_inst_bf16_du.hip",
           "conv_inst_f32_c3.hip", "conv_inst_f32_du.hip", "conv_inst_bf16_c3b.hip", "conv_inst_bf16_wreg.hip", "conv_inst_bf16_pipe.hip", "conv_inst_f32_c3b.hip", "train_kernels.hip", "wgrad_inst_bf16.hip", "wgrad_inst_f32.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-Wno-unused-result"]
# conv_pipe.h: MFMA accumulators in VGPRs (the epilogue reads them with plain vector instructions, no v_accvgpr_read per element)
# and no SLP packing of its scalar f32 arithmetic into v_pk_*_f32 (an anti-lever beside MFMAs, MI355X_MICROARCH.md)
EXTRA_FLAGS = {"conv_inst_bf16_pipe.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, jobs=None, stamp=False):
    """stamp=True builds the diagnostic variant libddimx_stamp.so (-DDDIMX_STAMP: in-kernel phase timers)."""
    global OUT, FLAGS
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    bdir = os.path.join(CSRC, "build_stamp" if stamp else "build")
    if stamp:
        OUT = os.path.join(HERE, "libddimx_stamp.so")
        FLAGS = FLAGS + ["-DDDIMX_STAMP"]
    os.makedirs(bdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "ddimx.h"))
    todo, objs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        ob = os.path.join(bdir, src.rsplit(".", 1)[0] + ".o")
        objs.append(ob)
        if force or _stale(ob, [sp] + headers):
            todo.append((sp, ob))

    def cc(job):
        sp, ob = job
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(sp), []) + ["-c", sp, "-o", ob]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (sp, r.stderr[-4000:]))
        return sp

    if todo:
        with ThreadPoolExecutor(max_workers=jobs or min(8, len(todo))) as ex:
            for sp in ex.map(cc, todo):
                if verbose:
                    print("[ddimx build] compiled", os.path.basename(sp), flush=True)
    if todo or not os.path.exists(OUT):
        r = subprocess.run([hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", OUT] + objs,
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-4000:])
        if verbose:
            print("[ddimx build] linked", OUT, flush=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, stamp="--stamp" in sys.argv)
