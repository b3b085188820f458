#!/bin/bash
# rocprofv3 --pmc over the sampling step (one SQ pass): per-kernel VALU / MFMA / wait shares.  tools/step_pmc.sh OUTDIR
OUT=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS \
  --kernel-trace --output-format csv -d "$OUT" -- python3 "$ROOT/bench.py" --steps 12 --warmup 2 --no-cpu-baseline --no-train-leg --no-extra-legs --no-roofline "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
csv.field_size_limit(1 << 30)
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float); cnt = collections.defaultdict(int); seen = set()
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    m = re.search(r"(conv3_wreg_kernel<ddimx::WregCfg<\d+)|(conv3_pipe_kernel<ddimx::PipeCfg<\d+, \d+, \d+, \d)|(ConvCfgIDF16bLi\d+ELi\d+ELi\d+ELi\d)|(resid_kernel)|(gemm_nt)|(fnet_mix2)|(fnet_dense_kernel<[^>]*>)|(fnet_mix)|(gemm_splitk)|(layernorm)|(gemm_reduce_ln)|(conv_out)|(conv_in)", n)
    key = m.group(0) if m else n[:40]
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
    did = r["Dispatch_Id"]
    if did not in seen:
        seen.add(did); cnt[key] += 1
        if r.get("End_Timestamp"): dur[key] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
tot = sum(dur.values())
print(f"{'kernel':44s} {'calls':>6s} {'ms':>8s} {'%':>5s} {'VALU%':>6s} {'MFMA%':>6s} {'LDS%':>5s} {'w/SIMD':>6s} {'issue%':>6s} {'stall%':>6s} {'park%':>6s}")
for k in sorted(dur, key=lambda k: -dur[k])[:22]:
    c = agg[k]; d = dur[k]  # ns
    cyc = d * 2.2 * 1024    # SIMD-cycles at ~2.2 GHz
    wc = c["SQ_WAVE_CYCLES"] or 1
    print(f"{k:44s} {cnt[k]:6d} {d/1e6:8.2f} {100*d/tot:5.1f} {100*4*c['SQ_ACTIVE_INST_VALU']/cyc:6.1f} {100*c['SQ_VALU_MFMA_BUSY_CYCLES']/cyc:6.1f} {100*4*c['SQ_ACTIVE_INST_LDS']/cyc:5.1f} {4*wc/cyc:6.2f} {100*c['SQ_ACTIVE_INST_ANY']/wc:6.1f} {100*c['SQ_WAIT_INST_ANY']/wc:6.1f} {100*c['SQ_WAIT_ANY']/wc:6.1f}")
PY
