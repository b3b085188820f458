#!/bin/bash
# rocprofv3 PMC passes over one level's fused 3x3 conv (tools/conv_one.py): SQ activity, LDS, clock and HBM traffic.
# usage: tools/pmc_conv.sh OUTDIR LEVEL [B] [REPS]      (run on the GPU box; counters in separate passes, with
# --kernel-trace only, as MI355X_MICROARCH.md "rocprofv3 PMC slots" prescribes: SQ 8 slots, TCC 4, GRBM 2)
set -o pipefail
OUT=$1; LVL=$2; B=${3:-8}; REPS=${4:-6}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pass() {  # name, counters...
    local name=$1; shift
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/l${LVL}_$name" -- python3 "$ROOT/tools/conv_one.py" "$LVL" "$B" "$REPS" \
        > "$OUT/l${LVL}_$name.log" 2>&1 || echo "pass $name failed (see $OUT/l${LVL}_$name.log)"
}
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES
pass sq2 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS
pass sq3 SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_WAVES
pass grbm GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
pass write WRITE_SIZE
echo "pmc level $LVL done"
