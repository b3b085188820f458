#!/bin/bash
# Round-closing measurements on one box: the default bench line (with cpu_baseline and every secondary leg), the other BASELINE
# shapes, B = 8-only kernel statistics (forked step and single-stream step: roofline.frac is recomputable from these), the training
# step's kernel statistics, the GPU suite's summary, the per-tensor gradient parity lines.
#   gpurun --timeout 1200 -- 'bash tools/final_bench.sh profiles/rNN/final'
out=$1; mkdir -p $out; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
python bench.py > $out/bench_b8_t1024_bf16.json 2> $out/bench_b8.err
python bench.py --batch 64 --steps 30 --no-cpu-baseline --no-train-leg --no-extra-legs > $out/bench_b64_bf16.json 2>> $out/bench_b8.err
python bench.py --batch 32 --steps 50 --no-cpu-baseline --no-train-leg --no-extra-legs > $out/bench_b32_bf16.json 2>> $out/bench_b8.err
python bench.py --batch 4 --t-size 4096 --steps 50 --no-cpu-baseline --no-train-leg --no-extra-legs > $out/bench_b4_t4096_bf16.json 2>> $out/bench_b8.err
python bench.py --batch 1 --t-size 8192 --steps 50 --no-cpu-baseline --no-train-leg --no-extra-legs > $out/bench_b1_t8192_bf16.json 2>> $out/bench_b8.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pfin /tmp/pfin1 /tmp/pfint
# (--no-extra-legs: the traces hold the B = 8 sampling loop ONLY)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pfin -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-train-leg --no-extra-legs > /tmp/pfin.log 2>&1
cp $(find /tmp/pfin -name "*kernel_stats.csv" | head -n1) $R/$out/bench_b8_t1024_bf16_kernel_stats.csv
DDIMX_FORK_MASK=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pfin1 -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-train-leg --no-extra-legs > /tmp/pfin1.log 2>&1
cp $(find /tmp/pfin1 -name "*kernel_stats.csv" | head -n1) $R/$out/bench_b8_t1024_bf16_1stream_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pfint -- python3 $R/tools/train_bench.py 32 1024 3 bf16 > /tmp/pfint.log 2>&1
cp $(find /tmp/pfint -name "*kernel_stats.csv" | head -n1) $R/$out/train_b32_t1024_bf16_kernel_stats.csv
cd $R; for f in $out/bench_*.json; do echo $f; tail -n1 $f | cut -c1-260; done
