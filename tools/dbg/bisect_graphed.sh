#!/bin/bash
# One box, the same test file under different settings, one process each: which setting makes the intermittent host crash around
# test_graphed_train_step_is_bit_identical_to_eager go away?   usage: bisect_graphed.sh <outdir>
R=$(cd "$(dirname "$0")/../.." && pwd)
out=$1; mkdir -p $out
export AMD_LOG_LEVEL=1 PYTHONFAULTHANDLER=1 LIBC_FATAL_STDERR_=1 LD_PRELOAD=$R/tools/dbg/abort_bt.so
run() {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python3 -m pytest $R/tests/test_gpu_configs.py -x -q -m gpu -p no:cacheprovider > $out/$name.log 2>&1
  echo "$name rc=$?"
}
# round 4's runs (profiles/r04/graph_fork_crash/): the captured backward with its second-stream branch (DDIMX_CAPTURE_FORK=1) 5 aborts /
# segfaults in 12 runs, always in the eager steps that follow GraphedTrainStep.close(); on one stream (the default since) 0 in 8
for k in 1 2 3; do run capfork1_$k DDIMX_CAPTURE_FORK=1; done
for k in 1 2 3; do run capfork0_$k DDIMX_CAPTURE_FORK=0; done
