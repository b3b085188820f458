#!/bin/bash
# The GPU test suite, once, in one process, with a log that no later run overwrites.
#   gpurun --timeout 1200 -- 'bash tools/gpu_suite.sh [extra pytest args]'
# stdout+stderr (faulthandler dumps, AMD_LOG_LEVEL=1 messages of the HIP runtime) -> gpurun_out/r04/gputest_<time>.log;
# tests/conftest.py adds gpurun_out/crash/<time>_<pid>.txt with one flushed line per test start.
set -o pipefail
mkdir -p gpurun_out/r04
log=gpurun_out/r04/gputest_$(date +%Y%m%d_%H%M%S).log
export AMD_LOG_LEVEL=${AMD_LOG_LEVEL:-1}
export PYTHONFAULTHANDLER=1
export LIBC_FATAL_STDERR_=1   # glibc's heap-corruption messages go to stderr (the log), not to a controlling terminal
# native backtrace of whichever thread aborts / segfaults (faulthandler shows Python frames only): tools/dbg/abort_bt.c
R=$(cd "$(dirname "$0")/.." && pwd)
[ -f $R/tools/dbg/abort_bt.so ] || gcc -shared -fPIC -O1 -o $R/tools/dbg/abort_bt.so $R/tools/dbg/abort_bt.c -ldl 2>/dev/null
[ -f $R/tools/dbg/abort_bt.so ] && export LD_PRELOAD=$R/tools/dbg/abort_bt.so
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -p no:cacheprovider "$@" > "$log" 2>&1
rc=$?
tail -n 25 "$log"
echo "exit=$rc log=$log"
exit $rc
