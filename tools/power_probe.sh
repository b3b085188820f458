#!/bin/bash
# Board power / shader clock while one conv kernel loops for ~4 s: watts x microseconds per launch = energy per launch.
#   tools/power_probe.sh LEVEL B XF PIPE(0|1) [LAUNCHES]      (DDIMX_LIB / DDIMX_PIPE_DBG pass through: stage-less variants)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
lvl=$1; b=$2; xf=$3; pipe=$4; n=${5:-50000}
rm -f /tmp/power_probe.out
DDIMX_ONE_TIME=$n DDIMX_ONE_XF=$xf DDIMX_ONE_PIPE=$pipe python3 -u $R/tools/conv_one.py $lvl $b 5 > /tmp/power_probe.out 2>&1 &
pid=$!
while ! grep -q "^done" /tmp/power_probe.out 2>/dev/null; do sleep 0.2; done   # the warm-up launches are through: the timed loop runs
sleep 0.8
for i in 1 2 3; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks -d 0 2>/dev/null | grep -E "Power|sclk" | sed 's/.*: //' | tr '\n' ' ' | sed 's/=* Power Consumption =*//'; echo
  sleep 0.4
done
wait $pid
grep sustained /tmp/power_probe.out
