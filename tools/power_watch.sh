#!/bin/bash
# Sample board power / shader clock (rocm-smi) every ~0.2 s while a workload runs: is a kernel's lower effective clock
# the POWER cap?  usage: tools/power_watch.sh <tag> -- <command...>      -> gpurun_out/<tag>/power.log
TAG=$1; shift; [ "$1" = "--" ] && shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
( while true; do
    echo "t=$(date +%s.%N) $(rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E 'Power|sclk|mclk|fclk|socclk|Temperature \(Sensor junction' | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';')"
    sleep 0.15
  done ) > $OUT/power.log 2>&1 &
SAMPLER=$!
"$@"
RC=$?
kill $SAMPLER 2>/dev/null
exit $RC
