#!/bin/bash
# one gpurun call of round 3.  usage: tools/r3_round.sh <tag> step...
# a step is a keyword (tests | bench | smoke) or "SECONDS:shell command" (its own timeout; output -> gpurun_out/<tag>/stepN.log)
TAG=${1:-r3a}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OUT
set +e
N=0
for w in "$@"; do
  t0=$(date +%s)
  case $w in
    tests) timeout 1500 python -m pytest tests -m gpu -q --maxfail=8 --durations=12 > $OUT/pytest.log 2>&1; echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)" | tee -a $OUT/steps.log; tail -30 $OUT/pytest.log;;
    smoke) timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$? ($(( $(date +%s) - t0 )) s)" | tee -a $OUT/steps.log; tail -3 $OUT/smoke.log;;
    bench) timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? ($(( $(date +%s) - t0 )) s)" | tee -a $OUT/steps.log; cat $OUT/bench.json; tail -5 $OUT/bench.err;;
    *) N=$((N+1)); T=${w%%:*}; C=${w#*:}; timeout $T bash -c "$C" < /dev/null > $OUT/step$N.log 2>&1; echo "step$N [$C] rc=$? ($(( $(date +%s) - t0 )) s)" | tee -a $OUT/steps.log; tail -40 $OUT/step$N.log;;
  esac
done
