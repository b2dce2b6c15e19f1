#!/bin/bash
# one gpurun call of round 2: GPU test suite, the bench line, a variant sweep.  usage: tools/r2_round.sh <tag> [what...]
# every further argument is ONE step: a keyword (tests newtests bench sweep) or a whole shell command (quote it)
TAG=${1:-r2a}; shift
if [ $# -eq 0 ]; then set -- tests bench sweep; fi
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
set +e
N=0
for w in "$@"; do
  case $w in
    tests) timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/steps.log; tail -25 $OUT/pytest.log;;
    newtests) timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_bench.py -m gpu -x -q --durations=10 > $OUT/pytest_new.log 2>&1; echo "pytest-new rc=$?" | tee -a $OUT/steps.log; tail -25 $OUT/pytest_new.log;;
    bench) timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/steps.log; cat $OUT/bench.json; tail -5 $OUT/bench.err;;
    sweep) timeout 600 python bench.py --sweep --quick --no-attention > $OUT/sweep.json 2> $OUT/sweep.err; echo "sweep rc=$?" | tee -a $OUT/steps.log; grep sweep $OUT/sweep.err;;
    *) N=$((N+1)); timeout 300 bash -c "$w" < /dev/null > $OUT/extra$N.log 2>&1; echo "extra$N [$w] rc=$?" | tee -a $OUT/steps.log; tail -60 $OUT/extra$N.log;;
  esac
done
