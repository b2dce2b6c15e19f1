#!/bin/bash
# final validation + profile of a round: full GPU suite, smoke, default bench, then tools/profile_round.sh <tag>
TAG=${1:-final}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -m pytest tests -m gpu -q 2>&1 | tail -12 > $OUT/pytest.log; tail -4 $OUT/pytest.log
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
bash tools/profile_round.sh $TAG > $OUT/profile_round.log 2>&1; tail -5 $OUT/steps.log
ls profiles/${TAG}_* 2>/dev/null
cp profiles/${TAG}_* $OUT/ 2>/dev/null
