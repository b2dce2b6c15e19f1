#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r6f; mkdir -p $OUT
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6f/bench_default.json"))
r = d["roofline"]
print("value", d["value"], "keys:", list(r)[:16])
print({k: r[k] for k in list(r)[:16] if not isinstance(r[k], (dict, list))})
print("per_rank", d.get("per_rank"))
print("cfg4 per_rank", d["attention_cfg4"]["per_rank"], d["attention_cfg4"].get("checksum"))
PY
timeout 3000 python -m pytest tests/test_gpu_bench.py tests/test_gpu_dist.py -q -x > $OUT/pytest_bench.log 2>&1; tail -8 $OUT/pytest_bench.log
