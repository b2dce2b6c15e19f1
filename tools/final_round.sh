#!/bin/bash
# final verification round: the driver's own commands + evidence logs
TAG=${1:-r01z}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
set +e
( time timeout 1200 python -m pytest tests/ -x -q -m gpu ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/steps.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/steps.log
( time timeout 900 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/steps.log
timeout 600 python bench.py --workload attn --steps 20 --warmup 2 --no-cpu-baseline > $OUT/bench_attn.json 2> $OUT/bench_attn.err; echo "bench attn rc=$?" | tee -a $OUT/steps.log
timeout 600 python bench.py --layout nn --steps 30 --warmup 3 --no-cpu-baseline --no-attention > $OUT/bench_nn.json 2> $OUT/bench_nn.err; echo "bench nn rc=$?" | tee -a $OUT/steps.log
PYTHONPATH=leetcuda_amd timeout 600 python tools/hgemm_bench.py --mma --mma-all --mma-tn --cute-tn --wmma --cuda --torch --MNK 8192 > $OUT/hgemm_bench_8192.log 2>&1; echo "hgemm_bench rc=$?" | tee -a $OUT/steps.log
PYTHONPATH=leetcuda_amd timeout 600 python tools/hgemm_bench.py --mma --mma-tn --MNK 4096 > $OUT/hgemm_bench_4096.log 2>&1
PYTHONPATH=leetcuda_amd timeout 600 python tools/flash_attn_bench.py --B 4 --H 32 --N 4096 --D 128 --check > $OUT/flash_attn_bench_d128.log 2>&1; echo "fa_bench rc=$?" | tee -a $OUT/steps.log
PYTHONPATH=leetcuda_amd timeout 600 python tools/flash_attn_bench.py --B 1 --H 8 --N 8192 --D 64 --check > $OUT/flash_attn_bench_d64.log 2>&1
PYTHONPATH=leetcuda_amd timeout 600 python tools/flash_attn_bench.py --B 1 --H 48 --N 8192 --D 512 --check --iters 2 > $OUT/flash_attn_bench_d512.log 2>&1
LC_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 2 --steps 5 --warmup 1 > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err; echo "bench2 rc=$?" | tee -a $OUT/steps.log
du -sh $OUT | tee -a $OUT/steps.log
