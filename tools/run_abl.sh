mkdir -p gpurun_out/r01g; export TMPDIR=/tmp
python tools/attn_ablate.py > gpurun_out/r01g/ablate.log 2>&1
rocprofv3 -L > gpurun_out/r01g/counters.txt 2>&1
cat > /tmp/attn_only.py <<'PY'
import sys; sys.path.insert(0,'.')
import torch
from leetcuda_amd import capi, host
capi.load(); capi.tune("attn_nw", 8)
q,k,v,o,_ = host.get_qkvo(4,32,4096,128,seed=0)
for _ in range(3): capi.attn_fwd(q,k,v,o)
torch.cuda.synchronize()
PY
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES -d gpurun_out/r01g/pmc_a -o pmc -- python /tmp/attn_only.py > gpurun_out/r01g/pmc_a.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU_TRANS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT -d gpurun_out/r01g/pmc_b -o pmc -- python /tmp/attn_only.py > gpurun_out/r01g/pmc_b.log 2>&1
