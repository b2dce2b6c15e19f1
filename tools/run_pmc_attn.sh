mkdir -p gpurun_out/$1; export TMPDIR=/tmp
cat > /tmp/attn_var.py <<'PY'
import sys; sys.path.insert(0,'.')
import torch
from leetcuda_amd import capi, host
capi.load()
q,k,v,o,_ = host.get_qkvo(4,32,4096,128,seed=0)
for nw in (8,16,32):
    capi.tune("attn_nw", nw)
    for _ in range(3): capi.attn_fwd(q,k,v,o)
    torch.cuda.synchronize()
PY
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES -d gpurun_out/$1/pmc_c -o pmc -- python /tmp/attn_var.py > gpurun_out/$1/pmc_c.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d gpurun_out/$1/pmc_d -o pmc -- python /tmp/attn_var.py > gpurun_out/$1/pmc_d.log 2>&1
