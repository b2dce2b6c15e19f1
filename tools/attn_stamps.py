#!/usr/bin/env python3
"""Per-phase cycle stamps of the lock-step attention kernel (diagnosis; lc_tune_set attn_ablate=32).
Needs the diagnostic kernel instantiations: `LC_DIAG=1 python -m leetcuda_amd.build --force` first."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from leetcuda_amd import capi, host  # noqa: E402
capi.load()
q, k, v, o, _ = host.get_qkvo(4, 32, 4096, 128, seed=0)
capi.tune("attn_nw", 8)
for _ in range(5):
    capi.attn_fwd(q, k, v, o)
torch.cuda.synchronize()
capi.tune("attn_ablate", 32)
capi.attn_fwd(q, k, v, o)
torch.cuda.synchronize()
st = q.view(-1)[:2 * 4 * 8 * 4].view(torch.int64).cpu().numpy().reshape(2, 4, 8).astype("int64")
names = ["top->QKdone", "QK->max", "max..rescale->exp start", "exp+PV", "ds_write", "barrier"]
for w in range(2):
    for t in range(4):
        r = st[w, t]
        d = [int(r[i + 1] - r[i]) for i in range(5)]
        nxt = int(st[w, t + 1, 0] - r[5]) if t < 3 else -1
        print(f"wave{w*4} tile{16+t}: start={int(r[0]-st[0,0,0])} QK={d[0]} max={d[1]} dec={d[2]} exp+PV={d[3]} dswrite={d[4]} barrier={int(r[5]-r[4])} loopback={nxt}")
capi.tune("attn_ablate", 0)
capi.tune("attn_nw", 0)
