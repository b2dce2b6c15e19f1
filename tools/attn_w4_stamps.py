#!/usr/bin/env python3
"""Phase cycle stamps of the 4-wave x 64-row attention kernel (attn_nw=128 + attn_ablate=32).
Needs the diagnostic kernel instantiations: `LC_DIAG=1 python -m leetcuda_amd.build --force` first."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from leetcuda_amd import capi, host  # noqa: E402
capi.load()
q, k, v, o, _ = host.get_qkvo(4, 32, 4096, 128, seed=0)
capi.tune("attn_nw", 128)
for _ in range(5):
    capi.attn_fwd(q, k, v, o)
torch.cuda.synchronize()
capi.tune("attn_ablate", 32)
capi.attn_fwd(q, k, v, o)
torch.cuda.synchronize()
st = q.view(-1)[:4 * 4 * 4].view(torch.int64).cpu().numpy().reshape(4, 4).astype("int64")
for t in range(4):
    r = st[t]
    nxt = int(st[t + 1, 0] - r[3]) if t < 3 else -1
    print(f"tile{16+t}: start={int(r[0]-st[0,0]):6d} wait+barrier={int(r[1]-r[0]):5d} phase1={int(r[2]-r[1]):5d} phase2={int(r[3]-r[2]):5d} loop={nxt}")
capi.tune("attn_ablate", 0)
capi.tune("attn_nw", 0)
