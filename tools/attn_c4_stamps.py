#!/usr/bin/env python3
"""Segment cycle stamps of the four-cluster attention kernel (diagnosis; attn_nw=64 + attn_ablate=32).
Needs the diagnostic kernel instantiations: `LC_DIAG=1 python -m leetcuda_amd.build --force` first."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from leetcuda_amd import capi, host  # noqa: E402
capi.load()
q, k, v, o, _ = host.get_qkvo(4, 32, 4096, 128, seed=0)
capi.tune("attn_nw", 64)
for _ in range(5):
    capi.attn_fwd(q, k, v, o)
torch.cuda.synchronize()
names = ["L1", "bar", "C1", "bar", "SM", "bar", "C2", "bar+loop"]
q0 = q.clone()
for abl in (32,):
    q.copy_(q0)
    capi.tune("attn_ablate", abl)
    capi.attn_fwd(q, k, v, o)
    torch.cuda.synchronize()
    st = q.view(-1)[:2 * 4 * 8 * 4].view(torch.int64).cpu().numpy().reshape(2, 4, 8).astype("int64")
    for w in range(2):
        for t in (1, 2):
            r = st[w, t]
            d = [int(r[i + 1] - r[i]) for i in range(7)]
            d.append(int(st[w, t + 1, 0] - r[7]) if t < 3 else -1)
            print(f"abl {abl} wave{w*4} tile{16+t}: start={int(r[0]-st[0,0,0]):6d} " + " ".join(f"{n}={x:5d}" for n, x in zip(names, d)))
fl = host.mha_matmul_flops(4, 32, 4096, 128)
for abl in (0, 4, 8, 16):
    q.copy_(q0)
    capi.tune("attn_ablate", abl)
    ms = capi.attn_time(q, k, v, o, False, capi.ATTN_SPLIT_Q, 2, warmup=3, iters=20)
    print(f"abl {abl}: {ms:.4f} ms {fl / ms * 1e-9:7.1f} TF-equivalent")
capi.tune("attn_ablate", 0)
capi.tune("attn_nw", 0)
