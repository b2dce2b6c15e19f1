#!/usr/bin/env python3
"""Condense the log of the reference's UNMODIFIED hgemm.py default sweep (tools/run_f1.sh step 2b) into one line per size: the best and the
worst of this library's rows against the script's own `tn(cublas)` row (= hipBLASLt behind the cuBLAS entry names).
    tools/f1_sweep_summary.py gpurun_out/r6Z_f1/f1_hgemm_default_sweep.log"""
import re
import sys

txt = open(sys.argv[1]).read()
blocks = re.split(r"\n\s+M=(\d+), N=\d+, K=\d+, Warmup", txt)
rows = []
for i in range(1, len(blocks), 2):
    n, body = int(blocks[i]), blocks[i + 1]
    vals = {m.group(1).strip(): float(m.group(2)) for m in re.finditer(r"^\s*(\S.*?): \[.*?TFLOPS: ([\d.]+)", body, re.M)}
    ours = {k: v for k, v in vals.items() if "cublas" not in k}
    if not ours or "tn(cublas)" not in vals:
        continue
    kb = max(ours, key=ours.get)
    rows.append((n, ours[kb], kb, min(ours.values()), vals["tn(cublas)"]))
print("# the reference's own bench (hgemm.py --mma-tn --cute-tn --plot, defaults: M = N = K = 256 ... 12800 step 256, warmup 2, iters 10 per row — short bursts, so the")
print("# absolute figures sit below the sustained ones of tools/hgemm_sizes.py), unmodified, against this library's toy_hgemm module; tn(cublas) = hipBLASLt")
print(f"# best row >= tn(cublas) at {sum(b >= c for _, b, _, _, c in rows)} of {len(rows)} sizes; min ratio {min(b / c for _, b, _, _, c in rows):.3f}; "
      f"geometric mean {__import__('math').exp(sum(__import__('math').log(b / c) for _, b, _, _, c in rows) / len(rows)):.3f}")
print("#     n   best row TFLOPS  (row)                                                        worst row   tn(cublas)   best / cublas")
for n, b, kb, w, c in rows:
    print(f"{n:7d}   {b:9.2f}  {kb:66s} {w:9.2f}   {c:9.2f}   {b / c:6.3f}")
