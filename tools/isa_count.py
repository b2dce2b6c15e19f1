#!/usr/bin/env python3
"""Instruction mix of a kernel's steady-state loop, from the audited device assembly (leetcuda_amd/lib/obj/*.s — written by
every build).  VERDICT round 1, item 3 asks for "an ISA count showing <= 2.5 VALU per score element" for the attention kernel:
  tools/isa_count.py                         the default table (attention w4n / w4m, GEMM w4y)
  tools/isa_count.py <file.s> <kernel-regex> <elements-per-lane-per-iteration>
See loop_mix() for how the steady-state path is separated from the overflow guard's slow path."""
import re
import sys
from collections import Counter
from pathlib import Path

OBJ = Path(__file__).resolve().parent.parent / "leetcuda_amd" / "lib" / "obj"


def kernel_lines(path, rx):
    lines = Path(path).read_text().splitlines()
    out, name, on = [], None, False
    for raw in lines:
        m = re.match(r"\s*\.type\s+(\S+),@function", raw)
        if m:
            on = re.search(rx, m.group(1)) is not None and name is None
            if on:
                name = m.group(1)
            continue
        if on and raw.strip().startswith(".Lfunc_end"):
            break
        if on:
            out.append(raw)
    return name, out


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq")):
        return "valu_trans"
    if op.startswith("v_accvgpr"):
        return "valu_accvgpr"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_")):
        return "vmem"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_barrier"):
        return "s_barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


LABEL = re.compile(r"(\.L[\w$]+):")
BRANCH = re.compile(r"s_c?branch\w*\s+(\.L[\w$]+)")


COLD = ("ds_bpermute", "v_permlane", "global_atomic", "flat_atomic")


def loop_mix(lines, label=None):
    """(instructions on the steady-state path of one loop iteration, number of instructions in the loop's cold blocks).
    The loop = the largest label..backward-branch span (label: regex — only spans whose label matches count: a persistent kernel's
    outer tile / block walk spans the whole kernel); its header = the last label in front of the span's first s_barrier
    (hipcc lays the overflow slow path of the second phase out IN FRONT of the header).  From the header the walk follows
    the control flow; at a conditional branch it takes the side that does not run into a cross-lane reduction or an atomic
    (the overflow guard's slow path) within the next 80 instructions and stays inside the span."""
    ins, labels = [], {}
    for raw in lines:
        if raw.lstrip().startswith(";;#"):
            continue
        s = raw.split(";", 1)[0].strip()
        m = LABEL.match(s)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if s and not s.startswith("."):
            ins.append(s)
    best = None
    for i, s in enumerate(ins):
        m = BRANCH.match(s)
        if m and labels.get(m.group(1), 1 << 30) <= i and (label is None or re.search(label, m.group(1))):
            span = (labels[m.group(1)], i + 1)
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    if best is None:
        return [], 0
    bar = next((i for i in range(best[0], best[1]) if ins[i].startswith("s_barrier")), best[0])
    header = max([v for v in labels.values() if best[0] <= v <= bar], default=best[0])

    def cold(i):
        return any(x.startswith(COLD) for x in ins[i:i + 80])

    body, i, seen = [], header, set()
    while i not in seen and len(body) < 20000:
        seen.add(i)
        s = ins[i]
        m = BRANCH.match(s)
        if not m:
            body.append(s)
            i += 1
            continue
        body.append(s)
        tgt = labels[m.group(1)]
        if s.startswith("s_branch"):
            nxt = tgt
        else:
            inside = best[0] <= tgt < best[1]
            if tgt == header:
                break
            if not inside or (cold(tgt) and not cold(i + 1)):
                nxt = i + 1
            elif cold(i + 1) and not cold(tgt):
                nxt = tgt
            else:
                nxt = i + 1
        if nxt == header:
            break
        i = nxt
    return body, (best[1] - best[0]) - len(body)


def report(path, rx, elems, label=None):
    name, lines = kernel_lines(path, rx)
    if not lines:
        print(f"{rx}: not found in {path}")
        return
    body, ncold = loop_mix(lines, label)
    mix = Counter(classify(x.split()[0]) for x in body)
    valu = mix["valu"] + mix["valu_trans"] + mix["valu_accvgpr"]
    nops = sum(int(x.split()[1]) + 1 for x in body if x.startswith("s_nop"))
    print(f"{name}\n  steady-state loop: {len(body)} instructions (+ {ncold} in the loop's overflow-guard blocks, not on the path): "
          + ", ".join(f"{k} {v}" for k, v in sorted(mix.items())) + f"; s_nop wait states {nops}")
    if elems:
        print(f"  per lane and iteration {elems:g} score elements: {valu / elems:.2f} VALU per score element "
              f"({mix['valu_trans'] / elems:.2f} transcendental), {mix['mfma']} MFMA, {valu / max(mix['mfma'], 1):.2f} VALU per MFMA")
    else:
        print(f"  {mix['mfma']} MFMA, {valu} VALU, {mix['lds']} LDS, {mix['vmem']} VMEM, {mix['salu']} SALU per iteration")


if __name__ == "__main__":
    if len(sys.argv) >= 4:
        report(sys.argv[1], sys.argv[2], float(sys.argv[3]))
    else:
        # w4u: one loop iteration = one 64-key tile for 64 query rows per wave = 4096 scores / 64 lanes
        report(OBJ / "tu_attn_w4u_d128.s", r"attn_fwd_w4u_kernelILi128ELb0ELi0", 64)
        report(OBJ / "tu_attn_w4u_d128t.s", r"attn_fwd_w4u_kernelILi128ELb1ELi0", 64)
        report(OBJ / "tu_attn_w4u_d64.s", r"attn_fwd_w4u_kernelILi64ELb0ELi0", 64)
        report(OBJ / "tu_w4.s", r"hgemm_w4y_kernelILb0ELi1E", 0, r"w4y_loop")   # (the K loop, not the persistent tile walk)
