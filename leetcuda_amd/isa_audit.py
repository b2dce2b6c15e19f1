"""Post-build ISA audit of the literal-AGPR / asm-load kernels.

The 4-wave kernels keep state where hipcc cannot see it: accumulators (and, in the attention kernel, Q / K fragments)
live in LITERAL AGPRs across separate asm statements, and `ds_read_b64_tr_b16` / `ds_read_b128` issued from asm
statements return asynchronously although hipcc believes their "=v" outputs are written at `;;#ASMEND`.  Correctness
therefore depends on properties of the EMITTED code that a ROCm or flag change could silently break.  This module
re-derives them from the device assembly of every translation unit that contains such kernels
(`hipcc -S --cuda-device-only`, same flags as the shipped objects) and `leetcuda_amd.build` fails the build when
one does not hold:

  R1  no compiler-emitted `v_accvgpr_*` (i.e. outside ;;#ASMSTART / ;;#ASMEND) touches an AGPR the kernel owns
      (w4 GEMM / fp8 kernels: a0-a255; attention w4 kernels: the ranges named in their clobber lists)
  R2  no scratch: `.private_segment_fixed_size` == 0 (a spilled tuple would be reloaded around asm statements
      without the wait states the asm needs)
  R4  kernels whose asm owns LITERAL arch VGPRs ACROSS statements (attn_fwd_w4i_kernel: v[LB:255] are reserved registers that carry
      the softmax state from one phase statement to the next): no compiler-emitted instruction names one of them.  (hgemm_w4y_kernel's
      literal fragment registers live INSIDE its one K-loop statement — initialised there, all named as clobbers, nothing read by
      another statement — so hipcc may use them between two executions of it; until round 4 the rule covered them too, vacuously.)
  R3  no compiler instruction reads or writes the destination registers of an asm-issued LDS / global load between
      the load and the next `s_waitcnt ... lgkmcnt(0)` / `vmcnt(0)` that retires it
  R5  no non-MFMA instruction names an arch VGPR — and (round 6) no compiler-emitted v_accvgpr_* an AGPR — written by an asm-issued MFMA before the MFMA's result latency has
      passed (MFMA_STATES wait states, counted conservatively: s_nop N = N + 1, an MFMA = 4, anything else = 1).  hipcc
      does not know that the statement is an MFMA and inserts no wait states; the hardware does not interlock; a bare
      `asm volatile("s_nop ...")` is no fence for compiler-scheduled VALU code (the round-2 attention kernels read Sᵀ one
      k-step short that way — only when issue was back to back, so results depended on instruction-cache state)
  R7  counted-wait protocol: LDS operations return in order, so `s_waitcnt lgkmcnt(N)` retires all but the N youngest; an
      instruction that names the destination of an asm-issued LDS read which is still among the outstanding ones at that
      point (queue replayed in file order; reads issued by hipcc itself count in the queue but are hipcc's business)
      is a violation — this is what checks the `lgkmcnt(6)` / `lgkmcnt(14)`-style waits inside the attention statements
  Control flow: the rules replay their hazard state (MFMA results in flight, fresh VALU writes, the in-order LDS queue,
  un-retired asm loads) in FILE order; in addition, at every BACKWARD branch the first REPLAY_WINDOW instructions behind the
  branch target are re-checked with the state as it stands at the branch — the hazard between the last instructions of a loop
  body and the first of the next iteration.  Forward branches (the if / else of the overflow slow paths)
  are still followed in file order only: the fall-through state reaches the join, the taken-branch state does not; the
  bit-equality / determinism GPU tests remain the guard for those.  One exception (round 6): a label that
  file order cannot reach by fall-through — the instruction in front of it is an unconditional `s_branch` / `s_endpgm` — starts from
  the state recorded at the forward branch(es) that target it (union of their hazards; the longest LDS queue) instead of the state of
  the unrelated block that happens to precede it in the file (hipcc lays cold prologue / tail blocks out behind the loop they belong
  in front of: hgemm_mid_kernel's K < NS tiles prologue sits behind its loop body).
  R6  no asm-issued MFMA reads an arch VGPR that a VALU instruction wrote fewer than VALU_TO_MFMA_STATES wait states
      earlier (hipcc pads this for its own MFMAs — LLVM's "legacy VALU write VGPR -> MFMA read" rule — but not in front of
      an asm statement: a v_cvt_pk of the P fragment scheduled right in front of the statement that consumes it made
      the pipelined D = 512 kernel read a half-written operand)
"""
from __future__ import annotations

import re
from dataclasses import dataclass, field
from pathlib import Path

# kernel-name regex -> AGPR ranges owned by the kernel's asm statements (inclusive)
OWNED_AGPRS = [
    (re.compile(r"hgemm_w4b_kernel|hgemm_w4x_kernel|hgemm_w4y_kernel|gemm_fp8_w4_kernel|gemm_fp8_w4k_kernel"), [(0, 255)]),
    (re.compile(r"attn_fwd_w4u_kernel|attn_fwd_w4i_kernel|attn_fwd_bigd2_kernel|attn_fwd_bigd3_kernel|attn_fwd_bigd4_kernel|attn_fwd_bigd6_kernel|attn_fwd_bigd7_kernel"), [(0, 255)]),
]

# kernel-name regex -> literal arch VGPR range owned by the kernel's asm (inclusive)
OWNED_VGPRS = [
    # attn_w4i: v[LB:255] are RESERVED registers (amdgpu_num_vgpr(LB)) holding the softmax state under literal names across
    # statements (tools/gen_attn_w4i.py register map): no compiler instruction may ever name one
    (re.compile(r"attn_fwd_w4i_kernelILi32E"), (104, 255)),
    (re.compile(r"attn_fwd_w4i_kernelILi64E"), (88, 255)),
    (re.compile(r"attn_fwd_w4i_kernelILi96E"), (72, 255)),
    (re.compile(r"attn_fwd_w4i_kernelILi128E"), (64, 255)),
]

VALU_TO_MFMA_STATES = 2
REPLAY_WINDOW = 96      # instructions re-checked behind the target of a backward branch (covers the longest latency budget)
MFMA_STATES = {"16x16": 12, "32x32": 20, "4x4": 8}   # result latency budget per MFMA shape family (wait states)

_REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")
_ASM_LOAD = re.compile(r"^\s*(ds_read\w*|ds_load\w*|global_load_(?!lds)\w+|buffer_load_\w+)\s+(.*)$")


def _regs(text: str, kind: str) -> set[int]:
    out: set[int] = set()
    for m in _REG.finditer(text):
        if m.group(1) == kind:
            out.add(int(m.group(2)))
        elif m.group(3) == kind:
            out.update(range(int(m.group(4)), int(m.group(5)) + 1))
    return out


@dataclass
class KernelReport:
    name: str
    scratch: int = 0
    vgpr_count: int = 0
    agpr_count: int = 0
    violations: list[str] = field(default_factory=list)
    mfma: int = 0
    compiler_accvgpr: int = 0
    asm_loads: int = 0


def _owned(name: str):
    for rx, ranges in OWNED_AGPRS:
        if rx.search(name):
            s: set[int] = set()
            for lo, hi in ranges:
                s.update(range(lo, hi + 1))
            return s
    return None


_BRANCH = re.compile(r"s_c?branch\w*\s+(\.L[\w$]+)")


def _viol(cur, tag, msg):
    msg += tag
    if msg not in cur.violations:
        cur.violations.append(msg)


def _step(st, cur, path, rec, owned, owned_v, count=True, tag=""):
    """Apply the hazard rules to ONE instruction.  st = {"pending", "mfma_busy", "valu_fresh", "lgkm"} (module docstring);
    count=False on the loop-back-edge replay (statistics are counted once)."""
    s, in_asm, ln = rec
    pending, mfma_busy, valu_fresh, lgkm = st["pending"], st["mfma_busy"], st["valu_fresh"], st["lgkm"]
    # ---- R7: counted lgkmcnt waits vs asm-issued LDS reads
    m7 = re.search(r"lgkmcnt\((\d+)\)", s) if s.startswith("s_waitcnt") else None
    if m7:
        keep = int(m7.group(1))
        if len(lgkm) > keep:
            # (round 6) what a counted wait retires is retired for R3 too: hgemm_mid_kernel settles one k-step's fragments with
            # lgkmcnt(N) while the next k-step's reads stay outstanding, and hipcc may recycle the settled registers behind the MFMAs
            for dst, from_asm in lgkm[:len(lgkm) - keep]:
                if from_asm:
                    pending -= {r for k, r in dst if k == "v"}
            st['lgkm'] = lgkm = lgkm[len(lgkm) - keep:] if keep else []
    elif s.startswith("ds_"):
        if lgkm:
            named = {("v", r) for r in _regs(s, "v")} | {("a", r) for r in _regs(s, "a")}
            for dst, from_asm in lgkm:
                if from_asm and dst & named:
                    _viol(cur, tag, f"R7 {path.name}:{ln}: `{s}` names {sorted(k + str(n) for k, n in dst & named)[:4]} while the asm LDS read "
                                          "that writes it is still outstanding")
        is_read = s.startswith(("ds_read", "ds_load", "ds_bpermute", "ds_permute", "ds_swizzle"))
        d0 = s.split(None, 1)[1].split(",")[0]
        dst = frozenset({("v", r) for r in _regs(d0, "v")} | {("a", r) for r in _regs(d0, "a")}) if is_read else frozenset()
        lgkm.append((dst, in_asm))
    elif lgkm and not s.startswith("s_"):
        named = {("v", r) for r in _regs(s, "v")} | {("a", r) for r in _regs(s, "a")}
        for dst, from_asm in lgkm:
            if from_asm and dst & named:
                _viol(cur, tag, f"R7 {path.name}:{ln}: `{s}` names {sorted(k + str(n) for k, n in dst & named)[:4]} while the asm LDS read "
                                      "that writes it is still outstanding")
                break
    # ---- R6: VALU write (arch VGPR, or AGPR through v_accvgpr_write) -> asm MFMA read
    if s.startswith("v_mfma"):
        if in_asm and valu_fresh:
            ops = s.split(None, 1)[1]
            hit = ({("v", r) for r in _regs(ops, "v")} | {("a", r) for r in _regs(ops, "a")}) & valu_fresh.keys()
            if hit:
                _viol(cur, tag, f"R6 {path.name}:{ln}: asm `{s}` reads {sorted(k + str(n) for k, n in hit)[:4]} "
                                      f"{max(valu_fresh[h] for h in hit)} wait states too early after a VALU write")
        st['valu_fresh'] = valu_fresh = {}
    else:
        m6 = re.match(r"s_nop\s+(\d+)", s)
        adv6 = int(m6.group(1)) + 1 if m6 else 1
        st['valu_fresh'] = valu_fresh = {r: n - adv6 for r, n in valu_fresh.items() if n - adv6 > 0}
        if s.startswith("v_") and not s.startswith(("v_cmp", "v_cmpx")):
            d6 = s.split(None, 1)[1].split(",")[0]
            for r in _regs(d6, "v"):
                valu_fresh[("v", r)] = VALU_TO_MFMA_STATES
            for r in _regs(d6, "a"):
                valu_fresh[("a", r)] = VALU_TO_MFMA_STATES
    # ---- R5: result latency of asm-issued MFMAs that write arch VGPRs
    if mfma_busy:
        if not in_asm and s.startswith("v_accvgpr_"):
            # (round 6) AGPR results of asm MFMAs (keys 1000 + n): hipcc's own accumulator reads behind an asm MFMA — hgemm_mid_kernel's
            # epilogue was hoisted above a bare `s_nop` statement and read one 16 x 16 block an MFMA early
            hit_a = {1000 + r for r in _regs(s, "a")} & mfma_busy.keys()
            if hit_a:
                _viol(cur, tag, f"R5 {path.name}:{ln}: compiler `{s}` names a{sorted(h - 1000 for h in hit_a)[:4]} "
                                      f"{max(mfma_busy[h] for h in hit_a)} wait states before the asm MFMA result is there")
                for h in hit_a:
                    del mfma_busy[h]
        if not s.startswith("v_mfma") and not s.startswith("s_"):
            hit = _regs(s, "v") & mfma_busy.keys()
            if hit:
                _viol(cur, tag, f"R5 {path.name}:{ln}: `{s}` names v{sorted(hit)[:4]} "
                                      f"{max(mfma_busy[h] for h in hit)} wait states before the asm MFMA result is there")
                for h in hit:
                    del mfma_busy[h]
        m = re.match(r"s_nop\s+(\d+)", s)
        adv = int(m.group(1)) + 1 if m else (4 if s.startswith("v_mfma") else 1)
        st['mfma_busy'] = mfma_busy = {r: n - adv for r, n in mfma_busy.items() if n - adv > 0}
    if s.startswith("v_mfma"):
        cur.mfma += int(count)
        if in_asm:
            dst = s.split(None, 1)[1].split(",")[0]
            need = next((v for k, v in MFMA_STATES.items() if k in s.split()[0]), 20)
            for r in _regs(dst, "v"):
                mfma_busy[r] = need
            for r in _regs(dst, "a"):
                mfma_busy[1000 + r] = need
    is_wait = s.startswith("s_waitcnt") and ("lgkmcnt(0)" in s or "vmcnt(0)" in s)
    if is_wait:
        # a counted wait retires everything older in program order; a plain lgkmcnt(0) retires LDS reads, a
        # vmcnt(0) global ones — asm loads in these kernels are LDS reads, global asm loads are followed by
        # vmcnt waits written next to them
        pending.clear()
        return
    if in_asm:
        m = _ASM_LOAD.match(s)
        if m and " lds" not in s and not s.rstrip().endswith("lds"):
            dst = m.group(2).split(",")[0]
            pending |= _regs(dst, "v")
            cur.asm_loads += int(count)
        return
    # ---- compiler-emitted instruction
    if owned_v:
        hit = _regs(s, "v") & owned_v
        if hit:
            _viol(cur, tag, f"R4 {path.name}:{ln}: compiler `{s}` names asm-owned VGPR(s) v{sorted(hit)[:4]}")
    if s.startswith("v_accvgpr_"):
        cur.compiler_accvgpr += int(count)
        if owned is not None:
            hit = _regs(s, "a") & owned
            if hit:
                _viol(cur, tag, f"R1 {path.name}:{ln}: compiler `{s}` touches asm-owned AGPR(s) {sorted(hit)[:4]}")
    if pending:
        hit = _regs(s, "v") & pending
        if hit:
            _viol(cur, tag, f"R3 {path.name}:{ln}: compiler `{s}` uses v{sorted(hit)[:4]} before the wait "
                                  "that retires the asm load writing it")
            pending -= hit     # report each register once


def audit_asm(path: Path) -> list[KernelReport]:
    """Two walks over the file: the first only records the hazard state at EVERY branch (keyed by function and target label), the second
    is the audit proper and starts every label that file order cannot reach by fall-through from the merged state of the branches that
    do reach it — forward ones and loop back-edges alike (hipcc places a rotated loop's latch block in FRONT of its header, entered
    only by a backward branch: hgemm_mid_kernel)."""
    lines = Path(path).read_text().splitlines()
    _, known = _walk(path, lines, None)
    out, _ = _walk(path, lines, known)
    return out


def _walk(path: Path, lines, known):
    reports: dict[str, KernelReport] = {}
    seen: dict[tuple, list] = {}   # (function, label) -> hazard states at the branches that target it (returned for the second walk)
    fn_name = ""
    cur: KernelReport | None = None
    owned: set[int] | None = None
    owned_v: set[int] = set()
    in_asm = False
    pending: set[int] = set()     # VGPR destinations of asm loads not yet retired by a wait
    mfma_busy: dict[int, int] = {}   # arch VGPR written by an asm MFMA -> wait states until its result is readable
    valu_fresh: dict[tuple, int] = {}  # ("v" | "a", n) written by a VALU instruction -> wait states until an MFMA may read it
    # (the four hazard states live in `st`, one dict per function, so that a loop back-edge can replay with a copy)
    lgkm: list[tuple[frozenset, bool]] = []   # outstanding LDS operations in issue order: (VGPR destinations, asm-issued)
    st = {"pending": pending, "mfma_busy": mfma_busy, "valu_fresh": valu_fresh, "lgkm": lgkm}
    fn_ins: list[tuple[str, bool, int]] = []     # instructions of the current function in file order
    fn_labels: dict[str, int] = {}               # label -> index into fn_ins of the first instruction behind it
    fwd_states: dict[str, list[dict]] = {}       # label not yet seen -> hazard states at the forward branches that target it

    def _copy(x):
        return {"pending": set(x["pending"]), "mfma_busy": dict(x["mfma_busy"]), "valu_fresh": dict(x["valu_fresh"]), "lgkm": list(x["lgkm"])}

    def _merge(states):
        out = _copy(states[0])
        for y in states[1:]:
            out["pending"] |= y["pending"]
            for key in ("mfma_busy", "valu_fresh"):
                for r, n in y[key].items():
                    out[key][r] = max(out[key].get(r, 0), n)
            if len(y["lgkm"]) > len(out["lgkm"]):
                out["lgkm"] = list(y["lgkm"])
        return out

    for ln, raw in enumerate(lines, 1):
        line = raw.split(";", 1)[0] if not raw.lstrip().startswith(";;#") else raw
        s = line.strip()
        m = re.match(r"\.type\s+(\S+),@function", s)
        if m:
            name = m.group(1)
            cur = reports.setdefault(name, KernelReport(name))
            owned = _owned(name)
            owned_v = set()
            for rx, (lo, hi) in OWNED_VGPRS:
                if rx.search(name):
                    owned_v = set(range(lo, hi + 1))
            in_asm = False
            st = {"pending": set(), "mfma_busy": {}, "valu_fresh": {}, "lgkm": []}
            fn_ins, fn_labels, fwd_states = [], {}, {}
            fn_name = name
            continue
        if s.startswith(".Lfunc_end"):
            cur = None
            continue
        m = re.match(r"\.amdhsa_kernel\s+(\S+)", s)
        if m:
            cur = reports.setdefault(m.group(1), KernelReport(m.group(1)))
            owned = None
            continue
        if cur is not None and s.startswith(".amdhsa_private_segment_fixed_size"):
            cur.scratch = int(s.split()[-1])
            continue
        if s.startswith(".end_amdhsa_kernel"):
            cur = None
            continue
        if cur is None:
            continue
        if raw.lstrip().startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if raw.lstrip().startswith(";;#ASMEND"):
            in_asm = False
            continue
        ml = re.match(r"(\.L[\w$]+):", s)
        if ml:
            fn_labels[ml.group(1)] = len(fn_ins)
            if fn_ins and not in_asm and fn_ins[-1][0].split()[0] in ("s_branch", "s_endpgm"):
                # no fall-through into this label: the state of the branches that reach it (second walk: every branch of the function)
                reach = (known or {}).get((fn_name, ml.group(1))) or fwd_states.get(ml.group(1))
                if reach:
                    st = _merge(reach)
        if not s or s.endswith(":") or s.startswith("."):
            continue
        rec = (s, in_asm, ln)
        fn_ins.append(rec)
        _step(st, cur, path, rec, owned, owned_v, count=True)
        mb = _BRANCH.match(s)
        # (an edge taken with EXEC == 0 carries no hazard: no lane is active to observe one)
        if mb and not s.startswith("s_cbranch_execz"):
            seen.setdefault((fn_name, mb.group(1)), []).append(_copy(st))
        if mb and mb.group(1) not in fn_labels and not s.startswith("s_cbranch_execz"):
            fwd_states.setdefault(mb.group(1), []).append(_copy(st))
        if mb and mb.group(1) in fn_labels:
            # backward branch: re-check the head of the loop with the state carried over from its tail
            st2 = {"pending": set(st["pending"]), "mfma_busy": dict(st["mfma_busy"]), "valu_fresh": dict(st["valu_fresh"]),
                   "lgkm": list(st["lgkm"])}
            t0 = fn_labels[mb.group(1)]
            for rec2 in fn_ins[t0:t0 + REPLAY_WINDOW]:
                if _BRANCH.match(rec2[0]) and rec2 is not rec and not rec2[0].startswith("s_cbranch"):
                    break                      # an unconditional jump inside the window: file order says nothing beyond it
                _step(st2, cur, path, rec2, owned, owned_v, count=False, tag=" [loop back-edge]")
    # metadata block: .name / .vgpr_count / .agpr_count
    txt = "\n".join(lines)
    for blk in re.split(r"\n  - \.agpr_count:", txt)[1:]:
        nm = re.search(r"\.name:\s+(\S+)", blk)
        if not nm or nm.group(1) not in reports:
            continue
        r = reports[nm.group(1)]
        r.agpr_count = int(blk.strip().split()[0])
        m = re.search(r"\.vgpr_count:\s+(\d+)", blk)
        r.vgpr_count = int(m.group(1)) if m else 0
        m = re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk)
        if m:
            r.scratch = max(r.scratch, int(m.group(1)))
    out = []
    for r in reports.values():
        if _owned(r.name) is None and r.asm_loads == 0:
            continue      # ordinary compiler-scheduled kernel: nothing hidden from hipcc
        if r.scratch != 0:
            r.violations.append(f"R2 {path.name}: {r.name} uses {r.scratch} B of scratch")
        out.append(r)
    return out, seen


def audit_files(paths) -> tuple[list[KernelReport], list[str]]:
    reps, bad = [], []
    for p in paths:
        for r in audit_asm(Path(p)):
            reps.append(r)
            bad.extend(r.violations)
    return reps, bad


if __name__ == "__main__":
    import sys
    reps, bad = audit_files(sys.argv[1:])
    for r in reps:
        print(f"{r.name}: vgpr {r.vgpr_count} agpr {r.agpr_count} scratch {r.scratch} mfma {r.mfma} "
              f"compiler v_accvgpr {r.compiler_accvgpr} violations {len(r.violations)}")
    for b in bad:
        print("VIOLATION", b)
    sys.exit(1 if bad else 0)
