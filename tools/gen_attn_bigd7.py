#!/usr/bin/env python3
"""Emit leetcuda_amd/csrc/attn_bigd7_stmts.inc: the asm statements of attn_fwd_bigd7_kernel (attn_bigd7.hip) that touch the score registers.

Why a generator: the softmax of tile t runs as filler behind the P·V MFMAs of tile t − 1.  Written in C++ between two 8-MFMA statements
the fillers issue AFTER the statement's MFMAs (which go out back to back, 12 of every 16 issue cycles unused) and the matrix pipe idles
while they run: 0.53 MFMA-busy (profiles/r4n_pmc_bigd7.txt).  Here the three VALU instructions per score (v_fma_f32 s sl2 − m, v_exp_f32,
v_add_f32 into the row sum) sit IN the gaps between the MFMAs of the statement, software-pipelined one gap apart (a transcendental's
result may not be read by the very next VALU instruction).  That needs the scores under literal register names:

  Sᵀ block (kvb, qb) = v[208 + 4 (4 kvb + qb) .. + 3]  — score element e = 16 kvb + 4 qb + r lives in v[208 + e]
  Vᵀ quads           = v[240:255]                      — as attn_bigd6.hip

Emitted (C++ function templates, BF16 selects the MFMA opcode):
  bd7_qk8f<BF16, FIRST>(s, k0, k1, q0..q3)      the eight MFMAs of a d-step on the pinned score registers
  bd7_rd<VT, OFF, HOFF>(...)                    the reads of a tile's first P·V step
  bd7_pvn<X, BF16, VT, R0, OFF, HOFF>(...)      statement X = 0 .. 7 of a tile's eight P·V statements (X >> 1 = step, X & 1 = quad pair): two Vᵀ
                                                 fragments x four query blocks, the reads of the NEXT step behind each fragment's four
                                                 MFMAs (X < 6), counted lgkmcnt waits (LDS reads return in order; a step starts with its 8 —
                                                 VT: 4 — reads outstanding in fragment order)
  bd7_pvf<X, BF16, VT, R0, OFF, HOFF>(...)      the same for X = 2 .. 7 with the fillers of score elements e, e * 6 / 32 == X − 2 (two score
                                                 blocks, two query blocks: ps / m operands A and B)
VT: V handed over as [B,H,D,N] (the reference's *_swizzle_qkv entries): the LDS tile is [256 d][32 kv], a fragment ONE ds_read_b128.
Register indices inside the statements are assembler expressions on ONE "n" operand (a[%R + 4 : %R + 7]): the 30-operand limit of an asm
statement would not hold sixteen of them.

usage: tools/gen_attn_bigd7.py [--check]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "leetcuda_amd" / "csrc" / "attn_bigd7_stmts.inc"
S0 = 208          # first score register
QUADS = {0: (240, 244), 1: (248, 252)}


def elements(x):
    return [e for e in range(32) if e * 6 // 32 == x - 2]


def asm_lines(lines):
    return "\n".join(f'               "{ln}\\n\\t"' for ln in lines)


def gen_qk():
    out = []
    out.append("// the eight MFMAs of one d-step on the pinned score registers: Sᵀ block (kvb, qb) += K fragment (kvb) x Q fragment (qb).  FIRST: the first")
    out.append("// d-step of a tile — the accumulator input is the inline constant 0 (no VALU zeroing of the 32 score registers), outputs write-only")
    out.append("template <bool BF16, bool FIRST>")
    out.append("LC_DEVINL void bd7_qk8f(f32x4_t (&s)[2][4], half8_t k0, half8_t k1, half8_t q0, half8_t q1, half8_t q2, half8_t q3) {")
    for bf in (True, False):
        for first in (True, False):
            op = "v_mfma_f32_16x16x32_bf16" if bf else "v_mfma_f32_16x16x32_f16"
            lines = []
            for kvb in range(2):
                for qb in range(4):
                    r = S0 + 4 * (4 * kvb + qb)
                    lines.append(f"{op} v[{r}:{r + 3}], %{8 + kvb}, %{10 + qb}, " + ("0" if first else f"v[{r}:{r + 3}]"))
            mod = "=" if first else "+"
            cons = ", ".join(f'"{mod}{{v[{S0 + 4 * t}:{S0 + 4 * t + 3}]}}"(s[{t >> 2}][{t & 3}])' for t in range(8))
            out.append(f"  if constexpr ({'BF16' if bf else '!BF16'} && {'FIRST' if first else '!FIRST'}) {{")
            out.append("    asm volatile(\n" + asm_lines(lines) + "\n"
                       f"               : {cons}\n"
                       '               : "v"(k0), "v"(k1), "v"(q0), "v"(q1), "v"(q2), "v"(q3)\n'
                       "               : LC_AGPR_ALL);")
            out.append("  }")
    out.append("}")
    return out


def gen_pv(x, fill, vt):
    """Statement x (0 .. 7) of a tile's eight P·V statements: the two Vᵀ fragments of quad pair x & 1 x four query blocks.  fill: with the
    softmax fillers of its score elements (x >= 2).  vt: V handed over as [B,H,D,N] — a fragment is ONE ds_read_b128 of the Vᵀ tile's row
    (counted waits on 4 outstanding reads instead of 8)."""
    st, hq = x >> 1, x & 1
    rd = st + 1 < 4
    per = 1 if vt else 2                     # LDS reads per fragment
    if rd:
        w0 = w1 = 3 * per
    else:
        w0, w1 = ((per, 0) if hq else (3 * per, 2 * per))
    qx, qy = QUADS[hq]
    gaps = [[] for _ in range(8)]
    ta = tb = None
    if fill:
        es = elements(x)
        ta, tb = sorted({e >> 2 for e in es})
        # operands: %0 fx %1 fy %2 sa %3 sb %4 psa %5 psb | %6..%9 p0..p3 %10 ax %11 ay %12 sl2 %13 ma %14 mb | %15 R0 %16 OFF %17 OFF+HOFF
        def filler(kind, e):
            reg = f"v[{S0 + e}]"
            first = (e >> 2) == ta
            ps, m = ("%4", "%13") if first else ("%5", "%14")
            if kind == "F":
                return f"v_fma_f32 {reg}, {reg}, %12, -{m}"
            if kind == "X":
                return f"v_exp_f32 {reg}, {reg}"
            return f"v_add_f32 {ps}, {ps}, {reg}"
        for i, e in enumerate(es):
            gaps[i].append(filler("F", e))
            gaps[i + 1].append(filler("X", e))
            gaps[i + 2].append(filler("A", e))
        assert len(es) + 2 <= 8
        P0, AX, R, OF = 6, 10, 15, 16
    else:
        # operands: %0 fx %1 fy | %2..%5 p0..p3 %6 ax %7 ay | %8 R0 %9 OFF %10 OFF+HOFF
        P0, AX, R, OF = 2, 6, 8, 9
    out = []
    for bf in (True, False):
        op = "v_mfma_f32_16x16x32_bf16" if bf else "v_mfma_f32_16x16x32_f16"
        lines = ["s_nop 1", f"s_waitcnt lgkmcnt({w0})"]
        for j, (quad, base) in enumerate(((qx, 0), (qy, 16))):
            if j == 1:
                lines.append(f"s_waitcnt lgkmcnt({w1})")
            for q in range(4):
                lines.append(f"{op} a[%{R}+{base + 4 * q}:%{R}+{base + 4 * q + 3}], v[{quad}:{quad + 3}], %{P0 + q}, "
                             f"a[%{R}+{base + 4 * q}:%{R}+{base + 4 * q + 3}]")
                g = 4 * j + q
                if q == 3 and rd:
                    if vt:      # fragment X: row block at OFF, fragment Y: the next row block (OFF + HOFF), same lane address
                        lines.append(f"ds_read_b128 v[{quad}:{quad + 3}], %{AX + j} offset:%{OF + j}")
                    else:       # kv rows 4 g4 .. of kv block 0 (OFF), then kv block 1 (OFF + HOFF); fragment X / Y: lane address ax / ay
                        lines.append(f"ds_read_b64_tr_b16 v[{quad}:{quad + 1}], %{AX + j} offset:%{OF}")
                        lines.append(f"ds_read_b64_tr_b16 v[{quad + 2}:{quad + 3}], %{AX + j} offset:%{OF + 1}")
                lines.extend(gaps[g])
        cons_out = f'"+{{v[{qx}:{qx + 3}]}}"(fx), "+{{v[{qy}:{qy + 3}]}}"(fy)'
        cons_in = '"v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(ax), "v"(ay)'
        if fill:
            cons_out += (f', "+{{v[{S0 + 4 * ta}:{S0 + 4 * ta + 3}]}}"(sa), "+{{v[{S0 + 4 * tb}:{S0 + 4 * tb + 3}]}}"(sb), "+v"(psa), "+v"(psb)')
            cons_in += ', "v"(sl2), "v"(ma), "v"(mb)'
        cons_in += ', "n"(R0), "n"(OFF), "n"(OFF + HOFF)'
        out.append(f"    if constexpr ({'BF16' if bf else '!BF16'}) {{")
        out.append("      asm volatile(\n" + asm_lines(lines).replace("               ", "                   ") + "\n"
                   f"                   : {cons_out}\n"
                   f"                   : {cons_in}\n"
                   "                   : LC_AGPR_ALL);")
        out.append("    }")
    return out, (ta, tb)


def gen_pv_check(vt):
    """Statement 1 (step 0, quad pair 1) with the running-maximum check of tile t in its MFMA gaps: per query block the largest of the lane's
    eight raw scores (three v_max3_f32 + one v_max_f32 on the pinned score registers — final since the whole of statement 0 lies between
    them and the Sᵀ MFMAs), its excess over the running maximum (v_fma_f32 mx sl2 − m), and the largest excess of the four into `over`."""
    qx, qy = QUADS[1]
    per = 1 if vt else 2
    w0 = w1 = 3 * per
    # operands: %0 fx %1 fy %2 over %3 tmp | %4..%7 p0..p3 %8 ax %9 ay %10 sl2 %11..%14 m0..m3 %15..%22 score blocks | %23 R0 %24 OFF %25 OFF+HOFF
    ops = []
    for qb in range(4):
        r = [S0 + 4 * qb + i for i in range(4)] + [S0 + 16 + 4 * qb + i for i in range(4)]      # blocks (0, qb) and (1, qb)
        ops.append(f"v_max3_f32 %3, v[{r[0]}], v[{r[1]}], v[{r[2]}]")
        ops.append(f"v_max3_f32 %3, %3, v[{r[3]}], v[{r[4]}]")
        ops.append(f"v_max3_f32 %3, %3, v[{r[5]}], v[{r[6]}]")
        ops.append(f"v_max_f32 %3, %3, v[{r[7]}]")
        ops.append(f"v_fma_f32 %3, %3, %10, -%{11 + qb}")
        ops.append("v_mov_b32 %2, %3" if qb == 0 else "v_max_f32 %2, %2, %3")
    gaps = [ops[3 * g:3 * g + 3] for g in range(8)]
    out = []
    for bf in (True, False):
        op = "v_mfma_f32_16x16x32_bf16" if bf else "v_mfma_f32_16x16x32_f16"
        lines = ["s_nop 1", f"s_waitcnt lgkmcnt({w0})"]
        for j, (quad, base) in enumerate(((qx, 0), (qy, 16))):
            if j == 1:
                lines.append(f"s_waitcnt lgkmcnt({w1})")
            for q in range(4):
                lines.append(f"{op} a[%23+{base + 4 * q}:%23+{base + 4 * q + 3}], v[{quad}:{quad + 3}], %{4 + q}, a[%23+{base + 4 * q}:%23+{base + 4 * q + 3}]")
                if q == 3:
                    if vt:
                        lines.append(f"ds_read_b128 v[{quad}:{quad + 3}], %{8 + j} offset:%{24 + j}")
                    else:
                        lines.append(f"ds_read_b64_tr_b16 v[{quad}:{quad + 1}], %{8 + j} offset:%24")
                        lines.append(f"ds_read_b64_tr_b16 v[{quad + 2}:{quad + 3}], %{8 + j} offset:%25")
                lines.extend(gaps[4 * j + q])
        cons_out = f'"+{{v[{qx}:{qx + 3}]}}"(fx), "+{{v[{qy}:{qy + 3}]}}"(fy), "=&v"(over), "=&v"(tmp)'
        cons_in = ('"v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(ax), "v"(ay), "v"(sl2), "v"(m0), "v"(m1), "v"(m2), "v"(m3), '
                   + ", ".join(f'"{{v[{S0 + 4 * t}:{S0 + 4 * t + 3}]}}"(s[{t >> 2}][{t & 3}])' for t in range(8))
                   + ', "n"(R0), "n"(OFF), "n"(OFF + HOFF)')
        out.append(f"    if constexpr ({'BF16' if bf else '!BF16'}) {{")
        out.append("      asm volatile(\n" + asm_lines(lines).replace("               ", "                   ") + "\n"
                   f"                   : {cons_out}\n"
                   f"                   : {cons_in}\n"
                   "                   : LC_AGPR_ALL);")
        out.append("    }")
    return out


def gen_rd(vt):
    """The reads of a tile's first P·V step (fragments db = 0 .. 3 into the four quads), in fragment order."""
    lines = []
    for j in range(4):
        q = 240 + 4 * j
        if vt:
            lines.append(f"ds_read_b128 v[{q}:{q + 3}], %4 offset:%{5 + j}")
        else:
            lines.append(f"ds_read_b64_tr_b16 v[{q}:{q + 1}], %{4 + j} offset:%8")
            lines.append(f"ds_read_b64_tr_b16 v[{q + 2}:{q + 3}], %{4 + j} offset:%9")
    cons = '"={v[240:243]}"(f0), "={v[244:247]}"(f1), "={v[248:251]}"(f2), "={v[252:255]}"(f3)'
    ins = ('"v"(a0), "n"(OFF), "n"(OFF + HOFF), "n"(OFF + 2 * HOFF), "n"(OFF + 3 * HOFF)' if vt
           else '"v"(a0), "v"(a1), "v"(a2), "v"(a3), "n"(OFF), "n"(OFF + HOFF)')
    return ["    asm volatile(\n" + asm_lines(lines).replace("               ", "                 ") + "\n"
            f"                 : {cons}\n                 : {ins});"]


def render():
    L = ["// GENERATED by tools/gen_attn_bigd7.py — do not edit.  Statements of attn_fwd_bigd7_kernel on the pinned score registers v[208:239]",
         "// (score element e = 16 kvb + 4 qb + r in v[208 + e]) and the Vᵀ quads v[240:255]; see the generator for the schedule.", ""]
    L += gen_qk()
    L.append("")
    L.append("// the reads of a tile's first P·V step.  VT = false: two transpose reads per fragment (lane addresses a0 .. a3, kv blocks at OFF / OFF + HOFF);")
    L.append("// VT = true (V as [B,H,D,N]): one ds_read_b128 per fragment, row blocks at OFF + j HOFF of the one lane address a0")
    L.append("template <bool VT, int OFF, int HOFF>")
    L.append("LC_DEVINL void bd7_rd(half8_t& f0, half8_t& f1, half8_t& f2, half8_t& f3, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3) {")
    L.append("  if constexpr (VT) {")
    L += gen_rd(True)
    L.append("  } else {")
    L += gen_rd(False)
    L.append("  }")
    L.append("}")
    L.append("")
    L.append("// statement X of a tile's eight P·V statements WITHOUT fillers (X = 0, 1: the scores are not ready; the tail; the exact path)")
    L.append("template <int X, bool BF16, bool VT, int R0, int OFF, int HOFF>")
    L.append("LC_DEVINL void bd7_pvn(half8_t& fx, half8_t& fy, half8_t p0, half8_t p1, half8_t p2, half8_t p3, uint32_t ax, uint32_t ay) {")
    L.append("  static_assert(X >= 0 && X <= 7);")
    for x in range(8):
        for vt in (False, True):
            body, _ = gen_pv(x, False, vt)
            L.append(f"  if constexpr (X == {x} && {'VT' if vt else '!VT'}) {{")
            L += body
            L.append("  }")
    L.append("}")
    L.append("")
    L.append("// statement 1 with the running-maximum check in its gaps: over = max over the four query blocks of (largest raw score of the lane) sl2 − m")
    L.append("template <bool BF16, bool VT, int R0, int OFF, int HOFF>")
    L.append("LC_DEVINL void bd7_pvc(half8_t& fx, half8_t& fy, half8_t p0, half8_t p1, half8_t p2, half8_t p3, uint32_t ax, uint32_t ay,")
    L.append("                       const f32x4_t (&s)[2][4], float& over, float sl2, float m0, float m1, float m2, float m3) {")
    L.append("  float tmp;")
    for vt in (False, True):
        L.append(f"  if constexpr ({'VT' if vt else '!VT'}) {{")
        L += gen_pv_check(vt)
        L.append("  }")
    L.append("}")
    L.append("")
    L.append("// statement X with the softmax fillers of its score elements in the MFMA gaps.  sa / sb: the two score blocks it touches (bd7_pvf_ta /")
    L.append("// _tb), psa / psb and ma / mb: row sum and running maximum of their query blocks.")
    L.append("template <int X, bool BF16, bool VT, int R0, int OFF, int HOFF>")
    L.append("LC_DEVINL void bd7_pvf(half8_t& fx, half8_t& fy, half8_t p0, half8_t p1, half8_t p2, half8_t p3, uint32_t ax, uint32_t ay,")
    L.append("                       f32x4_t& sa, f32x4_t& sb, float& psa, float& psb, float sl2, float ma, float mb) {")
    L.append("  static_assert(X >= 2 && X <= 7);")
    table = {}
    for x in range(2, 8):
        for vt in (False, True):
            body, info = gen_pv(x, True, vt)
            table[x] = info
            L.append(f"  if constexpr (X == {x} && {'VT' if vt else '!VT'}) {{")
            L += body
            L.append("  }")
    L.append("}")
    L.append("// score blocks (t = 4 kvb + qb) statement X touches")
    L.append("constexpr int bd7_pvf_ta(int x) { return " + " : ".join(f"x == {x} ? {table[x][0]}" for x in range(2, 7)) + f" : {table[7][0]}; }}")
    L.append("constexpr int bd7_pvf_tb(int x) { return " + " : ".join(f"x == {x} ? {table[x][1]}" for x in range(2, 7)) + f" : {table[7][1]}; }}")
    return "\n".join(L) + "\n"


def main():
    text = render()
    if "--check" in sys.argv:
        if not OUT.exists() or OUT.read_text() != text:
            print(f"{OUT} is stale: run tools/gen_attn_bigd7.py", file=sys.stderr)
            return 1
        return 0
    OUT.write_text(text)
    print(f"wrote {OUT} ({len(text.splitlines())} lines)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
