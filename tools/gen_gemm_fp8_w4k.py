#!/usr/bin/env python3
"""Emit leetcuda_amd/csrc/gemm_fp8_w4k_loop{,_mx}.inc: the whole K loop of gemm_fp8_w4k_kernel (gemm_fp8_w4k.hip) as ONE asm statement.

The fp8 (e4m3) GEMM on v_mfma_scale_f32_16x16x128_f8f6f4: one instruction contracts a whole 128-deep K tile, so a wave's 128 x 128 C tile
is 8 x 8 = 64 MFMAs (32 cycles each) per K tile on 8 + 8 operand fragments of EIGHT registers — 128 VGPRs for one K tile, which leaves no
room for hgemm_w4y's "both k-steps resident" double buffer.  Instead the 64 MFMAs walk the four quadrants of the wave tile in Gray-code
order, so that every quadrant boundary frees exactly one operand half (4 fragments = 32 registers) for the data needed two quadrants on:

  slots   S0 = v[128:159] A rows 0..63 ("A lo")   S1 = v[160:191] B lo   S2 = v[192:223] B hi   S3 = v[224:255] A hi
  even K tile  Q1 (A lo, B lo)  Q2 (A lo, B hi)  | barrier |  Q3 (A hi, B hi)  Q4 (A hi, B lo)
  odd  K tile  Q1 (A lo, B hi)  Q2 (A lo, B lo)  | barrier |  Q3 (A hi, B lo)  Q4 (A hi, B hi)
  reads        Q1: this tile's other B half, then its A hi          (16 ds_read_b128)
               Q3: NEXT tile's A lo -> S0 (dead since Q2)           (8)
               Q4: NEXT tile's first B half -> the B slot Q3 used   (8)   — the order of the quadrants alternates, the slot <-> rows map is fixed
  DMA          8 B pieces of tile t + 2 through the first half, 8 A pieces of tile t + 2 behind the barrier (into this tile's A slot)

The LDS images, the source-side swizzle, the ring (A 2 + B 3 slots of 32 KiB), the piece map, the K-loop stagger and the SALU bookkeeping
are hgemm_w4y's (tools/gen_hgemm_w4y.py) byte for byte: a 128-byte LDS row holds 128 k values instead of 64, and an fp8 fragment's two
register halves are the 16-byte chunks (lane >> 4) and 4 + (lane >> 4) of its row — the very reads of hgemm_w4y's k-steps 0 and 1
(the hardware's k order inside the instruction, tools/cpp/mx_probe.cpp).  s_waitcnt lgkmcnt(N) is computed per MFMA from the reads it consumes.

MX (…_mx.inc): real E8M0 block scales (one per row and 32 k).  Lane l of an instruction supplies the scale of row l % 16, k block l / 16
(mx_probe.cpp) in a byte of a scale register chosen by op_sel; the kernel takes the scales PACKED so that one dword per lane carries that
byte for four fragments (gemm_fp8_w4k.hip: mx_pack_scales), i.e. four dwords per lane and K tile: (A lo, A hi) and (B lo, B hi), loaded with
one buffer_load_dwordx2 each, one K tile ahead at the top of the tile (before the B pieces, so the barrier's vmcnt(8) covers them), two register sets
alternating with the tile parity.

usage: tools/gen_gemm_fp8_w4k.py [--check]"""
import re
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from gen_hgemm_w4y import PIECE_STEP, STAGGER, place  # noqa: E402

ROOT = Path(__file__).resolve().parent.parent
SLOT = {"alo": 128, "blo": 160, "bhi": 192, "ahi": 224}
VA0, VA1, VB0, VB1 = "v120", "v121", "v122", "v123"   # fragment read addresses: slot base + lane part of register half 0 / 1
SC0 = 112                                             # MX: scale registers v[112:119] = 2 parity sets x (A lo, A hi, B lo, B hi)
VCLOB = list(range(112, 256))


def frag(kind, f):
    b = SLOT[kind] + 8 * f
    return b


def fa(i):
    return frag("alo" if i < 4 else "ahi", i & 3)


def fb(j):
    return frag("blo" if j < 4 else "bhi", j & 3)


def acc(i, j):
    b = 4 * (8 * i + j)
    return f"a[{b}:{b + 3}]"


def mfma(i, j, mx, par):
    # SrcA = B fragment, SrcB = A fragment (hgemm_w4y's accumulator layout: lane holds C[m = 16 i + (l & 15)][n = 16 j + 4 (l >> 4) + r])
    a, b = fa(i), fb(j)
    if not mx:
        return (f"v_mfma_scale_f32_16x16x128_f8f6f4 {acc(i, j)}, v[{b}:{b + 7}], v[{a}:{a + 7}], {acc(i, j)}, %[one], %[one] "
                "op_sel:[0,0,0] op_sel_hi:[0,0,0]")
    sa = SC0 + 4 * par + (0 if i < 4 else 1)
    sb = SC0 + 4 * par + (2 if j < 4 else 3)
    ba, bb = i & 3, j & 3      # byte of the scale register = fragment index inside its half
    return (f"v_mfma_scale_f32_16x16x128_f8f6f4 {acc(i, j)}, v[{b}:{b + 7}], v[{a}:{a + 7}], {acc(i, j)}, v{sb}, v{sa} "
            f"op_sel:[{bb & 1},{ba & 1},0] op_sel_hi:[{bb >> 1},{ba >> 1},0]")


def reads(kind, which):
    """The 8 ds_read_b128 of operand half `kind` (alo / ahi / blo / bhi); which = 'a' / 'b' picks the address registers."""
    hi = kind.endswith("hi")
    v0, v1 = (VA0, VA1) if which == "a" else (VB0, VB1)
    out = []
    for f in range(4):
        b = frag(kind, f)
        off = (4 * hi + f) * 2048
        out.append((f"ds_read_b128 v[{b}:{b + 3}], {v0} offset:{off}", b))
        out.append((f"ds_read_b128 v[{b + 4}:{b + 7}], {v1} offset:{off}", b))
    return out


def quadrants(par):
    """MFMA order of a K tile of parity par: list of (i, j).  The operand half that was read last is the outer loop."""
    lo, hi = range(0, 4), range(4, 8)
    if par == 0:
        q1 = [(i, j) for j in lo for i in lo]
        q2 = [(i, j) for j in hi for i in lo]
        q3 = [(i, j) for i in hi for j in hi]
        q4 = [(i, j) for j in lo for i in hi]
    else:
        q1 = [(i, j) for j in hi for i in lo]
        q2 = [(i, j) for j in lo for i in lo]
        q3 = [(i, j) for i in hi for j in lo]
        q4 = [(i, j) for j in hi for i in hi]
    return q1 + q2 + q3 + q4


class Stream:
    """Instruction list with an LDS-read ledger: wait_for(regs) emits the s_waitcnt lgkmcnt(N) that makes the named fragments valid."""

    def __init__(self):
        self.L = []
        self.issued = 0          # ds_reads issued so far
        self.done = 0            # reads known complete (by an emitted wait)
        self.last = {}           # fragment base register -> index (1-based) of the last read that writes it

    def e(self, ins, frag_base=None):
        self.L.append(ins)
        if ins.startswith("ds_read"):
            self.issued += 1
            self.last[frag_base] = self.issued
        if "lgkmcnt(0)" in ins:
            self.done = self.issued

    def wait_for(self, *bases):
        need = max((self.last.get(b, 0) for b in bases), default=0)
        if need > self.done:
            n = self.issued - need
            assert n <= 15
            self.L.append(f"s_waitcnt lgkmcnt({n})")
            self.done = need


def scale_loads(par, mx):
    """MX: the scale dwords of the NEXT tile into parity par ^ 1's registers: (A lo, A hi) and (B lo, B hi), one dwordx2 each."""
    if not mx:
        return []
    base = SC0 + 4 * (par ^ 1)
    return [f"buffer_load_dwordx2 v[{base}:{base + 1}], %[slo], %[rsa], %[s1off] offen",
            f"buffer_load_dwordx2 v[{base + 2}:{base + 3}], %[slo], %[rsb], %[s1off] offen"]


def gen_init(mx):
    s = Stream()
    e = s.e
    e("s_mov_b32 %[t], 0")
    e("s_mov_b32 %[acur], %[a0]")
    e("s_add_u32 %[anxt], %[a0], 0x8000")
    e("s_add_u32 %[b0], %[a0], 0x10000")
    e("s_add_u32 %[b1], %[a0], 0x18000")
    e("s_add_u32 %[b2], %[a0], 0x20000")
    e(f"v_add_u32_e32 {VA0}, %[acur], %[ar0]")
    e(f"v_add_u32_e32 {VA1}, %[acur], %[ar1]")
    e(f"v_add_u32_e32 {VB0}, %[b0], %[br0]")
    e(f"v_add_u32_e32 {VB1}, %[b0], %[br1]")
    for ins, b in reads("alo", "a") + reads("blo", "b"):
        e(ins, b)
    if mx:
        # scales of tile 0 (memory tile stg) and tile 1 (memory tile (min(1, KT - 1) + stg) mod KT): 512 bytes per (128-row block, K tile)
        e("s_lshl_b32 %[s1off], %[stg], 9")
        for k, ins in enumerate(scale_loads(1, True)):   # parity-0 registers
            e(ins)
        e("s_sub_u32 %[swp], %[kt], 1")
        e("s_min_u32 %[swp], %[swp], 1")
        for ins in STAGGER:
            e(ins)
        e("s_lshl_b32 %[s1off], %[swp], 9")
    e("s_sub_u32 %[swp], %[kt], 1")
    e("s_min_u32 %[swp], %[swp], 2")
    for ins in STAGGER:
        e(ins)
    e("s_lshl_b32 %[t2off], %[swp], 7")
    if mx:
        e("s_waitcnt vmcnt(0)")
    return s


def gen_tile(s, par, mx):
    """One K tile of parity par.  Ends with the loop counter compare (SCC = more tiles)."""
    order = quadrants(par)
    fill = {}

    def after(m, *ins):
        fill.setdefault(m, []).extend(ins)

    # ---- first half: reads of this tile's second B half (gaps 1..8) and A hi (9..16); the next tile's scales; B pieces of tile t + 2
    bsecond = "bhi" if par == 0 else "blo"
    for k, rd in enumerate(reads(bsecond, "b")):
        after(1 + k, rd)
    for k, rd in enumerate(reads("ahi", "a")):
        after(9 + k, rd)
    for k, ins in enumerate(scale_loads(par, mx)):
        after(k, ins)
    after(3, "s_add_u32 %[tmp], %[b2], %[wv]")
    for p in range(8):
        after(4 + 3 * p, f"s_add_u32 m0, %[tmp], {p * PIECE_STEP}",
              "s_mov_b32 %[soff], %[t2off]" if p == 0 else "s_add_u32 %[soff], %[soff], %[blk]")
        after(5 + 3 * p, f"buffer_load_dwordx4 %[ao], %[rb], %[soff] offen lds")
    # ---- barrier behind MFMA 32; addresses of the next tile's slots; A pieces of tile t + 2 into this tile's A slot
    bar = 32
    after(bar, "s_barrier", f"v_add_u32_e32 {VA0}, %[anxt], %[ar0]", "s_add_u32 %[tmp], %[acur], %[wv]")
    after(bar + 1, f"v_add_u32_e32 {VA1}, %[anxt], %[ar1]", f"v_add_u32_e32 {VB0}, %[b1], %[br0]", f"v_add_u32_e32 {VB1}, %[b1], %[br1]")
    for g in range(8):
        after(bar + 2 + 3 * g, f"s_add_u32 m0, %[tmp], {g * PIECE_STEP}",
              "s_mov_b32 %[soff], %[t2off]" if g == 0 else "s_add_u32 %[soff], %[soff], %[blk]")
        after(bar + 3 + 3 * g, f"buffer_load_dwordx4 %[ao], %[ra], %[soff] offen lds")
    # ---- second half: the next tile's A lo (S0 is dead behind MFMA 31) and its first B half (the slot Q3 used, dead behind MFMA 47)
    for k, rd in enumerate(reads("alo", "a")):
        after(bar + 2 + k, rd)
    bnext = "bhi" if par == 0 else "blo"
    for k, rd in enumerate(reads(bnext, "b")):
        after(49 + k, rd)
    # ring rotation and the next tile's source offsets, in the gaps behind the last A piece
    rot = [["s_mov_b32 %[swp], %[b0]"], ["s_mov_b32 %[b0], %[b1]"], ["s_mov_b32 %[b1], %[b2]"], ["s_mov_b32 %[b2], %[swp]"],
           ["s_mov_b32 %[swp], %[acur]"], ["s_mov_b32 %[acur], %[anxt]"], ["s_mov_b32 %[anxt], %[swp]"]]
    if mx:
        rot.append(["s_lshl_b32 %[s1off], %[t2off], 2"])   # this tile's t + 2 is the next tile's t + 1; 512 scale bytes per 128 data bytes
    rot += [["s_add_u32 %[swp], %[t], 3"], ["s_sub_u32 %[t2off], %[kt], 1"], ["s_min_u32 %[swp], %[swp], %[t2off]"],
            [STAGGER[0]], STAGGER[1:3], [STAGGER[3]], ["s_lshl_b32 %[t2off], %[swp], 7"]]
    slots = list(range(bar + 4, 62))   # (behind the first A piece: acur / anxt / b1 / t2off have been consumed by then)
    place(rot, slots, after, f"fp8 parity {par}")
    after(62, "s_add_u32 %[t], %[t], 1", "s_cmp_lt_u32 %[t], %[kt]")

    for m, (i, j) in enumerate(order):
        if m == bar:
            s.e("s_waitcnt vmcnt(8) lgkmcnt(0)")
        else:
            s.wait_for(fa(i), fb(j))
        s.e(mfma(i, j, mx, par))
        for ins in fill.get(m, []):
            if isinstance(ins, tuple):
                s.e(ins[0], ins[1])
            else:
                s.e(ins)


def gen(mx):
    s = gen_init(mx)
    lab, end = ".Lw4k_loop_%=", ".Lw4k_end_%="
    s.e(lab + ":")
    gen_tile(s, 0, mx)
    s.e(f"s_cbranch_scc0 {end}")
    gen_tile(s, 1, mx)
    s.e(f"s_cbranch_scc1 {lab}")
    s.e(end + ":")
    s.e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    return s.L


def check_literal_vgprs(lines, clob, what):
    """Round-4 advisor: the audit's rule R4 (no compiler instruction may name an asm-owned literal VGPR) cannot cover registers that are
    owned only INSIDE one statement, so safety rests on the clobber list naming every literal v-register of the body — enforced here, at
    generation time and on every build (`--check`): hipcc may then keep nothing of its own in them across the statement."""
    used = set()
    for ln in lines:
        for m in re.finditer(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]", ln):
            if m.group(1) is not None:
                used.add(int(m.group(1)))
            else:
                used.update(range(int(m.group(2)), int(m.group(3)) + 1))
    missing = sorted(used - set(clob))
    if missing:
        raise SystemExit(f"{what}: literal VGPRs {missing[:8]} ... are not in the statement's clobber list")


def render(mx):
    lines = gen(mx)
    check_literal_vgprs(lines, VCLOB, f"gemm_fp8_w4k loop (mx = {mx})")
    body = "\n".join(f'    "{ln}\\n\\t"' for ln in lines)
    vclob = ", ".join(f'"v{r}"' for r in VCLOB)
    n_mfma = sum(ln.startswith("v_mfma") for ln in lines)
    head = (f"// GENERATED by tools/gen_gemm_fp8_w4k.py ({'MX block scales' if mx else 'unit scales'}) — do not edit ({len(lines)} instructions, "
            f"{n_mfma} MFMAs per two K tiles).\n"
            "// Operands (gemm_fp8_w4k.hip): kt, stg, a0, wv, blk, ra / rb, ao (DMA lane offset), ar0 / ar1 / br0 / br1 (fragment read lane offsets of\n"
            "// register half 0 / 1)" + ("; MX: rsa / rsb (scale descriptors), slo (scale lane offset).\n" if mx else "; one (0x7f7f7f7f: unit E8M0 scales).\n"))
    outs = ("    : [t] \"=&s\"(w4k_t), [acur] \"=&s\"(w4k_acur), [anxt] \"=&s\"(w4k_anxt), [b0] \"=&s\"(w4k_b0), [b1] \"=&s\"(w4k_b1),\n"
            "      [b2] \"=&s\"(w4k_b2), [soff] \"=&s\"(w4k_soff), [t2off] \"=&s\"(w4k_t2off), [tmp] \"=&s\"(w4k_tmp), [swp] \"=&s\"(w4k_swp)"
            + (", [s1off] \"=&s\"(w4k_s1off)\n" if mx else "\n"))
    ins = ("    : [kt] \"s\"(KT), [stg] \"s\"(w4k_stg), [a0] \"s\"(w4k_a0), [wv] \"s\"(w4k_wv), [blk] \"s\"(w4k_blk), [ra] \"s\"(w4k_ra),\n"
           "      [rb] \"s\"(w4k_rb), [ao] \"v\"(w4k_ao), [ar0] \"v\"(fr.a_ad[0]), [ar1] \"v\"(fr.a_ad[1]), [br0] \"v\"(fr.b_ad[0]),\n"
           "      [br1] \"v\"(fr.b_ad[1])"
           + (", [rsa] \"s\"(w4k_rsa), [rsb] \"s\"(w4k_rsb), [slo] \"v\"(w4k_slo)\n" if mx else ", [one] \"v\"(w4k_one)\n"))
    return head + "asm volatile(\n" + body + "\n" + outs + ins + f"    : \"memory\", \"scc\", {vclob}, LC_AGPR_ALL);\n"


def outputs():
    d = ROOT / "leetcuda_amd" / "csrc"
    return [(render(False), d / "gemm_fp8_w4k_loop.inc"), (render(True), d / "gemm_fp8_w4k_loop_mx.inc")]


def main():
    rc = 0
    for text, out in outputs():
        if "--check" in sys.argv:
            if not out.exists() or out.read_text() != text:
                print(f"{out} is stale: run tools/gen_gemm_fp8_w4k.py", file=sys.stderr)
                rc = 1
        else:
            out.write_text(text)
            print(f"wrote {out} ({len(text.splitlines())} lines)")
    return rc


if __name__ == "__main__":
    sys.exit(main())
