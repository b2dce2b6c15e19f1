#!/usr/bin/env python3
"""Emit leetcuda_amd/csrc/hgemm_w4y_loop.inc: the whole K loop of hgemm_w4y_kernel (hgemm_w4y.hip) as ONE asm statement.

Why a generator: with one wave per SIMD and 16-cycle MFMAs (v_mfma_f32_16x16x32_f16) every instruction hipcc adds between
two MFMAs (an s_nop at each asm-statement boundary, its own s_waitcnt placement, SALU address arithmetic per DMA piece)
opens a bubble in the matrix pipe — hgemm_w4x_kernel (the same algorithm as C++ + small asm statements) measures 77 %
MFMA-busy where a hand-ordered stream of the same work reaches > 90 %.  This script writes the stream by hand-rule:

  per K tile (64) and wave: 128 MFMAs = 2 k-steps x (8 A fragments x 8 B fragments), accumulators a[0:255];
  fragments fully double-buffered per k-step in LITERAL VGPRs: A(ks, i) = v[128 + 64 ks + 4 i ..], B(ks, j) = v[160 + 64 ks + 4 j ..];
  k-step 0 carries: 16 ds_read_b128 of (tile t, k-step 1)           (one per 2 MFMAs, first half)
                    8 LDS-DMA pieces B(t+2) -> B ring slot t+2       (one per 4 MFMAs, second half)
  k-step 1 starts with  s_waitcnt vmcnt(8) lgkmcnt(0) | MFMA | s_barrier   (tile t fully read, tile t+1 landed)
           carries: 16 ds_read_b128 of (tile t+1, k-step 0), 8 LDS-DMA pieces A(t+2) -> A ring slot t (dead now),
                    the ring rotation, the next iteration's source offset / read addresses
                    and the loop counter (the loop top is one s_waitcnt).
  A DMA piece = s_add m0 / s_add soffset behind one MFMA, buffer_load_dwordx4 ... offen lds behind the next (an M0 write
  needs one wait state before the LDS-DMA that uses it).
Ring, swizzles, piece order and the vmcnt(8) count are those of hgemm_w4b_kernel (hgemm_w4.hip).

usage: tools/gen_hgemm_w4y.py [--check]     (--check: exit 1 if the committed .inc differs from what would be generated)
       tools/gen_hgemm_w4y.py --diag DIR   (LC_DIAG builds only: write the ablation loops 3..5 — results WRONG by design, never
                                            committed — into DIR, which leetcuda_amd/build.py puts on the include path)"""
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
NSCHED = 3
NDIAG = 3   # LC_DIAG-only ablations of schedule 1 (results are WRONG): 3 = no DMA in the loop, 4 = DMA issued but never waited
            # for, 5 = no fragment reads (profiles/r2_w4y_ablation.log)


def out_path(sched, diag_dir=None):
    return (Path(diag_dir) if diag_dir else ROOT / "leetcuda_amd" / "csrc") / f"hgemm_w4y_loop{sched}.inc"

# LDS-DMA piece map of the K-contiguous operands (A; B of TN): piece g of wave w = rows 32 g + 8 w .. + 8 of the 256-row tile — at
# any moment the four waves of a workgroup ask for 32 CONSECUTIVE rows (the vendor kernel's map; with one 64-row band per wave the
# concurrent requests sat 64 rows = 1 MiB apart at K = 8192 and held their L2 reads 1.6x longer, profiles/r3n_pmc_vmem.txt).  The
# LDS image is row-major either way: piece g lands at g * 4096 + w * 1024 (operand wv = w * 1024).
PIECE_STEP = 4096
VA, VB = "v124", "v125"          # fragment read addresses (slot base + lane part)
FRAG0 = 128                      # first literal fragment VGPR
VCLOB = list(range(124, 256))    # literal VGPRs owned by the statement


def fa(ks, i):
    b = FRAG0 + 64 * ks + 4 * i
    return f"v[{b}:{b + 3}]"


def fb(ks, j):
    b = FRAG0 + 32 + 64 * ks + 4 * j
    return f"v[{b}:{b + 3}]"


def acc(i, j):
    b = 4 * (8 * i + j)
    return f"a[{b}:{b + 3}]"


def mfma(ks, i, j):
    # SrcA = B fragment, SrcB = A fragment: lane holds C[m = 16 i + (l & 15)][n = 16 j + 4 (l >> 4) + r]
    return f"v_mfma_f32_16x16x32_f16 {acc(i, j)}, {fb(ks, j)}, {fa(ks, i)}, {acc(i, j)}"


# K-loop stagger (hgemm_w4y.hip: the workgroup walks the K tiles stg, stg + 1, ..., KT - 1, 0, ..., stg - 1): logical tile index in
# swp -> memory tile index (swp + stg) mod KT, stg < KT.  The s_cmp / s_cselect pair must stay adjacent (SCC).
STAGGER = ["s_add_u32 %[swp], %[swp], %[stg]", "s_cmp_ge_u32 %[swp], %[kt]", "s_cselect_b32 %[t2off], %[kt], 0",
           "s_sub_u32 %[swp], %[swp], %[t2off]"]


def place(rot, slots, after, what):
    """rot: instruction groups (lists; a group stays in one gap), slots: free MFMA gaps in stream order.  Groups are merged from
    the front while there are more groups than gaps."""
    groups = [list(g) for g in rot]
    while len(groups) > len(slots):
        assert len(groups) >= 2, what
        groups[0:2] = [groups[0] + groups[1]]
    for g, m in zip(groups, slots):
        after(m, *g)


def gen_init():
    L = []
    e = L.append
    e("s_mov_b32 %[t], 0")
    e("s_mov_b32 %[acur], %[a0]")
    e("s_add_u32 %[anxt], %[a0], 0x8000")
    e("s_add_u32 %[b0], %[a0], 0x10000")
    e("s_add_u32 %[b1], %[a0], 0x18000")
    e("s_add_u32 %[b2], %[a0], 0x20000")
    e(f"v_add_u32_e32 {VA}, %[acur], %[ar0]")
    e(f"v_add_u32_e32 {VB}, %[b0], %[br0]")
    for i in range(8):
        e(f"ds_read_b128 {fa(0, i)}, {VA} offset:{i * 2048}")
    for j in range(8):
        e(f"ds_read_b128 {fb(0, j)}, {VB} offset:{j * 2048}")
    # source offset of tile min(t + 2, KT - 1) and the read addresses of (t, k-step 1): computed here for t = 0 and in
    # the tail of every iteration for the next one, so the loop top is the wait alone
    e("s_sub_u32 %[swp], %[kt], 1")
    e("s_min_u32 %[swp], %[swp], 2")
    for ins in STAGGER:
        e(ins)
    e("s_lshl_b32 %[t2off], %[swp], 7")
    e(f"v_add_u32_e32 {VA}, %[acur], %[ar1]")
    e(f"v_add_u32_e32 {VB}, %[b0], %[br1]")
    return L


def gen_body(sched):
    """One K-tile iteration, from the loop label to the backward branch.  (Tried and removed: one copy of the body per wave
    with the DMA pieces shifted by the wave's index, so that the four lock-stepped waves of a CU do not hand their pieces to
    the texture-address path in the same gap — no effect, profiles/r2_w4y_ablation.log.)"""
    L = []
    e = L.append
    lab = ".Lw4y_loop_%="
    e(lab + ":")
    e("s_waitcnt lgkmcnt(0)")
    fill = {}   # MFMA index -> instructions issued right behind it

    def after(m, *ins):
        fill.setdefault(m, []).extend(ins)

    # k-step 0: reads of (t, ks 1), B pieces of tile t + 2
    for r in range(16):
        after(2 * r, f"ds_read_b128 {fa(1, r)}, {VA} offset:{r * 2048}" if r < 8
              else f"ds_read_b128 {fb(1, r - 8)}, {VB} offset:{(r - 8) * 2048}")
    d0, dstep = (3, 8) if sched == 2 else (32, 4)
    after(d0 - 1, "s_add_u32 %[tmp], %[b2], %[wv]")
    for p in range(8):
        after(d0 + dstep * p, f"s_add_u32 m0, %[tmp], {p * PIECE_STEP}",
              "s_mov_b32 %[soff], %[t2off]" if p == 0 else "s_add_u32 %[soff], %[soff], %[blk]")
        after(d0 + 1 + dstep * p, f"buffer_load_dwordx4 %[ao{p & 1}], %[rb], %[soff] offen lds")
    assert d0 + 1 + dstep * 7 < 64
    # k-step 1: addresses and reads of (t + 1, ks 0), A pieces of tile t + 2, ring rotation, loop counter.
    #   sched 0: reads first (66..96), A pieces late (97..126)
    #   sched 1: A pieces first, right behind the barrier (their data is needed one tile later, at the next barrier:
    #            late pieces leave < 1100 MFMA cycles of latency cover), reads interleaved with them
    #   sched 2: the 16 pieces spread evenly over the tile period (B pieces one per 8 MFMAs through k-step 0, A pieces one per 7
    #            behind the barrier): the texture-address unit serves the four waves at 16 cycles per piece, so bursts of one
    #            piece per 4 MFMAs and wave run it at 100 % and fill its FIFO (SQ_VMEM_TA_CMD_FIFO_FULL); spread, it idles at 50 %.
    #            (Round 2's schedule 2 — barrier 8 MFMAs into the k-step — measured level with 1 and was dropped.)
    bar = 64
    after(bar, "s_barrier", f"v_add_u32_e32 {VA}, %[anxt], %[ar0]", "s_add_u32 %[tmp], %[acur], %[wv]")
    after(bar + 1, f"v_add_u32_e32 {VB}, %[b1], %[br0]")
    if sched == 0:
        rd = [66 + 2 * r for r in range(16)]
        dma = [97 + 4 * g for g in range(8)]
    elif sched == 2:
        dma = [bar + 2 + 7 * g for g in range(8)]
        busy = {m for d in dma for m in (d, d + 1)}
        rd = [m for m in range(bar + 4, 126) if m not in busy][:16]
    else:
        dma = [bar + 2 + 4 * g for g in range(8)]
        rd = [bar + 4 + 4 * (r >> 1) + (r & 1) for r in range(16)]
    for r, m in enumerate(rd):
        after(m, f"ds_read_b128 {fa(0, r)}, {VA} offset:{r * 2048}" if r < 8
              else f"ds_read_b128 {fb(0, r - 8)}, {VB} offset:{(r - 8) * 2048}")
    for g, m in enumerate(dma):
        after(m, f"s_add_u32 m0, %[tmp], {g * PIECE_STEP}",
              "s_mov_b32 %[soff], %[t2off]" if g == 0 else "s_add_u32 %[soff], %[soff], %[blk]")
        after(m + 1, f"buffer_load_dwordx4 %[ao{g & 1}], %[ra], %[soff] offen lds")
    # ring rotation, then the next iteration's t2off = 128 min(t + 3, KT - 1) and its k-step-1 read addresses: in the
    # empty gaps behind the last read / the first A piece (acur, b*, t2off, VA, VB are dead from there on)
    rot = [["s_mov_b32 %[swp], %[b0]"], ["s_mov_b32 %[b0], %[b1]"], ["s_mov_b32 %[b1], %[b2]"], ["s_mov_b32 %[b2], %[swp]"],
           ["s_mov_b32 %[swp], %[acur]"], ["s_mov_b32 %[acur], %[anxt]"], ["s_mov_b32 %[anxt], %[swp]"],
           ["s_add_u32 %[swp], %[t], 3"], ["s_sub_u32 %[t2off], %[kt], 1"], ["s_min_u32 %[swp], %[swp], %[t2off]"],
           [STAGGER[0]], STAGGER[1:3], [STAGGER[3]],
           ["s_lshl_b32 %[t2off], %[swp], 7"], [f"v_add_u32_e32 {VA}, %[acur], %[ar1]"], [f"v_add_u32_e32 {VB}, %[b0], %[br1]"]]
    first = max(max(rd), dma[0] + 1) + 1
    slots = [m for m in range(first, 126) if m not in fill]
    place(rot, slots, after, sched)
    after(126, "s_add_u32 %[t], %[t], 1", "s_cmp_lt_u32 %[t], %[kt]")
    for m in range(128):
        ks, i, j = m >> 6, (m >> 3) & 7, m & 7
        if m == 64:
            e("s_waitcnt lgkmcnt(0)" if bar != 64 else "s_waitcnt vmcnt(8) lgkmcnt(0)")
        if m == bar and bar != 64:
            e("s_waitcnt vmcnt(8)")
        e(mfma(ks, i, j))
        for ins in fill.get(m, []):
            e(ins)
    e(f"s_cbranch_scc1 {lab}")
    return L


def gen(sched):
    ablate = sched - NSCHED + 1 if sched >= NSCHED else 0   # 1 no DMA, 2 no vmcnt wait, 3 no reads
    base = 1 if ablate else sched
    L = gen_init()
    L += gen_body(base)
    L.append("s_waitcnt vmcnt(0) lgkmcnt(0)")
    if ablate == 1:
        L = [ln for ln in L if not ln.startswith("buffer_load")]
    if ablate in (1, 2):
        L = [ln.replace("s_waitcnt vmcnt(8) lgkmcnt(0)", "s_waitcnt lgkmcnt(0)") for ln in L]
    if ablate == 3:
        keep_until = next(i for i, ln in enumerate(L) if ln.startswith(".Lw4y_loop"))
        L = L[:keep_until] + [ln for ln in L[keep_until:] if not ln.startswith("ds_read")]
    return L


def gen_nn():
    """NN (B stored [K][N]): the B image is two sub-images of 128 contiguous columns, [64 k][256 B] each, 32-B column pairs
    XOR-ed by key(k) = ((k & 3) << 1) | ((k >> 3) & 1); a B fragment (16 columns x 32 k) is two ds_read_b64_tr_b16 (4 k
    rows each) per lane group.  Per k-step: 8 A reads + 16 transpose reads.  Literal VGPRs: v108..v115 = B read lane
    offsets per column pair j (without the slot base), v116..v123 = the same + the base of the B slot of the tile being
    read, v124 = A read address."""
    L = []
    e = L.append
    BJ0, BJ = 108, 116

    def rd_a(ks, i):
        return f"ds_read_b128 {fa(ks, i)}, {VA} offset:{i * 2048}"

    def rd_b(ks, j, x):
        b = FRAG0 + 32 + 64 * ks + 4 * j + 2 * x
        return f"ds_read_b64_tr_b16 v[{b}:{b + 1}], v{BJ + j} offset:{ks * 8192 + x * 1024}"

    def reads(ks):
        return [rd_a(ks, i) for i in range(8)] + [rd_b(ks, j, x) for j in range(8) for x in range(2)]

    e("s_mov_b32 %[t], 0")
    e("s_mov_b32 %[acur], %[a0]")
    e("s_add_u32 %[anxt], %[a0], 0x8000")
    e("s_add_u32 %[b0], %[a0], 0x10000")
    e("s_add_u32 %[b1], %[a0], 0x18000")
    e("s_add_u32 %[b2], %[a0], 0x20000")
    for j in range(8):
        e(f"v_xor_b32_e32 v{BJ0 + j}, {j}, %[key]")
    for j in range(8):
        e(f"v_lshl_add_u32 v{BJ0 + j}, v{BJ0 + j}, 5, %[bk]")
    for j in range(8):
        e(f"v_add_u32_e32 v{BJ + j}, %[b0], v{BJ0 + j}")
    e(f"v_add_u32_e32 {VA}, %[acur], %[ar0]")
    for ins in reads(0):
        e(ins)
    e("s_sub_u32 %[swp], %[kt], 1")
    e("s_min_u32 %[swp], %[swp], 2")
    for ins in STAGGER:
        e(ins)
    e("s_lshl_b32 %[t2off], %[swp], 7")
    e("s_mul_i32 %[t2offb], %[swp], %[bkt]")
    e(f"v_add_u32_e32 {VA}, %[acur], %[ar1]")
    e(".Lw4y_loop_%=:")
    e("s_waitcnt lgkmcnt(0)")
    fill = {}

    def after(m, *ins):
        fill.setdefault(m, []).extend(ins)

    # k-step 0: the 24 reads of (t, ks 1) two per three MFMAs, then the B pieces of tile t + 2 (sub-image h = p >> 2,
    # piece 4 wave + (p & 3))
    slots0 = [m for m in range(36) if m % 3 != 2]
    for ins, m in zip(reads(1), slots0):
        after(m, ins)
    after(36, "s_add_u32 %[tmp], %[b2], %[wvb]")
    for p in range(8):
        h, p2 = p >> 2, p & 3
        soff = ("s_mov_b32 %[soff], %[t2offb]" if p == 0 else "s_add_u32 %[soff], %[t2offb], 256" if p == 4
                else "s_add_u32 %[soff], %[soff], %[bq]")
        after(37 + 3 * p, f"s_add_u32 m0, %[tmp], {h * 16384 + p2 * 1024}", soff)
        after(38 + 3 * p, f"buffer_load_dwordx4 %[bo{(p2 >> 1) & 1}], %[rb], %[soff] offen lds")
    # k-step 1: barrier, addresses of tile t + 1, A pieces of tile t + 2 right behind the barrier, reads of (t + 1, ks 0)
    bar = 64
    after(bar, "s_barrier", f"v_add_u32_e32 {VA}, %[anxt], %[ar0]", "s_add_u32 %[tmp], %[acur], %[wv]")
    for j in range(8):
        after(bar + 1 + (j >> 1), f"v_add_u32_e32 v{BJ + j}, %[b1], v{BJ0 + j}")
    dma = [bar + 2 + 4 * g for g in range(8)]
    for g, m in enumerate(dma):
        after(m, f"s_add_u32 m0, %[tmp], {g * PIECE_STEP}",
              "s_mov_b32 %[soff], %[t2off]" if g == 0 else "s_add_u32 %[soff], %[soff], %[blk]")
        after(m + 1, f"buffer_load_dwordx4 %[ao{g & 1}], %[ra], %[soff] offen lds")
    slots1 = [m for m in range(bar + 5, 126) if m not in [d + 1 for d in dma]][:24]
    for ins, m in zip(reads(0), slots1):
        after(m, ins)
    rot = [["s_mov_b32 %[swp], %[b0]"], ["s_mov_b32 %[b0], %[b1]"], ["s_mov_b32 %[b1], %[b2]"], ["s_mov_b32 %[b2], %[swp]"],
           ["s_mov_b32 %[swp], %[acur]"], ["s_mov_b32 %[acur], %[anxt]"], ["s_mov_b32 %[anxt], %[swp]"],
           ["s_add_u32 %[swp], %[t], 3"], ["s_sub_u32 %[t2off], %[kt], 1"], ["s_min_u32 %[swp], %[swp], %[t2off]"],
           [STAGGER[0]], STAGGER[1:3], [STAGGER[3]],
           ["s_lshl_b32 %[t2off], %[swp], 7"], ["s_mul_i32 %[t2offb], %[swp], %[bkt]"], [f"v_add_u32_e32 {VA}, %[acur], %[ar1]"]]
    first = max(slots1) + 1
    slots = [m for m in range(first, 126) if m not in fill]
    place(rot, slots, after, "nn")
    after(126, "s_add_u32 %[t], %[t], 1", "s_cmp_lt_u32 %[t], %[kt]")
    for m in range(128):
        ks, i, j = m >> 6, (m >> 3) & 7, m & 7
        if m == 64:
            e("s_waitcnt vmcnt(8) lgkmcnt(0)")
        e(mfma(ks, i, j))
        for ins in fill.get(m, []):
            e(ins)
    e("s_cbranch_scc1 .Lw4y_loop_%=")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    return L


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


def render_nn():
    lines = gen_nn()
    check_literal_vgprs(lines, range(108, 256), "hgemm_w4y NN loop")
    body = "\n".join(f'    "{ln}\\n\\t"' for ln in lines)
    vclob = ", ".join(f'"v{r}"' for r in range(108, 256))
    n_mfma = sum(ln.startswith("v_mfma") for ln in lines)
    head = (f"// GENERATED by tools/gen_hgemm_w4y.py (NN) — do not edit ({len(lines)} instructions, {n_mfma} MFMAs per K tile).\n"
            "// Operands (hgemm_w4y.hip): as the TN statement + wvb (wave * 4096: NN B pieces), bq (bytes between 4-row B pieces), bkt (bytes\n"
            "// per B K tile), bo0 / bo1 (B DMA lane offsets), bk / key (B transpose-read lane offset and swizzle key).\n")
    return (head + "asm volatile(\n" + body + "\n"
            "    : [t] \"=&s\"(w4y_t), [acur] \"=&s\"(w4y_acur), [anxt] \"=&s\"(w4y_anxt), [b0] \"=&s\"(w4y_b0), [b1] \"=&s\"(w4y_b1),\n"
            "      [b2] \"=&s\"(w4y_b2), [soff] \"=&s\"(w4y_soff), [t2off] \"=&s\"(w4y_t2off), [t2offb] \"=&s\"(w4y_t2offb),\n"
            "      [tmp] \"=&s\"(w4y_tmp), [swp] \"=&s\"(w4y_swp)\n"
            "    : [kt] \"s\"(KT), [stg] \"s\"(w4y_stg), [a0] \"s\"(w4y_a0), [wv] \"s\"(w4y_wv), [wvb] \"s\"(w4y_wvb), [blk] \"s\"(w4y_blk), [bq] \"s\"(w4y_bq),\n"
            "      [bkt] \"s\"(w4y_bkt), [ra] \"s\"(w4y_ra), [rb] \"s\"(w4y_rb), [ao0] \"v\"(w4y_ao0), [ao1] \"v\"(w4y_ao1),\n"
            "      [bo0] \"v\"(w4y_bo0), [bo1] \"v\"(w4y_bo1), [ar0] \"v\"(fr.a_ad[0]), [ar1] \"v\"(fr.a_ad[1]), [bk] \"v\"(w4y_bk),\n"
            "      [key] \"v\"(w4y_key)\n"
            f"    : \"memory\", \"scc\", {vclob}, LC_AGPR_ALL);\n")


def render(sched):
    lines = gen(sched)
    check_literal_vgprs(lines, VCLOB, f"hgemm_w4y TN loop, schedule {sched}")
    body = "\n".join(f'    "{ln}\\n\\t"' for ln in lines)
    vclob = ", ".join(f'"v{r}"' for r in VCLOB)
    n_mfma = sum(ln.startswith("v_mfma") for ln in lines)
    head = (f"// GENERATED by tools/gen_hgemm_w4y.py (schedule {sched}) — do not edit ({len(lines)} instructions, {n_mfma} MFMAs per K tile).\n"
            "// Operands (hgemm_w4y.hip): kt, stg (K-loop stagger in tiles, < kt), a0 (LDS address of A ring slot 0), wv (wave * 1024), blk (bytes between 32-row\n"
            "// blocks), ra / rb (buffer descriptors, u32x4 SGPR tuples), ao0 / ao1 (DMA lane offsets), ar0 / ar1 / br0 / br1\n"
            "// (fragment read lane offsets of k-step 0 / 1).\n")
    return (head + "asm volatile(\n" + body + "\n"
            "    : [t] \"=&s\"(w4y_t), [acur] \"=&s\"(w4y_acur), [anxt] \"=&s\"(w4y_anxt), [b0] \"=&s\"(w4y_b0), [b1] \"=&s\"(w4y_b1),\n"
            "      [b2] \"=&s\"(w4y_b2), [soff] \"=&s\"(w4y_soff), [t2off] \"=&s\"(w4y_t2off), [tmp] \"=&s\"(w4y_tmp), [swp] \"=&s\"(w4y_swp)\n"
            "    : [kt] \"s\"(KT), [stg] \"s\"(w4y_stg), [a0] \"s\"(w4y_a0), [wv] \"s\"(w4y_wv), [blk] \"s\"(w4y_blk), [ra] \"s\"(w4y_ra),\n"
            "      [rb] \"s\"(w4y_rb),\n"
            "      [ao0] \"v\"(w4y_ao0), [ao1] \"v\"(w4y_ao1), [ar0] \"v\"(fr.a_ad[0]), [ar1] \"v\"(fr.a_ad[1]), [br0] \"v\"(fr.b_ad[0]),\n"
            "      [br1] \"v\"(fr.b_ad[1])\n"
            f"    : \"memory\", \"scc\", {vclob}, LC_AGPR_ALL);\n")


def main():
    rc = 0
    if "--diag" in sys.argv:
        d = Path(sys.argv[sys.argv.index("--diag") + 1])
        d.mkdir(parents=True, exist_ok=True)
        for sched in range(NSCHED, NSCHED + NDIAG):
            out_path(sched, d).write_text(render(sched))
        return 0
    todo = [(render(sched), out_path(sched)) for sched in range(NSCHED)]
    todo.append((render_nn(), ROOT / "leetcuda_amd" / "csrc" / "hgemm_w4y_loop_nn.inc"))
    for text, out in todo:
        if "--check" in sys.argv:
            if not out.exists() or out.read_text() != text:
                print(f"{out} is stale: run tools/gen_hgemm_w4y.py", file=sys.stderr)
                rc = 1
        else:
            out.write_text(text)
            print(f"wrote {out} ({len(text.splitlines())} lines)")
    return rc


if __name__ == "__main__":
    sys.exit(main())
