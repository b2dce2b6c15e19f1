#!/usr/bin/env python3
"""Emit leetcuda_amd/csrc/attn_w4i_d<D>.inc: the merged phase of attn_fwd_w4i_kernel (attn_w4i.hip) as ONE hand-ordered asm
statement per phase, plus the register map and the tail statement.

Why: attn_w4n / attn_w4g issue one asm statement per MFMA slot and leave the softmax (v_exp_f32, row-sum v_add_f32,
v_cvt_pk_f16_f32) to hipcc between them.  With ONE wave per SIMD the in-order instruction stream is the critical path (a
v_mfma_f32_16x16x32_f16 holds the matrix core for 16 cycles ~ 4 issue slots) and hipcc adds to it: an s_nop at statement
boundaries (57 / 23 per 64-key tile at D = 128 / 64), copies of the exp results, address adds per transpose read, its own
order.  Moving the fillers INTO per-slot statements does not help (hipcc pads every boundary: measured 101 s_nop per tile).
What does: the kernel is compiled with __attribute__((amdgpu_num_vgpr(LB))), which makes v[LB:255] RESERVED registers — hipcc
never allocates them — and the whole softmax state lives there under literal names, so a phase can be a single statement:

    v[...]  Sᵀ buffers A / B (32 + 32), P fragments A / B (16 + 16), Vᵀ fragments (4 NDB), −m tuples (16), row sums (8),
            their per-block totals (4), l (4), two exp register pairs (4), odd-pair Vᵀ addresses (8, D = 128), temporaries

Per phase (one 32-row KV half-tile j = 2 t + H, 64 query rows per wave, NDS = D / 32, NDB = D / 16, NS = 16 NDS slots), in issue order:
    s_waitcnt lgkmcnt(0)                     K(j+1) fragments / Vᵀ set A of the previous phase's reads
    slot s:  MFMA | at most one LDS read | <= 3 VALU | LDS-DMA piece (H = 0: s_add m0 / soffset behind one MFMA, buffer_load ... lds behind the next)
       even slot 2 i   Sᵀ(j+1) block (kvb, qb) (+)= K(j+1) fragment (kvb, ds) x Q~ fragment (qb, ds)   ds = i >> 3, kvb = (i >> 2) & 1, qb = i & 3
       odd  slot 2 i+1 Oᵀ block (db, qb) += Vᵀ(j−1) fragment db x Pᵀ(j−1) fragment qb                   db = i >> 2, qb = i & 3
       reads: K(j+2) fragment c -> AGPR buffer H and Vᵀ(j−1) set B (db >= NDB / 2) alternating from slot 0; Vᵀ(j) set A from slot NS / 2
       softmax(j): pair p = (kvb = p >> 3, qb = (p >> 1) & 3, k2 = p & 1) owns NS / 16 slots: v_exp of its two values into exp set
              p & 1; row sums (unrounded P, split_q.cu:467-468) and RNE pack of pair p − 1 from the other set; l += the previous phase's
              totals and the per-block totals of this phase as soon as a block's last pair is summed
    v_max of the totals' bit patterns -> %[worst]   (the overflow guard's input: attn_w4i.hip decides and runs the slow path in C++)
Arithmetic order is attn_w4n's (MFMA order per accumulator block, the same exp2 / add / pack sequence per row): bit-identical
results, asserted on the GPU.  Hazards inside the stream (nothing is padded by hipcc): a v_exp result is first read >= 1 slot
later; Sᵀ(j) was written by the previous phase's MFMAs >= 16 slots before its first v_exp; P(j) is packed in this phase and read by
the next phase's MFMAs; an M0 write is followed by an MFMA before the LDS-DMA that uses it; MFMA operands written by VALU in
front of the statement (slow path, prologue) are covered by the s_nop the C++ side issues.

usage: tools/gen_attn_w4i.py [--check]
       tools/gen_attn_w4i.py --diag DIR    (liblc_diag.so only: ablated copies attn_w4i_d<D>_abl<K>.inc of the phase statements —
                                            K bits: 1 no LDS-DMA, 2 no LDS reads, 4 no softmax VALU, 8 no MFMA, 16 no per-tile wait + barrier, 32 no guard decision
                                            (the last two act in attn_w4i.hip); results WRONG by
                                            design, timing only: tools/attn_w4i_ablate.py)"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
KVB = 64


class Cfg:
    def __init__(self, D):
        self.D = D
        self.NDS, self.NDB = D // 32, D // 16
        self.GROWB = 2 * D                          # bytes per K / V row in global memory
        self.ROWB = 256 if D > 64 else 128          # ... and in LDS (D = 96 / 32 keep the 256-B / 128-B rows of D = 128 / 64: attn_mp.h W4G.hip W4G)
        self.NS = 16 * self.NDS
        self.NRV, self.NRK = self.NDB, 2 * self.NDS
        self.TILE = KVB * self.ROWB
        self.PPW = self.TILE // 1024 // 4
        self.GPIECE = (1024 // self.ROWB) * self.GROWB   # one LDS-DMA piece (1 KiB of LDS rows) in global memory
        self.KBUF = 8 * self.NDS
        self.O, self.K = 0, 16 * self.NDB
        self.Q = self.K + 2 * self.KBUF
        self.SPP = self.NS // 16
        self.NVX = 4                       # Vᵀ address registers per tile (D = 128: pair 2 u; D = 64: pair u)
        self.ODD = D > 64                  # odd column pairs sit at ±32 B from the even pair's slot (lane-dependent sign)
        # ---- literal register map, from v255 downwards
        top = [256]

        def alloc(n):
            top[0] -= n
            return top[0]
        self.SA, self.SB = alloc(32), alloc(32)
        self.PA, self.PB = alloc(16), alloc(16)
        self.VF = alloc(4 * self.NDB)
        self.NEGM = alloc(16)
        self.PS = alloc(8)
        self.SUM = alloc(4)
        self.LRUN = alloc(4)
        self.E = alloc(4)
        self.VCO = alloc(4) if self.ODD else None     # vc[u] + vodd
        self.VPO = alloc(4) if self.ODD else None     # vp[u] + vodd
        self.KA = alloc(self.NDS)                     # schedule 1: this tile period's K(t+1) fragment addresses ...
        self.VC = alloc(4)                            # ... Vᵀ addresses of tile t ...
        self.VP = alloc(4)                            # ... and of tile t − 1 (kept in registers, advanced inside the H = 1 statement)
        self.T = alloc(3)                             # temporaries
        self.LB = top[0] & ~7                         # first reserved register = amdgpu_num_vgpr


def out_path(D):
    return ROOT / "leetcuda_amd" / "csrc" / f"attn_w4i_d{D}.inc"


def v(n, cnt=1):
    return f"v{n}" if cnt == 1 else f"v[{n}:{n + cnt - 1}]"


def a(n, cnt=4):
    return f"a[{n}:{n + cnt - 1}]"


NSCHED = 2   # 0: softmax pairs spread over the whole phase, LDS addresses of a tile period computed by hipcc between the statements;
             # 1: (a) the softmax finishes early (D = 128: a pair every 3 slots instead of 4), so the block totals and the guard's
             #    v_max sit in MFMA shadows and the statement ends with MFMAs only; (b) the tile period's LDS addresses live in
             #    reserved registers and are advanced for tile t + 1 inside the H = 1 statement (no address arithmetic in the gap
             #    between two tiles); (c) LDS-DMA pieces only in slots that carry no LDS read


def read_schedule(c):
    rd, first = {}, []
    for i in range(max(c.NRK, c.NRV)):
        if i < c.NRK:
            first.append(("k", i))
        if i < c.NRV:
            first.append(("vb", i))
    assert len(first) <= c.NS // 2 - 2
    for s, r in enumerate(first):
        rd[s] = r
    for i in range(c.NRV):
        rd[c.NS // 2 + i] = ("va", i)
    return rd, len(first)


def pair(p):
    return p >> 3, (p >> 1) & 3, p & 1      # kvb, qb, k2


def gen_phase(c, H, sched=0):
    """-> (asm lines, operand description).  Operands: [kaN] [vcN] [vpN] [vodd] VGPR inputs; H = 0 only: [koff] [voff] VGPR,
    [rk] [rv] SGPR descriptor tuples, [m0b] [sob] SGPR, [st] SGPR temporary; [worst] VGPR output."""
    L = []
    e = L.append
    rd, nfirst = read_schedule(c)
    sr, sw = (c.SA, c.SB) if H == 0 else (c.SB, c.SA)
    pw, pr = (c.PA, c.PB) if H == 0 else (c.PB, c.PA)
    KQ = c.K + c.KBUF * (1 - H)
    VB_H = 1 if H == 0 else 0
    vb_arr = "vp" if H == 0 else "vc"

    def S(base, kvb, qb, r=None):
        b = base + 4 * (4 * kvb + qb)
        return v(b, 4) if r is None else v(b + r)

    def P(base, qb, dw=None):
        return v(base + 4 * qb, 4) if dw is None else v(base + 4 * qb + dw)

    def PSr(qb, w):
        return v(c.PS + 2 * qb + w)

    def Er(setp, i):
        return v(c.E + 2 * (setp & 1) + i)

    def areg(arr, u):
        """address register u of `arr` (ka / vc / vp): an operand (schedule 0) or a reserved register (schedule 1)"""
        if sched == 0:
            return f"%[{arr}{u}]"
        return v({"ka": c.KA, "vc": c.VC, "vp": c.VP}[arr] + u)

    def vaddr(arr, db):
        """register holding the transpose-read address of column block db of the tile `arr` points at"""
        if not c.ODD:
            return areg(arr, db)
        if db & 1:
            return v((c.VCO if arr == "vc" else c.VPO) + (db >> 1))
        return areg(arr, db >> 1)

    fill = {s: [] for s in range(-1, c.NS + 1)}     # slot -> filler instructions behind its MFMA (−1: before the first MFMA)

    # ---- l += totals of the previous phase (SUM), per block, anywhere before the block's totals are rewritten (its last pair's sums)
    for qb in range(4):
        fill[(3 if (sched == 1 and c.SPP == 4) else c.SPP) * 2 * qb].append(f"v_add_f32 {v(c.LRUN + qb)}, {v(c.LRUN + qb)}, {v(c.SUM + qb)}")
    # ---- odd-pair Vᵀ addresses of this tile period (D = 128): H = 0 computes both sets, H = 1 reuses VCO
    if c.ODD and H == 0 and sched == 0:
        for u in range(4):
            fill[-1].append(f"v_add_u32 {v(c.VPO + u)}, %[vp{u}], %[vodd]")      # needed from slot 1 on (set B reads)
        for u in range(4):
            fill[2 + 2 * u].append(f"v_add_u32 {v(c.VCO + u)}, %[vc{u}], %[vodd]")   # needed from slot NS / 2 on
    # ---- schedule 1, H = 1: advance the address registers to tile t + 1 once their last readers of this tile period are issued
    # (K fragments: first half; Vᵀ set A of tile t: slots NS / 2 .. NS / 2 + NRV − 1; VP / VPO are read by H = 0 only)
    if sched == 1 and H == 1:
        upd = [f"v_mov_b32 {v(c.VP + u)}, {v(c.VC + u)}" for u in range(4)]
        if c.ODD:
            upd += [f"v_mov_b32 {v(c.VPO + u)}, {v(c.VCO + u)}" for u in range(4)]
        upd += [f"v_add_u32 {v(c.VC + u)}, %[sbn1], %[vx{u}]" for u in range(4)]
        if c.ODD:
            upd += [f"v_add_u32 {v(c.VCO + u)}, {v(c.VC + u)}, %[vodd]" for u in range(4)]
        upd += [f"v_add_u32 {v(c.KA + d)}, %[sbn2], %[kx{d}]" for d in range(c.NDS)]
        s0 = c.NS // 2 + c.NRV + 1
        nslots = c.NS - s0
        per = -(-len(upd) // nslots)                 # instructions per slot (1, or 2 where the phase is short: D = 96)
        for i, ins in enumerate(upd):                # (order matters: VP <- VC before VC is advanced, VCO after VC)
            fill[s0 + i // per].append(ins)
    # ---- softmax.  A block's FIRST pair (p = 2 qb) exponentiates straight into the block's two row-sum registers PS[qb][0 / 1]
    # (0 + e = e: no move); the later pairs go through the exp sets and are added on.  The pack of the first pair reads the PS
    # registers before the second pair's sums change them (pack of pair p and sums of pair p + 1 sit in different pairs' slots).
    def Edst(p, i):
        kvb, qb, k2 = pair(p)
        return PSr(qb, i) if (kvb == 0 and k2 == 0) else Er(p, i)

    spp = 3 if (sched == 1 and c.SPP == 4) else c.SPP
    for p in range(17):
        base = spp * p
        if c.SPP == 4:
            plan = {"a0": base, "x0": base, "a1": base + 1, "c": base + 2, "x1": base + 2}
        elif c.SPP == 3:
            plan = {"a0": base, "x0": base, "a1": base + 1, "c": base + 2, "x1": base + 1}
        elif c.SPP == 2:
            plan = {"a0": base, "x0": base, "a1": base + 1, "c": base + 1, "x1": base + 1}
        else:          # one slot per pair (D = 32): everything of pair p − 1, then both exps of pair p
            plan = {"a0": base, "x0": base, "a1": base, "c": base, "x1": base}
        if p >= 1:       # sums and pack of pair p − 1
            kvb, qb, k2 = pair(p - 1)
            first = kvb == 0 and k2 == 0
            if not first:
                for w, key in ((0, "a0"), (1, "a1")):
                    fill[min(plan[key], c.NS)].append(f"v_add_f32 {PSr(qb, w)}, {PSr(qb, w)}, {Er(p - 1, w)}")
            fill[min(plan["c"], c.NS)].append(f"v_cvt_pk_f16_f32 {P(pw, qb, 2 * kvb + k2)}, {Edst(p - 1, 0)}, {Edst(p - 1, 1)}")
            if kvb == 1 and k2 == 1:                # the block's last pair: its totals are final
                fill[min(plan["c"], c.NS)].append(f"v_add_f32 {v(c.SUM + qb)}, {PSr(qb, 0)}, {PSr(qb, 1)}")
        if p <= 15:      # exps of pair p
            kvb, qb, k2 = pair(p)
            fill[plan["x0"]].append(f"v_exp_f32 {Edst(p, 0)}, {S(sr, kvb, qb, 2 * k2)}")
            fill[plan["x1"]].append(f"v_exp_f32 {Edst(p, 1)}, {S(sr, kvb, qb, 2 * k2 + 1)}")
    # (order inside a slot: sums, pack, totals first, then the exps — the lists above were filled in that order per pair; an exp of
    #  pair p never precedes a sum of pair p − 1 that reads the register it overwrites: different exp sets)
    # ---- LDS-DMA pieces of tile t + 2 (H = 0): piece i = K pieces 0 .. PPW − 1, then V pieces; wave w stages piece w + 4 i'
    dma = {}
    if H == 0:
        step0 = c.NS // (2 * c.PPW)
        starts = [step0 * i + (3 if step0 >= 5 else step0 - 2) for i in range(2 * c.PPW)]      # schedule 0: evenly from the phase start
        if sched == 1:       # read-free slots: between the first-half reads and NS / 2, then behind the set A reads (when the phase has enough of them)
            free = [s for s in range(nfirst + 1, c.NS // 2 - 1)] + [s for s in range(c.NS // 2 + c.NRV + 1, c.NS - 2)]
            step = max(2, len(free) // (2 * c.PPW))
            cand = [free[min(i * step, len(free) - 2)] for i in range(2 * c.PPW)] if len(free) >= 2 else []
            if len(set(cand)) == 2 * c.PPW and all(b - a >= 2 for a, b in zip(cand, cand[1:])):
                starts = cand
        for i, s0 in enumerate(starts):
            is_v, ii = i >= c.PPW, i % c.PPW
            dma.setdefault(s0, []).append(f"s_add_u32 m0, %[m0b], {(c.TILE if is_v else 0) + 4096 * ii}")
            dma.setdefault(s0, []).append(f"s_add_u32 %[st], %[sob], {4 * c.GPIECE * ii}")
            dma.setdefault(s0 + 1, []).append(f"buffer_load_dwordx4 %[{'voff' if is_v else 'koff'}], %[{'rv' if is_v else 'rk'}], %[st] offen lds")
        assert max(starts) + 1 < c.NS

    e("s_waitcnt lgkmcnt(0)")
    for ins in fill[-1]:
        e(ins)
    for s in range(c.NS):
        i = s >> 1
        if s == c.NS // 2 + 1:
            # Vᵀ set B (and every K(j+2) fragment) was requested in slots 0 .. nfirst − 1; younger: the set A reads of slots NS / 2, NS / 2 + 1 (in-order returns)
            e("s_waitcnt lgkmcnt(1)")
        if s % 2 == 0:
            ds, kvb, qb = i >> 3, (i >> 2) & 1, i & 3
            k_r, q_r = KQ + 4 * (c.NDS * kvb + ds), c.Q + 4 * (c.NDS * qb + ds)
            cop = v(c.NEGM + 4 * qb, 4) if ds == 0 else S(sw, kvb, qb)
            e(f"v_mfma_f32_16x16x32_f16 {S(sw, kvb, qb)}, {a(k_r)}, {a(q_r)}, {cop}")
        else:
            db, qb = i >> 2, i & 3
            o_r = c.O + 4 * (4 * db + qb)
            e(f"v_mfma_f32_16x16x32_f16 {a(o_r)}, {v(c.VF + 4 * db, 4)}, {P(pr, qb)}, {a(o_r)}")
        if s in rd:
            kind, ci = rd[s]
            if kind == "k":
                kr = c.K + c.KBUF * H + 4 * ci
                e(f"ds_read_b128 {a(kr)}, {areg('ka', ci % c.NDS)} offset:{H * 32 * c.ROWB + (ci // c.NDS) * 16 * c.ROWB}")
            else:
                set_a = kind == "va"
                rdb, rx = (0 if set_a else c.NDB // 2) + (ci >> 1), ci & 1
                vof = (H if set_a else VB_H) * 32 * c.ROWB + rx * 16 * c.ROWB
                e(f"ds_read_b64_tr_b16 {v(c.VF + 4 * rdb + 2 * rx, 2)}, {vaddr('vc' if set_a else vb_arr, rdb)} offset:{vof}")
        for ins in fill[s]:
            e(ins)
        for ins in dma.get(s, []):
            e(ins)
    for ins in fill[c.NS]:
        e(ins)
    # ---- the overflow guard's input: the largest bit pattern of the four block totals (non-negative floats order like integers);
    # when the softmax finished inside the phase (schedule 1, D = 128) the two v_max are moved up behind the last total
    gmax = [f"v_max3_u32 {v(c.T)}, {v(c.SUM)}, {v(c.SUM + 1)}, {v(c.SUM + 2)}", f"v_max_u32 %[worst], {v(c.T)}, {v(c.SUM + 3)}"]
    last_total = max(i for i, ln in enumerate(L) if ln.startswith(f"v_add_f32 {v(c.SUM + 3)},"))
    nxt = [i for i, ln in enumerate(L) if i > last_total and ln.startswith("v_mfma")]
    if sched == 1 and len(nxt) >= 4:
        L.insert(nxt[0] + 1, gmax[0])          # behind the next MFMA ...
        L.insert(nxt[1] + 2, gmax[1])          # ... and the one after it (indices shift by the first insertion)
    else:
        L.extend(gmax)
    return L


def gen_tail(c, sched=0):
    """Oᵀ += Vᵀ(2T−1)·Pᵀ(2T−1) after the loop: P in buffer B, set A already in registers, set B (second half of the last tile) read here.
    (schedule 1: the last H = 1 statement has advanced VC to the next tile; the last tile's addresses are in VP / VPO)"""
    L = ["s_waitcnt lgkmcnt(0)"]
    for ci in range(c.NRV):
        rdb, rx = c.NDB // 2 + (ci >> 1), ci & 1
        if sched == 0:
            addr = f"%[vc{rdb}]" if not c.ODD else (v(c.VCO + (rdb >> 1)) if rdb & 1 else f"%[vc{rdb >> 1}]")
        else:
            addr = v(c.VP + rdb) if not c.ODD else (v(c.VPO + (rdb >> 1)) if rdb & 1 else v(c.VP + (rdb >> 1)))
        L.append(f"ds_read_b64_tr_b16 {v(c.VF + 4 * rdb + 2 * rx, 2)}, {addr} offset:{32 * c.ROWB + rx * 16 * c.ROWB}")
    L.append("s_waitcnt lgkmcnt(0)")
    for i in range(4 * c.NDB):
        db, qb = i >> 2, i & 3
        o_r = c.O + 4 * (4 * db + qb)
        L.append(f"v_mfma_f32_16x16x32_f16 {a(o_r)}, {v(c.VF + 4 * db, 4)}, {v(c.PB + 4 * qb, 4)}, {a(o_r)}")
    return L


def cstr(lines):
    return "\n".join(f'    "{ln}\\n\\t"' for ln in lines)


ABLATIONS = (1, 2, 3, 4, 7, 8, 23, 55)   # + 16: no per-tile wait + barrier, + 32: no guard decision (attn_w4i.hip W4I_ABL)


def ablate(lines, abl):
    out = []
    for ln in lines:
        op = ln.split()[0]
        if abl & 1 and (op.startswith("buffer_load") or ln.startswith("s_add_u32 m0") or ln.startswith("s_add_u32 %[st]")):
            continue
        if abl & 2 and op.startswith("ds_read"):
            continue
        if abl & 4 and op in ("v_exp_f32", "v_add_f32", "v_cvt_pk_f16_f32", "v_mov_b32"):
            continue
        if abl & 8 and op.startswith("v_mfma"):
            continue
        out.append(ln)
    return out


def render(D, abl=0):
    c = Cfg(D)
    vclob = ", ".join(f'"v{r}"' for r in range(c.LB, 256))
    out = []
    w = out.append
    w(f"// GENERATED by tools/gen_attn_w4i.py (D = {D}) — do not edit.  Register map + the two phase statements + the tail statement.")
    w(f"#if W4I_PART == 0   // ---- register map (v[{c.LB}:255] are reserved: the kernel is compiled with amdgpu_num_vgpr({c.LB}))")
    for name in ("LB", "SA", "SB", "PA", "PB", "VF", "NEGM", "PS", "SUM", "LRUN", "KA", "VC", "VP") + (("VCO", "VPO") if c.ODD else ()):
        w(f"static constexpr int {name} = {getattr(c, name)};")
    w(f"static constexpr int NSCHED = {NSCHED};")
    w(f"#define W4I_VCLOB_{D} {vclob}")
    for H in (0, 1):
        w(f"#elif W4I_PART == {1 + H}   // ---- phase H = {H} (one statement per schedule, selected by the kernel's SCHED)")
        for sched in range(NSCHED):
            lines = ablate(gen_phase(c, H, sched), abl)
            n_mfma = sum(x.startswith("v_mfma") for x in lines)
            n_valu = sum(x.startswith("v_") and not x.startswith("v_mfma") for x in lines)
            w(("if" if sched == 0 else "} else if") + f" constexpr (SCHED == {sched}) {{   // {len(lines)} instructions, {n_mfma} MFMA, {n_valu} VALU")
            w("asm volatile(")
            w(cstr(lines))
            outs = ['[worst] "=&v"(worst)']      # (written before the statement's last operand reads: early clobber)
            if sched == 0:
                ops_in = [f'[ka{i}] "v"(ka[{i}])' for i in range(c.NDS)] + [f'[vc{i}] "v"(vc[{i}])' for i in range(c.NVX)]
                if H == 0:
                    ops_in += [f'[vp{i}] "v"(vp[{i}])' for i in range(c.NVX)]
                if c.ODD and H == 0:
                    ops_in.append('[vodd] "v"(vodd)')
            elif H == 1:
                ops_in = [f'[kx{i}] "v"(kx[{i}])' for i in range(c.NDS)] + [f'[vx{i}] "v"(vx[{i}])' for i in range(c.NVX)]
                ops_in += ['[sbn1] "s"(w4i_sbn1)', '[sbn2] "s"(w4i_sbn2)'] + (['[vodd] "v"(vodd)'] if c.ODD else [])
            else:
                ops_in = []
            if H == 0:
                ops_in += ['[koff] "v"(k_off)', '[voff] "v"(v_off)', '[rk] "s"(w4i_rk)', '[rv] "s"(w4i_rv)', '[m0b] "s"(w4i_m0b)', '[sob] "s"(w4i_sob)']
                outs.append('[st] "=&s"(w4i_st)')
            w("    : " + ", ".join(outs))
            w("    : " + ", ".join(ops_in))
            w(f'    : "memory", "scc", W4I_VCLOB_{D}, LC_AGPR_ALL);')
        w("}")
    w("#elif W4I_PART == 3   // ---- tail")
    for sched in range(NSCHED):
        lines = ablate(gen_tail(c, sched), abl & 10)
        w(("if" if sched == 0 else "} else if") + f" constexpr (SCHED == {sched}) {{   // {len(lines)} instructions")
        w("asm volatile(")
        w(cstr(lines))
        w("    :")
        w("    : " + (", ".join(f'[vc{i}] "v"(vc[{i}])' for i in range(c.NVX)) if sched == 0 else '[zero] "n"(0)'))
        w(f'    : "memory", W4I_VCLOB_{D}, LC_AGPR_ALL);')
    w("}")
    w("#endif")
    return "\n".join(out) + "\n"


def main():
    rc = 0
    if "--diag" in sys.argv:
        d = Path(sys.argv[sys.argv.index("--diag") + 1])
        d.mkdir(parents=True, exist_ok=True)
        for D in (32, 64, 96, 128):
            for k in ABLATIONS:
                (d / f"attn_w4i_d{D}_abl{k}.inc").write_text(render(D, k))
        return 0
    for D in (32, 64, 96, 128):
        text, out = render(D), out_path(D)
        if "--check" in sys.argv:
            if not out.exists() or out.read_text() != text:
                print(f"{out} is stale: run tools/gen_attn_w4i.py", file=sys.stderr)
                rc = 1
        else:
            out.write_text(text)
            print(f"wrote {out} ({len(text.splitlines())} lines)")
    return rc


if __name__ == "__main__":
    sys.exit(main())
