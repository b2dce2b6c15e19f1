"""CPU tests of the host-side mirrors (leetcuda_amd/host.py) against the reference-generated fixtures,
the entry-point tables against the reference's pybind files, and the attention sharding logic."""
import re
from pathlib import Path

import numpy as np
import pytest
import torch

from leetcuda_amd import host

REF = Path("/root/reference/kernels")

# kernels/hgemm/pybind/hgemm.cc:126-181 and kernels/flash-attn/pybind/flash_attn.cc:170-223, frozen here so
# the check also runs where /root/reference is absent (the GPU box).
HGEMM_NAMES = """hgemm_naive_f16 hgemm_sliced_k_f16 hgemm_t_8x8_sliced_k_f16x4 hgemm_t_8x8_sliced_k_f16x4_pack
hgemm_t_8x8_sliced_k_f16x4_bcf hgemm_t_8x8_sliced_k_f16x4_pack_bcf hgemm_t_8x8_sliced_k_f16x8_pack_bcf
hgemm_t_8x8_sliced_k_f16x8_pack_bcf_dbuf hgemm_t_8x8_sliced_k16_f16x8_pack_dbuf
hgemm_t_8x8_sliced_k16_f16x8_pack_dbuf_async hgemm_t_8x8_sliced_k32_f16x8_pack_dbuf
hgemm_t_8x8_sliced_k32_f16x8_pack_dbuf_async hgemm_t_16x8_sliced_k32_f16x8_pack_dbuf
hgemm_t_16x8_sliced_k32_f16x8_pack_dbuf_async init_cublas_handle destroy_cublas_handle
hgemm_cublas_tensor_op_nn hgemm_cublas_tensor_op_tn hgemm_wmma_m16n16k16_naive hgemm_wmma_m16n16k16_mma4x2
hgemm_wmma_m16n16k16_mma4x2_warp2x4 hgemm_wmma_m16n16k16_mma4x2_warp2x4_dbuf_async
hgemm_wmma_m32n8k16_mma2x4_warp2x4_dbuf_async hgemm_wmma_m16n16k16_mma4x2_warp2x4_stages
hgemm_wmma_m16n16k16_mma4x2_warp2x4_stages_dsmem hgemm_wmma_m16n16k16_mma4x2_warp4x4_stages_dsmem
hgemm_wmma_m16n16k16_mma4x4_warp4x4_stages_dsmem hgemm_mma_m16n8k16_naive hgemm_mma_m16n8k16_mma2x4_warp4x4
hgemm_mma_m16n8k16_mma2x4_warp4x4_stages hgemm_mma_m16n8k16_mma2x4_warp4x4_stages_dsmem
hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_x4
hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_rr hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_swizzle
hgemm_mma_m16n8k16_mma2x4_warp4x4_stages_dsmem_tn hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_tn_swizzle_x4
hgemm_mma_stages_block_swizzle_tn_cute""".split()
_P = "flash_attn_mma_stages_split_q"
ATTN_NAMES = (["flash_attn_mma_stages_split_kv", _P] +
              [f"{_P}_{s}" for s in ("shared_kv", "shared_qkv", "tiling_qk", "tiling_qkv", "shared_kv_acc_f32",
                                     "shared_qkv_acc_f32", "tiling_qk_acc_f32", "tiling_qkv_acc_f32")] +
              [f"{_P}_{f}_swizzle_{w}" for f in ("shared_kv", "shared_qkv", "tiling_qk", "tiling_qkv",
                                                 "tiling_qkv_acc_f32") for w in ("q", "qk", "qkv")] +
              ["flash_attn_cute", f"{_P}_shared_qkv_Os2g", f"{_P}_shared_kv_acc_f32_rr",
               f"{_P}_shared_qkv_acc_f32_rr"])


def _parse_bindings(path: Path):
    txt = path.read_text()
    return re.findall(r"TORCH_BINDING_COMMON_EXTENSION\(\s*(\w+)\s*\)", txt[txt.index("PYBIND11_MODULE"):])


@pytest.mark.skipif(not REF.exists(), reason="reference not mounted")
def test_frozen_name_lists_match_reference_pybind():
    assert _parse_bindings(REF / "hgemm/pybind/hgemm.cc") == HGEMM_NAMES
    assert _parse_bindings(REF / "flash-attn/pybind/flash_attn.cc") == ATTN_NAMES


def test_abi_entry_tables_cover_every_reference_export(built):
    from leetcuda_amd import capi
    he = capi.hgemm_entries()
    assert [n for n, _, _ in he] == HGEMM_NAMES and len(he) == 38
    ae = capi.attn_entries()
    assert [e[0] for e in ae] == ATTN_NAMES and len(ae) == 29
    lay = {n: l for n, l, _ in he}
    assert all((lay[n] == capi.LAYOUT_TN) == ("_tn" in n) for n in HGEMM_NAMES)
    nargs = {n: a for n, _, a in he}
    assert nargs["init_cublas_handle"] == 0 and nargs["hgemm_naive_f16"] == 3
    assert nargs["hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem"] == 6
    info = {e[0]: e[1:] for e in ae}
    vt = [n for n in ATTN_NAMES if info[n][1]]
    assert vt == [f"{_P}_shared_kv_swizzle_qkv", f"{_P}_shared_qkv_swizzle_qkv", f"{_P}_tiling_qk_swizzle_qkv"]
    assert info["flash_attn_cute"][5] == 4 and info[_P][5] == 5
    assert info[_P][3:5] == (128, 128) and info[f"{_P}_shared_qkv"][3:5] == (128, 256)
    assert info[f"{_P}_tiling_qkv"][3] == 1024


def test_host_helpers_match_reference(golden):
    for N, K, f, want in golden["host"]["swizzle_stride"]:
        assert host.make_block_swizzle_stride(N, K, f) == want
    for B, H, N, D, secs, om, want in golden["host"]["mha_tflops"]:
        assert host.get_mha_tflops(B, H, N, D, secs, om) == pytest.approx(want, rel=1e-12)
    x = torch.from_numpy(golden["colmajor"]["x"].view(np.float16))
    y = host.as_col_major(x)
    assert (y.numpy().view(np.uint16) == golden["colmajor"]["y"]).all()
    assert y.shape == x.shape and (y.reshape(x.shape[1], x.shape[0]) == x.t()).all()
    assert host.hgemm_tflops(8192, 8192, 8192, 1.0) == pytest.approx(1.099511627776)
    assert host.mha_matmul_flops(4, 32, 4096, 128) == 4 * 4 * 32 * 4096 ** 2 * 128


@pytest.mark.parametrize("total,world", [(32, 1), (32, 2), (32, 8), (5, 2), (7, 8), (128, 3)])
def test_shard_bounds_partition(total, world):
    spans = [host.shard_bounds(total, world, r) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == total
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    sizes = [hi - lo for lo, hi in spans]
    assert max(sizes) - min(sizes) <= 1


def test_attn_shard_prefers_batch_axis():
    assert host.attn_shard(32, 32, 8, 3) == (4, 32, 3 * 4 * 32)
    b, h, first = host.attn_shard(3, 8, 2, 1)   # 2 does not divide 3 -> flattened B*H axis
    assert (b, h, first) == (1, 12, 12)
    with pytest.raises(ValueError):
        host.shard_bounds(4, 2, 2)


def test_generated_hgemm_loops_are_current_and_well_formed():
    """hgemm_w4y's K loops are generated (tools/gen_hgemm_w4y.py): the committed .inc files must equal the generator's
    output, and every schedule must hold the invariants the hand-ordering relies on."""
    import importlib.util
    import re
    root = Path(__file__).resolve().parent.parent
    spec = importlib.util.spec_from_file_location("lc_gen_w4y", root / "tools" / "gen_hgemm_w4y.py")
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    for sched in range(gen.NSCHED):
        assert gen.out_path(sched).read_text() == gen.render(sched)
    assert (root / "leetcuda_amd" / "csrc" / "hgemm_w4y_loop_nn.inc").read_text() == gen.render_nn()
    # the K-loop stagger sequence (gen.STAGGER), interpreted: swp <- (swp + stg) mod kt for swp < kt, stg < kt; the compare and
    # the select that reads its SCC are adjacent in every generated stream
    def run_stagger(swp, kt, stg):
        reg = {"swp": swp, "kt": kt, "stg": stg, "t2off": 0xdead}
        scc = 0
        val = lambda tok: reg[tok.strip("%[]")] if tok.startswith("%[") else int(tok, 0)
        for ins in gen.STAGGER:
            op, args = ins.split(None, 1)
            d, *src = [a.strip() for a in args.split(",")]
            if op == "s_add_u32":
                reg[d.strip("%[]")] = (val(src[0]) + val(src[1])) & 0xffffffff
            elif op == "s_sub_u32":
                reg[d.strip("%[]")] = (val(src[0]) - val(src[1])) & 0xffffffff
            elif op == "s_cmp_ge_u32":
                scc = int(val(d) >= val(src[0]))
            elif op == "s_cselect_b32":
                reg[d.strip("%[]")] = val(src[0]) if scc else val(src[1])
            else:
                raise AssertionError(ins)
        return reg["swp"]
    for kt in (1, 2, 3, 16, 128):
        for stg in range(kt):
            assert [run_stagger(i, kt, stg) for i in range(kt)] == [(i + stg) % kt for i in range(kt)]
        bodies = [gen.gen_body(s) for s in range(3)]
    nn = gen.gen_nn()
    bodies.append(nn[nn.index(".Lw4y_loop_%=:"):-1])
    for body in bodies:
        mf = [l for l in body if l.startswith("v_mfma")]
        blocks = [l.split(",")[0] for l in mf]
        assert len(mf) == 128 and len(set(blocks)) == 64 and blocks[:64] == blocks[64:]   # 64 blocks x 2 k-steps
        assert sum(l.startswith("buffer_load_dwordx4") for l in body) == 16 and body.count("s_barrier") == 1
        # an M0 write is never directly followed by the LDS-DMA that uses it (one wait state needed), and the loop counter's
        # s_cmp is followed only by MFMAs (nothing that writes SCC) up to the branch
        assert sum(l.startswith("s_cmp_ge_u32") for l in body) == 1
        for a, b in zip(body, body[1:]):
            assert not (a.startswith("s_add_u32 m0") and b.startswith("buffer_load"))
            assert not a.startswith("s_cmp_ge_u32") or b.startswith("s_cselect_b32")
        tail = body[[i for i, l in enumerate(body) if l.startswith("s_cmp_lt_u32")][-1] + 1:]
        assert tail[-1].startswith("s_cbranch_scc1") and all(l.startswith("v_mfma") for l in tail[:-1])
        # the counted wait: 8 pieces (B of tile t+2) are issued between the loop top and the barrier's wait, 8 (A) behind it
        w = [i for i, l in enumerate(body) if "vmcnt(8)" in l][0]
        assert sum(l.startswith("buffer_load") for l in body[:w]) == 8
        assert body.index("s_barrier") > w
        # fragment registers stay inside the statement's literal range
        dst = [int(re.search(r"v\[(\d+):", l).group(1)) for l in body if l.startswith("ds_read")]
        assert min(dst) >= 128 and max(dst) <= 254
        # every ds_read of a k-step's fragments is issued before the wait that precedes their first MFMA
        assert all(not l.startswith("ds_read") for l in body[w - 2:w])


def test_generated_fp8_k128_loop_is_current_and_consumes_the_right_tiles():
    """gemm_fp8_w4k's K loop (tools/gen_gemm_fp8_w4k.py): committed .inc == generator output, and a replay of the generated stream as a
    dataflow machine — LDS ring slots hold tile numbers, ds_reads copy (tile, operand, fragment, half) tags into registers, DMA pieces
    overwrite slots, the barrier publishes — must show every MFMA of K tile t consuming exactly A fragment i / B fragment j of tile t,
    with every register half written by a read that an s_waitcnt lgkmcnt has retired (LDS reads return in order)."""
    import importlib.util
    import re
    root = Path(__file__).resolve().parent.parent
    spec = importlib.util.spec_from_file_location("lc_gen_w4k", root / "tools" / "gen_gemm_fp8_w4k.py")
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    for text, out in gen.outputs():
        assert out.read_text() == text, out
    for mx in (False, True):
        L = gen.gen(mx)
        for KT in (1, 2, 3, 4, 7):
            # SGPR / VGPR state.  LDS slot addresses: A ring 0x0 / 0x8000, B ring 0x10000 / 0x18000 / 0x20000 relative to a0 = 0.
            sg = {"kt": KT, "stg": 0, "a0": 0, "wv": 0, "blk": 0}
            slot_tile = {0x0: 0, 0x8000: 1, 0x10000: 0, 0x18000: 1, 0x20000: None}   # what the prologue staged
            pending_dma = []          # (slot address, tile) issued, not yet published by a vmcnt wait
            vaddr = {}                # address VGPR -> (slot address, register half)
            reg = {}                  # fragment register base -> {half: (tile, op, frag) or None}
            lds_q = []                # in-flight reads: (base, half, value)
            scale_reg, scale_q, m0 = {}, [], None
            dirty = set()             # ring slots read since the last barrier (another wave may still be reading them)
            mf_seen, pc, scc, steps = [], 0, 0, 0
            labels = {l[:-1]: i for i, l in enumerate(L) if l.endswith(":")}
            sval = lambda tok: sg[tok.strip("%[]")] if tok.startswith("%[") else int(tok, 0)
            while pc < len(L):
                ins = L[pc]
                pc += 1
                steps += 1
                assert steps < 20000
                if ins.endswith(":"):
                    continue
                op, _, rest = ins.partition(" ")
                args = [a.strip() for a in rest.split(",")]
                if op in ("s_mov_b32",):
                    sg[args[0].strip("%[]")] = sval(args[1])
                elif op in ("s_add_u32", "s_sub_u32", "s_min_u32", "s_lshl_b32"):
                    x, y = sval(args[1]), sval(args[2])
                    v = {"s_add_u32": x + y, "s_sub_u32": x - y, "s_min_u32": min(x, y), "s_lshl_b32": x << y}[op] & 0xffffffff
                    if args[0] == "m0":
                        m0 = v
                    else:
                        sg[args[0].strip("%[]")] = v
                elif op == "s_cmp_ge_u32":
                    scc = int(sval(args[0]) >= sval(args[1]))
                elif op == "s_cmp_lt_u32":
                    scc = int(sval(args[0]) < sval(args[1]))
                elif op == "s_cselect_b32":
                    sg[args[0].strip("%[]")] = sval(args[1]) if scc else sval(args[2])
                elif op == "v_add_u32_e32":
                    vaddr[args[0]] = (sval(args[1]), {"ar0": ("a", 0), "ar1": ("a", 1), "br0": ("b", 0), "br1": ("b", 1)}[args[2].strip("%[]")])
                elif op == "ds_read_b128":
                    m = re.match(r"v\[(\d+):(\d+)\]", args[0])
                    lo = int(m.group(1))
                    base, half = lo & ~7, (lo >> 2) & 1
                    an, off = args[1].split(" offset:")
                    slot, (which, h) = vaddr[an]
                    assert h == half
                    # the slot holds published data and no LDS-DMA into it is in flight or unpublished
                    assert not isinstance(slot_tile[slot], tuple) and all(sl != slot for sl, _ in pending_dma), (ins, hex(slot))
                    dirty.add(slot)
                    lds_q.append((base, half, (slot_tile[slot], which, int(off) // 2048)))
                    reg.setdefault(base, {})[half] = None        # in flight: unusable until retired
                elif op == "buffer_load_dwordx4":
                    is_a = args[1].strip("%[]") == "ra"
                    slot = m0 & ~0x7fff
                    assert (slot < 0x10000) == is_a, (ins, hex(m0))
                    assert slot not in dirty, (ins, hex(slot))     # every wave is past its reads of the slot (barrier since the last one)
                    pending_dma.append((slot, sg["soff"] >> 7))
                elif op == "buffer_load_dwordx2":
                    lo = int(re.match(r"v\[(\d+):", args[0]).group(1))
                    scale_q.append((lo, sg["s1off"] >> 9))
                elif op == "s_waitcnt":
                    m = re.search(r"lgkmcnt\((\d+)\)", ins)
                    if m:
                        while len(lds_q) > int(m.group(1)):
                            base, half, val = lds_q.pop(0)
                            reg[base][half] = val
                    m = re.search(r"vmcnt\((\d+)\)", ins)
                    if m:
                        keep = int(m.group(1))
                        assert len(scale_q) == 0 or keep <= 8      # scale loads are older than the 8 pieces that may stay in flight
                        for lo, t in scale_q:
                            scale_reg[lo] = t
                        scale_q.clear()
                        done, pending_dma[:] = pending_dma[:len(pending_dma) - keep], pending_dma[len(pending_dma) - keep:]
                        for slot, t in done:
                            slot_tile[slot] = ("landed", t)
                elif op == "s_barrier":
                    assert not lds_q                               # reads retired before the barrier: the slots they read are free behind it
                    dirty.clear()
                    for k, v in slot_tile.items():
                        if isinstance(v, tuple):
                            slot_tile[k] = v[1]
                elif op.startswith("v_mfma"):
                    m = re.match(r"a\[(\d+):", args[0])
                    blk = int(m.group(1)) // 4
                    i, j = blk >> 3, blk & 7
                    fb_, fa_ = int(re.match(r"v\[(\d+):", args[1]).group(1)), int(re.match(r"v\[(\d+):", args[2]).group(1))
                    t = len(mf_seen) // 64
                    want_t = min(t, KT - 1)
                    assert reg[fa_] == {0: (want_t, "a", i), 1: (want_t, "a", i)}, (KT, t, i, j, reg[fa_])
                    assert reg[fb_] == {0: (want_t, "b", j), 1: (want_t, "b", j)}, (KT, t, i, j, reg[fb_])
                    if mx:
                        sb_, sa_ = int(args[4][1:]), int(args[5].split()[0][1:])
                        assert scale_reg[sa_ & ~1] == want_t and scale_reg[sb_ & ~1] == want_t
                        assert (sa_ & 1) == (i >> 2) and (sb_ & 3) == 2 + (j >> 2)
                        sel = re.search(r"op_sel:\[(\d),(\d),0\] op_sel_hi:\[(\d),(\d),0\]", ins)
                        assert int(sel.group(1)) + 2 * int(sel.group(3)) == (j & 3) and int(sel.group(2)) + 2 * int(sel.group(4)) == (i & 3)
                    mf_seen.append((i, j))
                elif op == "s_cbranch_scc0":
                    if not scc:
                        pc = labels[args[0]]
                elif op == "s_cbranch_scc1":
                    if scc:
                        pc = labels[args[0]]
                else:
                    raise AssertionError(ins)
            assert len(mf_seen) == 64 * KT
            for t in range(KT):
                assert sorted(mf_seen[64 * t:64 * t + 64]) == [(i, j) for i in range(8) for j in range(8)]
            assert not pending_dma and not lds_q
        # stream hygiene, as for the fp16 loops
        for a, b in zip(L, L[1:]):
            assert not (a.startswith("s_add_u32 m0") and b.startswith("buffer_load"))
            assert not a.startswith("s_cmp_ge_u32") or b.startswith("s_cselect_b32")
        for k, l in enumerate(L):
            if l.startswith("s_cmp_lt_u32"):
                nxt = next(x for x in L[k + 1:] if not x.startswith("v_mfma") and not x.startswith("s_waitcnt"))
                assert nxt.startswith("s_cbranch")


def test_generated_bigd7_statements_are_current_and_pipeline_the_softmax():
    """attn_bigd7's statements on the pinned score registers (tools/gen_attn_bigd7.py): committed .inc == generator output; the six P·V
    statements with fillers cover every score element v[208 + e], e = 0 .. 31, exactly once with v_fma -> v_exp -> v_add one MFMA gap
    apart (a transcendental's result is never read by the next VALU instruction), on the row sum / maximum operand of the element's
    query block; the counted lgkmcnt waits are (6, 6) while transpose reads are re-issued and (6, 4) / (2, 0) on the last step."""
    import importlib.util
    import re
    root = Path(__file__).resolve().parent.parent
    spec = importlib.util.spec_from_file_location("lc_gen_bigd7", root / "tools" / "gen_attn_bigd7.py")
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    assert gen.OUT.read_text() == gen.render()
    seen = []
    for x in range(2, 8):
        body, (ta, tb) = gen.gen_pv(x, True, False)
        text = "\n".join(body)
        half = text[:text.index("if constexpr (!BF16)")]
        ins = re.findall(r'"([^"]+?)\\n\\t"', half)
        mf = [i for i, l in enumerate(ins) if l.startswith("v_mfma")]
        assert len(mf) == 8
        gap_of = lambda i: sum(1 for m in mf if m < i) - 1          # index of the MFMA an instruction follows
        for e in gen.elements(x):
            reg = f"v[{208 + e}]"
            f = next(i for i, l in enumerate(ins) if l.startswith(f"v_fma_f32 {reg},"))
            xx = next(i for i, l in enumerate(ins) if l.startswith(f"v_exp_f32 {reg},"))
            a = next(i for i, l in enumerate(ins) if l.startswith("v_add_f32") and l.endswith(reg))
            assert gap_of(f) + 1 == gap_of(xx) and gap_of(xx) + 1 == gap_of(a)
            first = (e >> 2) == ta
            assert ins[f].endswith("-%13" if first else "-%14") and ins[a].startswith("v_add_f32 %4" if first else "v_add_f32 %5")
            assert not ins[xx + 1].endswith(reg)                      # the instruction behind the v_exp does not read its result
            seen.append(e)
        waits = [int(w) for w in re.findall(r"s_waitcnt lgkmcnt\((\d)\)", "\n".join(ins))]
        assert waits == ([6, 6] if x < 6 else ([6, 4] if x == 6 else [2, 0]))
        assert sum(l.startswith("ds_read_b64_tr_b16") for l in ins) == (4 if x < 6 else 0)
        # V as [B,H,D,N]: one ds_read_b128 per fragment, waits on 4 outstanding reads; same fillers
        bodyt, _ = gen.gen_pv(x, True, True)
        textt = "\n".join(bodyt)
        inst = re.findall(r'"([^"]+?)\\n\\t"', textt[:textt.index("if constexpr (!BF16)")])
        waitst = [int(w) for w in re.findall(r"s_waitcnt lgkmcnt\((\d)\)", "\n".join(inst))]
        assert waitst == ([3, 3] if x < 6 else ([3, 2] if x == 6 else [1, 0]))
        assert sum(l.startswith("ds_read_b128") for l in inst) == (2 if x < 6 else 0) and not any("tr_b16" in l for l in inst)
        assert [l for l in inst if l.startswith(("v_fma", "v_exp", "v_add"))] == [l for l in ins if l.startswith(("v_fma", "v_exp", "v_add"))]
    assert sorted(seen) == list(range(32))
    # statement 1 carries the running-maximum check: every one of the 32 score registers is read exactly once, per query block through
    # three v_max3 + one v_max, the excess against that block's maximum operand
    for vt in (False, True):
        text = "\n".join(gen.gen_pv_check(vt))
        ins = re.findall(r'"([^"]+?)\\n\\t"', text[:text.index("if constexpr (!BF16)")])
        regs = [int(r) for l in ins if l.startswith(("v_max3", "v_max_f32 %3")) for r in re.findall(r"v\[(\d+)\]", l)]
        assert sorted(regs) == list(range(208, 240))
        fmas = [l for l in ins if l.startswith("v_fma_f32 %3")]
        assert [l.split("-")[-1] for l in fmas] == ["%11", "%12", "%13", "%14"]
        assert sum(l.startswith("v_mfma") for l in ins) == 8


def test_bench_traffic_keys_exist_in_committed_pmc_summary(built):
    """bench.py labels its roofline rows with the kernel name the DISPATCHER reports (lc_*_kernel_name) and looks the
    fabric bytes of that kernel up in profiles/latest_pmc.json (tools/summarize_prof.py, separate rocprofv3 --pmc
    passes).  The names must be in the form summarize_prof.py derives from the mangled symbols, and a key that is
    present must be what bench.py reports; a profile that predates a kernel rename degrades to traffic = null."""
    import importlib.util
    import json
    from leetcuda_amd import capi
    root = Path(__file__).resolve().parent.parent
    spec = importlib.util.spec_from_file_location("lc_bench", root / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    spec2 = importlib.util.spec_from_file_location("lc_sumprof", root / "tools" / "summarize_prof.py")
    sump = importlib.util.module_from_spec(spec2)
    spec2.loader.exec_module(sump)
    # dispatcher names == demangled symbol names of the kernels actually in the library
    assert capi.hgemm_kernel_name(8192, 8192, 8192, capi.LAYOUT_TN) == "hgemm_w4y_kernel<false,2>"
    assert capi.hgemm_kernel_name(8192, 8192, 8192, capi.LAYOUT_NN) == "hgemm_w4y_kernel<true,1>"
    assert capi.hgemm_kernel_name(8192, 8192, 8192, capi.LAYOUT_NN, capi.HGEMM_MFMA256P2) == "hgemm_pingpong2_kernel<true,false>"
    assert capi.hgemm_kernel_name(1024, 1024, 1024, capi.LAYOUT_NN) == "hgemm_mid_kernel<true,1,2,3>"       # (round 6: 64 x 128 tiles, one round)
    assert capi.hgemm_kernel_name(512, 512, 512, capi.LAYOUT_NN) == "hgemm_mfma128_kernel<true,2>"
    assert capi.hgemm_kernel_name(1000, 1024, 1024, capi.LAYOUT_NN) == "hgemm_mid_edge_kernel<true,1,2,3>"       # (late round 6: clamped tiles of the mid-size kernel; K % 32: hgemm_edge_kernel; K % 8: hgemm_generic_kernel)
    assert capi.hgemm_kernel_name(1000, 1024, 200, capi.LAYOUT_NN) == "hgemm_edge_kernel<true>"
    assert capi.hgemm_kernel_name(1000, 1024, 1032, capi.LAYOUT_NN) == "hgemm_pad_copy_kernel + hgemm_mid_edge_kernel<true,1,2,3>"   # (K % 32 != 0, K >= 256: zero-padded copies + the tuned kernels)
    assert capi.hgemm_kernel_name(1000, 1024, 1028, capi.LAYOUT_NN) == "hgemm_generic_kernel<true>"
    assert capi.attn_kernel_name(4096, 128) == "attn_fwd_w4u_kernel<128,false,1>"            # config 3: the persistent workgroup, static walk
    assert capi.attn_kernel_name(8192, 128) == "attn_fwd_w4u_kernel<128,false,0>"            # config 4: one block per workgroup (the dispatcher balances 16+ blocks per CU best)
    assert capi.attn_kernel_name(4096, 128, True) == "attn_fwd_w4u_kernel<128,true,1>"       # V handed over as [B,H,D,N]: the same kernel
    assert capi.attn_kernel_name(4096 + 128, 128) == "attn_fwd_w4u_kernel<128,false,0>"    # N % 256 == 128 from N = 1152 on: the merged-phase kernel, last block half real
    assert capi.attn_kernel_name(896, 64) == "attn_fwd_kernel<64,4,false,0>"               # ... below: lock-step (a wasted half block would cost more)
    assert capi.attn_kernel_name(4096 + 64, 128) == "attn_fwd_w4u_kernel<128,false,0>"     # ... and any other N % 64 == 0 (the last block a quarter real)
    assert capi.attn_kernel_name(1024 + 64, 128) == "attn_fwd_kernel<128,2,false,0>"       # N % 128 != 0 below 1152: lock-step
    assert capi.attn_kernel_name(192, 64, True) == "attn_fwd_kernel<64,2,true,0>"
    assert capi.attn_kernel_name(8192, 64) == "attn_fwd_w4u_kernel<64,false,1>"               # the reference's published shapes: D = 64 keeps the persistent walk up to N = 8192 (+ 1.4 %)
    assert capi.attn_kernel_name(8192, 64, True) == "attn_fwd_w4u_kernel<64,true,1>"
    assert capi.attn_kernel_name(16384, 64) == "attn_fwd_w4u_kernel<64,false,0>"
    assert capi.attn_kernel_name(8192, 96, True) == "attn_fwd_kernel<96,8,true,0>"            # D = 96 / 32 with V transposed: lock-step
    assert capi.attn_kernel_name(8192, 96) == "attn_fwd_w4i_kernel<96,1>"                     # only the generated kernel has a D = 96 instantiation
    assert capi.attn_kernel_name(8192 + 64, 96) == "attn_fwd_kernel<96,2,false,0>"            # N % 256 != 0: lock-step
    assert capi.attn_kernel_name(8192, 32) == "attn_fwd_w4i_kernel<32,1>"
    assert capi.attn_kernel_name(8192, 512, False, True) == "attn_fwd_bigd6_kernel<true>"             # config 5a: D = 512 on the 16x16x32 MFMA
    assert capi.attn_kernel_name(192, 512, False, False).startswith("attn_fwd_bigd_kernel<512,")     # N % 128 != 0
    assert sump.short("_ZN2lc16hgemm_w4b_kernelILb0ELb1ELi0EEEvPKDF16_S2_PDF16_iiiiii") == \
        "hgemm_w4b_kernel<false,true,0>"
    assert sump.short("_ZN2lc19attn_fwd_w4u_kernelILi128ELb1ELi2EEEvPKDF16_S2_S2_PDF16_iifiii") == "attn_fwd_w4u_kernel<128,true,2>"
    pmc = json.loads((root / "profiles" / "latest_pmc.json").read_text())
    for key, wl, other in ((capi.hgemm_kernel_name(8192, 8192, 8192, capi.LAYOUT_TN), "hgemm_8192", "hgemm_4096"),
                           (capi.attn_kernel_name(4096, 128), "attn_cfg3", "attn_cfg4"),
                           (capi.attn_kernel_name(8192, 512), "attn_d512_fp16", "attn_d512_bf16")):
        if key in pmc and "hbm_bytes_per_launch" in pmc[key]:
            assert bench.pmc_traffic(key) == pmc[key]["hbm_bytes_per_launch"] > 0
            r = bench.roofline(key, 1.0e12, 4.0e8, 1.0, workload=wl)
            assert r["traffic"] == pmc[key]["hbm_bytes_per_launch"] and "profiles/" in r["traffic_source"]
            assert r["traffic_ratio"] == pytest.approx(r["traffic"] / 4.0e8)
            # a per-launch byte count is only printed next to the shape it was measured on
            assert bench.roofline(key, 1.0e12, 4.0e8, 1.0, workload=other)["traffic"] is None
            assert bench.roofline(key, 1.0e12, 4.0e8, 1.0)["traffic"] is None
        else:
            assert bench.pmc_traffic(key) is None and bench.roofline(key, 1.0e12, 4.0e8, 1.0, workload=wl)["traffic_source"] is None
    # the large-head-dim "2.5x / 1.5x traffic" of the round-4 verdict is the tile model, not re-reads: 32 CUs of an XCD share one pass over a
    # head's K / V, a head needs N / rows_per_workgroup / 32 passes (bench.attn_traffic_model) — the committed counters to < 2 % (D = 1024 / 512: < 0.1 %)
    for key, models in (("attn_fwd_bigd4_kernel<8>", (bench.attn_traffic_model(48, 8192, 1024, 64),                       # round 4's XCD-contiguous order
                                                      bench.attn_traffic_model(48, 8192, 1024, 64, round_robin=True))),    # round 5's default
                        ("attn_fwd_bigd6_kernel<false>", (bench.attn_traffic_model(48, 8192, 512, 128),)),
                        ("attn_fwd_bigd7_kernel<false,false>", (bench.attn_traffic_model(48, 8192, 256, 256),))):
        if key in pmc and "hbm_bytes_per_launch" in pmc[key]:
            assert any(pmc[key]["hbm_bytes_per_launch"] == pytest.approx(m, rel=0.03) for m in models), key     # (D = 256: 1.6 ... 2.1 % above the model over two rounds of counters)
    assert bench.attn_traffic_model(48, 8192, 1024, 64, round_robin=True) == pytest.approx(14.5e9, rel=0.005)


def test_steady_state_loops_keep_their_instruction_mix(built):
    """tools/isa_count.py on the audited device assembly of the shipped kernels: the generated GEMM
    loop is exactly its 128 MFMAs + 32 LDS reads + 16 DMA pieces with no nop and (almost) no VALU; the attention loops keep
    their MFMA count per tile and stay near 3 VALU per score element.  A compiler or flag change that moves these shows up
    here before it shows up as TFLOP/s."""
    import sys
    from collections import Counter
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
    import isa_count as ic
    obj = built["abi"].parent / "obj"

    def mix(unit, rx, label=None):
        name, lines = ic.kernel_lines(obj / unit, rx)
        assert lines, rx
        body, _ = ic.loop_mix(lines, label)
        c = Counter(ic.classify(x.split()[0]) for x in body)
        return c, c["valu"] + c["valu_trans"] + c["valu_accvgpr"]

    g, gv = mix("tu_w4.s", r"hgemm_w4y_kernelILb0ELi1E", r"w4y_loop")   # the K loop (the kernel's outer loop is the persistent tile walk)
    assert (g["mfma"], g["lds"], g["vmem"], g["s_barrier"], g["s_nop"]) == (128, 32, 16, 1, 0), g
    assert gv <= 8, g
    g64, g64v = mix("tu_attn_w4u_d64.s", r"attn_fwd_w4u_kernelILi64ELb0ELi0")   # D = 64: one tile = 64 MFMAs for the same 64 score elements
    assert g64["mfma"] == 64 and g64["valu_trans"] == 64 and g64["s_barrier"] == 1, g64
    assert g64v / 64 <= 3.0, (g64v, g64)
    for unit, rx in (("tu_attn_w4u_d128.s", r"attn_fwd_w4u_kernelILi128ELb0ELi0"), ("tu_attn_w4u_d128t.s", r"attn_fwd_w4u_kernelILi128ELb1ELi0")):
        n, nv = mix(unit, rx)                                        # one 64-key tile: 64 score elements per lane, either V layout
        assert n["mfma"] == 128 and n["valu_trans"] == 64 and n["s_barrier"] == 1 and n["lds"] == 48, n
        assert nv / 64 <= 3.2, (nv, n)
    b, _ = mix("tu_attn_big.s", r"attn_fwd_bigd2_kernelILi512ELb0")  # two 64-key tiles per loop iteration
    assert b["mfma"] == 256 and b["s_barrier"] == 4 and b["valu_trans"] == 64, b


def test_large_head_dim_kernel_names(built):
    """Round 4: D = 1024 runs the pair kernel (attn_bigd4.hip) when N % 64 == 0, D = 256 with V as [B,H,D,N] the full-width kernel's
    V-transposed instantiation; the round-1 column-split kernel keeps ragged N, D = 512 with V transposed and the cross-check knob."""
    from leetcuda_amd import capi
    capi.load()
    assert capi.attn_kernel_name(8192, 1024) == "attn_fwd_bigd4_kernel<8>"
    assert capi.attn_kernel_name(64, 1024) == "attn_fwd_bigd4_kernel<8>"
    assert capi.attn_kernel_name(8192, 256, True) == "attn_fwd_bigd7_kernel<false,true>"          # V as [B,H,D,N]
    assert capi.attn_kernel_name(384, 256, True) == "attn_fwd_bigd2_kernel<256,false,true>"
    assert capi.attn_kernel_name(8192, 256) == "attn_fwd_bigd7_kernel<false,false>"
    assert capi.attn_kernel_name(8192, 256, False, True) == "attn_fwd_bigd7_kernel<true,false>"
    assert capi.attn_kernel_name(384, 256) == "attn_fwd_bigd2_kernel<256,false,false>"      # N % 256 == 128 below 1152: the 32-rows-per-wave kernel
    assert capi.attn_kernel_name(4224, 256) == "attn_fwd_bigd7_kernel<false,false>"          # ... from 1152 on: the ring kernel, last block half real
    # D = 256 by grid size (lc_attn_kernel_name_bh; the rounds rule of use_bigd7 on the 256 CUs this test assumes when no GPU is present):
    # 8 heads x 4 blocks = 32 workgroups and 16 x 8 = 128 go to the 128-row kernel, 96 x 2 = 192 and everything from 256 up to the ring kernel
    if capi.attn_kernel_name(1024, 256, bh=64) == "attn_fwd_bigd7_kernel<false,false>":     # (a 256-CU device or the no-device default)
        assert capi.attn_kernel_name(1024, 256, bh=8) == "attn_fwd_bigd2_kernel<256,false,false>"
        assert capi.attn_kernel_name(2048, 256, bh=16) == "attn_fwd_bigd2_kernel<256,false,false>"
        assert capi.attn_kernel_name(1024, 256, True, bh=8) == "attn_fwd_bigd2_kernel<256,false,true>"
        assert capi.attn_kernel_name(512, 256, bh=96) == "attn_fwd_bigd7_kernel<false,false>"
        assert capi.attn_kernel_name(8192, 256, bh=48) == "attn_fwd_bigd7_kernel<false,false>"
    assert capi.attn_kernel_name(8192, 512, True).startswith("attn_fwd_bigd_kernel<512,")
    assert capi.attn_kernel_name(192, 256, True).startswith("attn_fwd_bigd_kernel<256,")
    assert capi.attn_kernel_name(8192, 512) == "attn_fwd_bigd6_kernel<false>"
    capi.tune("attn_d512", 3)
    try:
        assert capi.attn_kernel_name(8192, 512) == "attn_fwd_bigd2_kernel<512,false,false>"     # the other MFMA shape: the cross-check
        assert capi.attn_kernel_name(8192, 256) == "attn_fwd_bigd2_kernel<256,false,false>"
        assert capi.attn_kernel_name(8192, 256, True) == "attn_fwd_bigd2_kernel<256,false,true>"
    finally:
        capi.tune("attn_d512", 0)
    capi.tune("attn_d512", 4)      # auto, but attn_bigd7 on any grid: nothing else changes
    try:
        assert capi.attn_kernel_name(8192, 256) == "attn_fwd_bigd7_kernel<false,false>"
        assert capi.attn_kernel_name(8192, 512) == "attn_fwd_bigd6_kernel<false>"
        assert capi.attn_kernel_name(8192, 1024) == "attn_fwd_bigd4_kernel<8>"
    finally:
        capi.tune("attn_d512", 0)
    capi.tune("attn_d512", 1)
    try:
        assert capi.attn_kernel_name(8192, 1024).startswith("attn_fwd_bigd_kernel<1024,")
        assert capi.attn_kernel_name(8192, 256, True).startswith("attn_fwd_bigd_kernel<256,")
    finally:
        capi.tune("attn_d512", 0)


def test_bench_compact_attention_scalars_and_traffic_models():
    """bench.py's second headline is emitted as FLAT scalars inside `roofline` (the driver's parsed record drops nested objects): the helper
    that builds them, and the two traffic models (GEMM tiles, attention passes) at the shapes DESIGN.md quotes."""
    import sys
    root = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root))
    import bench
    blk = {"value": 1250.0, "ms_per_step": 0.88, "steps": 10,
           "roofline": {"peak": 2500.0, "kernel": "attn_fwd_w4u_kernel<128,false,1>", "kernel_ms": 0.87, "frac": 0.5, "traffic_ratio": 1.03}}
    c = bench.compact_attn(blk, 2)
    assert c["tflops"] == 1250.0 and c["frac"] == pytest.approx(1250.0 / 5000.0) and c["n_ranks"] == 2 and c["kernel_ms"] == 0.87
    assert all(not isinstance(v, (dict, list)) for v in c.values())
    assert bench.compact_attn({"skipped": "x"}, 1) is None and bench.compact_attn(None, 1) is None
    assert bench.hgemm_traffic_model(8192, 8192, 8192) == 1744830464
    assert bench.attn_traffic_model(48, 8192, 1024, 64) == pytest.approx(8.053e9, rel=1e-3)                     # 4 passes, XCD-contiguous
    assert bench.attn_traffic_model(48, 8192, 1024, 64, round_robin=True) == pytest.approx(14.4955e9, rel=1e-4)   # 8 passes, round-robin
    assert bench.attn_traffic_model(48, 8192, 512, 128) == pytest.approx(2.4159e9, rel=1e-4)
    assert bench.attn_traffic_model(128, 4096, 128, 256) == 4 * 128 * 4096 * 128 * 2                            # config 3: one pass = algorithmic
