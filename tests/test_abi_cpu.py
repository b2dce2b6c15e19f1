"""CPU tests of the boundary: the C-ABI library loads, exports every symbol include/lc_abi.h declares,
rejects bad arguments without touching a GPU, and the two drop-in PyTorch modules export the reference
names and refuse CPU tensors loudly (there is no CPU path)."""
import ctypes as C
import re
import subprocess
import sys
from pathlib import Path

import pytest
import torch

from tests.test_host import ATTN_NAMES, HGEMM_NAMES

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols(header="lc_abi.h"):
    txt = (ROOT / "include" / header).read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lc_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_all_exported(built):
    from leetcuda_amd import capi
    lib = capi.load()
    declared = _declared_symbols()
    assert declared == sorted(capi.SYMBOLS)          # the ctypes table mirrors the header exactly
    for s in declared:
        assert getattr(lib, s) is not None
    nm = subprocess.run(["nm", "-D", "--defined-only", str(built["abi"])], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (lc_\w+)", nm))
    assert set(declared) <= exported
    assert lib.lc_abi_version() == 2
    # the drop-in boundary carries no diagnosis entry points (round-1 verdict): probes live in liblc_diag.so
    assert not [s for s in exported if s.startswith("lc_probe")]
    info, diag = capi.build_info()
    assert "gfx950" in info and diag is False          # the tests refuse a LC_DIAG=1 library
    capi.require_production()
    # diagnosis keys are rejected by a production library; selection keys validate their values
    for key in (b"attn_ablate", b"w4_abl", b"hgemm_stamps", b"no_such_key"):
        assert lib.lc_tune_set(key, 1) == capi.LC_ERR_ARG
    assert lib.lc_tune_set(b"attn_nw", 16) == capi.LC_ERR_ARG       # retired ping-pong schedule
    assert lib.lc_tune_set(b"attn_nw", 128) == capi.LC_ERR_ARG      # retired round-1 4-wave kernel
    for retired in (256, 260, 516):                                 # round 4: attn_w4m.hip / attn_w8g.hip retired
        assert lib.lc_tune_set(b"attn_nw", retired) == capi.LC_ERR_ARG
    assert lib.lc_tune_set(b"attn_nw", 517) == capi.LC_OK and lib.lc_tune_set(b"attn_nw", 512) == capi.LC_OK
    assert lib.lc_tune_set(b"attn_nw", 0) == capi.LC_OK
    assert lib.lc_tune_set(b"hgemm_auto", 7) == capi.LC_ERR_ARG     # retired 2-slot w4 kernel
    assert lib.lc_tune_set(b"hgemm_auto", capi.HGEMM_MFMA256W4C) == capi.LC_OK
    assert lib.lc_tune_set(b"hgemm_auto", capi.HGEMM_MFMA256W4Y) == capi.LC_OK     # (back to the default)


def test_knob_registry_reports_defaults_and_validates(built):
    """lc_tune_get / lc_tune_count / lc_tune_key: every knob of a freshly loaded library sits at its default; the K-loop stagger's
    mask field is 7 bits and 1 << 27 alone means "off" (round-3 advisor: a mask >= 128 used to switch the stagger off silently)."""
    from leetcuda_amd import capi
    items = capi.tune_items()
    assert {"attn_nw", "attn_d512", "fp8_mx", "hgemm_persist", "hgemm_stagger", "hgemm_tail", "hgemm_raster", "hgemm_auto",
            "w4y_sched", "attn_w4i_sched"} <= set(items)
    assert not {"attn_ablate", "w4_abl", "hgemm_stamps"} & set(items)           # diagnosis keys: not in a production library
    assert all(cur == dflt for cur, dflt in items.values()), items
    assert items["hgemm_persist"] == (1, 1) and items["hgemm_tail"] == (1, 1) and items["hgemm_stagger"] == (0, 0)
    lib = capi.load()
    off = 1 << 27
    assert lib.lc_tune_set(b"hgemm_stagger", off) == capi.LC_OK
    assert capi.tune_get("hgemm_stagger") == (off, 0)
    for bad in (off | 1, off | 7 << 20, 129 << 20, 255 << 20, 1 << 28, -1):     # (128 << 20 IS the off value)
        assert lib.lc_tune_set(b"hgemm_stagger", bad) == capi.LC_ERR_ARG, bad
    assert capi.tune_get("hgemm_stagger") == (off, 0)                           # a refused value changes nothing
    assert lib.lc_tune_set(b"hgemm_stagger", 15 | 15 << 4 | 15 << 8 | 255 << 12 | 127 << 20) == capi.LC_OK
    assert lib.lc_tune_set(b"hgemm_stagger", 0) == capi.LC_OK
    assert lib.lc_tune_get(b"no_such_key", None, None) == capi.LC_ERR_ARG
    assert lib.lc_tune_key(-1) is None and lib.lc_tune_key(lib.lc_tune_count()) is None


def test_diag_library_exports_its_header(built):
    from leetcuda_amd import build, capi
    p = build.build_diag()
    declared = _declared_symbols("lc_diag.h")
    assert declared == sorted(capi.DIAG_SYMBOLS)
    nm = subprocess.run(["nm", "-D", "--defined-only", str(p)], capture_output=True, text=True).stdout
    assert set(declared) <= set(re.findall(r" T (lc_\w+)", nm))


def test_isa_audit_report_is_clean(built):
    """leetcuda_amd/build.py runs leetcuda_amd/isa_audit.py on every build and refuses to link on a violation; the
    report it leaves behind must cover the kernels that keep state in literal AGPRs / asm loads."""
    import json
    rep = json.loads((built["abi"].parent / "obj" / "isa_audit.json").read_text())
    names = " ".join(r["kernel"] for r in rep)
    for k in ("hgemm_w4b_kernel", "hgemm_w4x_kernel", "hgemm_w4y_kernel", "gemm_fp8_w4_kernel", "gemm_fp8_w4k_kernelILb0E", "gemm_fp8_w4k_kernelILb1E", "attn_fwd_w4u_kernelILi128ELb0ELi0",
              "attn_fwd_w4u_kernelILi128ELb1ELi2", "attn_fwd_w4u_kernelILi64ELb0ELi1", "attn_fwd_w4u_kernelILi64ELb1ELi0",
              "attn_fwd_w4u_kernelILi128ELb0ELi3", "attn_fwd_w4u_kernelILi64ELb1ELi3",            # split-KV (round 5)
              "attn_fwd_w4i_kernel", "attn_fwd_bigd2_kernel", "hgemm_pingpong2_kernel"):
        assert k in names, k
    assert all(r["scratch"] == 0 and not r["violations"] for r in rep)
    w4 = [r for r in rep if "hgemm_w4b_kernel" in r["kernel"] or "hgemm_w4y_kernel" in r["kernel"]]
    assert w4 and all(r["agpr"] == 256 and r["compiler_accvgpr"] == 0 for r in w4)


def test_isa_audit_detects_planted_hazards(tmp_path):
    from leetcuda_amd import isa_audit
    asm = """
\t.type\t_ZN2lc16hgemm_w4b_kernelILb0EEEvv,@function
_ZN2lc16hgemm_w4b_kernelILb0EEEvv:
\t;;#ASMSTART
\tds_read_b64_tr_b16 v[4:5], v9 offset:0
\t;;#ASMEND
\tv_add_u32_e32 v4, v4, v1
\t;;#ASMSTART
\ts_waitcnt lgkmcnt(0)
\t;;#ASMEND
\tv_accvgpr_write_b32 a17, v2
\tv_add_u32_e32 v5, v5, v1
.Lfunc_end0:
\t.amdhsa_kernel _ZN2lc16hgemm_w4b_kernelILb0EEEvv
\t\t.amdhsa_private_segment_fixed_size 64
\t.end_amdhsa_kernel
"""
    f = tmp_path / "planted.s"
    f.write_text(asm)
    reps, bad = isa_audit.audit_files([f])
    kinds = sorted(b.split()[0] for b in bad)
    assert kinds == ["R1", "R2", "R3", "R7"], bad      # v5 after the wait is fine; v4 before it is not (R3 and its counted twin R7)


def test_isa_audit_detects_early_read_of_asm_mfma_result(tmp_path):
    """Rule R5 (the round-2 reproducibility bug): hipcc scheduled `v_max` reads of the Sᵀ blocks directly
    behind the asm MFMA that writes them because the drain was a bare asm volatile.  A bare s_nop drain AFTER the read does
    not help; enough wait states BEFORE it do; MFMA -> MFMA accumulation chains are the hardware's business."""
    from leetcuda_amd import isa_audit
    head = "\t.type\t_ZN2lc19attn_fwd_w4u_kernelILi128ELb0ELi0EEEvv,@function\n_ZN2lc19attn_fwd_w4u_kernelILi128ELb0ELi0EEEvv:\n"
    tail = ".Lfunc_end0:\n"
    mfma = ("\t;;#ASMSTART\n\tv_mfma_f32_16x16x32_f16 v[2:5], a[0:3], a[4:7], v[2:5]\n\t;;#ASMEND\n"
            "\t;;#ASMSTART\n\tv_mfma_f32_16x16x32_f16 v[2:5], a[8:11], a[12:15], v[2:5]\n\t;;#ASMEND\n")
    early = head + mfma + "\ts_nop 0\n\tv_max_f32_e32 v1, v2, v3\n\t;;#ASMSTART\n\ts_nop 15\n\ts_nop 15\n\t;;#ASMEND\n" + tail
    late = head + mfma + "\t;;#ASMSTART\n\ts_nop 15\n\ts_nop 15\n\t;;#ASMEND\n\tv_max_f32_e32 v1, v2, v3\n" + tail
    (tmp_path / "early.s").write_text(early)
    (tmp_path / "late.s").write_text(late)
    _, bad = isa_audit.audit_files([tmp_path / "early.s"])
    assert len(bad) == 1 and bad[0].startswith("R5") and "v_max_f32_e32 v1, v2, v3" in bad[0], bad
    _, bad = isa_audit.audit_files([tmp_path / "late.s"])
    assert bad == [], bad


def test_isa_audit_replays_hazards_across_a_loop_back_edge(tmp_path):
    """the hazard state used to be replayed in file order only, so a hazard between the LAST instruction of
    a loop body and the FIRST of the next iteration was invisible.  A loop whose tail is an asm MFMA writing v[2:5] and whose
    head reads v2 with a VALU instruction is clean in file order (the head comes first) and wrong on the back edge; with the
    wait states in front of the branch it is clean on both."""
    from leetcuda_amd import isa_audit
    head = "\t.type\t_ZN2lc19attn_fwd_w4u_kernelILi128ELb0ELi0EEEvv,@function\n_ZN2lc19attn_fwd_w4u_kernelILi128ELb0ELi0EEEvv:\n"
    tail = ".Lfunc_end0:\n"
    loop = (".LBB0_1:\n\tv_max_f32_e32 v1, v2, v3\n\ts_nop 7\n"
            "\t;;#ASMSTART\n\tv_mfma_f32_16x16x32_f16 v[2:5], a[0:3], a[4:7], v[2:5]\n\t;;#ASMEND\n"
            "{pad}\ts_add_i32 s4, s4, -1\n\ts_cmp_lg_u32 s4, 0\n\ts_cbranch_scc1 .LBB0_1\n")
    (tmp_path / "bad.s").write_text(head + loop.format(pad="") + tail)
    (tmp_path / "good.s").write_text(head + loop.format(pad="\t;;#ASMSTART\n\ts_nop 15\n\t;;#ASMEND\n") + tail)
    _, bad = isa_audit.audit_files([tmp_path / "bad.s"])
    assert len(bad) == 1 and bad[0].startswith("R5") and "[loop back-edge]" in bad[0] and "v_max_f32_e32 v1, v2, v3" in bad[0], bad
    assert isa_audit.audit_files([tmp_path / "good.s"])[1] == []
    # the same for an asm LDS read left outstanding at the loop end (R3 / R7 on the back edge)
    loop2 = (".LBB0_2:\n\tv_add_u32_e32 v8, v6, v1\n\t;;#ASMSTART\n\ts_waitcnt lgkmcnt(0)\n\t;;#ASMEND\n"
             "\t;;#ASMSTART\n\tds_read_b64_tr_b16 v[6:7], v9 offset:0\n\t;;#ASMEND\n"
             "\ts_add_i32 s4, s4, -1\n\ts_cmp_lg_u32 s4, 0\n\ts_cbranch_scc1 .LBB0_2\n")
    (tmp_path / "bad2.s").write_text(head + loop2 + tail)
    _, bad = isa_audit.audit_files([tmp_path / "bad2.s"])
    assert bad and all("[loop back-edge]" in b for b in bad) and {b.split()[0] for b in bad} == {"R3", "R7"}, bad


def test_isa_audit_detects_valu_write_in_front_of_asm_mfma(tmp_path):
    """Rule R6: hipcc pads two wait states between a VALU write and an MFMA that reads the register only for its OWN
    MFMAs; a v_cvt_pk of the P fragment scheduled in front of the asm statement that consumes it must be caught."""
    from leetcuda_amd import isa_audit
    head = "\t.type\t_ZN2lc21attn_fwd_bigd2_kernelILi512ELb0EEEvv,@function\n_ZN2lc21attn_fwd_bigd2_kernelILi512ELb0EEEvv:\n"
    tail = ".Lfunc_end0:\n"
    mfma = "\t;;#ASMSTART\n{pad}\tv_mfma_f32_32x32x16_f16 a[0:15], v[240:243], v[112:115], a[0:15]\n\t;;#ASMEND\n"
    bad_s = head + "\tv_cvt_pk_f16_f32 v112, v112, v126\n" + mfma.format(pad="\ts_waitcnt lgkmcnt(6)\n") + tail
    good_s = head + "\tv_cvt_pk_f16_f32 v112, v112, v126\n" + mfma.format(pad="\ts_nop 1\n\ts_waitcnt lgkmcnt(6)\n") + tail
    other = head + "\tv_cvt_pk_f16_f32 v116, v116, v126\n" + mfma.format(pad="") + tail
    for name, text in (("bad.s", bad_s), ("good.s", good_s), ("other.s", other)):
        (tmp_path / name).write_text(text)
    _, bad = isa_audit.audit_files([tmp_path / "bad.s"])
    assert len(bad) == 1 and bad[0].startswith("R6") and "v112" in bad[0], bad
    assert isa_audit.audit_files([tmp_path / "good.s"])[1] == []
    assert isa_audit.audit_files([tmp_path / "other.s"])[1] == []


def test_isa_audit_checks_counted_lgkmcnt_waits(tmp_path):
    """Rule R7: LDS operations return in order; `s_waitcnt lgkmcnt(N)` retires all but the N youngest.  Eight transpose
    reads in fragment order, then the MFMA of fragment 0: lgkmcnt(6) is enough, lgkmcnt(7) is one read short; a younger
    LDS operation of hipcc's in between counts in the queue — it makes a counted wait retire MORE of the older reads."""
    from leetcuda_amd import isa_audit
    head = "\t.type\t_ZN2lc21attn_fwd_bigd2_kernelILi512ELb0EEEvv,@function\n_ZN2lc21attn_fwd_bigd2_kernelILi512ELb0EEEvv:\n"
    tail = ".Lfunc_end0:\n"
    reads = "\t;;#ASMSTART\n" + "".join(
        f"\tds_read_b64_tr_b16 v[{240 + 2 * i}:{241 + 2 * i}], v{100 + i // 2} offset:{4096 * (i & 1)}\n" for i in range(8)) + "\t;;#ASMEND\n"
    step = "\t;;#ASMSTART\n\ts_nop 1\n\ts_waitcnt lgkmcnt({n})\n\tv_mfma_f32_32x32x16_f16 a[0:15], v[240:243], v[112:115], a[0:15]\n\t;;#ASMEND\n"
    ok = head + reads + step.format(n=6) + tail
    short = head + reads + step.format(n=7) + tail
    shifted = head + reads + "\tds_read_b128 v[20:23], v9\n" + step.format(n=7) + tail     # 9 outstanding: 7 retires both of fragment 0
    for name, text in (("ok.s", ok), ("short.s", short), ("shifted.s", shifted)):
        (tmp_path / name).write_text(text)
    assert isa_audit.audit_files([tmp_path / "ok.s"])[1] == []
    assert isa_audit.audit_files([tmp_path / "shifted.s"])[1] == []
    _, bad = isa_audit.audit_files([tmp_path / "short.s"])
    assert len(bad) == 1 and bad[0].startswith("R7") and "v242" in bad[0], bad


def test_status_strings_and_argument_errors(built):
    from leetcuda_amd import capi
    lib = capi.load()
    assert capi.status_string(capi.LC_ERR_HEADDIM) == "headdim not support!"     # split_q.cu:793
    assert capi.status_string(capi.LC_ERR_SHAPE) == "Tensor size mismatch!"      # hgemm_mma_stage.cu:2055
    # argument validation happens before any HIP call
    assert lib.lc_hgemm_f16(None, None, None, 256, 256, 256, 0, 0, 2, 1, None) == capi.LC_ERR_ARG
    one = C.c_void_p(16)
    assert lib.lc_hgemm_f16(one, one, one, 256, 256, 256, 7, 0, 2, 1, None) == capi.LC_ERR_ARG
    for retired in (2, 5, 7, 8, 11, 18, 19, 31):       # round-1 experiment variants / out of range (14 = LC_HGEMM_MID, 15 = LC_HGEMM_EDGE, 16 = LC_HGEMM_RAGGED, 17 = LC_HGEMM_KPAD since round 6)
        assert lib.lc_hgemm_f16(one, one, one, 256, 256, 256, 0, retired, 2, 1, None) == capi.LC_ERR_ARG
    assert lib.lc_hgemm_f16(one, one, one, 0, 256, 256, 0, 0, 2, 1, None) == capi.LC_ERR_SHAPE
    assert lib.lc_hgemm_f16(one, one, one, 128, 256, 256, 0, capi.HGEMM_MFMA256, 2, 1, None) == capi.LC_ERR_SHAPE
    assert lib.lc_hgemm_call(b"no_such_entry", one, one, one, 256, 256, 256, 2, 0, 1, None) == capi.LC_ERR_ARG
    assert lib.lc_attn_fwd_f16(one, one, one, one, 1, 1, 100, 64, 0, 0, 0, 2, None) == capi.LC_ERR_SHAPE
    assert lib.lc_attn_fwd_f16(one, one, one, one, 1, 1, 128, 64, 0, 99, 0, 2, None) == capi.LC_ERR_ARG
    # one head's K / V must fit the 32-bit buffer offsets of the LDS-DMA kernels: N * D * 2 >= 2^31 is refused,
    # by the entry points and by the kernel-name query alike, before anything is launched
    buf = C.create_string_buffer(128)
    assert lib.lc_attn_fwd_f16(one, one, one, one, 1, 1, 1 << 23, 128, 0, 0, 0, 2, None) == capi.LC_ERR_SHAPE
    assert lib.lc_attn_fwd_bf16(one, one, one, one, 1, 1, 1 << 21, 512, None) == capi.LC_ERR_SHAPE
    assert lib.lc_attn_kernel_name(1 << 23, 128, 0, 0, buf, 128) == capi.LC_ERR_SHAPE
    assert lib.lc_attn_kernel_name((1 << 23) - 256, 128, 0, 0, buf, 128) == capi.LC_OK and buf.value.startswith(b"attn_fwd_w4u_kernel<128,false,0>")
    # head-dim limits of the reference dispatchers (split_q: 128; share_qkv stage2: 128, stage1: 256)
    assert lib.lc_attn_call(b"flash_attn_mma_stages_split_q", one, one, one, one, 1, 1, 128, 256, 2, None) \
        == capi.LC_ERR_HEADDIM
    assert lib.lc_attn_call(b"flash_attn_mma_stages_split_q_shared_qkv", one, one, one, one, 1, 1, 128, 256, 2,
                            None) == capi.LC_ERR_HEADDIM
    # the vendor entry initialises its handle lazily (hgemm_cublas.cu:44-46); with no GPU that fails cleanly
    assert lib.lc_hgemm_vendor_f16(one, one, one, 256, 256, 256, 0, None) in (capi.LC_ERR_VENDOR, capi.LC_ERR_DEVICE)
    assert lib.lc_timer_stop(None, None) == capi.LC_ERR_ARG
    assert lib.lc_clock_probe(None, None) == capi.LC_ERR_ARG


def test_hgemm_dispatch_covers_the_reference_legal_shapes(built):
    """Kernel selection needs no GPU (lc_hgemm_kernel_name shares resolve_hgemm_variant with the launcher).  The reference's kernels
    are legal on M, N multiples of 128 and K multiples of 32 (hgemm_mma_stage.cu:650,675-676): with a large 256-tileable interior those
    shapes run the flagship kernel (+ border strips / a half K-step), small grids the 128-tile kernel, anything else the edge kernel;
    the cross-check kernels keep their 256 / K % 64 contract and say so."""
    from leetcuda_amd import capi
    capi.load()
    for lay, nn in ((capi.LAYOUT_NN, "true"), (capi.LAYOUT_TN, "false")):
        sch = 1 if nn == "true" else 2      # (the NN loop has one schedule; TN: "w4y_sched", 2 since round 6)
        for shp in ((8320, 8320, 8320), (8192, 8320, 8192), (8320, 8192, 8192), (8192, 8192, 8224), (3200, 3200, 96)):
            assert capi.hgemm_kernel_name(*shp, lay) == f"hgemm_w4y_kernel<{nn},{sch}>", shp
            assert capi.hgemm_kernel_name(*shp, lay, capi.HGEMM_MFMA128) == f"hgemm_mfma128_kernel<{nn},1>"     # thousands of blocks: four waves, two blocks per CU
            for v in (capi.HGEMM_MFMA256, capi.HGEMM_MFMA256W4C, capi.HGEMM_MFMA256W4X):
                with pytest.raises(capi.LcError, match="Tensor size mismatch"):
                    capi.hgemm_kernel_name(*shp, lay, v)
        for shp in ((384, 384, 128), (256, 256, 96), (128, 640, 160), (768, 768, 800)):         # <= 48 blocks of 128 x 128: the 128-tile kernel
            assert capi.hgemm_kernel_name(*shp, lay) == f"hgemm_mfma128_kernel<{nn},2>", shp                 # <= 0.6 blocks per CU: the eight-wave form
        # between the small grids and > 128 tiles of 256 x 256: the mid-size kernel (round 6) with the smallest tile whose grid is one
        # round of the 256 CUs this rule assumes without a device (three ring slots), else 128 x 128 at two workgroups per CU
        for shp, tile in (((1024, 1024, 1024), "1,2,3"), ((1280, 1280, 1280), "1,2,3"), ((1536, 1536, 1568), "1,3,3" if nn == "false" else "2,2,3"),
                          ((1792, 1792, 1792), "2,2,3"), ((2048, 2048, 2080), "2,2,3"), ((2304, 2304, 2304), "2,3,3" if nn == "false" else "3,2,3"),
                          ((2560, 2560, 2560), "2,2,2"), ((2816, 2816, 2816), "2,2,2"), ((2816, 2560, 96), "2,2,2")):
            assert capi.hgemm_kernel_name(*shp, lay) == f"hgemm_mid_kernel<{nn},{tile}>", shp
            assert capi.hgemm_kernel_name(*shp, lay, capi.HGEMM_MFMA128).startswith("hgemm_mfma128_kernel<")  # ... still there when asked for
        # 144 tiles of 256 x 256 leave 112 CUs idle in their one round: TN has a tile that fills all 256 in one round (192 x 192), NN has not
        assert capi.hgemm_kernel_name(3072, 3072, 3072, lay) == (f"hgemm_w4y_kernel<{nn},{sch}>" if nn == "true" else "hgemm_mid_kernel<false,3,3,3>")
        assert capi.hgemm_kernel_name(3328, 3328, 3328, lay) == f"hgemm_w4y_kernel<{nn},{sch}>"                   # 169 tiles, no mid-size tile in one round
        # split-K (round 6): a one-round grid on at most half the CUs with at least 32 K tiles per range; never at the reference sweep's sizes
        assert capi.hgemm_kernel_name(1024, 1024, 8192, lay) == f"hgemm_mid_sk_kernel<{nn},1,3> x2"
        assert capi.hgemm_kernel_name(512, 512, 8192, lay) == f"hgemm_mid_sk_kernel<{nn},1,3> x4"
        assert capi.hgemm_kernel_name(1024, 1024, 2048, lay) == f"hgemm_mid_kernel<{nn},1,2,3>"
        capi.tune("hgemm_mid_splitk", 1)
        try:
            assert capi.hgemm_kernel_name(1024, 1024, 8192, lay) == f"hgemm_mid_kernel<{nn},1,2,3>"
            assert capi.hgemm_kernel_name(512, 512, 8192, lay) == f"hgemm_mfma128_kernel<{nn},2>"
        finally:
            capi.tune("hgemm_mid_splitk", 0)
        # M, N multiples of 64 only (not legal in the reference): a mid-size tile where one divides the shape, else the edge kernel
        assert capi.hgemm_kernel_name(2880, 2880, 2880, lay) == ("hgemm_mid_kernel<false,3,3,3>" if nn == "false" else "hgemm_mid_edge_kernel<true,3,2,3>")
        assert capi.hgemm_kernel_name(8192, 8256, 8192, lay) == f"hgemm_w4y_kernel<{nn},{sch}> + hgemm_mid_edge_kernel<{nn},2,2,3>"   # (a flagship-size interior goes ahead of the 128 x 192 tile)
        if nn == "false":
            assert capi.hgemm_kernel_name(8192, 8256, 8192, lay, capi.HGEMM_MID) == "hgemm_mid_kernel<false,2,3,3>"
        # 192 x 192 only where 128 x 128 at two per CU needs more than one double round (3072^3: 576 blocks on 512 slots)
        assert capi.hgemm_kernel_name(3072, 2304, 3072, lay) == f"hgemm_mid_kernel<{nn},2,2,2>"
        assert capi.hgemm_kernel_name(1088, 1152, 512, lay) == f"hgemm_mid_kernel<{nn},1,2,3>"
        # the knobs: never / a forced tile / the explicit variant on shapes LC_HGEMM_AUTO keeps away from it
        capi.tune("hgemm_mid", 1)
        try:
            assert capi.hgemm_kernel_name(2048, 2048, 2080, lay) == f"hgemm_mfma128_kernel<{nn},1>"           # (round 5's choice: one block per CU, four waves)
        finally:
            capi.tune("hgemm_mid", 0)
        assert capi.hgemm_kernel_name(384, 384, 128, lay, capi.HGEMM_MID) == f"hgemm_mid_kernel<{nn},1,2,3>"
        assert capi.hgemm_kernel_name(8192, 8192, 8192, lay, capi.HGEMM_MID) == f"hgemm_mid_kernel<{nn},2,2,2>"
        with pytest.raises(capi.LcError, match="Tensor size mismatch"):
            capi.hgemm_kernel_name(192, 96, 64, lay, capi.HGEMM_MID)
        for shp in ((384, 384, 128), (256, 256, 96)):                                             # ... the flagship kernel when asked for
            assert capi.hgemm_kernel_name(*shp, lay, capi.HGEMM_MFMA256W4Y) == f"hgemm_w4y_kernel<{nn},{sch}>"
        # ragged M / N with K % 32 == 0 (LC_HGEMM_RAGGED): a flagship-sized interior on hgemm_w4y_kernel + the border on clamped 128 x 128 tiles of the
        # mid-size kernel; smaller problems entirely on those tiles (three ring slots while they fit one round of the CUs)
        assert capi.hgemm_kernel_name(8192, 8224, 8192, lay) == f"hgemm_w4y_kernel<{nn},{sch}> + hgemm_mid_edge_kernel<{nn},2,2,3>"
        assert capi.hgemm_kernel_name(4100, 4104, 4096, lay) == f"hgemm_w4y_kernel<{nn},{sch}> + hgemm_mid_edge_kernel<{nn},2,2,3>"
        assert capi.hgemm_kernel_name(12808, 12808, 4096, lay) == f"hgemm_w4y_kernel<{nn},{sch}> + hgemm_mid_edge_kernel<{nn},2,2,3>"
        assert capi.hgemm_kernel_name(1000, 3000, 4096, lay) == f"hgemm_mid_edge_kernel<{nn},2,2,3>"
        assert capi.hgemm_kernel_name(100, 4096, 1024, lay) == f"hgemm_mid_edge_kernel<{nn},1,2,3>"        # (the smallest tile whose grid fits one round)
        assert capi.hgemm_kernel_name(100, 4096, 4096, lay) == f"hgemm_mid_edge_sk_kernel<{nn},1,3> x2"    # (... split-K when that grid covers at most half the CUs and K is long)
        assert capi.hgemm_kernel_name(2888, 2880, 544, lay, capi.HGEMM_RAGGED) == ("hgemm_mid_edge_kernel<false,3,3,3>" if nn == "false" else "hgemm_mid_edge_kernel<true,3,2,3>")
        assert capi.hgemm_kernel_name(2500, 2504, 2560, lay) == f"hgemm_mid_edge_kernel<{nn},2,2,2>"      # (128 x 128 at two per CU fits one double round)
        capi.tune("hgemm_ragged_tile", 22)
        try:
            assert capi.hgemm_kernel_name(2888, 2880, 544, lay) == f"hgemm_mid_edge_kernel<{nn},2,2,2>"
            assert capi.hgemm_kernel_name(100, 4096, 1024, lay) == f"hgemm_mid_edge_kernel<{nn},2,2,3>"
        finally:
            capi.tune("hgemm_ragged_tile", 0)
        capi.tune("hgemm_ragged", 1)
        try:
            assert capi.hgemm_kernel_name(8192, 8224, 8192, lay) == f"hgemm_edge_kernel<{nn}>"
            assert capi.hgemm_kernel_name(8192, 8224, 8192, lay, capi.HGEMM_RAGGED) == f"hgemm_w4y_kernel<{nn},{sch}> + hgemm_mid_edge_kernel<{nn},2,2,3>"   # (explicit: whatever the knob)
        finally:
            capi.tune("hgemm_ragged", 0)
        for shp in ((4096, 4096, 4096), (8192, 8192, 8200), (100, 4096, 40), (4100, 4100, 4096)):     # tiled / K % 32 / K < 64 / N % 8: not a ragged-path shape
            with pytest.raises(capi.LcError, match="Tensor size mismatch"):
                capi.hgemm_kernel_name(*shp, lay, capi.HGEMM_RAGGED)
        # K % 32 != 0 (K % 8 == 0) on a large problem: zero-padded operand copies in the workspace + whatever the padded problem runs (LC_HGEMM_KPAD)
        assert capi.hgemm_kernel_name(8192, 8192, 8200, lay) == f"hgemm_pad_copy_kernel + hgemm_w4y_kernel<{nn},{sch}>"
        assert capi.hgemm_kernel_name(8200, 8200, 8200, lay) == f"hgemm_pad_copy_kernel + hgemm_w4y_kernel<{nn},{sch}> + hgemm_mid_edge_kernel<{nn},2,2,3>"
        assert capi.hgemm_kernel_name(2000, 2000, 2056, lay, capi.HGEMM_KPAD) == f"hgemm_pad_copy_kernel + hgemm_mid_edge_kernel<{nn},2,2,3>"
        capi.tune("hgemm_kpad", 1)
        try:
            assert capi.hgemm_kernel_name(8192, 8192, 8200, lay) == f"hgemm_edge_kernel<{nn}>"
        finally:
            capi.tune("hgemm_kpad", 0)
        for shp in ((8192, 8192, 8192), (8192, 8192, 8196), (8192, 8196, 8200), (4096, 4096, 40)):    # K % 32 == 0 / K % 8 / N % 8 / K < 256: not this path
            with pytest.raises(capi.LcError, match="Tensor size mismatch"):
                capi.hgemm_kernel_name(*shp, lay, capi.HGEMM_KPAD)
        assert capi.hgemm_kernel_name(2000, 2000, 2056, lay) == f"hgemm_pad_copy_kernel + hgemm_mid_edge_kernel<{nn},2,2,3>"
        for shp in ((256, 256, 32), (640, 640, 2056), (128, 128, 48), (1000, 3000, 200)):     # K < 64, K % 32 on small problems / with a short K: the vectorised edge kernel
            assert capi.hgemm_kernel_name(*shp, lay) == f"hgemm_edge_kernel<{nn}>", shp
            assert capi.hgemm_kernel_name(*shp, lay, capi.HGEMM_GENERIC) == f"hgemm_generic_kernel<{nn}>", shp     # ... the element-wise one when asked for
            with pytest.raises(capi.LcError, match="Tensor size mismatch"):
                capi.hgemm_kernel_name(*shp, lay, capi.HGEMM_MFMA256W4Y)
        for shp in ((256, 256, 36), (100, 100, 100), (128, 130, 64)):                               # K % 8 (NN: N % 8): only the element-wise kernel is left
            if shp[2] % 8 == 0 and nn == "false":
                assert capi.hgemm_kernel_name(*shp, lay) == "hgemm_edge_kernel<false>", shp                         # (TN needs K % 8 only)
                continue
            assert capi.hgemm_kernel_name(*shp, lay) == f"hgemm_generic_kernel<{nn}>", shp
            with pytest.raises(capi.LcError, match="Tensor size mismatch"):
                capi.hgemm_kernel_name(*shp, lay, capi.HGEMM_EDGE)


def test_python_wrappers_validate_shapes(built):
    """a mismatched tensor must raise the reference's 'Tensor size mismatch!', not reach the device."""
    from leetcuda_amd import capi
    with pytest.raises(RuntimeError, match="Tensor size mismatch!"):
        capi._gemm_dims(torch.zeros(256, 128), torch.zeros(64, 256), torch.zeros(256, 256))
    with pytest.raises(RuntimeError, match="Tensor size mismatch!"):
        capi._gemm_dims(torch.zeros(256, 128), torch.zeros(128, 256), torch.zeros(128, 256))
    assert capi._gemm_dims(torch.zeros(256, 128), torch.zeros(128, 512), torch.zeros(256, 512)) == (256, 512, 128)
    assert capi._gemm_dims(torch.zeros(256, 128), torch.zeros(512, 128), torch.zeros(256, 512), capi.LAYOUT_TN) \
        == (256, 512, 128)
    with pytest.raises(RuntimeError, match="Tensor size mismatch!"):      # NN does not take the [N,K] storage shape
        capi._gemm_dims(torch.zeros(256, 128), torch.zeros(512, 128), torch.zeros(256, 512), capi.LAYOUT_NN)
    q = torch.zeros(1, 2, 64, 32)
    assert capi._attn_dims(q, q, q, q) == (1, 2, 64, 32)
    assert capi._attn_dims(q, q, q.transpose(-1, -2), q, v_transposed=True) == (1, 2, 64, 32)
    with pytest.raises(RuntimeError, match="Tensor size mismatch!"):
        capi._attn_dims(q, q[:, :1], q, q)
    with pytest.raises(RuntimeError, match="Tensor size mismatch!"):
        capi._attn_dims(q, q, q, q, v_transposed=True)


def test_python_wrappers_refuse_cpu_tensors(built):
    from leetcuda_amd import capi
    a = torch.zeros(256, 256, dtype=torch.half)
    with pytest.raises(RuntimeError, match="no CPU path"):
        capi.hgemm(a, a, a)
    q = torch.zeros(1, 1, 64, 64, dtype=torch.half)
    with pytest.raises(RuntimeError, match="no CPU path"):
        capi.attn_fwd(q, q, q, q)


@pytest.fixture(scope="module")
def torch_mods():
    from leetcuda_amd import build
    build.build_torch_ext()
    sys.path.insert(0, str(ROOT / "leetcuda_amd"))
    import flash_attn_lib
    import toy_hgemm
    return toy_hgemm, flash_attn_lib


def test_torch_modules_export_reference_names(torch_mods):
    toy_hgemm, flash_attn_lib = torch_mods
    for n in HGEMM_NAMES:
        assert callable(getattr(toy_hgemm, n)), n
    for n in ATTN_NAMES:
        assert callable(getattr(flash_attn_lib, n)), n


def test_torch_modules_error_conventions(torch_mods, capfd):
    toy_hgemm, flash_attn_lib = torch_mods
    a = torch.zeros(256, 256, dtype=torch.half)
    with pytest.raises(RuntimeError, match="values must be torch::kHalf"):
        toy_hgemm.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem(a.float(), a, a, 2, False, 1)
    assert "Tensor Info:" in capfd.readouterr().out            # the reference prints options first
    with pytest.raises(RuntimeError, match="no CPU path"):
        toy_hgemm.hgemm_naive_f16(a, a, a)
    with pytest.raises(TypeError):                             # 3-arg entry called with 6 args
        toy_hgemm.hgemm_naive_f16(a, a, a, 2, False, 1)
    q = torch.zeros(1, 1, 64, 64, dtype=torch.half)
    with pytest.raises(RuntimeError, match="values must be torch::kHalf"):
        flash_attn_lib.flash_attn_mma_stages_split_q(q.float(), q, q, q, 2)
    with pytest.raises(RuntimeError, match="no CPU path"):
        flash_attn_lib.flash_attn_cute(q, q, q, q)
    # round-4 verdict (structure #12): a non-contiguous view is refused with a message of its own — the reference compares sizes only
    # and would read the base storage in the wrong order (the TN operand of the reference is as_col_major's CONTIGUOUS tensor)
    with pytest.raises(RuntimeError, match="must be contiguous"):
        toy_hgemm.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_tn_swizzle_x4(a, a.t(), a, 2, False, 1)
    with pytest.raises(RuntimeError, match="must be contiguous"):
        toy_hgemm.hgemm_naive_f16(a, a, torch.zeros(256, 512, dtype=torch.half)[:, ::2])
    with pytest.raises(RuntimeError, match="must be contiguous"):
        flash_attn_lib.flash_attn_mma_stages_split_q(q, q.transpose(2, 3), q, q, 2)
    with pytest.raises(RuntimeError, match="must be contiguous"):
        flash_attn_lib.flash_attn_mma_stages_split_q_shared_qkv_swizzle_qkv(q, q, q.transpose(-2, -1), q, 2)   # V^T must be materialised


def test_cpp_bench_harness_builds_and_fails_loudly_without_gpu(built):
    """tools/cpp/hgemm_bench.bin: torch-free, links only libleetcuda_amd.so; without an MI355X it must
    stop at lc_device_check with a non-zero exit code, never fall back to anything."""
    from leetcuda_amd import build
    exe = build.build_cpp_bench()
    assert exe.exists()
    ldd = subprocess.run(["ldd", str(exe)], capture_output=True, text=True).stdout
    assert "libleetcuda_amd.so" in ldd and "libtorch" not in ldd and "libc10" not in ldd
    p = subprocess.run([str(exe), "--help"], capture_output=True, text=True)
    assert p.returncode == 0 and "--layout nn|tn" in p.stdout
    if not torch.cuda.is_available():
        p = subprocess.run([str(exe), "--mnk", "256", "256", "256"], capture_output=True, text=True)
        assert p.returncode == 4 and "no gfx950 device" in p.stderr


def test_wheel_carries_the_drop_in_modules(built, tmp_path):
    """Round-5 verdict (next #9; the reference ships `toy-hgemm` as a wheel: kernels/hgemm/setup.py:11,48): `pip install .` must make
    `import toy_hgemm` / `import flash_attn_lib` work with no PYTHONPATH.  Build the wheel offline, unpack it and import the two TOP-LEVEL
    modules from the unpacked tree alone (a fresh interpreter whose sys.path holds no source directory): the 38 + 29 names of the
    reference's pybind tables must be there, resolved against the libleetcuda_amd.so INSIDE the wheel."""
    import subprocess
    import sys
    import zipfile
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    (tmp_path / "egg").mkdir()
    p = subprocess.run([sys.executable, "setup.py", "-q", "egg_info", "--egg-base", str(tmp_path / "egg"), "build", "--build-base", str(tmp_path / "b"),
                        "bdist_wheel", "-d", str(tmp_path / "dist"), "--bdist-dir", str(tmp_path / "bd")],
                       cwd=root, capture_output=True, text=True, timeout=900)       # (nothing is left behind in the source tree)
    assert p.returncode == 0, p.stderr[-3000:]
    whl = next((tmp_path / "dist").glob("leetcuda_amd-*.whl"))
    assert "linux" in whl.name and "none-any" not in whl.name          # a platform wheel (gfx950 code objects, CPython-ABI modules)
    site = tmp_path / "site"
    with zipfile.ZipFile(whl) as z:
        names = z.namelist()
        z.extractall(site)
    assert "leetcuda_amd/lib/libleetcuda_amd.so" in names and "leetcuda_amd/include/lc_abi.h" in names and "leetcuda_amd/capi.py" in names
    assert any(n.startswith("toy_hgemm.") and n.endswith(".so") for n in names) and any(n.startswith("flash_attn_lib.") and n.endswith(".so") for n in names)
    code = ("import sys; sys.path[:] = [p for p in sys.path if p and not p.startswith(%r)]; sys.path.insert(0, %r)\n"
            "import torch, toy_hgemm, flash_attn_lib, leetcuda_amd\n"
            "from leetcuda_amd import capi\n"
            "assert toy_hgemm.__file__.startswith(%r) and capi.LIB_PATH.as_posix().startswith(%r), (toy_hgemm.__file__, capi.LIB_PATH)\n"
            "lib = capi.load()\n"
            "h = [lib.lc_hgemm_entry_name(i).decode() for i in range(lib.lc_hgemm_entry_count())]\n"
            "a = [lib.lc_attn_entry_name(i).decode() for i in range(lib.lc_attn_entry_count())]\n"
            "assert len(h) == 38 and len(a) == 29\n"
            "assert all(callable(getattr(toy_hgemm, n)) for n in h) and all(callable(getattr(flash_attn_lib, n)) for n in a)\n"
            "print('ok', len(h), len(a))\n") % (str(root), str(site), str(site), str(site))
    env = {k: v for k, v in __import__("os").environ.items() if k != "PYTHONPATH"}
    q = subprocess.run([sys.executable, "-c", code], cwd=tmp_path, capture_output=True, text=True, timeout=600, env=env)
    assert q.returncode == 0 and q.stdout.strip().endswith("ok 38 29"), (q.stdout[-500:], q.stderr[-3000:])


def test_launch_rules_reason_with_any_cu_count(built):
    """Round-5 verdict (weak #13): the launch rules are CU-relative forms ("rule_cus" lets a test ask what a 128- or 304-CU device would be
    told; without a device the library assumes 256).  For every CU count: the split-KV factor never grows when more (batch, head) problems
    arrive, never splits a grid of two full rounds, and always leaves >= 4 KV tiles per range; the GEMM's mid-size tile never needs more
    workgroups than one round when a one-round tile exists; the eight-wave 128-tile kernel keeps the smallest grids."""
    from leetcuda_amd import capi
    capi.load()

    def split_of(bh, N, D):
        name = capi.attn_kernel_name(N, D, bh=bh)
        return name

    try:
        for cus in (128, 256, 304):
            capi.tune("rule_cus", cus)
            for D in (64, 128):
                for N in (1024, 2048, 4096, 8192):
                    was_split = True
                    for bh in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512):
                        name = split_of(bh, N, D)
                        is_split = name.endswith(",3>")
                        if bh * (N // 256) >= 2 * cus:
                            assert not is_split, (cus, bh, N, D, name)          # two full rounds of query blocks: nothing to gain
                        if bh * (N // 256) * 16 <= cus and N >= 2048:
                            assert is_split, (cus, bh, N, D, name)              # a sixteenth of the GPU and >= 32 tiles: always worth it
                        if is_split:
                            assert was_split or bh * (N // 256) > cus, (cus, bh, N, D)   # below one round the answer is monotone in bh
                        if bh * (N // 256) <= cus:
                            was_split = is_split
            # GEMM: the smallest grids stay on the eight-wave 128-tile kernel, one-round grids get a one-round tile, big ones the 256 tile
            assert capi.hgemm_kernel_name(512, 512, 512, capi.LAYOUT_TN).startswith("hgemm_mfma128_kernel<false,2>")
            n1 = 128 * int((cus * 0.9) ** 0.5)                                   # ~ 0.8 cus blocks of 128 x 128
            name = capi.hgemm_kernel_name(n1, n1, 2048, capi.LAYOUT_TN)
            assert name.startswith("hgemm_mid_kernel<false,") and name.endswith(",3>"), (cus, n1, name)      # one round: three ring slots
            assert capi.hgemm_kernel_name(8192, 8192, 8192, capi.LAYOUT_TN) == "hgemm_w4y_kernel<false,2>"
            big = 256 * (int((cus / 2) ** 0.5) + 1)                              # just over cus / 2 tiles of 256 x 256
            assert capi.hgemm_kernel_name(big, big, 4096, capi.LAYOUT_NN) == "hgemm_w4y_kernel<true,1>", (cus, big)
            name = capi.hgemm_kernel_name(big, big, 4096, capi.LAYOUT_TN)       # TN: unless 192 x 192 tiles fill more CUs in ONE round (256 CUs: 3072)
            fills = big % 192 == 0 and (big // 256) ** 2 < (big // 192) ** 2 <= cus
            assert name == ("hgemm_mid_kernel<false,3,3,3>" if fills else "hgemm_w4y_kernel<false,2>"), (cus, big, name)
    finally:
        capi.tune("rule_cus", 0)
    with pytest.raises(capi.LcError):
        capi.tune("rule_cus", 7)
