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


def test_bench_traffic_keys_exist_in_committed_pmc_summary():
    """bench.py looks the dominant kernels' HBM bytes up in profiles/latest_pmc.json (written by
    tools/summarize_prof.py from the separate rocprofv3 --pmc passes): the keys it asks for must be there, otherwise
    `roofline.traffic` silently degrades to null."""
    import importlib.util
    import json
    root = Path(__file__).resolve().parent.parent
    spec = importlib.util.spec_from_file_location("lc_bench", root / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    pmc = json.loads((root / "profiles" / "latest_pmc.json").read_text())
    for layout in ("tn", "nn"):
        key = bench.pmc_key_hgemm("auto", layout)
        assert key in pmc and pmc[key]["hbm_bytes_per_launch"] > 0, key
        assert bench.pmc_traffic(key) == pmc[key]["hbm_bytes_per_launch"]
    src = (root / "bench.py").read_text()
    m = re.search(r'pmc_traffic\("(attn_[^"]+)"\)', src)
    assert m and m.group(1) in pmc, m and m.group(1)
    # algorithmic bytes of the 8192^3 HGEMM are 3 * 8192^2 * 2; fabric traffic is a small multiple of it, never less
    assert pmc[bench.pmc_key_hgemm("auto", "tn")]["hbm_bytes_per_launch"] >= 3 * 8192 * 8192 * 2
