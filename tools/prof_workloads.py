"""Which launch shape tools/prof_kernels.py profiles each kernel on: kernel-name prefix (as tools/summarize_prof.py
shortens it) -> workload key.  bench.py prints a committed per-launch byte count only next to the same key
(roofline(..., workload=...)): a byte count of another shape would be meaningless."""
WORKLOADS = [
    ("hgemm_mid_edge_kernel", "hgemm_ragged"), ("hgemm_edge_kernel", "hgemm_ragged"), ("hgemm_mid_kernel", "hgemm_2048"), ("hipblaslt:Cijk_Alik_Bljk_MT128x128", "hgemm_2048"), ("hipblaslt:Cijk_Ailk_Bljk_MT128x128", "hgemm_2048"),
    ("hgemm_", "hgemm_8192"), ("hipblaslt:", "hgemm_8192"),
    ("gemm_fp8_", "fp8_16384"),
    # attn_fwd_w4u_kernel<D, VT, WALK>: config 3 (N = 4096) runs the static persistent walk (1), config 4 / the D = 64 shape (N = 8192) one
    # block per workgroup (0); prof_kernels.py launches WALK 0 only at config 4's shape
    ("attn_fwd_w4u_kernel<128,false,1", "attn_cfg3"), ("attn_fwd_w4u_kernel<128,false,0", "attn_cfg4"), ("attn_fwd_w4u_kernel<128,false,2", "attn_cfg3"),
    ("attn_fwd_w4u_kernel<128,false,3", "attn_split_1x4x4096"), ("attn_split_combine_kernel<128", "attn_split_1x4x4096"),
    ("attn_fwd_w4u_kernel<128,true", "attn_cfg3"), ("attn_fwd_w4i_kernel<128", "attn_cfg3"), ("attn_fwd_kernel<128", "attn_cfg3"),
    ("attn_fwd_w4u_kernel<64", "attn_d64"), ("attn_fwd_w4i_kernel<64", "attn_d64"), ("attn_fwd_kernel<64", "attn_d64"),
    ("attn_fwd_bigd2_kernel<512,false", "attn_d512_fp16"), ("attn_fwd_bigd2_kernel<512,true", "attn_d512_bf16"),
    ("attn_fwd_bigd3_kernel<512,false", "attn_d512_fp16"), ("attn_fwd_bigd3_kernel<512,true", "attn_d512_bf16"),
    ("attn_fwd_bigd7_kernel<false,false", "attn_d256_fp16"), ("attn_fwd_bigd7_kernel<true", "attn_d256_bf16"), ("attn_fwd_bigd7_kernel<false,true", "attn_d256_fp16"),
    ("attn_fwd_bigd4_kernel", "attn_d1024"), ("attn_fwd_bigd6_kernel<false", "attn_d512_fp16"), ("attn_fwd_bigd6_kernel<true", "attn_d512_bf16"),
]


def workload_of(short_name: str):
    return next((w for p, w in WORKLOADS if short_name.startswith(p)), None)
