"""leetcuda_amd — MI355X (gfx950) native HGEMM + FlashAttention-2 forward behind LeetCUDA's entry points.

Layout
  include/lc_abi.h                      the C-ABI (drop-in boundary)
  leetcuda_amd/csrc/*.hip               hand-written CDNA4 kernels + the C-ABI host code
  leetcuda_amd/lib/libleetcuda_amd.so   built in-tree by `python -m leetcuda_amd.build`
  leetcuda_amd/toy_hgemm*.so            PyTorch extension modules exporting the reference names
  leetcuda_amd/flash_attn_lib*.so
  leetcuda_amd/capi.py                  ctypes view of the C-ABI (tests, bench.py)
  leetcuda_amd/host.py                  host-side helpers mirrored from the reference benches

There is no CPU path: every compute entry fails loudly without the HIP library or without a gfx950 GPU.
"""
__version__ = "0.1.0"
