#!/usr/bin/env python3
"""Which kernel does hipBLASLt pick per size?  (data for the LC_HGEMM_AUTO rule: macro tile, wave tile, split-K (GSU), LDS use.)

    rocprofv3 --kernel-trace -d gpurun_out/vk -o vk -- python tools/vendor_kernels.py run 256,512,...   [tn|nn|both]
    python tools/vendor_kernels.py report gpurun_out/vk

`run` launches hipBLASLt AND this library's AUTO 20 times per (size, layout); `report` prints, per kernel name (Tensile's full name, which
spells out MT = macro tile, MIWT = MFMA blocks per wave, WG, GSU = global split-K, LDS bytes ...) and workgroup count, the
launches and the median duration, so the two libraries' choices can be read side by side."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def run():
    import torch
    from leetcuda_amd import capi, host
    capi.load()
    capi.vendor_init()
    sizes = [int(x) for x in sys.argv[2].split(",")]
    lays = sys.argv[3] if len(sys.argv) > 3 else "both"
    for n in sizes:
        a = torch.randn(n, n, dtype=torch.half, device="cuda")
        b = torch.randn(n, n, dtype=torch.half, device="cuda")
        c = torch.empty(n, n, dtype=torch.half, device="cuda")
        st = host.make_block_swizzle_stride(n, n)
        for lname, lay in (("tn", capi.LAYOUT_TN), ("nn", capi.LAYOUT_NN)):
            if lays not in ("both", lname):
                continue
            b2 = host.as_col_major(b) if lay == capi.LAYOUT_TN else b
            for _ in range(20):
                capi.hgemm_vendor(a, b2, c, lay)
            torch.cuda.synchronize()
            for _ in range(20):
                capi.hgemm(a, b2, c, layout=lay, variant=capi.HGEMM_AUTO, swizzle_stride=st)
            torch.cuda.synchronize()
            print(f"n={n} {lname}: {capi.hgemm_kernel_name(n, n, n, lay)}", flush=True)
    capi.vendor_destroy()


def report():
    import re
    import sqlite3
    from collections import defaultdict
    src = Path(sys.argv[2])
    db = src if src.suffix == ".db" else next(src.rglob("*.db"), None)
    if db is None:
        raise SystemExit(f"no .db under {src}")
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, duration, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, start from kernels order by start").fetchall()
    agg = defaultdict(list)
    first = {}
    for name, dur, gx, wx, lds, vg, ag, start in rows:
        if "Cijk" not in name and "_ZN2lc" not in name:
            continue
        key = (name, gx // max(wx, 1), wx, lds, vg, ag)
        agg[key].append(dur / 1000.0)
        first.setdefault(key, start)
    for key in sorted(agg, key=lambda k: first[k]):
        name, wgs, wx, lds, vg, ag = key
        v = sorted(agg[key])
        nm = name
        if "Cijk" in name:   # keep the tokens that describe the schedule
            toks = name.split("_")
            keep = [t for t in toks if re.match(r"(Cijk|A[il][il]k|B[lj][lj]k|MT\d|MI\d|MIWT|WG\d|GSU|DTL|PGR|PLR|LDSB|SU\d|WGM|1LDS|LBSPP|SK|STA|WSGR|WS\d|NTA|NTB|GRVW|TLDS|ULSGRO|CLR)", t)]
            nm = "_".join(keep)
        print(f"wgs {wgs:6d} x {wx:4d} lds {lds:6d} vgpr {vg:3d}+{ag:3d} calls {len(v):4d} median {v[len(v) // 2]:9.2f} us  {nm}")


if __name__ == "__main__":
    {"run": run, "report": report}[sys.argv[1]]()
