#!/usr/bin/env python3
"""Condense the rocprofv3 databases of one gpurun round (gpurun_out/<tag>/prof_stats, pmc_*) into small,
committed summaries under profiles/:

    profiles/<tag>_kernel_stats.{md,json}   per-kernel calls / avg / min / max duration, VGPR, LDS
    profiles/<tag>_pmc.json                 per-kernel counter averages + derived HBM bytes per launch

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are collected in
SEPARATE --pmc passes (TCC slots), both are reported in KiB, and on gfx950 FETCH_SIZE counts a wide
coalesced 128-B read as 64 B -> the read side is doubled ("fetch_x2") before comparing with a byte count.
"""
import json
import re
import sqlite3
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(Path(__file__).resolve().parent))
from prof_workloads import workload_of  # noqa: E402


def short(name: str) -> str:
    v = re.match(r"(Custom_)?Cijk_(\w+?)_(HHS|HSS|BBS)\w*?_(MT\d+x\d+x\d+)_(MI\d+x\d+x\d+)", name)
    if v:   # hipBLASLt (the comparator): operand layout, macro tile, MFMA shape
        return f"hipblaslt:{'Custom_' if v.group(1) else ''}Cijk_{v.group(2)}_{v.group(4)}_{v.group(5)}"
    m = re.match(r"_ZN2lc\d+(\w+?_kernel)I(.*?)EEv", name)
    if not m:
        m2 = re.match(r"_ZN2lc\d+(\w+?_kernel)E", name)
        return m2.group(1) if m2 else name[:60]
    args = m.group(2).replace("Lb0E", "false,").replace("Lb1E", "true,")
    args = re.sub(r"Li(\d+)E", r"\1,", args).rstrip(",")
    return f"{m.group(1)}<{args}>"


def main(tag: str):
    src = ROOT / "gpurun_out" / tag
    out = ROOT / "profiles"
    out.mkdir(exist_ok=True)
    stats = {}
    db = next((src / "prof_stats").glob("*.db"), None)
    if db:
        cur = sqlite3.connect(db).cursor()
        rows = cur.execute("select name, duration, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, grid_x, "
                           "workgroup_x from kernels where name like '_ZN2lc%' or name like '%Cijk_%'").fetchall()
        # one row per (kernel, grid): bench.py launches some kernels on several shapes (round 6: the sweep points next to 8192^3) and an average
        # over shapes would agree with nothing.  The grid that holds most of a kernel's time keeps the plain name (what bench.py's
        # roofline.kernel says), the others are suffixed with their workgroup count.
        by_grid = defaultdict(list)
        gmeta = {}
        for name, dur, vg, ag, sg, lds, gx, wx in rows:
            key = (short(name), gx // max(wx, 1))
            by_grid[key].append(dur / 1000.0)
            gmeta[key] = {"vgpr": vg, "agpr": ag, "sgpr": sg, "lds_bytes": lds, "grid_x": gx, "wg_x": wx}
        main_grid = {}
        for (nm, wgs), v in by_grid.items():
            if nm not in main_grid or sum(v) > sum(by_grid[(nm, main_grid[nm])]):
                main_grid[nm] = wgs
        agg = defaultdict(list)
        meta = {}
        for (nm, wgs), v in by_grid.items():
            key = nm if main_grid[nm] == wgs else f"{nm} @{wgs}wg"
            agg[key] = v
            meta[key] = gmeta[(nm, wgs)]
        for k, v in agg.items():
            v2 = sorted(v)
            stats[k] = {"calls": len(v), "avg_us": sum(v) / len(v), "min_us": v2[0], "max_us": v2[-1],
                        "median_us": v2[len(v2) // 2], **meta[k]}
        (out / f"{tag}_kernel_stats.json").write_text(json.dumps(stats, indent=1))
        lines = [f"# rocprofv3 --kernel-trace --stats — {tag}", "",
                 "command: see profiles/README.md (rounds r02p / r03z: `rocprofv3 --kernel-trace --stats -- python bench.py "
                 "--steps 20 --warmup 2 --no-cpu-baseline`, i.e. the bench command itself)", "",
                 "| kernel | calls | avg µs | median µs | min µs | max µs | VGPR | LDS B | grid |", "|---|---|---|---|---|---|---|---|---|"]
        for k, s in sorted(stats.items(), key=lambda kv: -kv[1]["avg_us"]):
            lines.append(f"| `{k}` | {s['calls']} | {s['avg_us']:.1f} | {s['median_us']:.1f} | {s['min_us']:.1f} | "
                         f"{s['max_us']:.1f} | {s['vgpr']} | {s['lds_bytes']} | {s['grid_x']} |")
        (out / f"{tag}_kernel_stats.md").write_text("\n".join(lines) + "\n")
    pmc = defaultdict(dict)
    for d in sorted(src.glob("pmc_*")):
        db = next(d.glob("*.db"), None) if d.is_dir() else None
        if not db:
            continue
        cur = sqlite3.connect(db).cursor()
        rows = cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                           "where kernel_name like '_ZN2lc%' or kernel_name like '%Cijk_%' group by kernel_name, counter_name").fetchall()
        for name, cname, val, cnt in rows:
            pmc[short(name)][cname] = val
    for k, c in pmc.items():
        c["workload"] = workload_of(k)      # the launch shape of tools/prof_kernels.py (bench.py matches on it)
        if "FETCH_SIZE" in c:
            c["hbm_read_bytes_fetch_x2"] = c["FETCH_SIZE"] * 1024 * 2
        if "WRITE_SIZE" in c:
            c["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            c["hbm_bytes_per_launch"] = c["hbm_read_bytes_fetch_x2"] + c["hbm_write_bytes"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"]:
            # busy cycles are summed over the 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
            # (11.7 M "cycles" for a 0.88 ms kernel = 8 x 1.47 M): wall cycles = GRBM_GUI_ACTIVE / 8
            wall = c["GRBM_GUI_ACTIVE"] / 8.0
            c["wall_cycles"] = wall
            c["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (wall * 256 * 4)
        if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
            c["lds_conflict_frac"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    (out / f"{tag}_pmc.json").write_text(json.dumps(pmc, indent=1))
    if pmc:
        (out / "latest_pmc.json").write_text(json.dumps(pmc, indent=1))
    print(json.dumps({"stats": stats, "pmc": pmc}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01a")
