#!/usr/bin/env python3
"""Per-kernel durations out of a rocprofv3 --kernel-trace database (rocpd sqlite), grouped by kernel AND grid size — a tool run of
several shapes through one kernel stays apart.
usage: kernel_durations.py <directory or .db> [substring filter]"""
import sqlite3
import sys
from collections import defaultdict
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from summarize_prof import short  # noqa: E402

src = Path(sys.argv[1])
db = src if src.suffix == ".db" else next(src.rglob("*.db"), None)
if db is None:
    raise SystemExit(f"no .db under {src}")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, duration, grid_x, workgroup_x from kernels").fetchall()
agg = defaultdict(list)
for name, dur, gx, wx in rows:
    if flt in name:
        agg[(short(name), gx // max(wx, 1))].append(dur / 1000.0)
print(f"# {db.name}: kernel, workgroups, calls, avg / median / min / max us")
for (k, wgs), v in sorted(agg.items(), key=lambda kv: (kv[0][0], kv[0][1])):
    v2 = sorted(v)
    print(f"{k:48s} wgs {wgs:6d} calls {len(v):6d} avg {sum(v) / len(v):9.2f} median {v2[len(v2) // 2]:9.2f} min {v2[0]:9.2f} max {v2[-1]:9.2f}")
