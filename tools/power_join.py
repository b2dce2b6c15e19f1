#!/usr/bin/env python3
"""Join a tools/power_watch.sh log with the MODE / MARK lines of the workload: mean board power and sclk per interval."""
import re
import sys
plog, rlog = sys.argv[1], sys.argv[2]
rows = []
for l in open(plog):
    m = re.match(r"t=([\d.]+) (.*)", l)
    if not m:
        continue
    p = re.search(r"Power \(W\): ([\d.]+)", m.group(2))
    c = re.search(r"sclk clock level: \w+: \((\d+)Mhz\)", m.group(2))
    mc = re.search(r"mclk clock level: \w+: \((\d+)Mhz\)", m.group(2))
    fc = re.search(r"fclk clock level: \w+: \((\d+)Mhz\)", m.group(2))
    if p and c:
        rows.append((float(m.group(1)), float(p.group(1)), int(c.group(1)), int(mc.group(1)) if mc else 0, int(fc.group(1)) if fc else 0))
for l in open(rlog):
    m = re.search(r"t0=([\d.]+) t1=([\d.]+)", l)
    if not m:
        continue
    a, b = float(m.group(1)), float(m.group(2))
    sel = [r for r in rows if a + 0.5 <= r[0] <= b]
    if sel:
        print(f"{l.split(' t0=')[0].strip():70s} | {sum(r[1] for r in sel) / len(sel):7.0f} W  sclk {sum(r[2] for r in sel) / len(sel):5.0f} MHz  mclk {sel[-1][3]} fclk {sel[-1][4]}  ({len(sel)} samples)")
    else:
        print(l.strip(), "| no samples")
