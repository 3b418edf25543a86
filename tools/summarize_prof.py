#!/usr/bin/env python3
"""Summarise rocprofv3 csv output of tools/prof.sh: per-kernel time stats and mean PMC values."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
lines = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    lines.append(f"== {os.path.relpath(f, out)}")
    for row in csv.DictReader(open(f)):
        lines.append("  {Name:60.60s} calls={Calls:>6s} avg_ns={AverageNs:>12s} min={MinNs:>10s} max={MaxNs:>10s} pct={Percentage}".format(**row))
for sub in ("pmc1", "pmc2", "pmc3", "pmc4"):
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"][:50]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        lines.append(f"== {os.path.relpath(f, out)}")
        for k, d in acc.items():
            if "k_assign" not in k and "k_finalize" not in k and "k_chunk" not in k and "k_threshold" not in k:
                continue
            lines.append(f"  {k}")
            for c, v in d.items():
                v2 = sorted(v)
                lines.append(f"     {c:28s} n={len(v):4d} median={v2[len(v2)//2]:.6g} mean={sum(v)/len(v):.6g}")
txt = "\n".join(lines)
open(os.path.join(out, "summary.txt"), "w").write(txt + "\n")
print(txt)
