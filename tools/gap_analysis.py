#!/usr/bin/env python3
"""Where the GPU idles inside one bench step: gaps between consecutive kernels of a rocprofv3 kernel trace
(tools/prof_bench.sh), attributed to the kernel that FOLLOWS the gap."""
import csv, glob, sys
from collections import defaultdict
f = glob.glob((sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_bench") + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
st = [int(r["Start_Timestamp"]) for r in rows]
en = [int(r["End_Timestamp"]) for r in rows]
# steps begin at every k_threshold (the prune); take the last complete step
starts = [i for i, n in enumerate(names) if n.startswith("void k_threshold")]
a, b = starts[-2], starts[-1]
busy = sum(en[i] - st[i] for i in range(a, b))
wall = st[b] - st[a]
print(f"one step: {b - a} launches, wall {wall/1e3:.0f} us, kernels {busy/1e3:.0f} us, idle {(wall-busy)/1e3:.0f} us")
gaps = defaultdict(lambda: [0, 0.0, 0.0])
prev_end = en[a]
for i in range(a + 1, b):
    g = st[i] - prev_end
    prev_end = max(prev_end, en[i])
    key = names[i][:48] + "  <-after-  " + names[i - 1][:32]
    gaps[key][0] += 1; gaps[key][1] += g; gaps[key][2] = max(gaps[key][2], g)
for k, (c, t, m) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"  {t/1e3:8.1f} us total  x{c:3d}  avg {t/c/1e3:6.1f}  max {m/1e3:6.1f}   {k}")
