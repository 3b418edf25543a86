#!/usr/bin/env python3
"""Timeline of the last bench step from a rocprofv3 --kernel-trace CSV: busy / idle time, the idle gaps and their neighbours."""
import csv, glob, os, sys
d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
short = lambda n: n.replace("void ", "").split("(")[0][:38]
starts = [i for i, e in enumerate(ev) if "k_threshold" in e[2]]
i0 = starts[-1]
step = ev[i0:]
t0, t1 = step[0][0], max(e[1] for e in step)
busy = sum(e[1] - e[0] for e in step)
print(f"last step: {len(step)} launches, span {(t1 - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us, idle {(t1 - t0 - busy) / 1e3:.1f} us")
gaps = []
for a, b in zip(step[:-1], step[1:]):
    g = b[0] - a[1]
    gaps.append((g, short(a[2]), short(b[2]), (a[1] - t0) / 1e3))
small = [g for g in gaps if g[0] <= 15000]
print(f"gaps <= 15 us: {len(small)}, total {sum(g[0] for g in small) / 1e3:.1f} us (mean {sum(g[0] for g in small) / max(1, len(small)) / 1e3:.2f} us)")
big = [g for g in gaps if g[0] > 15000]
print(f"gaps  > 15 us: {len(big)}, total {sum(g[0] for g in big) / 1e3:.1f} us")
for g in big:
    print(f"   {g[0] / 1e3:7.1f} us at +{g[3]:8.1f} us  after {g[1]:38s} before {g[2]}")
by = {}
for e in step:
    k = short(e[2]); c = by.setdefault(k, [0, 0]); c[0] += 1; c[1] += e[1] - e[0]
for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"   {k:40s} {c:4d} launches {t / 1e3:8.1f} us")
