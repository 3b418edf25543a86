#!/usr/bin/env python3
"""Where one bench step spends its time, from a rocprofv3 kernel trace (csv): kernels in launch order with start / end stamps,
the idle time between them, and the totals by kind.  Usage: step_timeline.py <kernel_trace.csv> [step index counted from the end]"""
import csv, sys, collections

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# a step starts with the pruning statistics: the first k_chunk_sums after a k_assign<1 ...> (or the trace's beginning)
starts = [i for i, nm in enumerate(names) if "k_chunk_sums" in nm and (i == 0 or "k_chunk_sums" not in names[i - 1]) and any("k_assign" in x or "k_huff" in x or "k_bincount" in x for x in names[max(0, i - 6): i])]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 2
s0 = starts[-which - 1]; s1 = starts[-which]
step = rows[s0:s1]
t0 = int(step[0]["Start_Timestamp"]); t1 = int(step[-1]["End_Timestamp"])
busy = collections.defaultdict(float); cnt = collections.Counter()
gaps = []
prev_end = None
for r in step:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"].replace("void ", "").split("(")[0][:40]
    busy[nm] += (en - st) / 1e3; cnt[nm] += 1
    if prev_end is not None:
        gaps.append(((st - prev_end) / 1e3, nm, prev_nm))
    prev_end = max(prev_end or 0, en); prev_nm = nm
print(f"step of {len(step)} launches: {(t1 - t0) / 1e3:.1f} us from first start to last end; kernels busy {sum(busy.values()):.1f} us; gaps {sum(g[0] for g in gaps):.1f} us")
for nm, b in sorted(busy.items(), key=lambda kv: -kv[1]):
    print(f"  {nm:42s} {cnt[nm]:4d} x {b / cnt[nm]:7.2f} = {b:8.1f} us")
h = collections.Counter()
for g, nm, pn in gaps:
    h["<1" if g < 1 else "1-2" if g < 2 else "2-3" if g < 3 else "3-5" if g < 5 else "5-10" if g < 10 else ">=10"] += 1
print("gap histogram (us):", dict(h))
big = sorted(gaps, reverse=True)[:15]
print("largest gaps:", ", ".join(f"{g:.1f} before {nm} (after {pn})" for g, nm, pn in big))
bygap = collections.defaultdict(float)
for g, nm, pn in gaps:
    bygap[nm] += g
print("gap time in front of:", ", ".join(f"{nm} {v:.0f}" for nm, v in sorted(bygap.items(), key=lambda kv: -kv[1])[:8]))
