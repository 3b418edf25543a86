"""Kernel-trace CSV -> the launches of the LAST fit (from its k_km_init on): name, duration, gap before it."""
import csv, glob, os, sys
d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
short = lambda n: n.replace("void ", "").split("(")[0][:34]
starts = [i for i, e in enumerate(ev) if "k_km_init" in e[2]]
fit = ev[starts[-1]:]
t0 = fit[0][0]
prev_end = fit[0][0]
tot = {}
for s, e, name in fit:
    k = short(name)
    c = tot.setdefault(k, [0, 0, 0]); c[0] += 1; c[1] += e - s; c[2] += max(0, s - prev_end)
    if len(sys.argv) > 2: print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {k}")
    prev_end = max(prev_end, e)
span = (max(e for _, e, _ in fit) - t0) / 1e3
print(f"span {span:.1f} us, launches {len(fit)}")
for k, (c, t, g) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:36s} {c:4d} launches  busy {t / 1e3:8.1f} us  gaps-before {g / 1e3:8.1f} us  mean dur {t / c / 1e3:6.2f}")
