#!/bin/bash
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_clock
mkdir -p $out
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/a -- python tools/run_lloyd.py --iters 30 --grid-log2 14 --rep-log2 2 > $out/a.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/b -- python tools/run_lloyd.py --iters 30 --grid-log2 14 --rep-log2 2 --accum-only > $out/b.log 2>&1
python - <<'PY'
import csv, glob, os
out = os.path.join(os.getcwd(), "gpurun_out/prof_clock")
for sub in ("a", "b"):
    cc = glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(cc)) if "k_assign" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
    kt = glob.glob(os.path.join(out, sub, "**", "*kernel_trace.csv"), recursive=True)[0]
    dur = {}
    for r in csv.DictReader(open(kt)):
        if "k_assign" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    vals = []
    for r in rows:
        d = dur.get(r["Dispatch_Id"])
        if d: vals.append((float(r["Counter_Value"]) / 8 / d, d))
    vals = vals[5:]
    print(sub, "n=", len(vals), "clock GHz median", sorted(v[0] for v in vals)[len(vals)//2], "dur ns median", sorted(v[1] for v in vals)[len(vals)//2])
PY
grep k_assign $out/a.log $out/b.log
