#!/bin/bash
# dynamic instruction counts per wave of the K-sized kernels (a lone wave issues one instruction every ~4 clocks: 1.7 ns each)
export TMPDIR=/tmp
out=$PWD/gpurun_out/insts
rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $out/pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-streaming-leg > /dev/null 2> $out/err.txt
f=$(find $out/pmc -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY' | tee $out/insts.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    nm = r["Kernel_Name"].replace("void ", "").split("(")[0][:44]
    acc[nm][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"{'kernel':46s} launches  waves  per wave: VALU   SALU   SMEM   VMEM    LDS   total   (median launch)")
for nm, c in sorted(acc.items(), key=lambda kv: -len(kv[1].get("SQ_WAVES", []))):
    w = sorted(c.get("SQ_WAVES", [0])); n = len(w)
    if not n or w[n // 2] == 0: continue
    def med(k):
        per = sorted(a / b for a, b in zip(c.get(k, [0] * n), c["SQ_WAVES"]) if b)
        return per[len(per) // 2] if per else 0
    v = [med(k) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM", "SQ_INSTS_LDS")]
    print(f"{nm:46s} {n:6d} {w[n // 2]:6.0f}        {v[0]:7.0f}{v[1]:7.0f}{v[2]:7.0f}{v[3]:7.0f}{v[4]:7.0f}{sum(v):8.0f}")
# per launch, the K-sized kernels of ONE step in launch order: what an event costs in instructions
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
by = collections.defaultdict(dict)
for r in rows:
    by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    by[int(r["Dispatch_Id"])]["name"] = r["Kernel_Name"].replace("void ", "").split("(")[0][:14]
ids = sorted(by)
inits = [i for i in ids if by[i]["name"].startswith("k_km_init")]
if len(inits) >= 2:
    seq = [i for i in ids if inits[-2] < i < inits[-1]]
    out_ = []
    for i in seq:
        d = by[i]
        if d["name"].startswith(("k_bounds", "k_finalize", "k_reloc")) and d.get("SQ_WAVES"):
            tot = sum(d.get(k, 0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM", "SQ_INSTS_LDS"))
            tag = {"k_bounds<8>": "B", "k_finalize<102": "F", "k_reloc_head": "h", "k_reloc_dist": "d", "k_reloc_select": "s"}.get(d["name"], d["name"])
            out_.append(f"{tag}{tot / d['SQ_WAVES']:.0f}")
    print("instructions per wave, launch by launch (B k_bounds, F k_finalize, h / d / s the relocation chain):")
    print(" ".join(out_))
PY
rm -rf $out/pmc
