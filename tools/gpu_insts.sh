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
PY
rm -rf $out/pmc
