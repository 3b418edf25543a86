#!/bin/bash
# rocprofv3 kernel-trace + stats of the benchmark itself (run on the GPU box from the repo root)
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_bench
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/bench.log 2>&1
f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms over {sum(int(r['Calls']) for r in rows)} launches")
for r in rows[:40]:
    print(f"  {r['Name'][:70]:70s} calls={r['Calls']:>6s} total_ms={float(r['TotalDurationNs'])/1e6:8.3f} avg_us={float(r['AverageNs'])/1e3:9.2f} pct={r['Percentage']}")
PY
tail -2 $out/bench.log | cut -c1-400
