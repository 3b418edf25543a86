#!/bin/bash
# per-kernel durations of the three sorts (rocprofv3 --kernel-trace --stats over tools/time_sort.py)
out=$PWD/gpurun_out/sortprof
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/time_sort.py > $out/run.log 2>&1
echo "rc=$?"
f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
cp "$f" $out/kernel_stats.csv
cut -c1-230 $out/kernel_stats.csv | head -16
rm -rf $out/trace
