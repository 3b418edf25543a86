#!/bin/bash
export TMPDIR=/tmp
out=$PWD/gpurun_out/timeline
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-streaming-leg > $out/bench.json 2>/dev/null
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py "$f" 2 | tee $out/timeline.txt
python tools/step_timeline.py "$f" 3 | head -3 | tee -a $out/timeline.txt
cp "$f" $out/kernel_trace.csv; gzip -f $out/kernel_trace.csv; rm -rf $out/trace
