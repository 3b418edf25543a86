#!/bin/bash
mkdir -p gpurun_out/sort
timeout -k 10 600 python -m pytest tests/test_gpu_edge.py -x -q -m gpu -k "sort" > gpurun_out/sort/tests.log 2>&1
echo "sort tests rc=$?" | tee gpurun_out/sort/summary.txt
tail -n 4 gpurun_out/sort/tests.log
timeout -k 10 300 python tools/time_sort.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/sort/time.log
NNC_DIAG=1 timeout -k 10 300 python tools/trace_sort.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/sort/trace.log
