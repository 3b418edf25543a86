#!/bin/bash
# where a radix pass spends its time: the sort with parts of k_os_pass switched off (diagnostics build; results are wrong on purpose)
out=$PWD/gpurun_out/sortabl
mkdir -p $out
NNC_DIAG=1 python -m neural_network_compression_amd.build > $out/build.log 2>&1 || { tail -5 $out/build.log; exit 1; }
for a in 0 1 2 4 3 7; do
  echo "== NNC_OS_ABLATE=$a" | tee -a $out/abl.log
  NNC_DIAG=1 NNC_OS_ABLATE=$a timeout -k 10 120 python tools/time_sort.py 2>&1 | grep -v amdgpu.ids | tee -a $out/abl.log
done
