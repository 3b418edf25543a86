#!/bin/bash
# variants of the radix pass (diagnostics build, rebuilt on the box): NNC_EXTRA_CXXFLAGS per line
out=$PWD/gpurun_out/sortgeom
mkdir -p $out; : > $out/geom.log
while read -r tile flags; do
  [ -z "$tile" ] && continue
  echo "=== tile $tile: $flags" | tee -a $out/geom.log
  touch neural_network_compression_amd/csrc/nnc_sort.hip
  NNC_DIAG=1 NNC_EXTRA_CXXFLAGS="$flags" python -m neural_network_compression_amd.build > $out/build.log 2>&1 || { tail -5 $out/build.log | tee -a $out/geom.log; continue; }
  NNC_DIAG=1 timeout -k 10 120 python tools/time_sort.py 2>&1 | grep -v amdgpu.ids | tee -a $out/geom.log
  OS_TILE=$tile NNC_DIAG=1 timeout -k 10 120 python tools/trace_sort.py 2>&1 | grep -v amdgpu.ids | grep "pass 1" -A2 | tee -a $out/geom.log
done <<'CFG'
8192 -DOS_THREADS=512 -DOS_ITEMS=16 -DOS_BLOCKS_PER_CU=3
16384 -DOS_THREADS=1024 -DOS_ITEMS=16 -DOS_BLOCKS_PER_CU=1
12288 -DOS_THREADS=768 -DOS_ITEMS=16 -DOS_BLOCKS_PER_CU=2
CFG
