#!/bin/bash
# variants of the streaming kernels (diagnostics build rebuilt on the box): NNC_EXTRA_CXXFLAGS and the grid knob per line
out=$PWD/gpurun_out/assign
mkdir -p $out; : > $out/assign.log
while read -r q flags; do
  [ -z "$q" ] && continue
  echo "=== grid quarters $q, flags: $flags" | tee -a $out/assign.log
  touch neural_network_compression_amd/csrc/nnc_hip.hip
  NNC_DIAG=1 NNC_EXTRA_CXXFLAGS="$flags" python -m neural_network_compression_amd.build > $out/build.log 2>&1 || { tail -5 $out/build.log | tee -a $out/assign.log; continue; }
  NNC_DIAG=1 NNC_KM_GRID_QUARTERS=$q timeout -k 10 200 python tools/time_assign.py 2>&1 | grep -v amdgpu.ids | tee -a $out/assign.log
done <<'CFG'
4 -DKM_RING=4
4 -DKM_NT_LOADS
4 -DKM_KU_FAST
4 -DKM_RING=8
4 -DKM_RING=2
CFG
for q in 2 3 6 8; do
  echo "=== grid quarters $q (last build)" | tee -a $out/assign.log
  NNC_DIAG=1 NNC_KM_GRID_QUARTERS=$q timeout -k 10 200 python tools/time_assign.py 2>&1 | grep -v amdgpu.ids | tee -a $out/assign.log
done
