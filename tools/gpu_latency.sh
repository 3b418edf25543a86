#!/bin/bash
# what a launch costs before its first useful load: phase stamps of the two K-sized kernels, then the bench step with the kernel
# arguments in device memory (HIP_FORCE_DEV_KERNARG) and preloaded into SGPRs (-amdgpu-kernarg-preload-count)
out=$PWD/gpurun_out/latency
mkdir -p $out; : > $out/latency.log
run_bench() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-streaming-leg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print('ms_per_step %.4f' % d['ms_per_step'], 'bounds %.1f fin %.1f reloc %.1f us' % (k['k_bounds']['avg_ms']*1e3, k['k_finalize']['avg_ms']*1e3, k['k_reloc_*']['avg_ms']*1e3))"; }
NNC_DIAG=1 timeout -k 10 300 python tools/trace_finalize.py 2>&1 | grep -v amdgpu.ids | tee -a $out/latency.log
for v in "" 0 1; do
  echo "=== HIP_FORCE_DEV_KERNARG='$v'" | tee -a $out/latency.log
  if [ -z "$v" ]; then run_bench; else HIP_FORCE_DEV_KERNARG=$v run_bench; fi 2>&1 | tee -a $out/latency.log
done
for c in 8 16; do
  echo "=== -amdgpu-kernarg-preload-count=$c" | tee -a $out/latency.log
  NNC_EXTRA_CXXFLAGS="-mllvm -amdgpu-kernarg-preload-count=$c" python -m neural_network_compression_amd.build > $out/build.log 2>&1 || { tail -5 $out/build.log | tee -a $out/latency.log; continue; }
  run_bench 2>&1 | tee -a $out/latency.log
  HIP_FORCE_DEV_KERNARG=1 run_bench 2>&1 | tee -a $out/latency.log
done
