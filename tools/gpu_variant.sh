#!/bin/bash
# bench step of build variants, one per line of flags (library rebuilt on the box; the last line should be the plain build)
out=$PWD/gpurun_out/variant
mkdir -p $out; : > $out/variant.log
while read -r flags; do
  echo "=== flags: $flags" | tee -a $out/variant.log
  touch neural_network_compression_amd/csrc/nnc_hip.hip
  NNC_EXTRA_CXXFLAGS="$flags" python -m neural_network_compression_amd.build > $out/build.log 2>&1 || { tail -5 $out/build.log | tee -a $out/variant.log; continue; }
  for i in 1 2; do
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-streaming-leg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print('ms_per_step %.4f' % d['ms_per_step'], 'iters', d['config']['lloyd_iterations'], 'bounds %.1f fin %.1f reloc %.1f us' % (k['k_bounds']['avg_ms']*1e3, k['k_finalize']['avg_ms']*1e3, k['k_reloc_*']['avg_ms']*1e3))" | tee -a $out/variant.log
  done
done
