#!/bin/bash
# bench step of build variants, one per line of flags (library rebuilt on the box; the last line should be the plain build)
out=$PWD/gpurun_out/variant
mkdir -p $out; : > $out/variant.log
while read -r flags; do
  echo "=== flags: $flags" | tee -a $out/variant.log
  touch neural_network_compression_amd/csrc/*.hip
  NNC_EXTRA_CXXFLAGS="$flags" python -m neural_network_compression_amd.build > $out/build.log 2>&1 || { tail -5 $out/build.log | tee -a $out/variant.log; continue; }
  for i in 1 2; do
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-streaming-leg --dump-durations 2> $out/dump.txt | python -c "
import json,sys,re,statistics as st
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
line=[l for l in open('$out/dump.txt') if l.startswith('iteration kernels')][-1]
pairs=[(int(a),float(b)) for a,b in re.findall(r'(\\d+):(\\d+)', line)]
B=[u for t,u in pairs if t==1]; F=[u for t,u in pairs if t==5]
print('ms_per_step %.4f' % d['ms_per_step'], 'iters', d['config']['lloyd_iterations'], 'bounds avg %.1f median %.1f last20 %.1f | fin avg %.1f median %.1f | reloc %.1f | assign %.1f us' % (k['k_bounds']['avg_ms']*1e3, st.median(B), st.mean(B[-20:]), k['k_finalize']['avg_ms']*1e3, st.median(F), k['k_reloc_*']['avg_ms']*1e3, k['k_assign<labels>']['avg_ms']*1e3))" | tee -a $out/variant.log
  done
done
