#!/bin/bash
# after a change to the K-sized kernels: the trajectory tests, the phase stamps, the bench step
out=$PWD/gpurun_out/kchain
mkdir -p $out
timeout -k 10 1000 python -m pytest ${KCHAIN_TESTS:-tests/test_gpu_lloyd.py tests/test_gpu_parity.py tests/test_gpu_edge.py} -x -q -m gpu > $out/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -n 3 $out/tests.log
[ $rc -ne 0 ] && exit $rc
NNC_DIAG=1 timeout -k 10 300 python tools/trace_finalize.py 2>&1 | grep -v amdgpu.ids | tee $out/trace.log | grep "25000000"
for i in 1 2; do
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-streaming-leg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print('ms_per_step %.4f' % d['ms_per_step'], 'iters', d['config']['lloyd_iterations'], 'bounds %.1f fin %.1f reloc %.1f us' % (k['k_bounds']['avg_ms']*1e3, k['k_finalize']['avg_ms']*1e3, k['k_reloc_*']['avg_ms']*1e3))" | tee -a $out/bench.log
done
