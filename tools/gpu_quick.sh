#!/bin/bash
# a quick look after a kernel change: the tests that pin it, then the bench line's per-kernel numbers
mkdir -p gpurun_out/quick
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py tests/test_gpu_lloyd.py -x -q -m gpu -k "${1:-moments or prune or fit_matches or layers_dealt or full_size or iteration}" > gpurun_out/quick/tests.log 2>&1
echo "tests rc=$?"; tail -n 4 gpurun_out/quick/tests.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-streaming-leg > gpurun_out/quick/bench.json 2> gpurun_out/quick/bench.err
echo "bench rc=$?"
python -c "
import json; d=json.loads(open('gpurun_out/quick/bench.json').read().strip().splitlines()[-1]); print('ms_per_step', round(d['ms_per_step'],4), 'iters', d['config']['lloyd_iterations'], 'reloc', d['config']['relocations']); print({k:(round(v['ms_per_step'],3), v['launches_per_step'], round(v['avg_ms']*1e3,1), round(v.get('frac_of_hbm_peak',0),3)) for k,v in d['kernels'].items()})"
