#!/bin/bash
mkdir -p gpurun_out/chain
timeout -k 10 900 python -m pytest tests/test_gpu_lloyd.py tests/test_gpu_stress.py -x -q -m gpu > gpurun_out/chain/tests1.log 2>&1
echo "tests1 rc=$?" | tee gpurun_out/chain/summary.txt; tail -n 5 gpurun_out/chain/tests1.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-streaming-leg --dump-durations > gpurun_out/chain/bench.json 2> gpurun_out/chain/bench.err
echo "bench rc=$?" | tee -a gpurun_out/chain/summary.txt
python -c "
import json; d=json.loads(open('gpurun_out/chain/bench.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'iters', d['config']['lloyd_iterations'], 'reloc', d['config']['relocations']); print({k:(round(v['ms_per_step'],3), v['launches_per_step']) for k,v in d['kernels'].items()})"
grep "iteration kernels" gpurun_out/chain/bench.err | cut -c1-500
