#!/bin/bash
# first GPU call of round 4: new tests, then both bench configurations, then the rest of the suite
mkdir -p gpurun_out/r4a
python -m pytest tests/test_gpu_parity.py tests/test_gpu_storage.py -x -q -m gpu -k "headline or sparse or smaller_form or store_report" -s > gpurun_out/r4a/new_tests.log 2>&1
echo "new tests rc=$?" | tee -a gpurun_out/r4a/summary.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r4a/bench3.json 2> gpurun_out/r4a/bench3.err
echo "bench3 rc=$?" | tee -a gpurun_out/r4a/summary.txt
python bench.py --config 4 --steps 10 --warmup 3 > gpurun_out/r4a/bench4.json 2> gpurun_out/r4a/bench4.err
echo "bench4 rc=$?" | tee -a gpurun_out/r4a/summary.txt
python -m pytest tests -x -q -m gpu > gpurun_out/r4a/all_tests.log 2>&1
echo "all tests rc=$?" | tee -a gpurun_out/r4a/summary.txt
tail -n 5 gpurun_out/r4a/new_tests.log gpurun_out/r4a/all_tests.log
