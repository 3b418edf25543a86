#!/bin/bash
mkdir -p gpurun_out/r4b
python -m pytest tests -x -q -m gpu > gpurun_out/r4b/all_tests.log 2>&1
echo "all tests rc=$?" | tee gpurun_out/r4b/summary.txt
tail -n 4 gpurun_out/r4b/all_tests.log
bash tools/make_profiles.sh > gpurun_out/r4b/profiles.log 2>&1
echo "profiles rc=$?" | tee -a gpurun_out/r4b/summary.txt
tail -n 6 gpurun_out/r4b/profiles.log | cut -c1-400
