#!/bin/bash
# usage: tools/prof.sh <tag> <args to run_lloyd.py...>   (run on the GPU box from the repo root)
set -e
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
python tools/run_lloyd.py "$@" > $out/plain.log 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python tools/run_lloyd.py "$@" > $out/trace.log 2>&1 || true
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $out/pmc1 -- python tools/run_lloyd.py "$@" > $out/pmc1.log 2>&1 || true
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAVES --kernel-trace --output-format csv -d $out/pmc2 -- python tools/run_lloyd.py "$@" > $out/pmc2.log 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc3 -- python tools/run_lloyd.py "$@" > $out/pmc3.log 2>&1 || true
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc4 -- python tools/run_lloyd.py "$@" > $out/pmc4.log 2>&1 || true
find $out -name "*.csv" | head -30
python tools/summarize_prof.py $out
