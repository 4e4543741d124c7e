#!/bin/bash
# usage: tools/gpu_pmc.sh "<counters>" <tag> -- <gemm_bench args>
export TMPDIR=/tmp
ctr="$1"; tag="$2"; shift 3
mkdir -p gpurun_out/pmc_$tag
timeout -s KILL 120 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -o p -- python3 tools/gemm_bench.py "$@" > gpurun_out/pmc_$tag/log.txt 2>&1
echo "rc=$?"; grep "TF" gpurun_out/pmc_$tag/log.txt
python3 tools/pmc_table.py gpurun_out/pmc_$tag/p_counter_collection.csv
