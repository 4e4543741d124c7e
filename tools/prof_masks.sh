#!/bin/bash
# usage: tools/prof_masks.sh "<split>,<fp32>" tag   -- kernel trace of the bench step under that conv precision setting
export TMPDIR=/tmp; mkdir -p gpurun_out/prof_$2
SLSMI_CONV_PRECISION=$1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$2 -o r -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt > gpurun_out/prof_$2/log.txt 2>&1
rc=$?; echo "prof $2 rc=$rc"; python3 tools/summarize_prof.py gpurun_out/prof_$2/r_kernel_stats.csv 7 > gpurun_out/prof_$2/summary.md; python3 tools/one_forward.py gpurun_out/prof_$2/r_kernel_trace.csv --seq > gpurun_out/prof_$2/one_forward.md; rm -f gpurun_out/prof_$2/r_kernel_trace.csv
exit $rc
