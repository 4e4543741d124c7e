#!/bin/bash
# usage: tools/prof_precision.sh f16|bf16   -- one-forward cut of the bench step with that operand format
export TMPDIR=/tmp; d=gpurun_out/prof_$1; mkdir -p $d
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $d -o r -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt --precision $1 > $d/log.txt 2>&1 || { tail -5 $d/log.txt; exit 1; }
python3 tools/one_forward.py $d/r_kernel_trace.csv > $d/one_forward.md; rm -f $d/r_kernel_trace.csv; head -24 $d/one_forward.md
