#!/bin/bash
# usage: tools/prof_train_b.sh <batch>  -- kernel table of one training step at another batch size
export TMPDIR=/tmp
b=${1:-63}
t=gpurun_out/prof_train_b$b; mkdir -p $t
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $t -o tr -- python3 bench.py --mode train --batch $b --steps 4 --warmup 2 > $t/log.txt 2>&1 || { tail -5 $t/log.txt; exit 1; }
python3 tools/train_step_prof.py $t/tr_kernel_trace.csv 45 > $t/one_step.md
rm -f $t/tr_kernel_trace.csv
head -34 $t/one_step.md
