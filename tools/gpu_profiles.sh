#!/bin/bash
# The round's profile set on one box: kernel statistics and a one-forward cut of the bench step, the counter passes,
# the training step.  usage: tools/gpu_profiles.sh   -> gpurun_out/prof_final/..., gpurun_out/pmcb_*/summary.txt
export TMPDIR=/tmp
d=gpurun_out/prof_final; mkdir -p $d
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o r -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt > $d/log.txt 2>&1 || { tail -5 $d/log.txt; exit 1; }
python3 tools/summarize_prof.py $d/r_kernel_stats.csv 12 > $d/kernel_stats.md
python3 tools/one_forward.py $d/r_kernel_trace.csv --seq > $d/one_forward.md
rm -f $d/r_kernel_trace.csv
echo "bench profile done"
bash tools/gpu_pmc_bench.sh || exit 1
echo "counter passes done"
t=gpurun_out/prof_train; mkdir -p $t
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $t -o tr -- python3 bench.py --mode train --steps 5 --warmup 2 > $t/log.txt 2>&1 || { tail -5 $t/log.txt; exit 1; }
python3 tools/summarize_prof.py $t/tr_kernel_stats.csv > $t/kernel_stats.md
python3 tools/train_step_prof.py $t/tr_kernel_trace.csv 45 > $t/one_step.md
rm -f $t/tr_kernel_trace.csv
echo "train profile done"
bash tools/prof_train_sls.sh && echo "sls train profile done"
