export TMPDIR=/tmp
t=gpurun_out/prof_train; mkdir -p $t
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $t -o tr -- python3 bench.py --mode train --steps 5 --warmup 2 > $t/log.txt 2>&1 || { tail -5 $t/log.txt; exit 1; }
python3 tools/summarize_prof.py $t/tr_kernel_stats.csv > $t/kernel_stats.md
python3 tools/train_step_prof.py $t/tr_kernel_trace.csv 45 > $t/one_step.md
rm -f $t/tr_kernel_trace.csv
head -32 $t/one_step.md
