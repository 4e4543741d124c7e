export TMPDIR=/tmp
d=gpurun_out/prof_varlen; mkdir -p $d
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o r -- python3 bench.py --mode varlen --steps 2 --warmup 1 > $d/log.txt 2>&1 || { tail -5 $d/log.txt; exit 1; }
python3 tools/summarize_prof.py $d/r_kernel_stats.csv 25 > $d/kernel_stats.md
rm -f $d/r_kernel_trace.csv
cat $d/kernel_stats.md | head -34; tail -1 $d/log.txt | cut -c1-300
