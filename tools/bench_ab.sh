#!/bin/bash
# usage: tools/bench_ab.sh lib1.so lib2.so ...  -- whole-forward A/B of library builds on one box, two rounds
mkdir -p gpurun_out
: > gpurun_out/bench_ab.log
for round in 1 2; do
  for lib in "$@"; do
    SLSMI_LIB=$PWD/$lib timeout -k 10 300 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_ab_tmp.log 2>&1 || { tail -5 gpurun_out/bench_ab_tmp.log; exit 1; }
    echo "$lib $(python tools/show_bench.py gpurun_out/bench_ab_tmp.log | head -1)" | tee -a gpurun_out/bench_ab.log
  done
done
