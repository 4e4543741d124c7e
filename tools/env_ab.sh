#!/bin/bash
# usage: tools/env_ab.sh VAR v1 v2 ... -- whole-forward A/B of one environment switch on one box, two rounds
var=$1; shift
mkdir -p gpurun_out
: > gpurun_out/env_ab.log
for round in 1 2; do
  for v in "$@"; do
    env $var=$v timeout -k 10 300 python -u bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-alt > gpurun_out/env_ab_tmp.log 2>&1 || { tail -5 gpurun_out/env_ab_tmp.log; exit 1; }
    echo "$var=$v $(python tools/show_bench.py gpurun_out/env_ab_tmp.log 2>/dev/null | head -1)" | tee -a gpurun_out/env_ab.log
  done
done
