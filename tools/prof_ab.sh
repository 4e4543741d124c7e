#!/bin/bash
# usage: tools/prof_ab.sh lib1.so lib2.so ... -- rocprofv3 kernel stats of the bench per library build
export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  mkdir -p gpurun_out/prof_$name
  SLSMI_LIB=$PWD/$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$name -o r -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_$name.log 2>&1 || { tail -5 gpurun_out/prof_$name.log; exit 1; }
  echo "== $name"; python tools/summarize_prof.py gpurun_out/prof_$name/r_kernel_stats.csv 7 | head -14
done
