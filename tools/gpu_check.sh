#!/bin/bash
# usage: tools/gpu_check.sh [kernels] [forward] [bench] [bench_bf16] [prof]
mkdir -p gpurun_out
for step in "$@"; do
  case $step in
    kernels)
      timeout -k 10 900 python -u -m pytest tests/test_kernels_gpu.py -q -x --timeout 180 --timeout-method thread -p no:cacheprovider > gpurun_out/t_kernels.log 2>&1
      rc=$?; echo "kernels rc=$rc"; tail -4 gpurun_out/t_kernels.log | cut -c1-300
      [ $rc -ne 0 ] && exit $rc ;;
    forward)
      timeout -k 10 900 python -u -m pytest tests/test_forward_gpu.py -q -x -s --timeout 300 --timeout-method thread -p no:cacheprovider > gpurun_out/t_forward.log 2>&1
      rc=$?; echo "forward rc=$rc"; grep -oE "\[(2|24) layers [a-z]+ [a-z0-9]+\]|score err [0-9.e-]+|\[B=64\].*|EER.*|[0-9]+ passed.*|[0-9]+ failed.*" gpurun_out/t_forward.log | paste -sd' ' | cut -c1-1500
      [ $rc -ne 0 ] && exit $rc ;;
    bench)
      timeout -k 10 600 python -u bench.py --steps 50 --warmup 3 > gpurun_out/bench_f16.log 2>&1
      rc=$?; echo "bench rc=$rc"; python tools/show_bench.py gpurun_out/bench_f16.log
      [ $rc -ne 0 ] && exit $rc ;;
    bench_bf16)
      timeout -k 10 600 python -u bench.py --steps 10 --warmup 3 --precision bf16 --no-cpu-baseline > gpurun_out/bench_bf16.log 2>&1
      rc=$?; echo "bench bf16 rc=$rc"; python tools/show_bench.py gpurun_out/bench_bf16.log
      [ $rc -ne 0 ] && exit $rc ;;
    prof)
      export TMPDIR=/tmp; mkdir -p gpurun_out/prof
      timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r1 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt > gpurun_out/prof_bench.log 2>&1
      rc=$?; echo "prof rc=$rc"; python tools/summarize_prof.py gpurun_out/prof/r1_kernel_stats.csv 7 | head -16
      [ $rc -ne 0 ] && exit $rc ;;
  esac
done
exit 0
