#!/bin/bash
# GPU pass: kernel parity, forward parity, short bench in both operand formats
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_kernels_gpu.py -q --timeout 180 --timeout-method thread -p no:cacheprovider > gpurun_out/t_kernels.log 2>&1
rc=$?; echo "kernels rc=$rc"; tail -5 gpurun_out/t_kernels.log
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 900 python -u -m pytest tests/test_forward_gpu.py -q -s --timeout 300 --timeout-method thread -p no:cacheprovider > gpurun_out/t_forward.log 2>&1
rc=$?; echo "forward rc=$rc"; grep -E "^\[|EER|passed|failed" gpurun_out/t_forward.log | tail -30
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 600 python -u bench.py --steps 5 --warmup 2 > gpurun_out/bench_f16.log 2>&1
rc=$?; echo "bench f16 rc=$rc"; tail -c 1500 gpurun_out/bench_f16.log | cut -c1-1500
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 600 python -u bench.py --steps 5 --warmup 2 --precision bf16 --no-cpu-baseline > gpurun_out/bench_bf16.log 2>&1
rc=$?; echo "bench bf16 rc=$rc"; tail -c 600 gpurun_out/bench_bf16.log | cut -c1-600
exit $rc
