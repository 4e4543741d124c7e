#!/bin/bash
# samples GPU clock and power while the bench loops
mkdir -p gpurun_out
timeout -k 10 200 python -u bench.py --steps 600 --warmup 3 --no-cpu-baseline > gpurun_out/clock_bench.log 2>&1 &
pid=$!
sleep 25
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk|fclk" | tr -s ' ' | paste -sd'|'; sleep 1; done | tee gpurun_out/clock_probe.log
wait $pid
python tools/show_bench.py gpurun_out/clock_bench.log | head -1
echo idle; sleep 3; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr -s ' ' | paste -sd'|'
