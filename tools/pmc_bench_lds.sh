#!/bin/bash
# LDS counters of the scoring forward's kernels (one rocprofv3 pass): bank conflicts against LDS activity.
export TMPDIR=/tmp
d=gpurun_out/pmc_bench_lds; mkdir -p $d
timeout -s KILL 170 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $d -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt > $d/log.txt 2>&1
rc=$?; echo "rc=$rc"; [ $rc -ne 0 ] && { tail -5 $d/log.txt; exit $rc; }
python3 tools/pmc_bench_table.py $d/p_counter_collection.csv > $d/summary.txt
rm -f $d/p_counter_collection.csv $d/p_kernel_trace.csv
grep -E "BANK_CONFLICT|IDX_ACTIVE" $d/summary.txt | grep -E "gemm16|attention|posconv|conv0" | cut -c1-160
