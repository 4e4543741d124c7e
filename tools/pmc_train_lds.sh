#!/bin/bash
# LDS counters of the training step's kernels (one rocprofv3 pass): bank conflicts against LDS activity.
export TMPDIR=/tmp
d=gpurun_out/pmc_train_lds; mkdir -p $d
timeout -s KILL 170 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $d -o p -- python3 bench.py --mode train --steps 2 --warmup 1 > $d/log.txt 2>&1
rc=$?; echo "rc=$rc"; [ $rc -ne 0 ] && { tail -5 $d/log.txt; exit $rc; }
python3 tools/pmc_bench_table.py $d/p_counter_collection.csv > $d/summary.txt
rm -f $d/p_counter_collection.csv $d/p_kernel_trace.csv
grep -E "gemm16_kernel<128|gemm_tn_kernel|attn_bwd|gemm16_wide" $d/summary.txt | cut -c1-200
