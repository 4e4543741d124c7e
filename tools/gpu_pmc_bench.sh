#!/bin/bash
# Counter passes over the bench workload, one rocprofv3 run per pass (the hardware cannot hold
# them together).  usage: tools/gpu_pmc_bench.sh   -> gpurun_out/pmcb_<pass>/p_counter_collection.csv
export TMPDIR=/tmp
run() {
  tag=$1; ctr=$2
  mkdir -p gpurun_out/pmcb_$tag
  timeout -s KILL 170 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmcb_$tag -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt > gpurun_out/pmcb_$tag/log.txt 2>&1
  rc=$?; echo "pass $tag rc=$rc"; [ $rc -ne 0 ] && { tail -5 gpurun_out/pmcb_$tag/log.txt; exit $rc; }
  # keep only what the summary needs (the raw csv is large)
  python3 tools/pmc_bench_table.py gpurun_out/pmcb_$tag/p_counter_collection.csv > gpurun_out/pmcb_$tag/summary.txt
  rm -f gpurun_out/pmcb_$tag/p_counter_collection.csv gpurun_out/pmcb_$tag/p_kernel_trace.csv
}
run fetch "FETCH_SIZE" && run write "WRITE_SIZE" && run mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
