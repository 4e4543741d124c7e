#!/bin/bash
mkdir -p gpurun_out
{
timeout -k 10 200 python -u tools/gemm_ab.py --rounds 15 --libs "$@" --mode 1 12864,4096,1024 || exit 1
timeout -k 10 200 python -u tools/gemm_ab.py --rounds 15 --libs "$@" --mode 0 12864,3072,1024 || exit 1
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gemm_ab_m1.log
