#!/bin/bash
# usage: tools/gemm_ab_set.sh lib1.so lib2.so ...   (A/B of GEMM builds on the forward's shapes)
mkdir -p gpurun_out
{
timeout -k 10 200 python -u tools/gemm_ab.py --libs "$@" --mode 0 12864,3072,1024 425984,512,1536,1024 53248,512,1536,1024 || exit 1
timeout -k 10 200 python -u tools/gemm_ab.py --libs "$@" --mode 1 12864,4096,1024 || exit 1
timeout -k 10 200 python -u tools/gemm_ab.py --libs "$@" --mode 3 12864,1024,1024 12864,1024,4096 || exit 1
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gemm_ab.log
