#!/bin/bash
# usage: tools/gemm_ab_probe.sh "modes" lib1.so lib2.so ...
mkdir -p gpurun_out
modes=$1; shift
{
for mode in $modes; do
timeout -k 10 200 python -u tools/gemm_ab.py --libs "$@" --mode $mode 8192,1024,1024 16384,1024,1024 12864,1024,1024 12864,3072,1024 12864,4096,1024 8192,1024,4096 || exit 1
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gemm_probe.log
