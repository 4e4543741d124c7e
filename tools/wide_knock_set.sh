#!/bin/bash
# usage: tools/wide_knock_set.sh lib1.so lib2.so ... -- K-loop time of the 208 x 256 kernel per library build
mkdir -p gpurun_out
: > gpurun_out/wide_knock.log
for lib in "$@"; do
  SLSMI_LIB=$PWD/$lib timeout -k 10 120 python -u tools/wide_knock.py 2>&1 | tail -1 | tee -a gpurun_out/wide_knock.log || exit 1
done
