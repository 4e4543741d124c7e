#!/usr/bin/env python3
"""Time of the 128-row GEMM kernel at the training step's row count (M = 1005) against the depth K: the slope is the
cost of a 64-deep K-tile, the intercept what a launch costs before and after its K loop."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slsmi  # noqa: E402,F401
from slsmi import ops  # noqa: E402

DEV = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1005
for N in (4096, 1024):
    for K in (128, 256, 512, 1024, 2048, 4096):
        a = (torch.randn(M, K, device=DEV) * 0.5).half()
        w = (torch.randn(N, K, device=DEV) * 0.05).half()
        bias = torch.zeros(N, device=DEV)
        out = torch.empty(M, N, dtype=torch.float16, device=DEV)
        us = ops.time_gemm_bf16(a, w, bias, out, 0, M, K, K, 200, bn=128) * 1e6
        print(f"M {M} N {N:5d} K {K:5d}: {us:7.2f} us  {K // 64:3d} K-tiles  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
