#!/usr/bin/env python3
"""The training step's GEMM shapes (M = 1005 rows) on the 128-row kernel with 128- and 64-column tiles, K cut by the
library where the tiles are few (its own rule): time per launch including the summing kernel of a cut."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slsmi  # noqa: E402,F401
from slsmi import ops  # noqa: E402

DEV = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1005
for (N, K, mode) in ((4096, 1024, 1), (3072, 1024, 0), (1024, 1024, 3), (1024, 4096, 3), (1024, 3072, 2)):
    a = (torch.randn(M, K, device=DEV) * 0.5).half()
    w = (torch.randn(N, K, device=DEV) * 0.05).half()
    bias = torch.zeros(N, device=DEV)
    out = torch.zeros(M, N, dtype=torch.float16 if mode in (0, 1) else torch.float32, device=DEV)
    row = []
    for bn in (128, 64):
        us = ops.time_gemm_bf16(a, w, bias, out, mode, M, K, K, 200, bn=bn) * 1e6
        row.append(f"bn {bn}: {us:6.2f} us {2.0 * M * N * K / us / 1e6:6.0f} TF")
    print(f"M {M} N {N} K {K} mode {mode}:  " + "   ".join(row), flush=True)
