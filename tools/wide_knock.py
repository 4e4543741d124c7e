#!/usr/bin/env python3
"""K-loop time of the 208 x 256 kernel (M = 12864, N = 1024, 16-bit output) at K = 1024 and 4096; run once per
diagnostic library (SLSMI_LIB=diag/libslsmi_k<n>.so) to see what each part of the loop costs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slsmi  # noqa: E402,F401
from slsmi import _lib, ops  # noqa: E402

DEV = torch.device("cuda:0")
res = {}
for K in (1024, 4096):
    M, N = 12864, 1024
    a = (torch.randn(M, K, device=DEV) * 0.5).half()
    w = (torch.randn(N, K, device=DEV) * K ** -0.5).half()
    b = torch.zeros(N, device=DEV)
    out = torch.zeros(M, N, dtype=torch.float16, device=DEV)
    ts = [ops.time_gemm_bf16(a, w, b, out, _lib.GEMM_BF16, M, K, K, reps=20, bn=256) for _ in range(3)]
    res[K] = min(ts)
print(f"{os.environ.get('SLSMI_LIB', 'as built'):28s} K=1024 {1e6 * res[1024]:6.1f} us  K=4096 {1e6 * res[4096]:6.1f} us  "
      f"per K-tile {1e6 * (res[4096] - res[1024]) / 48:5.3f} us", flush=True)
