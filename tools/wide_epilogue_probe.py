#!/usr/bin/env python3
"""Fixed cost of the 208 x 256 kernel's epilogues: M = 12864, N = 1024 at K = 1024 and K = 4096 in the 16-bit,
fp32 and fp32 + residual output modes (isolated launches, back to back)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slsmi  # noqa: E402,F401
from slsmi import _lib, ops  # noqa: E402

DEV = torch.device("cuda:0")
M, N = 12864, 1024
for mode, name in ((_lib.GEMM_BF16, "f16 out"), (_lib.GEMM_F32, "fp32 out"), (_lib.GEMM_F32_RESID, "fp32 + resid")):
    res = {}
    for K in (1024, 4096):
        a = (torch.randn(M, K, device=DEV) * 0.5).half()
        w = (torch.randn(N, K, device=DEV) * K ** -0.5).half()
        b = torch.zeros(N, device=DEV)
        out = torch.zeros(M, N, dtype=torch.float16 if mode == _lib.GEMM_BF16 else torch.float32, device=DEV)
        ts = [ops.time_gemm_bf16(a, w, b, out, mode, M, K, K, reps=20, bn=256) for _ in range(3)]
        res[K] = min(ts)
    kt = (res[4096] - res[1024]) / 48
    print(f"{name:14s} K=1024 {1e6 * res[1024]:6.1f} us  K=4096 {1e6 * res[4096]:6.1f} us  per K-tile {1e6 * kt:5.3f} us  "
          f"fixed {1e6 * (res[1024] - 16 * kt):5.1f} us", flush=True)
