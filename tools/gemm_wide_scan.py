#!/usr/bin/env python3
"""Times the 208 x 256 kernel (bn = 256) and the 256 x 128 kernel (bn = 1) over tile counts per CU and K depths."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slsmi  # noqa: E402,F401
from slsmi import _lib, ops  # noqa: E402

DEV = torch.device("cuda:0")
cases = []
for N in (1024, 2048, 3072, 4096):
    cases.append((12864, N, 1024, _lib.GEMM_BF16))
for K in (256, 512, 2048, 4096):
    cases.append((12864, 1024, K, _lib.GEMM_BF16))
cases += [(12864, 1024, 1024, _lib.GEMM_F32), (12864, 1024, 1024, _lib.GEMM_F32_RESID), (12864, 4096, 1024, _lib.GEMM_BF16_GELU),
          (12864, 4096, 1024, _lib.GEMM_F32_RELU)]
for (M, N, K, mode) in cases:
    a = (torch.randn(M, K, device=DEV) * 0.5).half()
    w = (torch.randn(N, K, device=DEV) * K ** -0.5).half()
    b = torch.zeros(N, device=DEV)
    odt = torch.float16 if mode in (_lib.GEMM_BF16, _lib.GEMM_BF16_GELU) else torch.float32
    out = torch.zeros(M, N, dtype=odt, device=DEV)
    line = f"{M}x{N}x{K} mode {mode}:"
    for bn in (1, 256, 1, 256):
        sec = ops.time_gemm_bf16(a, w, b, out, mode, M, K, K, reps=20, bn=bn)
        line += f"  bn={bn}: {1e6 * sec:7.1f} us {2.0 * M * N * K / sec / 1e12:6.0f} TF"
    print(line, flush=True)
