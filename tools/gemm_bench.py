#!/usr/bin/env python3
"""Time single GEMM shapes through the C ABI (HIP events inside the library).
usage: gemm_bench.py [--bn 0|128] [--reps N] M,N,K[,lda] ..."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slsmi  # noqa: E402
from slsmi import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("shapes", nargs="+")
ap.add_argument("--bn", type=int, nargs="+", default=[0])
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--mode", type=int, default=_lib.GEMM_BF16)
ap.add_argument("--dtype", default="f16")
a = ap.parse_args()
dev = torch.device("cuda:0")
dt = ops.op_dtype(a.dtype)
for sh in a.shapes:
    v = [int(x) for x in sh.split(",")]
    M, N, K = v[:3]
    lda = v[3] if len(v) > 3 else K
    A = (torch.randn((M - 1) * lda + K + 64, device=dev) * 0.5).to(dt)
    W = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
    b = torch.zeros(N, device=dev)
    odt = dt if a.mode in (0, 1) else torch.float32
    out = torch.zeros(M, N, dtype=odt, device=dev)
    for bn in a.bn:
        sec = ops.time_gemm_bf16(A, W, b, out, a.mode, M, K, lda, reps=a.reps, bn=bn)
        print(f"M={M} N={N} K={K} lda={lda} bn={bn}: {1e6 * sec:.1f} us  {2.0 * M * N * K / sec / 1e12:.0f} TF", flush=True)
