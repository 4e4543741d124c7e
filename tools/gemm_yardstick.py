#!/usr/bin/env python3
"""Yardstick for the library's GEMM: the same shapes through torch.nn.functional.linear (hipBLASLt / rocBLAS
on this image), timed with events on the current stream, next to the library's own event-timed launches.
Not part of the product: it only tells how far the hand-written kernel is from the vendor library.
usage: gemm_yardstick.py [--reps N] [M,N,K ...]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slsmi  # noqa: E402,F401
from slsmi import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("shapes", nargs="*", default=["12864,3072,1024", "12864,1024,1024", "12864,4096,1024", "12864,1024,4096",
                                               "425984,512,1536", "53248,512,1536"])
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--dtype", default="f16")
a = ap.parse_args()
dev = torch.device("cuda:0")
dt = ops.op_dtype(a.dtype)
for sh in a.shapes:
    M, N, K = [int(x) for x in sh.split(",")][:3]
    A = (torch.randn(M, K, device=dev) * 0.5).to(dt)
    W = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
    b = torch.zeros(N, device=dev)
    bh = b.to(dt)
    out = torch.zeros(M, N, dtype=dt, device=dev)
    for _ in range(3):
        F.linear(A, W, bh)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        F.linear(A, W, bh)
    e1.record()
    torch.cuda.synchronize()
    t_lib = e0.elapsed_time(e1) / a.reps * 1e-3
    t_own = ops.time_gemm_bf16(A.reshape(-1), W, b, out, _lib.GEMM_BF16, M, K, K, reps=a.reps, bn=0)
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K}: vendor {1e6 * t_lib:.1f} us {fl / t_lib / 1e12:.0f} TF | own {1e6 * t_own:.1f} us "
          f"{fl / t_own / 1e12:.0f} TF", flush=True)
