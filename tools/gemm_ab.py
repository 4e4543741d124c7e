#!/usr/bin/env python3
"""A/B several builds of libslsmi.so on the same GEMM shapes in ONE process, interleaved
rounds, median and min per build.  usage: gemm_ab.py --libs a.so b.so --mode 1 M,N,K[,lda] ..."""
import argparse
import ctypes as C
import os
import statistics
import sys

import torch

ap = argparse.ArgumentParser()
ap.add_argument("shapes", nargs="+")
ap.add_argument("--libs", nargs="+", required=True)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--mode", type=int, default=0)
ap.add_argument("--bn", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda:0")
P = C.c_void_p
libs = []
for path in a.libs:
    lib = C.CDLL(os.path.abspath(path))
    f = lib.slsmi_time_gemm_bf16
    f.restype = C.c_int
    f.argtypes = [P, C.c_int64, P, C.c_int64, P, P, C.c_int64] + [C.c_int] * 8 + [C.POINTER(C.c_float), P]
    libs.append((os.path.basename(path), f))
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for sh in a.shapes:
    v = [int(x) for x in sh.split(",")]
    M, N, K = v[:3]
    lda = v[3] if len(v) > 3 else K
    A = (torch.randn((M - 1) * lda + K + 64, device=dev) * 0.5).half()
    W = (torch.randn(N, K, device=dev) * K ** -0.5).half()
    b = torch.zeros(N, device=dev)
    out = torch.zeros(M, N, dtype=torch.float16 if a.mode in (0, 1) else torch.float32, device=dev)
    res = {n: [] for n, _ in libs}
    for r in range(a.rounds):
        for n, f in libs:
            ms = C.c_float()
            rc = f(P(A.data_ptr()), lda, P(W.data_ptr()), K, P(b.data_ptr()), P(out.data_ptr()), N, M, N, K, a.mode,
                   a.bn, 1, 1, a.reps, C.byref(ms), st)
            assert rc == 0, rc
            res[n].append(ms.value * 1e3 / a.reps)
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K} mode={a.mode}: " + "  ".join(
        f"{n}: med {statistics.median(t):.1f} us ({fl / statistics.median(t) / 1e6:.0f} TF) min {min(t):.1f}" for n, t in res.items()),
        flush=True)
