#!/usr/bin/env python3
"""A/B of library builds on one GEMM shape with operands that are NOT cache-resident: every launch
uses the next of `--bufs` A and C buffers (together larger than the 256 MiB Infinity Cache).
usage: gemm_ab_cold.py --libs a.so b.so --mode 3 M,N,K ..."""
import argparse
import ctypes as C
import os
import statistics

import torch

ap = argparse.ArgumentParser()
ap.add_argument("shapes", nargs="+")
ap.add_argument("--libs", nargs="+", required=True)
ap.add_argument("--rounds", type=int, default=40)
ap.add_argument("--mode", type=int, default=3)
ap.add_argument("--bufs", type=int, default=8)
a = ap.parse_args()
dev = torch.device("cuda:0")
P = C.c_void_p
libs = []
for path in a.libs:
    lib = C.CDLL(os.path.abspath(path))
    f = lib.slsmi_gemm_bf16
    f.restype = C.c_int
    f.argtypes = [P, C.c_int64, P, C.c_int64, P, P, C.c_int64] + [C.c_int] * 7 + [P]
    libs.append((os.path.basename(path), f))
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for sh in a.shapes:
    M, N, K = [int(x) for x in sh.split(",")]
    As = [(torch.randn(M, K, device=dev) * 0.5).half() for _ in range(a.bufs)]
    W = (torch.randn(N, K, device=dev) * K ** -0.5).half()
    b = torch.zeros(N, device=dev)
    Cs = [torch.zeros(M, N, dtype=torch.float16 if a.mode in (0, 1) else torch.float32, device=dev) for _ in range(a.bufs)]
    res = {n: [] for n, _ in libs}
    k = 0
    for r in range(a.rounds):
        for n, f in libs:
            A, out = As[k % a.bufs], Cs[k % a.bufs]
            k += 1
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = f(P(A.data_ptr()), K, P(W.data_ptr()), K, P(b.data_ptr()), P(out.data_ptr()), N, M, N, K, a.mode,
                   0, 1, 1, st)
            e1.record()
            assert rc == 0, rc
            e1.synchronize()
            if r >= 3:
                res[n].append(e0.elapsed_time(e1) * 1e3)
    fl = 2.0 * M * N * K
    print(f"cold M={M} N={N} K={K} mode={a.mode}: " + "  ".join(
        f"{n}: med {statistics.median(t):.1f} us ({fl / statistics.median(t) / 1e6:.0f} TF) min {min(t):.1f}" for n, t in res.items()),
        flush=True)
