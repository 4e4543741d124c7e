#!/usr/bin/env python3
"""Times GEMMs whose A operand repeats along K (slsmi_gemm_bf16_period) against plain ones of the same shape.
usage: gemm_period_probe.py M,N,K,lda,ka,mode,bn ..."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slsmi  # noqa: E402,F401
from slsmi import ops  # noqa: E402

dev = torch.device("cuda:0")
for sh in sys.argv[1:]:
    M, N, K, lda, ka, mode, bn = (int(x) for x in sh.split(","))
    A = (torch.randn((M - 1) * lda + max(K, ka) + 64, device=dev) * 0.5).half()
    W = (torch.randn(N, K, device=dev) * K ** -0.5).half()
    b = torch.zeros(N, device=dev)
    out = torch.zeros(M, N, dtype=torch.float16 if mode in (0, 1) else torch.float32, device=dev)
    def run():
        ops.gemm_bf16(A, W, b, mode, out=out, bn=bn, M=M, K=K, lda=lda, ka=ka)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"M={M} N={N} K={K} lda={lda} ka={ka} mode={mode} bn={bn}: {us:.1f} us  {2.0 * M * N * K / us / 1e6:.0f} TF", flush=True)
