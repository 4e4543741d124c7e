#!/usr/bin/env python3
"""Per-GEMM time inside the sequence of an encoder layer's four GEMMs (each reads what another just wrote),
208 x 256 kernel (bn = 256) against 256 x 128 (bn = 1) and the library's choice (bn = 0)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slsmi  # noqa: E402,F401
from slsmi import _lib, ops  # noqa: E402

DEV = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 12864
g = lambda *s, sc=1.0: (torch.randn(*s, device=DEV) * sc)
L = 4   # distinct weight sets, cycled (a forward has 24)
W = [{"wqkv": g(3072, 1024, sc=1 / 32).half(), "wo": g(1024, 1024, sc=1 / 32).half(), "w1": g(4096, 1024, sc=1 / 32).half(),
      "w2": g(1024, 4096, sc=1 / 64).half()} for _ in range(L)]
b3, b1, b4 = torch.zeros(3072, device=DEV), torch.zeros(1024, device=DEV), torch.zeros(4096, device=DEV)
h, ao, x = g(M, 1024, sc=0.5).half(), g(M, 1024, sc=0.5).half(), g(M, 1024)
qkv = torch.zeros(M, 3072, dtype=torch.float16, device=DEV)
f = torch.zeros(M, 4096, dtype=torch.float16, device=DEV)
for bn in (1, 256, 0, 1, 256):
    reps = 24
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(reps)]
    for r in range(reps + 4):
        w = W[r % L]
        e = ev[r - 4] if r >= 4 else None
        if e: e[0].record()
        ops.gemm_bf16(h, w["wqkv"], b3, _lib.GEMM_BF16, out=qkv, bn=bn)
        if e: e[1].record()
        ops.gemm_bf16(ao, w["wo"], b1, _lib.GEMM_F32_RESID, out=x, bn=bn)
        if e: e[2].record()
        ops.gemm_bf16(h, w["w1"], b4, _lib.GEMM_BF16_GELU, out=f, bn=bn)
        if e: e[3].record()
        ops.gemm_bf16(f, w["w2"], b1, _lib.GEMM_F32_RESID, out=x, bn=bn)
        if e: e[4].record()
    torch.cuda.synchronize()
    t = [sum(e[i].elapsed_time(e[i + 1]) for e in ev) / reps * 1e3 for i in range(4)]
    print(f"bn={bn}: qkv {t[0]:6.1f}  out {t[1]:6.1f}  fc1 {t[2]:6.1f}  fc2 {t[3]:6.1f}  sum {sum(t):6.1f} us", flush=True)
