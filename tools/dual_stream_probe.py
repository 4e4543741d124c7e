#!/usr/bin/env python3
"""Does running two half-batches on two streams, each on half of the CUs, beat one full batch on all CUs?
The four GEMMs of an encoder layer (QKV, out-proj with fp32 residual, fc1 + GELU, fc2 with residual), 208 x 256
kernel, 24 times: (a) one stream, 64 clips; (b) two streams, 32 clips each, launches capped at 128 workgroups."""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slsmi  # noqa: E402,F401
from slsmi import _lib, ops  # noqa: E402

DEV = torch.device("cuda:0")
lib = _lib.load()
cap_fn = lib.slsmi_debug_wide_grid_cap
cap_fn.argtypes = [ctypes.c_int]


def make(M):
    g = lambda *s, sc=1.0: (torch.randn(*s, device=DEV) * sc)
    d = {"h": g(M, 1024, sc=0.5).half(), "qkv": torch.zeros(M, 3072, dtype=torch.float16, device=DEV),
         "ao": g(M, 1024, sc=0.5).half(), "x": g(M, 1024), "f": torch.zeros(M, 4096, dtype=torch.float16, device=DEV),
         "wqkv": g(3072, 1024, sc=1 / 32).half(), "wo": g(1024, 1024, sc=1 / 32).half(), "w1": g(4096, 1024, sc=1 / 32).half(),
         "w2": g(1024, 4096, sc=1 / 64).half(), "b3": torch.zeros(3072, device=DEV), "b1": torch.zeros(1024, device=DEV),
         "b4": torch.zeros(4096, device=DEV)}
    return d


def layer(d, bn):
    ops.gemm_bf16(d["h"], d["wqkv"], d["b3"], _lib.GEMM_BF16, out=d["qkv"], bn=bn)
    ops.gemm_bf16(d["ao"], d["wo"], d["b1"], _lib.GEMM_F32_RESID, out=d["x"], bn=bn)
    ops.gemm_bf16(d["h"], d["w1"], d["b4"], _lib.GEMM_BF16_GELU, out=d["f"], bn=bn)
    ops.gemm_bf16(d["f"], d["w2"], d["b1"], _lib.GEMM_F32_RESID, out=d["x"], bn=bn)


def run_single(d, bn, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        layer(d, bn)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def run_dual(d0, d1, bn, reps, offset=False):
    s0, s1 = torch.cuda.Stream(DEV), torch.cuda.Stream(DEV)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if offset:   # the second stream starts half a layer late
        with torch.cuda.stream(s1):
            ops.gemm_bf16(d1["h"], d1["w1"], d1["b4"], _lib.GEMM_BF16_GELU, out=d1["f"], bn=bn)
            ops.gemm_bf16(d1["h"], d1["wqkv"], d1["b3"], _lib.GEMM_BF16, out=d1["qkv"], bn=bn)
    for _ in range(reps):
        with torch.cuda.stream(s0):
            layer(d0, bn)
        with torch.cuda.stream(s1):
            layer(d1, bn)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


full, h0, h1 = make(12864), make(6432), make(6432)
for bn in (256, 1):
    cap_fn(0)
    run_single(full, bn, 5)
    a = run_single(full, bn, 48)
    b = run_dual(h0, h1, bn, 48)
    line = f"bn={bn}: one stream 64 clips {1e6 * a:7.1f} us/layer   two streams 32+32 uncapped {1e6 * b:7.1f}"
    if bn == 256:
        for cap in (128, 192):
            cap_fn(cap)
            run_dual(h0, h1, bn, 5)
            c = run_dual(h0, h1, bn, 48)
            c2 = run_dual(h0, h1, bn, 48, offset=True)
            line += f"   cap {cap}: {1e6 * c:7.1f} (offset start {1e6 * c2:7.1f})"
        cap_fn(0)
    print(line, flush=True)
