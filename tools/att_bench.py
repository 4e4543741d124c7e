#!/usr/bin/env python3
"""Time the attention kernel of several libslsmi builds in one process (interleaved rounds).
usage: att_bench.py lib1.so lib2.so ..."""
import ctypes as C
import os
import statistics
import sys

import torch

dev = torch.device("cuda:0")
B, T = 64, 201
qkv = (torch.randn(B * T, 3072, device=dev) * 0.5).half()
out = torch.empty(B * T, 1024, device=dev, dtype=torch.half)
P = C.c_void_p
libs = []
for path in sys.argv[1:]:
    lib = C.CDLL(os.path.abspath(path))
    f = lib.slsmi_attention
    f.restype = C.c_int
    f.argtypes = [P, P, C.c_int, C.c_int, C.c_int, P]
    libs.append((os.path.basename(path), f))
st = P(torch.cuda.current_stream(dev).cuda_stream)
res = {n: [] for n, _ in libs}
for r in range(7):
    for n, f in libs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f(P(qkv.data_ptr()), P(out.data_ptr()), B, T, 1, st)
        e0.record()
        for _ in range(10):
            f(P(qkv.data_ptr()), P(out.data_ptr()), B, T, 1, st)
        e1.record()
        torch.cuda.synchronize()
        res[n].append(e0.elapsed_time(e1) * 100)
for n, t in res.items():
    print(f"{n}: med {statistics.median(t):.1f} us  min {min(t):.1f}")
