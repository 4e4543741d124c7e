#!/usr/bin/env python3
"""Read the cycle stamps of a SLSMI_STAMPS build for one GEMM launch (workgroup 0, wave 0)."""
import ctypes as C
import os
import sys
import torch
os.environ["SLSMI_LIB"] = os.path.abspath(sys.argv[1])
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slsmi
from slsmi import _lib, ops
M, N, K = [int(x) for x in sys.argv[2].split(",")]
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda:0")
A = (torch.randn(M, K, device=dev) * 0.5).half(); W = (torch.randn(N, K, device=dev) * K ** -0.5).half()
b = torch.zeros(N, device=dev)
for _ in range(3):
    out = ops.gemm_bf16(A, W, b, mode)
torch.cuda.synchronize()
lib = _lib.load()
buf = (C.c_ulonglong * 256)()
lib.slsmi_debug_read_stamps(buf)
prev = None
names = {1: "tile start (after R(0))", 2: "steady part done, last loads of the tile issued", 3: "boundary segments done (M(nt-1))", 4: "epilogue done", 5: "epilogue: activation + row indices done", 6: "row block 0 stored", 7: "row block 1 stored", 8: "row block 2 stored", 9: "row block 3 stored"}
for v in buf:
    if v == 0: break
    if (v >> 56) == 1: print('---')
    tag, t = v >> 56, v & 0xFFFFFFFFFFFFFF
    if prev is not None: print(f"  +{t - prev:7d} cyc -> {names.get(tag, tag)}")
    else: print("start", names.get(tag))
    prev = t
