#!/usr/bin/env python3
"""Time of the encoder's positional conv at the benchmark batch (64 clips of 201 frames) in both forms
(slsmi_posconv form 0: batched GEMM over the Toeplitz view, form 1: slab-resident kernel)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slsmi  # noqa: E402,F401
from slsmi import ops  # noqa: E402

DEV = torch.device("cuda:0")
B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 201
slab = (T + 128 + 7) // 8 * 8
xg = (torch.randn(B, 16, slab, 64, device=DEV) * 0.5).half()
w = (torch.randn(1024, 128, 64, device=DEV) * 0.01).half()
bias = torch.zeros(1024, device=DEV)
x = torch.zeros(B * T, 1024, device=DEV)
for form in (0, 1):
    for _ in range(3):
        ops.posconv(xg, w, bias, x, T, form)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.posconv(xg, w, bias, x, T, form)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 20
    print(f"form {form}: {us:7.1f} us  {2 * B * T * 1024 * 8192 / us / 1e6:7.1f} TFLOP/s", flush=True)
