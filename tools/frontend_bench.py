#!/usr/bin/env python3
"""Times of the first conv layer (conv + LayerNorm + GELU) and of the LayerNorm(512) + GELU pass over the
conv stack's row counts at the benchmark batch (64 clips of 64600 samples)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slsmi  # noqa: E402,F401
from slsmi import ops  # noqa: E402

DEV = torch.device("cuda:0")
B, L = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 64600


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / reps


wav = torch.randn(B, L, device=DEV)
w = torch.randn(512, 1, 10, device=DEV) * 0.3
bias = torch.randn(512, device=DEV) * 0.1
gamma = 1 + 0.1 * torch.randn(512, device=DEV)
beta = 0.1 * torch.randn(512, device=DEV)
T0 = (L - 10) // 5 + 1
# (ops.conv0_ln_gelu zero-fills its result first: that fill is timed alone and taken off)
fill = timed(lambda: torch.zeros(B, T0, 512, dtype=torch.float16, device=DEV))
us = timed(lambda: ops.conv0_ln_gelu(wav, w, bias, gamma, beta, 1e-5, T0, dtype=torch.float16)) - fill
print(f"conv0 {B} x {T0} frames: {us:7.1f} us  ({B * T0 * 1024 / us / 1e6:5.2f} TB/s written; zero fill of the "
      f"same bytes {fill:6.1f} us)", flush=True)
total = 0.0
T = T0
for k in (3, 3, 3, 3, 2, 2):
    T = (T - k) // 2 + 1
    x = torch.randn(B * T, 512, device=DEV).half()
    us = timed(lambda: ops.ln512_bf16(x, gamma, beta, 1e-5, True))
    total += us
    print(f"ln512 {B * T:7d} rows: {us:7.1f} us  ({B * T * 2048 / us / 1e6:5.2f} TB/s read + written)", flush=True)
print(f"ln512 total {total:7.1f} us", flush=True)
