#!/usr/bin/env python3
"""slsmi.training.Adam's one-launch step over a model-sized parameter set, by elements per workgroup."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slsmi  # noqa: E402,F401
from slsmi import training  # noqa: E402

dev = torch.device("cuda:0")
shapes = [(1024, 1024)] * 96 + [(4096, 1024)] * 48 + [(1024,)] * 200 + [(4096, 4096)]   # ~318 M values
for chunk in (2048, 4096, 8192, 16384):
    training.Adam.CHUNK = chunk
    ps = [torch.nn.Parameter(torch.randn(*s, device=dev) * 0.02) for s in shapes]
    for p in ps:
        p.grad = torch.randn_like(p) * 1e-3
    opt = training.Adam(ps, lr=1e-6, weight_decay=1e-4)
    for _ in range(3):
        opt.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        opt.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    n = sum(p.numel() for p in ps)
    print(f"chunk {chunk:7d}: {dt * 1e3:.3f} ms per step, {28.0 * n / dt / 1e12:.2f} TB/s", flush=True)
    del ps, opt
