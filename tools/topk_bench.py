#!/usr/bin/env python3
"""Time of the per-frame TopK of the SAE head at the benchmark batch (12864 rows of 4096 ReLU outputs, k = 128)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slsmi  # noqa: E402,F401
from slsmi import ops  # noqa: E402

DEV = torch.device("cuda:0")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 12864
src = torch.relu(torch.randn(rows, 4096, device=DEV) * 0.7 - 0.1)
work = src.clone()
for _ in range(3):
    work.copy_(src)
    ops.topk_rows(work, 128)
torch.cuda.synchronize()
e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
e0.record()
for _ in range(20):
    work.copy_(src)
e1.record()
for _ in range(20):
    work.copy_(src)
    ops.topk_rows(work, 128)
e2.record()
torch.cuda.synchronize()
copy_us = e0.elapsed_time(e1) * 1000 / 20
us = e1.elapsed_time(e2) * 1000 / 20 - copy_us
print(f"topk_rows {rows} x 4096, k = 128: {us:7.1f} us  ({rows * 4096 * 8 / us / 1e6:5.2f} TB/s read + written; "
      f"copy of the same bytes {copy_us:6.1f} us)", flush=True)
