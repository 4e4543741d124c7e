#!/usr/bin/env python3
"""Which torch operators launch GPU kernels during one training step (bench.py --mode train's step): counts per operator
from torch.profiler, to find elementwise launches that are not the library's."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slsmi  # noqa: E402,F401
from slsmi import training  # noqa: E402
from slsmi.config import XLSR_300M  # noqa: E402
from slsmi.model import Model  # noqa: E402
from slsmi.weights import synth_state_dict  # noqa: E402

dev = torch.device("cuda:0")
model = Model(None, dev, cp_path=None, precision="f16", use_sae=True).train()
model.load_state_dict(synth_state_dict(XLSR_300M, "sae", 201, 1234))
model.to(dev)
training.freeze_unused(model)
params = [p for p in model.parameters() if p.requires_grad]
opt = training.Adam(params, lr=1e-6, weight_decay=1e-4)
wav = (0.1 * torch.randn(5, 64600)).clamp(-1, 1).to(dev)
y = torch.arange(5, device=dev) % 2
for _ in range(4):
    training.train_step(model, opt, wav, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(2):
        training.train_step(model, opt, wav, y)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="count", row_limit=45, max_name_column_width=60))
