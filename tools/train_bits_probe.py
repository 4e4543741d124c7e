#!/usr/bin/env python3
"""Bits of one training forward and a few gradients (in-tree model, 2 layers, 3 clips): to compare library builds
(SLSMI_LIB=...)."""
import hashlib
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slsmi  # noqa: E402,F401
from slsmi.config import BackboneConfig  # noqa: E402
from slsmi.model import Model  # noqa: E402
from slsmi.weights import synth_state_dict  # noqa: E402

dev = torch.device("cuda:0")
cfg = BackboneConfig(layers=3)
m = Model(None, dev, cp_path=None, cfg=cfg, precision="f16", use_sae=True).train()
m.load_state_dict(synth_state_dict(cfg, "sae", 201, 1234))
m.to(dev)
g = torch.Generator().manual_seed(5)
wav = (0.1 * torch.randn(3, 64600, generator=g)).clamp(-1, 1).to(dev)
mask = torch.ones(3, 256, device=dev)
logp, loss = m(wav, dropout_mask=mask)
(F.nll_loss(logp, torch.tensor([1, 0, 1], device=dev)) + 0.1 * loss).backward()
h = lambda t: hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:12]  # noqa: E731
print("logp", h(logp), "loss", h(loss))
for n, p in m.named_parameters():
    if p.grad is not None and (".layers.2." in n or ".layers.1.fc1.weight" in n or "conv_layers.0.0.weight" in n):
        print(n, h(p.grad))
