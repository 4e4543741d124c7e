#!/usr/bin/env python3
"""Layer outputs of one small batch under the three LayerNorm forms of the scoring forward (SLSMI_LN_FOLD = 0 separate
kernels, 1 row-wise fold at this batch size, 2 the 208 x 256 fold): where do they part?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slsmi  # noqa: E402,F401
from slsmi.config import BackboneConfig  # noqa: E402
from slsmi.engine import Engine  # noqa: E402
from slsmi.weights import synth_state_dict  # noqa: E402

DEV = torch.device("cuda:0")
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = BackboneConfig(layers=layers)
sd = synth_state_dict(cfg, "sls", 201, 1234)
g = torch.Generator().manual_seed(99)
wav = (0.1 * torch.randn(3, 64600, generator=g)).clamp(-1, 1).to(DEV)
outs = {}
for form in (0, 1, 2):
    eng = Engine(DEV, cfg, "sls", precision="f16", ln_fold=form)
    eng.load_state_dict({k: v.to(DEV) for k, v in sd.items()})
    o = eng.forward(wav, want_layers=True)
    outs[form] = (o["layers"].clone(), o["logp"].clone())
    eng.close()
for a, b in ((0, 1), (0, 2), (1, 2)):
    d = [(outs[a][0][l] - outs[b][0][l]).abs().max().item() for l in range(layers)]
    print(f"forms {a} vs {b}: max |diff| per layer " + " ".join(f"{v:.2e}" for v in d) +
          f"   logp {(outs[a][1] - outs[b][1]).abs().max().item():.2e}", flush=True)
