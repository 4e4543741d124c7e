#!/usr/bin/env python3
"""Bits and time of the low-latency scoring forward (single clip, four clips; SLS and SAE heads): to compare
library builds (SLSMI_LIB=...)."""
import hashlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slsmi  # noqa: E402,F401
from slsmi.config import XLSR_300M  # noqa: E402
from slsmi.model import Model  # noqa: E402
from slsmi.weights import synth_state_dict  # noqa: E402

dev = torch.device("cuda:0")
for head in ("sls", "sae"):
    m = Model(None, dev, cp_path=None, head=head, precision="f16", low_latency=True).eval()
    m.load_state_dict(synth_state_dict(XLSR_300M, head, 201, 1234))
    m.to(dev)
    g = torch.Generator().manual_seed(5)
    wav = (0.1 * torch.randn(4, 64600, generator=g)).clamp(-1, 1).to(dev)
    with torch.no_grad():
        for B in (1, 4):
            out = m(wav[:B], return_sae_loss=False)
            h = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:12]
            for _ in range(5):
                m(wav[:B], return_sae_loss=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                m(wav[:B], return_sae_loss=False)
            torch.cuda.synchronize()
            print(f"{head} B={B}: logp {h}  {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms", flush=True)
