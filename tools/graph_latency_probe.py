#!/usr/bin/env python3
"""Single-clip (and small-batch) scoring latency: eager launches against replaying the captured forward.
usage: graph_latency_probe.py [low_latency=1]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slsmi.config import XLSR_300M  # noqa: E402
from slsmi.model import Model  # noqa: E402
from slsmi.weights import synth_state_dict  # noqa: E402

ll = bool(int(sys.argv[1])) if len(sys.argv) > 1 else True
dev = torch.device("cuda:0")
m = Model(None, dev, cp_path=None, head="sls", precision="f16", low_latency=ll).eval()
m.load_state_dict(synth_state_dict(XLSR_300M, "sls", 201, 1234))
m.to(dev)
g = torch.Generator().manual_seed(1234)
for B in (1, 4, 16):
    wav = (0.1 * torch.randn(B, 64600, generator=g)).clamp(-1, 1).to(dev)
    with torch.no_grad():
        for _ in range(3):
            want = m(wav, return_sae_loss=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            m(wav, return_sae_loss=False)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / 50 * 1e3
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            m(wav, return_sae_loss=False)
        torch.cuda.current_stream(dev).wait_stream(side)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            out = m(wav, return_sae_loss=False)
        for _ in range(3):
            gr.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            gr.replay()
        torch.cuda.synchronize()
        rep = (time.perf_counter() - t0) / 50 * 1e3
        print(f"B={B} low_latency={ll}: eager {eager:.3f} ms  graph replay {rep:.3f} ms  equal {torch.equal(out, want)}", flush=True)
