#!/usr/bin/env python3
"""Whole scoring steps of two engines alternating on two HIP streams against the same steps on one stream:
does the device fill the tails of one forward's launches with the other forward's kernels?
usage: two_stream_steps_probe.py [steps=40]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slsmi.config import XLSR_300M  # noqa: E402
from slsmi.model import Model  # noqa: E402
from slsmi.weights import synth_state_dict  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
sd = synth_state_dict(XLSR_300M, "sls", 201, 1234)
models = []
for _ in range(2):
    m = Model(None, dev, cp_path=None, head="sls", precision="f16").eval()
    m.load_state_dict(sd)
    m.to(dev)
    models.append(m)
g = torch.Generator().manual_seed(1234)
wav = (0.1 * torch.randn(64, 64600, generator=g)).clamp(-1, 1).to(dev)
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]


def run(n, two):
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    with torch.no_grad():
        for i in range(n):
            k = i % 2
            with torch.cuda.stream(streams[k if two else 0]):
                models[k](wav, return_sae_loss=False)
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / n * 1e3


for two in (False, True):
    run(6, two)
for r in range(3):
    a = run(steps, False)
    b = run(steps, True)
    print(f"round {r}: one stream {a:.3f} ms/step   two streams {b:.3f} ms/step", flush=True)
