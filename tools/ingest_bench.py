#!/usr/bin/env python3
"""Throughput of the native audio ingest on this machine's host cores, and (with --score, on a
GPU) of files -> scores end to end through BatchLoader + Model.

Writes a few synthetic 16 kHz FLAC clips (about 4 s, 16-bit, LPC + fixed subframes) with the test
encoder, links them under many names, then times slsmi_audio_load_batch.
usage: ingest_bench.py [--files 2048] [--threads 1 4 16] [--score]"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import slsmi  # noqa: E402
from slsmi import data  # noqa: E402
from flac_writer import encode_flac  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--files", type=int, default=2048)
ap.add_argument("--threads", type=int, nargs="+", default=[1, 4, 16])
ap.add_argument("--score", action="store_true")
a = ap.parse_args()

tmp = tempfile.mkdtemp(prefix="slsmi_ingest_")
os.makedirs(os.path.join(tmp, "flac"))
rng = np.random.default_rng(0)
t = np.arange(66000)
base = []
for i in range(8):
    f0 = 100 + 20 * i
    x = 5000 * np.sin(2 * np.pi * f0 * t / 16000) * (0.5 + 0.5 * np.sin(2 * np.pi * 2.5 * t / 16000))
    x += 1500 * np.sin(2 * np.pi * (1500 + 100 * i) * t / 16000) + rng.normal(0, 300, len(t))
    blob = encode_flac(np.clip(np.round(x), -32768, 32767).astype(np.int64), kinds=("lpc", "fixed2"), blocksize=4096,
                       lpc_order=8, porder=3)
    p = os.path.join(tmp, f"base{i}.flac")
    open(p, "wb").write(blob)
    base.append(p)
print(f"clip: {len(t)} samples, {os.path.getsize(base[0])} bytes ({8 * os.path.getsize(base[0]) / len(t):.2f} bits/sample)")
ids = []
for i in range(a.files):
    u = f"DF_E_{i:07d}"
    os.symlink(base[i % len(base)], os.path.join(tmp, "flac", u + ".flac"))
    ids.append(u)
ds = data.Dataset_ASVspoof2021_eval(ids, tmp)
paths = [ds.path_of(u) for u in ids]
out = torch.empty(a.files, data.CUT)
for th in a.threads:
    data.load_batch_into(paths[:64], out, threads=th)
    t0 = time.perf_counter()
    data.load_batch_into(paths, out, threads=th)
    dt = time.perf_counter() - t0
    print(f"decode + pad, {th:3d} threads: {a.files / dt:8.0f} utt/s  ({1e3 * dt / a.files:.3f} ms/utt)", flush=True)

if a.score:
    from slsmi.config import XLSR_300M
    from slsmi.model import Model
    from slsmi.weights import synth_state_dict
    from slsmi.scoring import scores_from_logp
    dev = torch.device("cuda:0")
    m = Model(None, dev, cp_path=None, head="sls").eval()
    m.load_state_dict(synth_state_dict(XLSR_300M, "sls", 201, 1234))
    m.to(dev)
    for th in a.threads[-1:]:
        loader = data.BatchLoader(ds, batch_size=64, device=dev, threads=th)
        n = 0
        with torch.no_grad():
            for k, (batch, bid) in enumerate(loader):
                if k == 2:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    n = 0
                s = m(batch, return_sae_loss=False)
                n += len(bid)
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"files -> scores, {th} decode threads: {n / dt:.0f} utt/s over {n} clips")
