#!/usr/bin/env python3
"""Score error of the SLS / f16 scoring forward over the 2048-clip fixture (tests/golden/oracle_scores_2048.npz) for a
list of feature-extractor precision settings (slsmi_set_conv_precision), with the forward's time at a batch of 64:
what each extra-precision choice buys and costs on the GPU.

usage: score_err_masks.py "<split mask>,<fp32 mask>" ...      e.g.  0,0  0x70,0x78  0x78,0x7e
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import slsmi  # noqa: E402,F401
from slsmi.config import XLSR_300M  # noqa: E402
from slsmi.model import Model  # noqa: E402
from slsmi.weights import synth_state_dict  # noqa: E402
import clips as clipgen  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    fx = np.load(os.path.join(ROOT, "tests", "golden", "oracle_scores_2048.npz"))
    n = int(os.environ.get("CLIPS", fx["n"]))
    ref = fx["sls_scores"][:n]
    t0 = time.time()
    wavs = [clipgen.clips(i, min(64, n - i)) for i in range(0, n, 64)]
    print(f"{n} clips generated in {time.time() - t0:.0f} s", flush=True)
    m = Model(None, dev, cp_path=None, head="sls", cfg=XLSR_300M, precision="f16").eval()
    m.load_state_dict(synth_state_dict(XLSR_300M, "sls", 201, 1234))
    for spec in sys.argv[1:] or ["0,0"]:
        sm, fm = (int(v, 0) for v in spec.split(","))
        with torch.no_grad():
            m(wavs[0].to(dev), return_sae_loss=False)      # builds the engine
            m._engine.set_conv_precision(sm, fm)
            out = []
            for w in wavs:
                out.append(torch.exp(m(w.to(dev), return_sae_loss=False).float())[:, 1].cpu())
            w0 = wavs[0].to(dev)
            for _ in range(3):
                m(w0, return_sae_loss=False)
            torch.cuda.synchronize()
            t1 = time.time()
            for _ in range(20):
                m(w0, return_sae_loss=False)
            torch.cuda.synchronize()
            ms = (time.time() - t1) / 20 * 1e3
        d = torch.cat(out).numpy().astype(np.float64) - ref
        err = np.abs(d)
        print(f"split {sm:#04x} fp32 {fm:#04x}: max {err.max():.3e}  99.9% {np.quantile(err, 0.999):.3e}  median "
              f"{np.median(err):.3e}  rms {np.sqrt((d ** 2).mean()):.3e}  mean {d.mean():+.3e}  over 1e-3: "
              f"{int((err > 1e-3).sum())}  |  {ms:.3f} ms per batch of 64", flush=True)


if __name__ == "__main__":
    main()
