#!/usr/bin/env python3
"""Scores of the fp32 CPU oracle (oracle/reference_cpu.py, itself pinned to the reference's Python by the other
fixtures here) on 2048 seeded clips at the full XLS-R-300M geometry, both heads, plus the in-tree head under
seven more weight seeds (64 clips each): the committed answer the GPU parity test at the benchmark batch size
checks against (tests/test_forward_gpu.py::test_2048_clips_*).  Trial labels for the EER comparison are drawn from
the oracle's own score (steeply rising probability around the median: an EER of a few per cent).  Runs for about half an hour on 8 cores.

usage: make_oracle_scores.py [n_clips=2048] [threads=6]   -> oracle_scores_2048.npz"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
import slsmi  # noqa: E402,F401
from slsmi.config import XLSR_300M  # noqa: E402
from slsmi.weights import synth_state_dict  # noqa: E402
from oracle import reference_cpu as orc  # noqa: E402
import clips as clipgen  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
torch.set_num_threads(int(sys.argv[2]) if len(sys.argv) > 2 else 6)
SAE_SEEDS = [1234, 11, 22, 33, 44, 55, 66, 77]
PER_SEED = int(os.environ.get("PER_SEED", "64"))
BATCH = 16


def score(logp):
    return torch.exp(logp.float())[:, 1].numpy()


@torch.no_grad()
def main():
    cfg = orc.OracleConfig()
    sd_sls = synth_state_dict(XLSR_300M, "sls", 201, 1234)
    sd_sae = synth_state_dict(XLSR_300M, "sae", 201, 1234)   # same backbone values (keys and seed decide them)
    sls, sae = np.zeros(N, np.float32), np.zeros(N, np.float32)
    t0 = time.time()
    for i in range(0, N, BATCH):
        wav = clipgen.clips(i, min(BATCH, N - i))
        bb = orc.backbone_forward(sd_sls, wav, cfg, pfx="ssl_model.model.")
        sls[i:i + len(wav)] = score(orc.sls_head(sd_sls, bb["layer_results"]))
        sae[i:i + len(wav)] = score(orc.sae_head(sd_sae, bb["x"], cfg))
        print(f"{i + len(wav)}/{N} clips, {time.time() - t0:.0f} s", flush=True)
    out = {"n": np.int64(N), "first_clip": np.int64(0), "sls_scores": sls, "sae_scores": sae}
    rng = np.random.default_rng(0)
    for head, s in (("sls", sls), ("sae", sae)):
        # P(bona fide) rises steeply with the score around its median: two classes that overlap a little (EER of
        # a few per cent, the regime of the reference's published results), so that an EER comparison is not
        # decided by the hundreds of trials a 30 % EER would put next to the threshold
        spread = np.quantile(s, 0.95) - np.quantile(s, 0.05)
        pr = 1.0 / (1.0 + np.exp(-(s - np.median(s)) / (0.04 * spread)))
        out[f"{head}_labels"] = rng.random(N) < pr
    seed_scores = np.zeros((len(SAE_SEEDS), PER_SEED), np.float32)
    seed_scores[0] = sae[:PER_SEED]
    for k, seed in enumerate(SAE_SEEDS):
        if k == 0:
            continue
        sd = synth_state_dict(XLSR_300M, "sae", 201, seed)
        for i in range(0, PER_SEED, BATCH):
            seed_scores[k, i:i + BATCH] = score(orc.model_forward(sd, clipgen.clips(i, BATCH), cfg, "sae"))
        print(f"seed {seed} done, {time.time() - t0:.0f} s", flush=True)
    out["sae_seeds"] = np.array(SAE_SEEDS, np.int64)
    out["sae_seed_scores"] = seed_scores
    np.savez_compressed(os.path.join(HERE, f"oracle_scores_{N}.npz"), **out)
    print("sls score quantiles", np.quantile(sls, [0, .05, .5, .95, 1]), "sae", np.quantile(sae, [0, .05, .5, .95, 1]))


if __name__ == "__main__":
    main()
