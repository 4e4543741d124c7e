#!/usr/bin/env python3
"""Golden vectors for the RawBoost mirror (slsmi/rawboost.py), made by running the reference's own
RawBoost.py and the dispatcher of data_utils_SSL.py (:141-203) in this container.

    python tests/golden/make_golden_rawboost.py        # writes tests/golden/rawboost_cases.npz

For every algorithm 0..8: np.random.seed(seed); y = process_Rawboost_feature(x, 16000, args, algo), with
the command-line defaults of main.py:443-459 and a seeded test waveform (a chirp plus noise, 4000 samples).
Stored: x, the seeds, every y, and the helper outputs genNotchCoeffs / filterFIR for one seed.
"""
import importlib.util
import os
import sys
import types
import warnings

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def load_ref_module(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def main():
    warnings.simplefilter("ignore")   # the reference converts one-element arrays with int()
    rb = load_ref_module("RawBoost", os.path.join(REF, "RawBoost.py"))
    sys.modules.setdefault("librosa", types.ModuleType("librosa"))   # only used inside the Dataset classes
    du = load_ref_module("ref_data_utils_SSL", os.path.join(REF, "data_utils_SSL.py"))
    args = types.SimpleNamespace(algo=3, nBands=5, minF=20, maxF=8000, minBW=100, maxBW=1000, minCoeff=10,
                                 maxCoeff=100, minG=0, maxG=0, minBiasLinNonLin=5, maxBiasLinNonLin=20, N_f=5, P=10,
                                 g_sd=2, SNRmin=10, SNRmax=40)                  # main.py:443-459
    n = 4000
    t = np.arange(n) / 16000.0
    rng = np.random.RandomState(7)
    x = 0.6 * np.sin(2 * np.pi * (200 + 1500 * t) * t) + 0.05 * rng.randn(n)
    out = {"x": x, "seeds": np.array([1234, 99])}
    for seed in (1234, 99):
        for algo in range(0, 9):
            np.random.seed(seed)
            y = du.process_Rawboost_feature(x.copy(), 16000, args, algo)
            out[f"y_seed{seed}_algo{algo}"] = np.asarray(y, dtype=np.float64)
    np.random.seed(5)
    b = rb.genNotchCoeffs(5, 20, 8000, 100, 1000, 10, 100, 0, 0, 16000)
    out["notch_seed5"] = np.asarray(b, dtype=np.float64)
    out["fir_seed5"] = np.asarray(rb.filterFIR(x, np.asarray(b).ravel()), dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "rawboost_cases.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
