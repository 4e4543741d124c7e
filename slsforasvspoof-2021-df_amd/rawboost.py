"""RawBoost waveform augmentation for the training loader: host-side numpy, as in the reference
(RawBoost.py:8-91, dispatched by data_utils_SSL.py:141-203 inside the DataLoader workers).

The three noise models and their eight combinations are restated here with the reference's
function names and argument order, so `process_Rawboost_feature(x, sr, args, algo)` is a drop-in.
Every random number is drawn from numpy's global generator with the same calls in the same order
as the reference, so that under `np.random.seed(s)` both produce the same waveform
(tests/test_rawboost.py compares against vectors made by the reference's own module).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
from scipy import signal

__all__ = ["RawBoostParams", "randRange", "normWav", "genNotchCoeffs", "filterFIR", "LnL_convolutive_noise",
           "ISD_additive_noise", "SSI_additive_noise", "process_Rawboost_feature"]


@dataclass
class RawBoostParams:
    """Defaults of the reference's command line (main.py:443-459); any object with these
    attributes (an argparse namespace) serves as `args`."""
    algo: int = 3
    nBands: int = 5
    minF: int = 20
    maxF: int = 8000
    minBW: int = 100
    maxBW: int = 1000
    minCoeff: int = 10
    maxCoeff: int = 100
    minG: int = 0
    maxG: int = 0
    minBiasLinNonLin: int = 5
    maxBiasLinNonLin: int = 20
    N_f: int = 5
    P: int = 10
    g_sd: int = 2
    SNRmin: int = 10
    SNRmax: int = 40


def randRange(x1, x2, integer):
    """One uniform draw in [x1, x2) (RawBoost.py:8-12); truncated to int when `integer`."""
    u = float(np.random.uniform(low=x1, high=x2, size=(1,))[0])
    return int(u) if integer else u


def normWav(x, always):
    """Peak-normalise: always, or only when the peak exceeds 1 (RawBoost.py:14-19)."""
    peak = np.amax(np.abs(x))
    return x / peak if (always or peak > 1) else x


def genNotchCoeffs(nBands, minF, maxF, minBW, maxBW, minCoeff, maxCoeff, minG, maxG, fs):
    """Cascade of `nBands` random band-stop FIR filters, peak gain set to a random G dB
    (RawBoost.py:22-42).  Draw order per band: centre, width, length; then the gain."""
    taps = np.ones(1)
    nyq = fs / 2
    for _ in range(nBands):
        centre = randRange(minF, maxF, 0)
        width = randRange(minBW, maxBW, 0)
        length = randRange(minCoeff, maxCoeff, 1)
        length += 1 - length % 2                       # odd length: a band-stop needs a type-I filter
        lo, hi = centre - width / 2, centre + width / 2
        lo = lo if lo > 0 else 1 / 1000
        hi = hi if hi < nyq else nyq - 1 / 1000
        taps = np.convolve(signal.firwin(length, [lo, hi], window="hamming", fs=fs), taps)
    gain_db = randRange(minG, maxG, 0)
    _, response = signal.freqz(taps, 1, fs=fs)
    return pow(10, gain_db / 20) * taps / np.amax(np.abs(response))


def filterFIR(x, b):
    """Zero-phase-aligned FIR filtering: pad, filter, cut the group delay off both ends
    (RawBoost.py:45-50)."""
    n = b.shape[0] + 1
    y = signal.lfilter(b, 1, np.pad(x, (0, n), "constant"))
    return y[int(n / 2):int(y.shape[0] - n / 2)]


def LnL_convolutive_noise(x, N_f, nBands, minF, maxF, minBW, maxBW, minCoeff, maxCoeff, minG, maxG,
                          minBiasLinNonLin, maxBiasLinNonLin, fs):
    """Linear and non-linear convolutive noise: sum over powers x^1 .. x^N_f, each through its own
    notch cascade; the gain range is lowered once, from the second power on (RawBoost.py:53-63)."""
    total = np.zeros(x.shape[0])
    for order in range(1, N_f + 1):
        if order == 2:
            minG, maxG = minG - minBiasLinNonLin, maxG - maxBiasLinNonLin
        b = genNotchCoeffs(nBands, minF, maxF, minBW, maxBW, minCoeff, maxCoeff, minG, maxG, fs)
        total = total + filterFIR(np.power(x, order), b)
    total = total - np.mean(total)
    return normWav(total, 0)


def ISD_additive_noise(x, P, g_sd):
    """Impulsive signal-dependent noise on a random beta % of the samples (RawBoost.py:67-79).
    Draw order: beta, the permutation, then two uniform vectors."""
    beta = randRange(0, P, 0)
    count = int(x.shape[0] * (beta / 100))
    where = np.random.permutation(x.shape[0])[:count]
    u1 = 2 * np.random.rand(where.shape[0]) - 1
    u2 = 2 * np.random.rand(where.shape[0]) - 1
    y = np.array(x, copy=True)
    y[where] = x[where] + g_sd * x[where] * np.multiply(u1, u2)
    return normWav(y, 0)


def SSI_additive_noise(x, SNRmin, SNRmax, nBands, minF, maxF, minBW, maxBW, minCoeff, maxCoeff, minG, maxG, fs):
    """Stationary signal-independent noise: white noise coloured by a notch cascade, added at a
    random SNR (RawBoost.py:83-91).  Draw order: the noise, the cascade, the SNR."""
    noise = np.random.normal(0, 1, x.shape[0])
    b = genNotchCoeffs(nBands, minF, maxF, minBW, maxBW, minCoeff, maxCoeff, minG, maxG, fs)
    noise = normWav(filterFIR(noise, b), 1)
    snr_db = randRange(SNRmin, SNRmax, 0)
    noise = noise / np.linalg.norm(noise, 2) * np.linalg.norm(x, 2) / 10.0 ** (0.05 * snr_db)
    return x + noise


def process_Rawboost_feature(feature, sr, args, algo):
    """Algorithm `algo` of the reference's dispatcher (data_utils_SSL.py:141-203): 1 convolutive,
    2 impulsive, 3 coloured additive, 4 = 1+2+3 in series, 5 = 1+2, 6 = 1+3, 7 = 2+3,
    8 = 1 and 2 in parallel; anything else leaves the waveform alone."""
    a = args

    def lnl(v):
        return LnL_convolutive_noise(v, a.N_f, a.nBands, a.minF, a.maxF, a.minBW, a.maxBW, a.minCoeff, a.maxCoeff,
                                     a.minG, a.maxG, a.minBiasLinNonLin, a.maxBiasLinNonLin, sr)

    def isd(v):
        return ISD_additive_noise(v, a.P, a.g_sd)

    def ssi(v):
        return SSI_additive_noise(v, a.SNRmin, a.SNRmax, a.nBands, a.minF, a.maxF, a.minBW, a.maxBW, a.minCoeff,
                                  a.maxCoeff, a.minG, a.maxG, sr)

    series = {1: (lnl,), 2: (isd,), 3: (ssi,), 4: (lnl, isd, ssi), 5: (lnl, isd), 6: (lnl, ssi), 7: (isd, ssi)}
    if algo in series:
        for stage in series[algo]:
            feature = stage(feature)
        return feature
    if algo == 8:
        first = lnl(feature)       # evaluated in this order: both consume the random stream
        second = isd(feature)
        return normWav(first + second, 0)
    return feature
