"""Seeded synthetic test clips with some variety (test infrastructure, shared by tests/golden/make_oracle_scores.py
and the GPU parity tests): white noise clips all score alike under seeded random weights, so each clip here mixes
noise with two tones under a slow envelope, every parameter drawn from the clip's own generator."""
import math

import torch


def clip(i: int, L: int = 64600) -> torch.Tensor:
    g = torch.Generator().manual_seed(7_000_000 + int(i))
    u = torch.rand(8, generator=g, dtype=torch.float64)
    noise = torch.randn(L, generator=g, dtype=torch.float32).double()
    t = torch.arange(L, dtype=torch.float64) / 16000.0
    amp = 0.02 + 0.4 * u[0]
    f1, f2 = 80.0 + 3000.0 * u[1], 200.0 + 6000.0 * u[2]
    x = (0.2 + 0.8 * u[3]) * 0.3 * noise + u[4] * torch.sin(2 * math.pi * f1 * t) \
        + u[5] * torch.sin(2 * math.pi * f2 * t + 2 * math.pi * u[6])
    env = 0.5 + 0.5 * torch.sin(2 * math.pi * (0.5 + 4.0 * u[7]) * t)
    return (amp * env * x).clamp(-1.0, 1.0).float()


def clips(first: int, count: int, L: int = 64600) -> torch.Tensor:
    return torch.stack([clip(first + j, L) for j in range(count)])
