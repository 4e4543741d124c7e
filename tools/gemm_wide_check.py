#!/usr/bin/env python3
"""Checks the 208 x 256 GEMM kernel (bn = 256) against the 128-row kernel (bn = 128; same bits expected) and an
fp32 product, then times it against the 256 x 128 kernel (bn = 1) on the forward's shapes.
usage: gemm_wide_check.py [check] [time]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slsmi  # noqa: E402,F401
from slsmi import _lib, ops  # noqa: E402

DEV = torch.device("cuda:0")
what = sys.argv[1:] or ["check", "time"]


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


if "check" in what:
    bad = 0
    for dt in (torch.float16, torch.bfloat16):
        for (M, N, K) in ((201, 256, 256), (208, 256, 256), (100, 256, 512), (12864, 1024, 1024), (12864, 3072, 1024),
                          (5001, 512, 512), (14070, 1024, 768), (629, 256, 4096), (12864, 4096, 1024)):
            a = rnd(M, K, seed=1).to(dt).to(DEV)
            w = rnd(N, K, seed=2, scale=K ** -0.5).to(dt).to(DEV)
            b = rnd(N, seed=3, scale=0.1).to(DEV)
            old = rnd(M, N, seed=4).to(DEV)
            for mode in (_lib.GEMM_BF16, _lib.GEMM_BF16_GELU, _lib.GEMM_F32, _lib.GEMM_F32_RESID, _lib.GEMM_F32_RELU):
                outs = []
                for bn in (128, 256):
                    odt = dt if mode in (_lib.GEMM_BF16, _lib.GEMM_BF16_GELU) else torch.float32
                    out = old.clone().to(odt)
                    ops.gemm_bf16(a, w, b, mode, out=out, bn=bn)
                    outs.append(out)
                torch.cuda.synchronize()
                same = torch.equal(outs[0], outs[1])
                d = (outs[0].float() - outs[1].float()).abs().max().item()
                nbad = (outs[0] != outs[1]).sum().item()
                if not same:
                    bad += 1
                    idx = (outs[0] != outs[1]).nonzero()
                    print(f"MISMATCH {dt} M={M} N={N} K={K} mode={mode}: {nbad} elements, max diff {d:.3e}, first {idx[0].tolist()} "
                          f"last {idx[-1].tolist()} rows {idx[:, 0].unique()[:12].tolist()} cols {idx[:, 1].unique()[:12].tolist()}",
                          flush=True)
            print(f"{dt} {M}x{N}x{K} done", flush=True)
    # row dots
    M, K = 12864, 4096
    a = rnd(M, K, seed=60).half().to(DEV)
    w = rnd(1024, K, seed=61, scale=K ** -0.5).half().to(DEV)
    bias, x0, w0 = rnd(1024, seed=62, scale=0.1).to(DEV), rnd(M, 1024, seed=63).to(DEV), rnd(1024, seed=64, scale=1 / 32).to(DEV)
    res = []
    for bn in (128, 256):
        c = torch.full((M, 1024), float("nan"), device=DEV)
        dots = ops.gemm_resid_rowdot(a, w, bias, c, w0, bn=bn, resid=x0)
        res.append((c, dots))
    torch.cuda.synchronize()
    print("rowdot C equal", torch.equal(res[0][0], res[1][0]), "dots equal", torch.equal(res[0][1], res[1][1]),
          "dots max diff", (res[0][1] - res[1][1]).abs().max().item())
    # conv view (overlapping rows)
    M, N, K, lda = 26624, 512, 1536, 1024
    x = rnd((M - 1) * lda + K + 64, seed=7).half().to(DEV)
    wk = rnd(N, K, seed=8, scale=K ** -0.5).half().to(DEV)
    o = [ops.gemm_bf16(x, wk, None, _lib.GEMM_BF16, bn=bn, M=M, K=K, lda=lda) for bn in (128, 256)]
    torch.cuda.synchronize()
    print("conv view equal", torch.equal(o[0], o[1]))
    print("MISMATCHES:", bad)

if "time" in what:
    for (M, N, K, lda, mode) in ((12864, 3072, 1024, 1024, _lib.GEMM_BF16), (12864, 1024, 1024, 1024, _lib.GEMM_F32_RESID),
                                 (12864, 4096, 1024, 1024, _lib.GEMM_BF16_GELU), (12864, 1024, 4096, 4096, _lib.GEMM_F32_RESID),
                                 (12864, 1024, 512, 512, _lib.GEMM_F32), (425984, 512, 1536, 1024, _lib.GEMM_BF16),
                                 (212992, 512, 1536, 1024, _lib.GEMM_BF16), (53248, 512, 1536, 1024, _lib.GEMM_BF16),
                                 (26624, 512, 1024, 1024, _lib.GEMM_BF16), (13312, 512, 1024, 1024, _lib.GEMM_BF16)):
        a = (torch.randn((M - 1) * lda + K + 64, device=DEV) * 0.5).half()
        w = (torch.randn(N, K, device=DEV) * K ** -0.5).half()
        b = torch.zeros(N, device=DEV)
        odt = torch.float16 if mode in (_lib.GEMM_BF16, _lib.GEMM_BF16_GELU) else torch.float32
        out = torch.zeros(M, N, dtype=odt, device=DEV)
        line = f"{M}x{N}x{K} mode {mode}:"
        for bn in (1, 256, 1, 256):
            sec = ops.time_gemm_bf16(a, w, b, out, mode, M, K, lda, reps=20, bn=bn)
            line += f"  bn={bn}: {1e6 * sec:7.1f} us {2.0 * M * N * K / sec / 1e12:6.0f} TF"
        print(line, flush=True)
