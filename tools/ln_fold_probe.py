#!/usr/bin/env python3
"""Time of the GEMMs that carry a folded LayerNorm against their plain forms (M = 12864, isolated launches back
to back): the residual producer (N = 1024, K = 1024 and 4096) and the consumers (N = 3072, N = 4096 + GELU)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slsmi  # noqa: E402,F401
from slsmi import _lib, ops  # noqa: E402

DEV = torch.device("cuda:0")
M = 12864


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


x_prev = torch.randn(M, 1024, device=DEV)
proxy = torch.stack([x_prev.mean(1), 1 / x_prev.std(1)], 1).contiguous()
gamma = torch.ones(1024, device=DEV)
fc0 = torch.randn(1024, device=DEV) * 0.05
for K in (1024, 4096):
    a = (torch.randn(M, K, device=DEV) * 0.5).half()
    w = (torch.randn(1024, K, device=DEV) * K ** -0.5).half()
    b = torch.zeros(1024, device=DEV)
    out = x_prev.clone()
    t_plain = timed(lambda: ops.gemm_bf16(a, w, b, _lib.GEMM_F32_RESID, out=out, bn=256))
    out2 = torch.empty_like(x_prev)
    lib = _lib.load()
    ln_a = torch.empty(M, 1024, dtype=torch.float16, device=DEV)
    slots = torch.zeros(16, M, 2, device=DEV)
    import ctypes as C
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
    S = C.c_void_p(torch.cuda.current_stream(DEV).cuda_stream)

    def emit():
        rc = lib.slsmi_gemm_bf16_ln(P(a), K, P(w), K, P(b), P(x_prev), P(out2), 1024, M, 1024, K, _lib.GEMM_F32_RESID, 1, 1,
                                    P(proxy), P(gamma), P(ln_a), P(slots), None, 0.0, None, None, S)
        assert rc == 0
    t_emit = timed(emit)
    print(f"producer K={K}: plain residual {t_plain:6.1f} us   with LayerNorm emit {t_emit:6.1f} us", flush=True)

ln_a, proxy0, slots0 = ops.ln_fold_prep(x_prev, gamma, 1e-5, torch.float16)
for N, gelu in ((3072, False), (4096, True)):
    w = (torch.randn(N, 1024, device=DEV) / 32).half()
    b = torch.zeros(N, device=DEV)
    g, c = ops.ln_fold_consts(w, b, gamma, torch.zeros(1024, device=DEV))
    mode = _lib.GEMM_BF16_GELU if gelu else _lib.GEMM_BF16
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    t_plain = timed(lambda: ops.gemm_bf16(ln_a, w, b, mode, out=out, bn=256))
    t_apply = timed(lambda: ops.gemm_ln_apply(ln_a, w, c, g, proxy0, slots0, 1e-5, gelu=gelu, want_stat=True))
    print(f"consumer N={N}: plain {t_plain:6.1f} us   with LayerNorm apply {t_apply:6.1f} us (includes two allocations)", flush=True)
