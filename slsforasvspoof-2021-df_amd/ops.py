"""Thin torch-tensor wrappers over libslsmi's single-kernel entry points.

Used by the parity tests (one kernel at a time against an fp32 reference) and
by AutoEncoderTopK's stand-alone encode/decode.  Every function takes ROCm
tensors, enqueues on torch's current stream and returns torch tensors; nothing
here computes on the host.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib


def _p(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _s(dev) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and t.device.type != "cuda":
            raise _lib.SlsmiError("slsmi ops need ROCm tensors; there is no CPU path")


def _fmt(t: torch.Tensor) -> int:
    """Operand format code from a tensor's dtype (bf16 -> 0, f16 -> 1)."""
    if t.dtype == torch.bfloat16:
        return _lib.FMT_BF16
    if t.dtype == torch.float16:
        return _lib.FMT_F16
    raise _lib.SlsmiError(f"16-bit operand expected, got {t.dtype}")


def op_dtype(precision: str) -> torch.dtype:
    return {"bf16": torch.bfloat16, "f16": torch.float16}[precision]


def gemm_bf16(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], mode: int,
              out: Optional[torch.Tensor] = None, bn: int = 0, splitk: int = 1,
              M: Optional[int] = None, K: Optional[int] = None, lda: Optional[int] = None,
              static_scratch: bool = False, ka: int = 0) -> torch.Tensor:
    """C[m][n] (op)= sum_k a[m][k] w[n][k] + bias[n]; a, w bf16.  `M`, `K`, `lda`
    let a caller describe an overlapping-row (Toeplitz) view of `a`.  static_scratch: go through
    slsmi_gemm_bf16_static (the entry point that allocates nothing); `out` must then be given.
    ka > 0: `a` repeats along K with that period (slsmi_gemm_bf16_period): column k reads a[m][k mod ka]."""
    _need_cuda(a, w, bias, out)
    assert a.dtype == w.dtype
    fmt = _fmt(a)
    lib = _lib.load()
    N = w.shape[0]
    K = w.shape[1] if K is None else K
    M = a.shape[0] if M is None else M
    lda = a.stride(0) if lda is None else lda
    if out is None:
        dt = a.dtype if mode in (_lib.GEMM_BF16, _lib.GEMM_BF16_GELU) else torch.float32
        out = (torch.zeros if mode == _lib.GEMM_F32_ATOMIC else torch.empty)(M, N, dtype=dt, device=a.device)
    if ka:
        assert not static_scratch and splitk == 1
        with torch.cuda.device(a.device):
            rc = lib.slsmi_gemm_bf16_period(_p(a), lda, ka, _p(w), w.stride(0), _p(bias), _p(out), out.stride(0), M, N, K,
                                            mode, bn, fmt, _s(a.device))
        _lib.check(rc, None, "slsmi_gemm_bf16_period")
        return out
    fn = lib.slsmi_gemm_bf16_static if static_scratch else lib.slsmi_gemm_bf16
    with torch.cuda.device(a.device):
        rc = fn(_p(a), lda, _p(w), w.stride(0), _p(bias), _p(out), out.stride(0), M, N, K, mode,
                bn, splitk, fmt, _s(a.device))
    _lib.check(rc, None, "slsmi_gemm_bf16_static" if static_scratch else "slsmi_gemm_bf16")
    return out


def mfma_rate(device, dtype: torch.dtype, iters: int = 20000) -> float:
    """slsmi_debug_mfma_rate: dense TFLOP/s of back-to-back 16x16x32 MFMAs from registers on every SIMD of
    `device` (a few milliseconds of full matrix load; synchronises)."""
    import ctypes
    dev = torch.device(device)
    sink = torch.empty(512 * torch.cuda.get_device_properties(dev).multi_processor_count, dtype=torch.float32, device=dev)
    out = ctypes.c_double(0.0)
    fmt = 1 if dtype == torch.float16 else 0
    with torch.cuda.device(dev):
        rc = _lib.load().slsmi_debug_mfma_rate(_p(sink), sink.numel(), fmt, iters, ctypes.byref(out), _s(dev))
    _lib.check(rc, None, "slsmi_debug_mfma_rate")
    return out.value


class gemm_timing:
    """Context manager over slsmi_debug_gemm_timing_begin / _end: every MFMA GEMM launch the library makes inside
    the block (on the current device) is timed where it runs, by two HIP events on its launch stream.  After the
    block: .launches, .dropped, .seconds (sum of the launches' durations), .flops (sum of their 2 M N K)."""

    def __init__(self, device, max_launches: int = 4096):
        self.device = torch.device(device)
        self.max_launches = max_launches
        self.launches = self.dropped = 0
        self.seconds = self.flops = 0.0

    def __enter__(self):
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().slsmi_debug_gemm_timing_begin(self.max_launches), None, "slsmi_debug_gemm_timing_begin")
        return self

    def __exit__(self, *exc):
        import ctypes
        n, d = ctypes.c_int(0), ctypes.c_int(0)
        sec, fl = ctypes.c_double(0.0), ctypes.c_double(0.0)
        with torch.cuda.device(self.device):
            rc = _lib.load().slsmi_debug_gemm_timing_end(ctypes.byref(n), ctypes.byref(d), ctypes.byref(sec), ctypes.byref(fl))
        if exc[0] is None:
            _lib.check(rc, None, "slsmi_debug_gemm_timing_end")
        self.launches, self.dropped, self.seconds, self.flops = n.value, d.value, sec.value, fl.value
        return False


def posconv(xg: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, x: torch.Tensor, T: int, form: int) -> torch.Tensor:
    """slsmi_posconv: x fp32 [B * T][1024] += gelu(grouped conv of the packed input xg [B][16][slab_rows][64]
    with w [1024][128][64] + bias), in place.  form 0: batched GEMM over the Toeplitz view; form 1: the
    slab-resident kernel (T <= 208)."""
    _need_cuda(xg, w, bias, x)
    assert xg.dtype == w.dtype and x.dtype == torch.float32 and xg.is_contiguous() and w.is_contiguous()
    B, groups, slab_rows, ch = xg.shape
    assert groups == 16 and ch == 64 and x.shape == (B * T, 1024) and w.numel() == 1024 * 128 * 64
    with torch.cuda.device(x.device):
        rc = _lib.load().slsmi_posconv(_p(xg), _p(w), _p(bias), _p(x), B, T, slab_rows, _fmt(xg), form, _s(x.device))
    _lib.check(rc, None, "slsmi_posconv")
    return x


def gemm_tn_grouped(dy: torch.Tensor, x: torch.Tensor, N: int, K: int, nz: int, group_rows: int, lda: int, za: int, ga: int,
                    ldx: int, zx: int, gx: int, alpha: float = 1.0) -> torch.Tensor:
    """dW[z][n][k] = alpha * sum_m dy_z[m][n] * x_z[m][k] over M = dy.shape[0] rows that come in groups of group_rows:
    row m of batch entry z starts at element z * za + (m // group_rows) * ga + (m % group_rows) * lda of the flat
    bfloat16 dy and at z * zx + (m // group_rows) * gx + (m % group_rows) * ldx of the flat 16-bit x
    (slsmi_gemm_tn_grouped) -> fp32 [nz, N, K]."""
    _need_cuda(dy, x)
    assert dy.dtype == torch.bfloat16 and dy.is_contiguous() and x.is_contiguous()
    M = dy.shape[0]
    out = torch.empty(nz, N, K, dtype=torch.float32, device=dy.device)
    with torch.cuda.device(dy.device):
        rc = _lib.load().slsmi_gemm_tn_grouped(_p(dy), lda, _p(x), ldx, _fmt(x), _p(out), K, M, N, K, nz, za, zx, N * K,
                                               float(alpha), group_rows, ga, gx, _s(dy.device))
    _lib.check(rc, None, "slsmi_gemm_tn_grouped")
    return out


def gemm_tn(dy: torch.Tensor, x: torch.Tensor, alpha: float = 1.0, K: Optional[int] = None, ldx: Optional[int] = None,
            zx: Optional[int] = None) -> torch.Tensor:
    """dW[z][n][k] = alpha * sum_m dy[z][m][n] * x[z][m][k]: dy bfloat16 [nz, M, N] (or [M, N]), x 16-bit [nz, M, K]
    -> fp32 [nz, N, K] (slsmi_gemm_tn).  K, ldx, zx describe an overlapping-row view of a flat x instead (row m of
    batch entry z starts at element z * zx + m * ldx and is K long: the view a strided Conv1d takes of its input)."""
    _need_cuda(dy, x)
    squeeze = dy.dim() == 2
    if squeeze:
        dy = dy[None]
        x = x[None] if K is None else x
    assert dy.dtype == torch.bfloat16 and dy.is_contiguous() and x.is_contiguous()
    nz, M, N = dy.shape
    if K is None:
        assert dy.shape[:2] == x.shape[:2]
        K = x.shape[2]
        ldx, zx = K, M * K
    else:
        assert (nz - 1) * zx + (M - 1) * ldx + K <= x.numel()
    out = torch.empty(nz, N, K, dtype=torch.float32, device=dy.device)
    with torch.cuda.device(dy.device):
        rc = _lib.load().slsmi_gemm_tn(_p(dy), N, _p(x), ldx, _fmt(x), _p(out), K, M, N, K, nz, M * N, zx, N * K,
                                       float(alpha), _s(dy.device))
    _lib.check(rc, None, "slsmi_gemm_tn")
    return out[0] if squeeze else out


def gemm_ln_emit(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], resid: torch.Tensor,
                 proxy: torch.Tensor, gamma: torch.Tensor, out: Optional[torch.Tensor] = None,
                 dot_w: Optional[torch.Tensor] = None):
    """The residual GEMM in front of a folded LayerNorm (slsmi_gemm_bf16_ln, ln_mode 1): returns
    (C fp32 [M][1024], ln_a 16-bit [M][1024], slots fp32 [16][M][2], row dots [16][M] or None)."""
    _need_cuda(a, w, bias, resid, proxy, gamma, out, dot_w)
    M, N = a.shape[0], w.shape[0]
    out = torch.empty(M, N, dtype=torch.float32, device=a.device) if out is None else out
    ln_a = torch.empty(M, N, dtype=a.dtype, device=a.device)
    slots = torch.zeros(16, M, 2, dtype=torch.float32, device=a.device)
    dots = torch.zeros(N // 64, M, dtype=torch.float32, device=a.device) if dot_w is not None else None
    with torch.cuda.device(a.device):
        rc = _lib.load().slsmi_gemm_bf16_ln(_p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(resid), _p(out),
                                            out.stride(0), M, N, w.shape[1], _lib.GEMM_F32_RESID, _fmt(a), 1, _p(proxy),
                                            _p(gamma), _p(ln_a), _p(slots), None, 0.0, _p(dot_w), _p(dots),
                                            _s(a.device))
    _lib.check(rc, None, "slsmi_gemm_bf16_ln (emit)")
    return out, ln_a, slots, dots


def gemm_ln_apply(ln_a: torch.Tensor, w: torch.Tensor, c: torch.Tensor, g: torch.Tensor, proxy: torch.Tensor,
                  slots: torch.Tensor, eps: float, gelu: bool = False, want_stat: bool = True):
    """The GEMM behind a folded LayerNorm (slsmi_gemm_bf16_ln, ln_mode 2): returns (C 16-bit [M][N], stat [M][2])."""
    _need_cuda(ln_a, w, c, g, proxy, slots)
    M, N = ln_a.shape[0], w.shape[0]
    out = torch.empty(M, N, dtype=ln_a.dtype, device=ln_a.device)
    stat = torch.zeros(M, 2, dtype=torch.float32, device=ln_a.device) if want_stat else None
    with torch.cuda.device(ln_a.device):
        rc = _lib.load().slsmi_gemm_bf16_ln(_p(ln_a), ln_a.stride(0), _p(w), w.stride(0), _p(c), None, _p(out),
                                            out.stride(0), M, N, w.shape[1],
                                            _lib.GEMM_BF16_GELU if gelu else _lib.GEMM_BF16, _fmt(ln_a), 2, _p(proxy),
                                            _p(g), None, _p(slots), _p(stat), eps, None, None, _s(ln_a.device))
    _lib.check(rc, None, "slsmi_gemm_bf16_ln (apply)")
    return out, stat


def ln_fold_prep(x: torch.Tensor, gamma: torch.Tensor, eps: float, dtype: torch.dtype):
    """slsmi_ln_fold_prep: x fp32 [rows][1024] -> (ln_a 16-bit, proxy [rows][2], slots [16][rows][2])."""
    _need_cuda(x, gamma)
    rows = x.shape[0]
    ln_a = torch.empty(rows, 1024, dtype=dtype, device=x.device)
    proxy = torch.empty(rows, 2, dtype=torch.float32, device=x.device)
    slots = torch.empty(16, rows, 2, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load().slsmi_ln_fold_prep(_p(x), _p(ln_a), _p(proxy), _p(slots), rows, _p(gamma), eps, _fmt(ln_a),
                                            _s(x.device))
    _lib.check(rc, None, "slsmi_ln_fold_prep")
    return ln_a, proxy, slots


def ln_fold_rows(x: torch.Tensor, proxy: Optional[torch.Tensor], gamma: torch.Tensor, eps: float, dtype: torch.dtype):
    """slsmi_ln_fold_rows: x fp32 [rows][1024] -> (ln_a 16-bit, aff [rows][2], stat [rows][2])."""
    _need_cuda(x, proxy, gamma)
    rows = x.shape[0]
    ln_a = torch.empty(rows, 1024, dtype=dtype, device=x.device)
    aff = torch.empty(rows, 2, dtype=torch.float32, device=x.device)
    stat = torch.empty(rows, 2, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load().slsmi_ln_fold_rows(_p(x), _p(proxy), _p(ln_a), _p(aff), _p(stat), rows, _p(gamma), eps, _fmt(ln_a),
                                            _s(x.device))
    _lib.check(rc, None, "slsmi_ln_fold_rows")
    return ln_a, aff, stat


def gemm_ln_apply_rows(ln_a: torch.Tensor, w: torch.Tensor, c: torch.Tensor, g: torch.Tensor, aff: torch.Tensor,
                       gelu: bool = False, bn: int = 0) -> torch.Tensor:
    """slsmi_gemm_bf16_ln_rows: the GEMM behind a folded LayerNorm with the rows' factors given."""
    _need_cuda(ln_a, w, c, g, aff)
    M, N = ln_a.shape[0], w.shape[0]
    out = torch.empty(M, N, dtype=ln_a.dtype, device=ln_a.device)
    with torch.cuda.device(ln_a.device):
        rc = _lib.load().slsmi_gemm_bf16_ln_rows(_p(ln_a), ln_a.stride(0), _p(w), w.stride(0), _p(c), _p(g), _p(aff), _p(out),
                                                 out.stride(0), M, N, w.shape[1],
                                                 _lib.GEMM_BF16_GELU if gelu else _lib.GEMM_BF16, bn, _fmt(ln_a),
                                                 _s(ln_a.device))
    _lib.check(rc, None, "slsmi_gemm_bf16_ln_rows")
    return out


def ln_fold_consts(w: torch.Tensor, bias: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor):
    """slsmi_ln_fold_consts: (g, c) fp32 [N] for a 16-bit w [N][1024]."""
    _need_cuda(w, bias, gamma, beta)
    N = w.shape[0]
    g = torch.empty(N, dtype=torch.float32, device=w.device)
    c = torch.empty(N, dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        rc = _lib.load().slsmi_ln_fold_consts(_p(w), _p(bias), _p(gamma), _p(beta), _p(g), _p(c), N, _fmt(w), _s(w.device))
    _lib.check(rc, None, "slsmi_ln_fold_consts")
    return g, c


def time_gemm_bf16(a, w, bias, out, mode: int, M: int, K: int, lda: int, reps: int, bn: int = 0,
                   splitk: int = 1) -> float:
    """Seconds per launch of one GEMM shape, from HIP events recorded inside the
    library on the launch stream (slsmi_time_gemm_bf16)."""
    _need_cuda(a, w, bias, out)
    ms = C.c_float(0.0)
    with torch.cuda.device(a.device):
        rc = _lib.load().slsmi_time_gemm_bf16(_p(a), lda, _p(w), w.stride(0), _p(bias), _p(out), out.stride(0), M,
                                              w.shape[0], K, mode, bn, splitk, _fmt(a), reps, C.byref(ms),
                                              _s(a.device))
    _lib.check(rc, None, "slsmi_time_gemm_bf16")
    return ms.value * 1e-3 / reps


def linear_bf16(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
                dtype: torch.dtype = torch.float16) -> torch.Tensor:
    """fp32 in/out linear layer on the 16-bit-operand MFMA GEMM (K padded to 64)."""
    _need_cuda(x, weight)
    K = x.shape[-1]
    Kp = (K + 63) // 64 * 64
    x2 = x.reshape(-1, K)
    a = torch.zeros(x2.shape[0], Kp, dtype=dtype, device=x.device)
    a[:, :K] = x2
    N = weight.shape[0]
    Np = (N + 63) // 64 * 64
    w = torch.zeros(Np, Kp, dtype=dtype, device=x.device)
    w[:N, :K] = weight
    b = None
    if bias is not None:
        b = torch.zeros(Np, dtype=torch.float32, device=x.device)
        b[:N] = bias
    out = gemm_bf16(a, w, b, _lib.GEMM_F32, bn=128 if Np % 128 == 0 else 64)
    return out[:, :N].reshape(*x.shape[:-1], N)


def sae_encode(x: torch.Tensor, enc_w: torch.Tensor, enc_b: torch.Tensor, b_dec: torch.Tensor, k: int,
               dtype: torch.dtype = torch.float16) -> torch.Tensor:
    _need_cuda(x, enc_w)
    x2 = (x.reshape(-1, x.shape[-1]).float() - b_dec).to(dtype).contiguous()
    enc = gemm_bf16(x2, enc_w.to(dtype).contiguous(), enc_b.float().contiguous(), _lib.GEMM_F32_RELU)
    topk_rows(enc, k)
    return enc.reshape(*x.shape[:-1], enc_w.shape[0])


def conv0_ln_gelu(wav, w, bias, gamma, beta, eps, T, rows_per_utt=None, dtype=torch.bfloat16) -> torch.Tensor:
    _need_cuda(wav, w)
    B, L = wav.shape
    rows = T if rows_per_utt is None else rows_per_utt
    out = torch.zeros(B, rows, 512, dtype=dtype, device=wav.device)
    with torch.cuda.device(wav.device):
        rc = _lib.load().slsmi_conv0_ln_gelu(_p(wav), B, L, T, _p(w), _p(bias), _p(gamma), _p(beta), eps, _p(out),
                                             rows, _fmt(out), _s(wav.device))
    _lib.check(rc, None, "slsmi_conv0_ln_gelu")
    return out


def ln512_f32(x: torch.Tensor, gamma, beta, eps, gelu: bool, dtype=torch.float16) -> torch.Tensor:
    """LayerNorm(512) (+ GELU) of fp32 rows -> 16-bit rows (slsmi_ln512_f32)."""
    _need_cuda(x)
    assert x.shape[-1] == 512 and x.is_contiguous() and x.dtype == torch.float32
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load().slsmi_ln512_f32(_p(x), _p(out), x.numel() // 512, _p(gamma), _p(beta), eps, int(gelu),
                                         _fmt(out), _s(x.device))
    _lib.check(rc, None, "slsmi_ln512_f32")
    return out


def ln512_split3(x: torch.Tensor, gamma, beta, eps) -> torch.Tensor:
    """LayerNorm(512) of 16-bit rows -> rows of 1536: [hi | lo | hi] (slsmi_ln512_split3)."""
    _need_cuda(x)
    assert x.shape[-1] == 512 and x.is_contiguous()
    out = torch.empty(x.shape[:-1] + (1536,), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load().slsmi_ln512_split3(_p(x), _p(out), x.numel() // 512, _p(gamma), _p(beta), eps, _fmt(x),
                                            _s(x.device))
    _lib.check(rc, None, "slsmi_ln512_split3")
    return out


def ln512_bf16(x, gamma, beta, eps, gelu: bool) -> torch.Tensor:
    _need_cuda(x)
    assert x.shape[-1] == 512 and x.is_contiguous()
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        rc = _lib.load().slsmi_ln512_bf16(_p(x), _p(out), x.numel() // 512, _p(gamma), _p(beta), eps, int(gelu),
                                          _fmt(x), _s(x.device))
    _lib.check(rc, None, "slsmi_ln512_bf16")
    return out


def ln1024_f32_bf16(x, gamma, beta, eps, dtype=torch.bfloat16) -> torch.Tensor:
    _need_cuda(x)
    assert x.dtype == torch.float32 and x.shape[-1] == 1024 and x.is_contiguous()
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load().slsmi_ln1024_f32_bf16(_p(x), _p(out), x.numel() // 1024, _p(gamma), _p(beta), eps,
                                               _fmt(out), _s(x.device))
    _lib.check(rc, None, "slsmi_ln1024_f32_bf16")
    return out


def attention(qkv: torch.Tensor, B: int, T: int, nframes: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nframes: None, or int32 [B] on the device: utterance b attends to its first nframes[b] keys."""
    _need_cuda(qkv)
    assert qkv.shape == (B * T, 3072) and qkv.is_contiguous()
    out = torch.empty(B * T, 1024, dtype=qkv.dtype, device=qkv.device)
    with torch.cuda.device(qkv.device):
        if nframes is None:
            rc = _lib.load().slsmi_attention(_p(qkv), _p(out), B, T, _fmt(qkv), _s(qkv.device))
        else:
            _need_cuda(nframes)
            assert nframes.dtype == torch.int32 and nframes.shape == (B,) and nframes.is_contiguous()
            rc = _lib.load().slsmi_attention_varlen(_p(qkv), _p(out), B, T, _p(nframes), _fmt(qkv), _s(qkv.device))
    _lib.check(rc, None, "slsmi_attention")
    return out


def gemm_resid_rowdot(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], c: torch.Tensor,
                      dot_w: torch.Tensor, bn: int = 0, resid: Optional[torch.Tensor] = None) -> torch.Tensor:
    """c = resid + a w^T + bias (fp32; resid None: c is updated in place), returning
    dots[s][m] = c_new[m][64 s : 64 s + 64] . dot_w[64 s : 64 s + 64]."""
    _need_cuda(a, w, bias, c, dot_w, resid)
    M, K = a.shape
    N = w.shape[0]
    if resid is not None:
        assert resid.shape == c.shape and resid.stride() == c.stride() and resid.dtype == torch.float32
    dots = torch.empty(N // 64, M, dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        rc = _lib.load().slsmi_gemm_resid_rowdot(_p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(resid), _p(c),
                                                 c.stride(0), M, N, K, bn, _fmt(a), _p(dot_w), _p(dots), _s(a.device))
    _lib.check(rc, None, "slsmi_gemm_resid_rowdot")
    return dots


def sls_weights(dots: torch.Tensor, fc0_b: torch.Tensor, B: int, T: int) -> torch.Tensor:
    """dots [layers][slices][B T] (row dots of every layer output with fc0.weight) -> w [layers][B]."""
    _need_cuda(dots, fc0_b)
    layers, slices, M = dots.shape
    assert M == B * T and dots.is_contiguous() and dots.dtype == torch.float32
    w = torch.empty(layers, B, dtype=torch.float32, device=dots.device)
    with torch.cuda.device(dots.device):
        rc = _lib.load().slsmi_sls_weights(_p(dots), slices * M, layers, slices, _p(fc0_b), _p(w), B, T, _s(dots.device))
    _lib.check(rc, None, "slsmi_sls_weights")
    return w


def sls_sum_pool(xs: torch.Tensor, w: torch.Tensor, B: int, T: int, scale: float, shift: float,
                 dtype=torch.bfloat16) -> torch.Tensor:
    """xs [layers][B T][1024] fp32, w [layers][B] -> pooled [B][ld]: max_pool2d(selu(scale * sum_l w_l x_l + shift), 3)."""
    _need_cuda(xs, w)
    layers = xs.shape[0]
    assert xs.shape[1:] == (B * T, 1024) and xs.is_contiguous() and xs.dtype == torch.float32
    assert w.shape == (layers, B) and w.is_contiguous() and w.dtype == torch.float32
    n = (T // 3) * (1024 // 3)
    ld = (n + 63) // 64 * 64
    out = torch.empty(B, ld, dtype=dtype, device=xs.device)
    with torch.cuda.device(xs.device):
        rc = _lib.load().slsmi_sls_sum_pool(_p(xs), B * T * 1024, layers, _p(w), _p(out), ld, B, T, scale, shift,
                                            _fmt(out), _s(xs.device))
    _lib.check(rc, None, "slsmi_sls_sum_pool")
    return out


def topk_rows(enc: torch.Tensor, k: int, want_bf16: bool = False, dtype=torch.bfloat16):
    _need_cuda(enc)
    assert enc.dtype == torch.float32 and enc.is_contiguous()
    rows, dict_ = enc.shape
    eb = torch.empty(rows, dict_, dtype=dtype, device=enc.device) if want_bf16 else None
    with torch.cuda.device(enc.device):
        rc = _lib.load().slsmi_topk_rows(_p(enc), _p(eb), rows, dict_, k,
                                         _fmt(eb) if eb is not None else 0, _s(enc.device))
    _lib.check(rc, None, "slsmi_topk_rows")
    return eb


def window_topk(acts: torch.Tensor, k: int, window: int, want_16bit: bool = False, dtype=torch.bfloat16):
    """acts fp32 [B, T, D] non-negative -> the window TopK sparse code (model_window_topk.py:118-203),
    computed on a copy; with want_16bit also its 16-bit copy."""
    _need_cuda(acts)
    B, T, D = acts.shape
    out = acts.to(torch.float32).contiguous().clone()
    lib = _lib.load()
    n = lib.slsmi_window_topk_scratch_bytes(B, T, D, window)
    if n == 0:
        raise _lib.SlsmiError(f"window TopK needs T >= window >= 2 (T={T}, window={window})")
    scratch = torch.empty(n, dtype=torch.uint8, device=acts.device)
    o16 = torch.empty(B, T, D, dtype=dtype, device=acts.device) if want_16bit else None
    with torch.cuda.device(acts.device):
        rc = lib.slsmi_window_topk(_p(out), _p(o16), _p(scratch), n, B, T, D, k, window,
                                   _lib.FORMATS["f16" if dtype == torch.float16 else "bf16"], _s(acts.device))
    _lib.check(rc, None, "slsmi_window_topk")
    return (out, o16) if want_16bit else out
