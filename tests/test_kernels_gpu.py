"""Single-kernel parity through the C ABI: each HIP kernel against a plain
PyTorch fp32 computation of the same op on the same (bf16-rounded) inputs."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import slsmi
from slsmi import _lib, ops

DEV = "cuda:0"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


DT = torch.bfloat16   # overridden per test by the `fmt` fixture


@pytest.fixture(params=["bf16", "f16"], autouse=True)
def fmt(request):
    """Every kernel test runs once per 16-bit operand format."""
    global DT
    DT = ops.op_dtype(request.param)
    yield request.param


def bf(x):
    return x.to(DT)


def tol16():
    """Rounding step of the storage format: 2^-8 (bf16) or 2^-11 (f16), with margin."""
    return 2e-2 if DT == torch.bfloat16 else 3e-3


@pytest.mark.parametrize("M,N,K,bn", [(128, 128, 64, 128), (201, 512, 1536, 128), (1000, 3072, 1024, 128),
                                      (77, 64, 8192, 64), (402, 1024, 4096, 128), (64, 1024, 512, 128),
                                      # bn = 0: library's choice -> 256 x 128 three-stage kernel (M >= 1024);
                                      # K = 64 / 128 / 192 exercise its one-, two- and three-tile pipelines
                                      (1100, 128, 64, 0), (1100, 256, 128, 0), (1300, 128, 192, 0),
                                      (2049, 1024, 1024, 0), (1024, 3072, 512, 0),
                                      # more output tiles than CUs: a workgroup's finished tile leaves during the
                                      # K loop of its next one (K = 768 is the shortest such loop); ragged M
                                      (12864, 1024, 1024, 0), (5001, 3072, 768, 0), (9000, 1024, 4096, 0)])
@pytest.mark.parametrize("mode", [_lib.GEMM_BF16, _lib.GEMM_BF16_GELU, _lib.GEMM_F32, _lib.GEMM_F32_RESID,
                                  _lib.GEMM_F32_RESID_GELU, _lib.GEMM_F32_RELU])
def test_gemm_modes(M, N, K, bn, mode):
    a, w, b = bf(rnd(M, K, seed=1)), bf(rnd(N, K, seed=2, scale=K ** -0.5)), rnd(N, seed=3, scale=0.1)
    ref = a.float() @ w.float().t() + b
    old = rnd(M, N, seed=4)
    if mode in (_lib.GEMM_BF16_GELU, _lib.GEMM_F32_RESID_GELU):
        ref = F.gelu(ref)
    if mode == _lib.GEMM_F32_RELU:
        ref = F.relu(ref)
    if mode in (_lib.GEMM_F32_RESID, _lib.GEMM_F32_RESID_GELU):
        ref = ref + old
    out = old.to(DEV).clone() if mode in (_lib.GEMM_F32_RESID, _lib.GEMM_F32_RESID_GELU) else None
    got = ops.gemm_bf16(a.to(DEV), w.to(DEV), b.to(DEV), mode, out=out, bn=bn).float().cpu()
    tol = tol16() if mode in (_lib.GEMM_BF16, _lib.GEMM_BF16_GELU) else 2e-4
    torch.testing.assert_close(got, ref, atol=tol, rtol=tol)


WIDE_SHAPES = [(12864, 1024, 1024), (12864, 3072, 1024), (5001, 512, 512), (14070, 1024, 768), (9000, 256, 4096),
               (4824, 2048, 256)]


@pytest.mark.parametrize("M,N,K", WIDE_SHAPES)
@pytest.mark.parametrize("mode", [_lib.GEMM_BF16, _lib.GEMM_BF16_GELU, _lib.GEMM_F32, _lib.GEMM_F32_RESID,
                                  _lib.GEMM_F32_RELU])
def test_gemm_208x256_kernel(M, N, K, mode):
    """The 208 x 256 kernel (bn = 256) against the fp32 product and, bit for bit, against the 128-row kernel
    (bn = 128; every shape here has too many tiles for the library to cut K): both start from the bias and add
    the K-tiles in the same order.  Shapes: one tile per workgroup;
    three tiles per workgroup (the operand stream runs on across output tiles); rows not a multiple of 201 (tiles
    every 208 rows, ragged last tile row); more tiles than CUs with a ragged count; the shortest and a long K loop."""
    a, w, b = bf(rnd(M, K, seed=1)), bf(rnd(N, K, seed=2, scale=K ** -0.5)), rnd(N, seed=3, scale=0.1)
    old = rnd(M, N, seed=4)
    ref = a.float() @ w.float().t() + b
    if mode == _lib.GEMM_BF16_GELU:
        ref = F.gelu(ref)
    if mode == _lib.GEMM_F32_RELU:
        ref = F.relu(ref)
    if mode == _lib.GEMM_F32_RESID:
        ref = ref + old
    outs = []
    for bn in (256, 128):
        out = old.to(DEV).clone() if mode == _lib.GEMM_F32_RESID else None
        outs.append(ops.gemm_bf16(a.to(DEV), w.to(DEV), b.to(DEV), mode, out=out, bn=bn))
    tol = tol16() if mode in (_lib.GEMM_BF16, _lib.GEMM_BF16_GELU) else 2e-4
    torch.testing.assert_close(outs[0].float().cpu(), ref, atol=tol, rtol=tol)
    assert torch.equal(outs[0], outs[1])


def test_gemm_208x256_rows_past_the_end_are_not_written():
    """M = 201 * 3 + 50: the last tile row holds 50 rows; the rows behind the matrix stay as they were."""
    M, N, K = 653, 256, 256
    a, w = bf(rnd(M, K, seed=5)), bf(rnd(N, K, seed=6, scale=K ** -0.5))
    buf = torch.full((M + 300, N), 7.0, device=DEV)
    ops.gemm_bf16(a.to(DEV), w.to(DEV), None, _lib.GEMM_F32, out=buf[:M], bn=256)
    torch.testing.assert_close(buf[:M].cpu(), a.float() @ w.float().t(), atol=2e-4, rtol=2e-4)
    assert bool((buf[M:] == 7.0).all())


def test_gemm_208x256_rejects_what_it_cannot_do():
    a, w = bf(rnd(300, 256)), bf(rnd(128, 256))
    with pytest.raises(_lib.SlsmiError):
        ops.gemm_bf16(a.to(DEV), w.to(DEV), None, _lib.GEMM_F32, bn=256)   # N % 256 != 0
    a, w = bf(rnd(300, 128)), bf(rnd(256, 128))
    with pytest.raises(_lib.SlsmiError):
        ops.gemm_bf16(a.to(DEV), w.to(DEV), None, _lib.GEMM_F32, bn=256)   # fewer than four K-tiles


def test_gemm_repeat_and_interleave():
    """The forward's large shapes (more output tiles than CUs, not a whole number of rounds), launched back to
    back in changing order: the same bits every time, and the fp32 product."""
    shapes = [(12864, 1024, 1024), (12864, 3072, 1024), (12864, 4096, 1024), (12864, 1024, 4096), (7000, 2048, 768)]
    data = []
    for i, (M, N, K) in enumerate(shapes):
        a, w, b = bf(rnd(M, K, seed=70 + i)), bf(rnd(N, K, seed=80 + i, scale=K ** -0.5)), rnd(N, seed=90 + i, scale=0.1)
        data.append((a.to(DEV), w.to(DEV), b.to(DEV), a.float() @ w.float().t() + b))
    first = {}
    for rep in range(3):
        for i in (0, 3, 1, 4, 2, 0, 3):
            a, w, b, ref = data[i]
            got = ops.gemm_bf16(a, w, b, _lib.GEMM_F32, bn=0)
            if i not in first:
                first[i] = got.clone()
                torch.testing.assert_close(got.cpu(), ref, atol=2e-4, rtol=2e-4)
            else:
                assert torch.equal(got, first[i]), shapes[i]


def test_gemm_operand_orientation_exact():
    """A = identity against an asymmetric W: catches a transposed accumulator map."""
    K = N = 128
    a = bf(torch.eye(K))
    w = bf((torch.arange(N)[:, None] * 3 + torch.arange(K)[None, :] * 7).float() % 61)
    got = ops.gemm_bf16(a.to(DEV), w.to(DEV), None, _lib.GEMM_F32).cpu()
    assert torch.equal(got, w.float().t())


def test_gemm_splitk_atomic():
    M, N, K = 64, 1024, 22848
    a, w = bf(rnd(M, K, seed=5)), bf(rnd(N, K, seed=6, scale=K ** -0.5))
    ref = a.float() @ w.float().t()
    got = ops.gemm_bf16(a.to(DEV), w.to(DEV), None, _lib.GEMM_F32_ATOMIC, splitk=17).cpu()
    torch.testing.assert_close(got, ref, atol=5e-4, rtol=5e-4)


def test_gemm_overlapping_rows_conv_view():
    """Conv1d(512,512,k=3,s=2) as a GEMM over overlapping rows (lda = 2*512 < K = 3*512)."""
    T_in, T_out = 41, 20
    x = bf(rnd(T_in, 512, seed=7))
    wt = rnd(512, 512, 3, seed=8, scale=(512 * 3) ** -0.5)
    ref = F.conv1d(x.float().t()[None], bf(wt).float(), stride=2)[0].t()  # [T_out, 512]
    wk = bf(wt.permute(0, 2, 1).reshape(512, 1536).contiguous())
    got = ops.gemm_bf16(x.to(DEV), wk.to(DEV), None, _lib.GEMM_F32, M=T_out, K=1536, lda=1024).cpu()
    torch.testing.assert_close(got, ref, atol=3e-4, rtol=3e-4)


@pytest.mark.parametrize("B,T", [(2, 201), (3, 201), (1, 208), (5, 49), (1, 1), (1, 209), (3, 300), (2, 416), (3, 499)])
def test_positional_conv_forms(B, T):
    """The encoder's grouped positional conv (wav2vec/wav2vec2.py:915-917) on packed input: the batched-GEMM form
    and the slab-resident kernel give the same bits, and both match fp32 conv1d + GELU + residual on the same
    16-bit-rounded operands."""
    slab = (T + 128 + 7) // 8 * 8
    x0 = rnd(B * T, 1024, seed=3)
    xin = bf(rnd(B, T, 1024, seed=4))                                   # the conv's input, already rounded
    w = bf(rnd(1024, 64, 128, seed=5, scale=(64 * 128) ** -0.5))         # torch layout [out][in / groups][tap]
    bias = rnd(1024, seed=6, scale=0.1)
    xg = torch.zeros(B, 16, slab, 64, dtype=DT)
    xg[:, :, 64:64 + T, :] = xin.view(B, T, 16, 64).permute(0, 2, 1, 3)
    wp = w.permute(0, 2, 1).contiguous()                                 # [out][tap][in]
    ref = F.conv1d(xin.float().transpose(1, 2), w.float(), bias, padding=64, groups=16)[:, :, :T]   # SamePad: even kernel
    ref = x0 + F.gelu(ref.transpose(1, 2)).reshape(B * T, 1024)
    outs = []
    for form in (0, 1):
        x = x0.clone().to(DEV)
        ops.posconv(xg.to(DEV), wp.to(DEV), bias.to(DEV), x, T, form)
        outs.append(x.cpu())
        assert (outs[-1] - ref).abs().max().item() < 2e-4, form
    assert torch.equal(outs[0], outs[1])


def test_positional_conv_slab_kernel_rejects_a_short_slab():
    T, slab = 300, 416      # the frames read 127 rows past themselves: 427 rows, in whole 8-row pieces 432
    with pytest.raises(slsmi.SlsmiError):
        ops.posconv(torch.zeros(1, 16, slab, 64, dtype=DT, device=DEV), torch.zeros(1024, 128, 64, dtype=DT, device=DEV),
                    torch.zeros(1024, device=DEV), torch.zeros(T, 1024, device=DEV), T, 1)


@pytest.mark.parametrize("nz,M,N,K,xdt", [(1, 1005, 1024, 128, torch.float16), (3, 70, 128, 256, torch.bfloat16),
                                          (2, 1, 128, 128, torch.float16), (1, 2211, 256, 1024, torch.float16)])
def test_weight_gradient_contraction_from_row_major_operands(nz, M, N, K, xdt):
    """slsmi_gemm_tn: dW = alpha dY^T X from row-major dY (bfloat16) and X (16-bit), against fp32 on the same rounded
    operands; rows past M contribute nothing (M is not a multiple of the 64-row K-tile)."""
    dy = rnd(nz, M, N, seed=21).to(torch.bfloat16)
    x = rnd(nz, M, K, seed=22).to(xdt)
    ref = 0.125 * torch.einsum("zmn,zmk->znk", dy.float(), x.to(torch.bfloat16).float())
    got = ops.gemm_tn(dy.to(DEV), x.to(DEV), alpha=0.125).cpu()
    torch.testing.assert_close(got, ref, atol=2e-3 * max(1.0, M ** 0.5 / 8), rtol=2e-3)
    with pytest.raises(_lib.SlsmiError):
        ops.gemm_tn(dy[:, :, :32].contiguous().to(DEV), x.to(DEV))      # N not a multiple of 64
    if K % 256:
        with pytest.raises(_lib.SlsmiError):
            ops.gemm_tn(dy[:, :, :64].contiguous().to(DEV), x.to(DEV))  # 64 columns need K in steps of 256
    else:   # 64 columns of dY per workgroup against 256 of X
        got64 = ops.gemm_tn(dy[:, :, :64].contiguous().to(DEV), x.to(DEV), alpha=0.125).cpu()
        torch.testing.assert_close(got64, ref[:, :64], atol=2e-3 * max(1.0, M ** 0.5 / 8), rtol=2e-3)


def test_weight_gradient_contraction_over_overlapping_rows():
    """slsmi_gemm_tn with rows of x that overlap (ldx < K): the weight gradient of a strided Conv1d taken straight
    from its input -- row m of clip z is frames 2 m .. 2 m + 2 (k = 3, stride 2, 128 channels) -- against fp32 on
    the unfolded rows; a clip's rows past its M valid ones are not read (they hold NaN here)."""
    nz, M, C, k, st, P = 3, 203, 128, 3, 2, 208
    dy = rnd(nz, M, 256, seed=31).to(torch.bfloat16)
    frames = bf(rnd(nz, P * st, C, seed=32))
    frames[:, (M - 1) * st + k:] = float("nan")
    rows = torch.stack([frames[:, m * st: m * st + k].reshape(nz, k * C) for m in range(M)], dim=1)   # [nz, M, k C]
    ref = torch.einsum("zmn,zmk->znk", dy.float(), rows.to(torch.bfloat16).float())   # (the kernel multiplies in bfloat16)
    got = ops.gemm_tn(dy.to(DEV), frames.to(DEV).reshape(-1), K=k * C, ldx=st * C, zx=P * st * C).cpu()
    assert torch.isfinite(got).all()
    torch.testing.assert_close(got, ref, atol=5e-3, rtol=2e-3)


@pytest.mark.parametrize("B,T,xdt", [(5, 201, torch.float16), (2, 99, torch.bfloat16), (1, 30, torch.float16)])
def test_positional_conv_weight_gradient_from_the_slabs(B, T, xdt):
    """slsmi_gemm_tn_grouped as the training step calls it for the encoder's positional conv (16 groups of 64
    channels, 128 taps): dW[g][co][(j, ci)] = sum over clips b and frames t of du[b T + t][64 g + co] *
    xg[b][g][t + j][ci], from the output gradient as it lies and the overlapping-row view of the padded input slabs,
    a group of T rows per clip -- against fp32 on the unfolded rows of the same rounded operands."""
    slab = (T + 128 + 7) // 8 * 8
    du = rnd(B * T, 1024, seed=41).to(torch.bfloat16)
    xg = rnd(B, 16, slab, 64, seed=42).to(xdt)
    rows = xg.to(torch.bfloat16).float().unfold(2, 128, 1)[:, :, :T]          # [B, 16, T, 64 ci, 128 j]
    rows = rows.permute(0, 1, 2, 4, 3).reshape(B, 16, T, 8192)                # column (j, ci)
    ref = torch.einsum("btgc,bgtk->gck", du.float().view(B, T, 16, 64), rows)
    got = ops.gemm_tn_grouped(du.to(DEV), xg.to(DEV).reshape(-1), N=64, K=8192, nz=16, group_rows=T, lda=1024, za=64,
                              ga=T * 1024, ldx=64, zx=slab * 64, gx=16 * slab * 64).cpu()
    torch.testing.assert_close(got, ref, atol=2e-3 * max(1.0, (B * T) ** 0.5 / 8), rtol=2e-3)


def test_gemm_rejects_bad_shapes():
    a, w = bf(rnd(8, 60)), bf(rnd(128, 60))
    with pytest.raises(_lib.SlsmiError):
        ops.gemm_bf16(a.to(DEV), w.to(DEV), None, _lib.GEMM_F32)  # K % 64 != 0


def test_gemm_timing_window_counts_launches_and_leaves_results_alone():
    """slsmi_debug_gemm_timing_begin / _end: launches inside the window are counted with their 2 M N K and a
    positive duration, launches beyond the capacity run untimed, results are the bits of an untimed launch,
    and a second begin (or an end without a begin) is refused."""
    M, N, K = 1300, 256, 512
    a, w, b = bf(rnd(M, K, seed=1)).to(DEV), bf(rnd(N, K, seed=2, scale=K ** -0.5)).to(DEV), rnd(N, seed=3).to(DEV)
    plain = ops.gemm_bf16(a, w, b, _lib.GEMM_F32)
    with ops.gemm_timing(DEV, max_launches=2) as gt:
        with pytest.raises(_lib.SlsmiError):
            _lib.check(_lib.load().slsmi_debug_gemm_timing_begin(4), None, "begin inside a window")
        outs = [ops.gemm_bf16(a, w, b, _lib.GEMM_F32) for _ in range(3)]
    assert (gt.launches, gt.dropped) == (2, 1)
    assert gt.flops == 2 * (2.0 * M * N * K)
    assert 0.0 < gt.seconds < 1.0
    for o in outs:
        assert torch.equal(o, plain)
    import ctypes
    n, d, sec, fl = ctypes.c_int(), ctypes.c_int(), ctypes.c_double(), ctypes.c_double()
    assert _lib.load().slsmi_debug_gemm_timing_end(ctypes.byref(n), ctypes.byref(d), ctypes.byref(sec), ctypes.byref(fl)) != 0
    with ops.gemm_timing(DEV, max_launches=8) as gt2:    # a window can be opened again
        ops.gemm_bf16(a, w, b, _lib.GEMM_F32)
    assert (gt2.launches, gt2.dropped) == (1, 0)


def test_conv0_ln_gelu():
    B, L = 3, 3205
    T = (L - 10) // 5 + 1
    wav = rnd(B, L, seed=9, scale=0.1)
    w, b = rnd(512, 1, 10, seed=10, scale=10 ** -0.5), rnd(512, seed=11, scale=0.05)
    g, be = 1 + rnd(512, seed=12, scale=0.1), rnd(512, seed=13, scale=0.05)
    y = F.conv1d(wav[:, None], w, b, stride=5).transpose(1, 2)
    ref = F.gelu(F.layer_norm(y, (512,), g, be, 1e-5))
    got = ops.conv0_ln_gelu(wav.to(DEV), w.reshape(512, 10).contiguous().to(DEV), b.to(DEV), g.to(DEV), be.to(DEV),
                            1e-5, T, rows_per_utt=T + 5, dtype=DT).float().cpu()
    torch.testing.assert_close(got[:, :T], ref, atol=tol16() / 2, rtol=tol16() / 2)
    assert torch.count_nonzero(got[:, T:]) == 0  # rows past T untouched


@pytest.mark.parametrize("gelu", [False, True])
def test_ln512(gelu):
    x = bf(rnd(1001, 512, seed=14, scale=2.0) + 0.3)
    g, b = 1 + rnd(512, seed=15, scale=0.1), rnd(512, seed=16, scale=0.05)
    ref = F.layer_norm(x.float(), (512,), g, b, 1e-5)
    if gelu:
        ref = F.gelu(ref)
    got = ops.ln512_bf16(x.to(DEV), g.to(DEV), b.to(DEV), 1e-5, gelu).float().cpu()
    torch.testing.assert_close(got, ref, atol=tol16(), rtol=tol16() / 2)


@pytest.mark.parametrize("gelu", [False, True])
def test_ln512_from_fp32_rows(gelu):
    """A conv block's LayerNorm fed unrounded fp32 values (the reference's Fp32LayerNorm, wav2vec2.py:807-812): only the
    result is rounded, so the error is half a unit of the format's last place of the OUTPUT."""
    x = rnd(1001, 512, seed=14, scale=2.0) + 0.3
    g, b = 1 + rnd(512, seed=15, scale=0.1), rnd(512, seed=16, scale=0.05)
    ref = F.layer_norm(x.double(), (512,), g.double(), b.double(), 1e-5)
    if gelu:
        ref = F.gelu(ref)
    got = ops.ln512_f32(x.to(DEV), g.to(DEV), b.to(DEV), 1e-5, gelu, dtype=DT).double().cpu()
    half_step = 2.0 ** -8 if DT == torch.bfloat16 else 2.0 ** -11   # half the spacing at a power of two, relative
    assert ((got - ref).abs() <= half_step * ref.abs() * 1.001 + 1e-5).all()   # correctly rounded, up to fp32 arithmetic noise


def test_ln512_split_rows():
    """[hi | lo | hi] rows: hi is exactly what the plain kernel writes, hi + lo carries the fp32 result to ~2^-19."""
    x = bf(rnd(700, 512, seed=24, scale=2.0) + 0.3)
    g, b = 1 + rnd(512, seed=25, scale=0.1), rnd(512, seed=26, scale=0.05)
    plain = ops.ln512_bf16(x.to(DEV), g.to(DEV), b.to(DEV), 1e-5, False)
    got = ops.ln512_split3(x.to(DEV), g.to(DEV), b.to(DEV), 1e-5)
    assert torch.equal(got[:, :512], plain) and torch.equal(got[:, 1024:], plain)
    ref = F.layer_norm(x.double(), (512,), g.double(), b.double(), 1e-5)
    both = (got[:, :512].double() + got[:, 512:1024].double()).cpu()
    one = plain.double().cpu()
    e2, e1 = (both - ref).abs().max().item(), (one - ref).abs().max().item()
    assert e2 < e1 / 50 and e2 < (6e-5 if DT == torch.bfloat16 else 2e-5), (e1, e2)   # bf16 halves: 16 bits of values up to ~8


@pytest.mark.parametrize("M,N,ka,lda,bn", [(12864, 512, 1024, 1024, 0), (12864, 512, 1536, 1024, 256), (3000, 512, 1536, 1024, 1),
                                           (500, 512, 1024, 1024, 128), (201, 256, 512, 512, 64), (26624, 512, 1536, 1024, 0)])
@pytest.mark.parametrize("mode", [_lib.GEMM_BF16, _lib.GEMM_F32])
def test_gemm_with_periodic_a_multiplies_by_both_weight_halves(M, N, ka, lda, bn, mode):
    """slsmi_gemm_bf16_period: C = A [W_hi | W_lo]^T with A repeating along K -- the product with a weight kept as
    its rounded value plus the remainder of that rounding (how the scoring forward runs conv layers 3-6, with the
    overlapping-row view lda < ka of a strided conv).  Against fp64 A (W_hi + W_lo)^T; against the single-half
    product the weights' rounding is gone (the error left is the fp32 accumulation's); and the same bits whichever
    kernel the tile request selects."""
    g = torch.Generator().manual_seed(7)
    a_flat = bf(torch.randn((M - 1) * lda + ka + 64, generator=g) * 0.5)
    w32 = torch.randn(N, ka, generator=g) * ka ** -0.5
    b = torch.randn(N, generator=g) * 0.1
    hi = bf(w32)
    lo = bf(w32 - hi.float())
    w2 = torch.cat([hi, lo], dim=1).contiguous()
    a_view = torch.as_strided(a_flat, (M, ka), (lda, 1)).double()
    exact = a_view @ w32.double().t() + b.double()
    got = ops.gemm_bf16(a_flat.to(DEV), w2.to(DEV), b.to(DEV), mode, bn=bn, M=M, K=2 * ka, lda=lda, ka=ka)
    one = ops.gemm_bf16(a_flat.to(DEV), hi.to(DEV).contiguous(), b.to(DEV), mode, bn=bn, M=M, K=ka, lda=lda)
    if mode == _lib.GEMM_F32:
        e2 = (got.double().cpu() - exact).abs().max().item()
        e1 = (one.double().cpu() - exact).abs().max().item()
        # bf16 halves leave 2^-17 of the weight, f16 halves 2^-22 (or the subnormal step): >= 30 x closer either way
        assert e2 < e1 / 30 and e2 < 3e-4, (e1, e2)
    else:
        torch.testing.assert_close(got.float().cpu(), exact.float(), atol=tol16(), rtol=tol16() / 2)
    if M < 12864:   # few output tiles: the 128-row kernel cuts K there (another summation order), nothing to compare
        return
    for other in (128, 1, 256, 0):   # every kernel adds the K-tiles in the same order
        if other == bn or (other == 256 and (2 * ka) % 128 != 0):
            continue
        again = ops.gemm_bf16(a_flat.to(DEV), w2.to(DEV), b.to(DEV), mode, bn=other, M=M, K=2 * ka, lda=lda, ka=ka)
        assert torch.equal(got, again), (bn, other)


def test_ln1024():
    x = rnd(777, 1024, seed=17, scale=3.0) - 0.5
    g, b = 1 + rnd(1024, seed=18, scale=0.1), rnd(1024, seed=19, scale=0.05)
    ref = F.layer_norm(x, (1024,), g, b, 1e-5)
    got = ops.ln1024_f32_bf16(x.to(DEV), g.to(DEV), b.to(DEV), 1e-5, dtype=DT).float().cpu()
    torch.testing.assert_close(got, ref, atol=tol16(), rtol=tol16() / 2)


@pytest.mark.parametrize("B,T", [(2, 201), (1, 224), (3, 17), (1, 1), (1, 256), (2, 300), (2, 380), (1, 440), (1, 499),
                                  (1, 512)])
def test_attention(B, T):
    qkv = bf(rnd(B * T, 3072, seed=20))
    q, k, v = [t.float().view(B, T, 16, 64).transpose(1, 2) for t in qkv.split(1024, dim=1)]
    # q arrives pre-scaled by 1/8 in the real pipeline; here the raw q is the scaled one
    ref = (torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v).transpose(1, 2).reshape(B * T, 1024)
    got = ops.attention(qkv.to(DEV), B, T).float().cpu()
    torch.testing.assert_close(got, ref, atol=tol16(), rtol=tol16())


@pytest.mark.parametrize("T,frames", [(201, [201, 124, 59]), (64, [1, 33, 64]), (499, [499, 250, 17]),
                                      (440, [440, 300, 20]), (380, [380, 2, 379]),
                                      (300, [257, 300, 16])])
def test_attention_key_padding(T, frames):
    """Utterance b attends to its first frames[b] keys (key-padding mask, wav2vec2.py:1050): its own
    rows must equal attention over those frames alone."""
    B = len(frames)
    qkv = bf(rnd(B * T, 3072, seed=22))
    nf = torch.tensor(frames, dtype=torch.int32)
    got = ops.attention(qkv.to(DEV), B, T, nf.to(DEV)).float().cpu().view(B, T, 1024)
    assert torch.isfinite(got).all()
    for b, f in enumerate(frames):
        own = qkv.view(B, T, 3072)[b, :f]
        q, k, v = [t.float().view(f, 16, 64).transpose(0, 1) for t in own.split(1024, dim=1)]
        ref = (torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v).transpose(0, 1).reshape(f, 1024)
        torch.testing.assert_close(got[b, :f], ref, atol=tol16(), rtol=tol16())


def test_attention_peaked_softmax():
    """One key dominates each query: the max-subtraction path must hold."""
    B, T = 1, 201
    qkv = rnd(B * T, 3072, seed=21)
    qkv[:, :1024] *= 6.0
    qkv = bf(qkv)
    q, k, v = [t.float().view(B, T, 16, 64).transpose(1, 2) for t in qkv.split(1024, dim=1)]
    ref = (torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v).transpose(1, 2).reshape(B * T, 1024)
    got = ops.attention(qkv.to(DEV), B, T).float().cpu()
    assert torch.isfinite(got).all()
    torch.testing.assert_close(got, ref, atol=1.5 * tol16(), rtol=1.5 * tol16())


@pytest.mark.parametrize("B,T,NL", [(3, 201, 4), (2, 50, 1), (1, 499, 24)])
def test_sls_weights_sum_pool(B, T, NL):
    """Layer weights from row dots, then the weighted layer sum + BatchNorm affine + SELU + 3 x 3 max pool in one
    pass, against the reference's order of operations (getAttenF: time mean -> fc0 -> sigmoid -> sum over layers)."""
    xs = torch.stack([rnd(B * T, 1024, seed=30 + l) for l in range(NL)])  # [NL][B T][1024]
    w0, b0 = rnd(1024, seed=40, scale=1 / 32), rnd(1, seed=41)
    # row dots in 16 slices of 64 columns, as the residual GEMM leaves them
    dots = (xs.view(NL, B * T, 16, 64) * w0.view(16, 64)).sum(-1).permute(0, 2, 1).contiguous()  # [NL][16][B T]
    xb = xs.view(NL, B, T, 1024)
    wgt_ref = torch.sigmoid(xb.mean(2) @ w0 + b0)  # [NL][B]
    wgt = ops.sls_weights(dots.to(DEV), b0.to(DEV), B, T)
    torch.testing.assert_close(wgt.cpu(), wgt_ref, atol=2e-5, rtol=2e-5)
    ref = (wgt_ref[:, :, None, None] * xb).sum(0)  # [B][T][1024]
    scale, shift = 0.83, -0.11
    pooled = ops.sls_sum_pool(xs.to(DEV), wgt, B, T, scale, shift, dtype=DT).float().cpu()
    pref = F.max_pool2d(F.selu(ref * scale + shift)[:, None], (3, 3)).flatten(1)
    n = (T // 3) * 341
    assert pref.shape[1] == n and pooled.shape[1] == (n + 63) // 64 * 64
    torch.testing.assert_close(pooled[:, :n], pref, atol=tol16(), rtol=tol16() / 2)
    assert torch.count_nonzero(pooled[:, n:]) == 0


@pytest.mark.parametrize("B,T,bn,K", [(3, 201, 0, 256), (8, 201, 0, 1024), (70, 201, 0, 768), (2, 50, 64, 256),
                                      (64, 201, 256, 512), (9, 50, 256, 256), (70, 201, 1, 768)])
@pytest.mark.parametrize("in_place", [True, False])
def test_gemm_resid_rowdot(B, T, bn, K, in_place):
    """Residual GEMM with row dots, in place or from a separate residual buffer.  B = 8 and B = 70 reach the
    256 x 128 kernel (B = 70: more output tiles than CUs, ragged last tile; bn = 1 asks for it), bn = 256 the
    208 x 256 kernel, the others the 128-row one."""
    M = B * T
    a, w = bf(rnd(M, K, seed=60)), bf(rnd(1024, K, seed=61, scale=K ** -0.5))
    bias, x0 = rnd(1024, seed=62, scale=0.1), rnd(M, 1024, seed=63)
    w0 = rnd(1024, seed=64, scale=1 / 32)
    x_ref = x0 + a.float() @ w.float().t() + bias
    if in_place:
        c, resid = x0.to(DEV).clone(), None
    else:
        c, resid = torch.full((M, 1024), float("nan"), device=DEV), x0.to(DEV)
    dots = ops.gemm_resid_rowdot(a.to(DEV), w.to(DEV), bias.to(DEV), c, w0.to(DEV), bn=bn, resid=resid)
    torch.testing.assert_close(c.cpu(), x_ref, atol=2e-3, rtol=2e-3)
    if resid is not None:
        assert torch.equal(resid.cpu(), x0)  # the source is left alone
    assert dots.shape == (16, M)
    torch.testing.assert_close(dots.sum(0).cpu(), c.cpu() @ w0, atol=2e-4, rtol=2e-4)
    wgt = ops.sls_weights(dots[None].contiguous(), rnd(1, seed=65).to(DEV), B, T)
    wgt_ref = torch.sigmoid(x_ref.view(B, T, 1024).mean(1) @ w0 + rnd(1, seed=65))
    torch.testing.assert_close(wgt.cpu()[0], wgt_ref, atol=2e-4, rtol=2e-4)


@pytest.mark.parametrize("dict_,k", [(4096, 128), (1024, 16), (4096, 1), (2048, 2048)])
def test_topk_rows(dict_, k):
    rows = 37
    x = F.relu(rnd(rows, dict_, seed=50) - 0.5)
    x[3] = 0.0                      # all-zero row
    x[4, 10:] = 0.0                 # fewer than k positives when k > 10
    vals, idx = x.topk(k, dim=-1, sorted=False)
    ref = torch.zeros_like(x).scatter_(-1, idx, vals)
    enc = x.to(DEV).clone()
    eb = ops.topk_rows(enc, k, want_bf16=True, dtype=DT)
    assert torch.equal(enc.cpu(), ref)
    assert torch.equal(eb.float().cpu(), ref.to(DT).float())


def _window_topk_expected(acts, k, w):
    """The reference's result where the votes determine it, and this library's rule elsewhere: the
    positive-vote features, then zero-vote features from the lowest feature number up to k in all."""
    from oracle import reference_cpu as orc
    votes = orc.window_votes(acts, k, w)
    B, T, D = acts.shape
    mask = torch.zeros_like(acts, dtype=torch.bool)
    for b in range(B):
        for t in range(T):
            v = votes[b, t]
            pos = int((v > 0).sum())
            if pos >= k:
                mask[b, t, v.topk(k).indices] = True
            else:
                mask[b, t] = v > 0
                mask[b, t, (v <= 0).nonzero()[:k - pos, 0]] = True
    return acts * mask, votes


@pytest.mark.parametrize("B,T,D,k,w,shift,seed", [(2, 37, 256, 16, 8, 0.0, 11), (1, 201, 512, 32, 8, 0.5, 12),
                                                   (2, 40, 1024, 8, 6, 0.0, 13), (3, 9, 256, 8, 8, 0.0, 14),
                                                   (1, 50, 256, 8, 2, 0.3, 15), (2, 201, 4096, 128, 8, 0.0, 16),
                                                   (1, 33, 4096, 128, 3, 1.0, 17)])
def test_window_topk(B, T, D, k, w, shift, seed, golden_dir):
    import os
    import numpy as np
    from oracle import reference_cpu as orc
    g = torch.Generator().manual_seed(seed)
    acts = torch.relu(torch.randn(B, T, D, generator=g) - shift)
    want, votes = _window_topk_expected(acts, k, w)
    got, got16 = ops.window_topk(acts.to(DEV), k, w, want_16bit=True, dtype=DT)
    got = got.cpu()
    assert torch.equal(got, want)
    torch.testing.assert_close(got16.float().cpu(), want, atol=tol16(), rtol=tol16())
    # where the reference's own result is determined (at least k positive votes), it is the same
    ref = orc.window_topk(acts, k, w)
    full = (votes > 0).sum(-1) >= k
    assert torch.equal(got[full], ref[full])
    z = np.load(os.path.join(golden_dir, "window_topk_cases.npz"))
    for name in ("t37_w8", "t201_w8"):   # the two committed reference outputs with dict >= 256
        if [B, T, D, k, w, seed] == [int(v) for v in z[name + "_meta"]]:
            fix = torch.from_numpy(z[name])
            assert torch.equal(got[full], fix[full])
            assert torch.equal((got * (votes > 0)), (fix * (votes > 0)))


@pytest.mark.parametrize("M,K1,N2,gelu", [(603, 1024, 3072, False), (1000, 4096, 4096, True), (12864, 1024, 3072, False),
                                          (12864, 4096, 4096, True), (77, 256, 1024, True)])
def test_layernorm_folded_into_gemm_pair(M, K1, N2, gelu):
    """LayerNorm(1024) folded into the residual GEMM in front of it and the GEMM behind it (slsmi_gemm_bf16_ln):
    the producer's C equals the plain residual GEMM bit for bit; its 16-bit operand and slice sums follow the stated
    formulas; the consumer's result is Linear(LayerNorm(x)) to the rounding of the 16-bit operand -- no worse than
    the separate LayerNorm kernel followed by the plain GEMM -- and the statistics it leaves are x's own.  The
    rows' scales differ by orders of magnitude and the (m, s) handed to the producer are deliberately off."""
    eps = 1e-5
    dev = DEV
    row_scale = torch.logspace(-1, 2, M).unsqueeze(1)                      # 0.1 .. 100
    x_prev = (rnd(M, 1024, seed=11) * row_scale + rnd(M, 1, seed=12) * row_scale).to(dev)
    a = bf(rnd(M, K1, seed=13)).to(dev)
    w1 = bf(rnd(1024, K1, seed=14, scale=K1 ** -0.5)).to(dev)
    b1 = rnd(1024, seed=15, scale=0.1).to(dev)
    gamma, beta = (1.0 + 0.3 * rnd(1024, seed=16)).to(dev), (0.2 * rnd(1024, seed=17)).to(dev)
    w2 = bf(rnd(N2, 1024, seed=18, scale=1 / 32)).to(dev)
    b2 = rnd(N2, seed=19, scale=0.1).to(dev)
    fc0 = rnd(1024, seed=20, scale=0.05).to(dev)
    # (m, s): the statistics of the residual input (what the forward hands over), pushed off a little
    m = x_prev.mean(1) * 1.1 + 0.01
    s = 0.8 / torch.sqrt(x_prev.var(1, unbiased=False) + eps)
    proxy = torch.stack([m, s], 1).contiguous()

    plain = ops.gemm_bf16(a, w1, b1, _lib.GEMM_F32_RESID, out=x_prev.clone(), bn=256)
    x, ln_a, slots, dots = ops.gemm_ln_emit(a, w1, b1, x_prev, proxy, gamma, dot_w=fc0)
    assert torch.equal(x, plain)
    u = (x - m[:, None]) * s[:, None]
    torch.testing.assert_close(ln_a.float(), (u * gamma).to(DT).float(), atol=1e-5, rtol=2 * tol16())
    want_slots = torch.stack([u.view(M, 16, 64).sum(2), (u * u).view(M, 16, 64).sum(2)], 2).transpose(0, 1)   # [16][M][2]
    torch.testing.assert_close(slots, want_slots, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(dots.sum(0), x @ fc0, rtol=1e-4, atol=1e-3)

    g, c = ops.ln_fold_consts(w2, b2, gamma, beta)
    torch.testing.assert_close(g, w2.float() @ gamma, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(c, b2 + w2.float() @ beta, rtol=1e-5, atol=1e-5)
    got, stat = ops.gemm_ln_apply(ln_a, w2, c, g, proxy, slots, eps, gelu=gelu)
    mean, rstd = x.mean(1), 1 / torch.sqrt(x.var(1, unbiased=False) + eps)
    torch.testing.assert_close(stat[:, 0], mean, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(stat[:, 1], rstd, rtol=1e-4, atol=0)

    def act(t):
        return F.gelu(t) if gelu else t
    ref = act(F.layer_norm(x.double(), (1024,), gamma.double(), beta.double(), eps) @ w2.double().t() + b2.double())
    h = ops.ln1024_f32_bf16(x, gamma, beta, eps, DT)
    unfused = ops.gemm_bf16(h, w2, b2, _lib.GEMM_BF16_GELU if gelu else _lib.GEMM_BF16, bn=256)
    err_fold = (got.double() - ref).abs().max().item()
    err_sep = (unfused.double() - ref).abs().max().item()
    print(f"[M={M} N={N2} {DT}] folded max err {err_fold:.3e}, separate kernels {err_sep:.3e}")
    assert err_fold < 1.5 * err_sep + 1e-3
    torch.testing.assert_close(got.float(), ref.float(), atol=tol16(), rtol=tol16())

    # the row-wise form on the other kernels (128-row two-stage: bn 128; 256 x 128 where the shape has the tiles for
    # it: bn 1; the library's pick: bn 0): the same bits -- operand, statistics and result -- because every form
    # adds the same terms in the same order (a score must not depend on which form its batch size selects)
    ln_a_r, aff, stat_r = ops.ln_fold_rows(x, proxy, gamma, eps, DT)
    assert torch.equal(ln_a_r, ln_a)
    assert torch.equal(stat_r, stat)
    for bn in (128, 1, 0):
        got_r = ops.gemm_ln_apply_rows(ln_a_r, w2, c, g, aff, gelu=gelu, bn=bn)
        assert torch.equal(got_r, got), bn

    # the chain's entry kernel stands in for a producer whose (m, s) are the row's own statistics
    ln_a0, proxy0, slots0 = ops.ln_fold_prep(x, gamma, eps, DT)
    torch.testing.assert_close(proxy0, torch.stack([mean, rstd], 1), rtol=1e-4, atol=1e-4)
    got0, _ = ops.gemm_ln_apply(ln_a0, w2, c, g, proxy0, slots0, eps, gelu=gelu)
    assert (got0.double() - ref).abs().max().item() < 1.5 * err_sep + 1e-3
