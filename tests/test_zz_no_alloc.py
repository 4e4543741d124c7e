"""slsmi_gemm_bf16_static (the GEMM entry point that allocates nothing: K-split partial sums in a
file-scope device array) against slsmi_gemm_bf16 (scratch allocated by the library), through the C ABI.
Bit-exact: both make the same K split and add the slices in the same order."""
import pytest
import torch

from slsmi import _lib, ops

pytestmark = pytest.mark.gpu


def _case(M, N, K, mode, seed, dtype=torch.float16, bn=0):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(seed)
    a = (torch.randn(M, K, generator=g) * 0.5).to(dtype).to(dev)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dtype).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    odt = dtype if mode in (_lib.GEMM_BF16, _lib.GEMM_BF16_GELU) else torch.float32
    base = torch.randn(M, N, generator=g).to(odt).to(dev)   # old C of the residual mode
    ref_out, new_out = base.clone(), base.clone()
    # the entry point under test runs FIRST, and device memory is compared around it: it must not allocate
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info(dev)[0]
    ops.gemm_bf16(a, w, b, mode, out=new_out, static_scratch=True, bn=bn)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info(dev)[0] == free_before, "slsmi_gemm_bf16_static changed the device's free memory"
    ops.gemm_bf16(a, w, b, mode, out=ref_out, bn=bn)
    torch.cuda.synchronize()
    fp = (a.float() @ w.float().t() + b).cpu()
    return ref_out.cpu(), new_out.cpu(), fp, base.cpu()


def test_static_scratch_gemm_two_calls_second_smaller():
    # one clip's rows (201) and then fewer: both are cut along K (16 and 8 output tiles on 256 CUs), so
    # the second call runs on an array that still holds the first call's partial sums
    for (M, N, K, mode, seed) in ((201, 1024, 4096, _lib.GEMM_F32_RESID, 1), (77, 1024, 1024, _lib.GEMM_F32_RESID, 2),
                                  (201, 3072, 1024, _lib.GEMM_BF16, 3), (50, 4096, 1024, _lib.GEMM_BF16_GELU, 4)):
        ref, new, fp, base = _case(M, N, K, mode, seed)
        assert torch.equal(ref, new), (M, N, K, mode, (ref.float() - new.float()).abs().max())
        if mode == _lib.GEMM_F32_RESID:   # and both are the GEMM: fp32 product of the 16-bit operands
            assert (new - (fp + base)).abs().max() < 2e-2


def test_static_scratch_gemm_same_split_for_every_tile_request():
    # bn = 1 (the 256 x 128 kernel asked for) on a shape with too few rows for it falls back to 128-row tiles cut
    # along K in BOTH entry points: same split, same bits
    ref, new, fp, base = _case(201, 1024, 2048, _lib.GEMM_F32_RESID, 7, bn=1)
    assert torch.equal(ref, new)
    assert (new - (fp + base)).abs().max() < 2e-2


# (The refusal path -- a K split that needs more than the 16 MiB array returns SLSMI_ERR_WORKSPACE -- cannot be
# reached on a 256-CU device: the library cuts K into S slices only while S x tiles <= CUs, and a tile is at most
# 128 x 128 outputs, so S x M x N x 4 bytes <= 256 x 16384 x 4 = 16 MiB.  It guards devices with more CUs.)


def test_static_scratch_gemm_without_split_has_no_limit():
    # a GEMM that is not cut along K uses no scratch, whatever its size
    dev = torch.device("cuda:0")
    lib = _lib.load()
    M = 4096
    a = torch.ones(M, 1024, dtype=torch.float16, device=dev)
    w = torch.ones(4096, 1024, dtype=torch.float16, device=dev)
    out = torch.zeros(M, 4096, dtype=torch.float32, device=dev)
    rc = lib.slsmi_gemm_bf16_static(a.data_ptr(), 1024, w.data_ptr(), 1024, None, out.data_ptr(), 4096, M, 4096, 1024,
                                    _lib.GEMM_F32, 0, 1, _lib.FMT_F16, None)
    assert rc == 0
    torch.cuda.synchronize()
    assert bool((out == 1024.0).all())
