// Diagnostic: the 16x16x32 MFMA rate this device sustains from registers alone (no LDS, no memory): every SIMD of
// every CU issues back-to-back MFMAs on sixteen independent accumulators for a few milliseconds.  Under that load the
// chip settles well below its nominal clock, so the figure (not the data-sheet peak) is what bounds any GEMM
// kernel's K loop on the box at hand.  bench.py reports it beside the roofline fraction.
#include "common.h"
#include "kernels.h"

namespace slsmi {

namespace {

template <int FMT>
__global__ __launch_bounds__(256) void mfma_rate_kernel(float* __restrict__ sink, int iters) {
    u16x8 a[4], b[4];
    // operands with varied bits (a constant operand toggles fewer wires and clocks higher): small normal values
    const unsigned seed = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const unsigned ra = seed * (2 * i + 3) + e * 0x9E3779B9u, rb = seed * (2 * i + 5) + e * 0x85EBCA6Bu;
            a[i][e] = Op<FMT>::pack(((int)(ra >> 20) - 2048) * (1.0f / 2048.0f));
            b[i][e] = Op<FMT>::pack(((int)(rb >> 20) - 2048) * (1.0f / 2048.0f));
        }
    f32x4 acc[4][4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = Op<FMT>::mfma(a[i], b[j], acc[i][j]);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    sink[(long)blockIdx.x * 256 + threadIdx.x] = s;
}

// Range of one 16-bit store: out[0] = largest |value| (as the bits of a non-negative float, so an unsigned maximum
// orders them), out[1] = number of values at or beyond the format's largest finite value (saturated, or inf / NaN).
template <int FMT>
__global__ __launch_bounds__(256) void range_stat_kernel(const bf16_t* __restrict__ x, long rows, int cols, long ld,
                                                         int grp_in, int grp_valid, unsigned* __restrict__ out) {
    const float top = FMT == FMT_F16 ? 65504.0f : 3.3895314e38f;
    const int c8 = cols >> 3;  // cols % 8 == 0 (launcher)
    const long n = rows * c8;
    unsigned mx = 0, sat = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long r = i / c8;
        const int c = (int)(i - r * c8) << 3;
        if (grp_in > 0 && r % grp_in >= grp_valid) continue;   // a row the producers never write
        const u16x8 v = *reinterpret_cast<const u16x8*>(x + r * ld + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float a = fabsf(Op<FMT>::unpack(v[e]));
            const bool bad = !(a < top);   // saturated, inf or NaN
            sat += bad ? 1u : 0u;
            mx = max(mx, __float_as_uint(bad ? top : a));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mx = max(mx, (unsigned)__shfl_xor((int)mx, o, 64));
        sat += (unsigned)__shfl_xor((int)sat, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (mx) atomicMax(out, mx);
        if (sat) atomicAdd(out + 1, sat);
    }
}

}  // namespace

hipError_t launch_range_stat(const bf16_t* x, long rows, int cols, long ld, int fmt, unsigned* out2, hipStream_t s,
                             int grp_in, int grp_valid) {
    if (!x || !out2 || rows < 1 || cols < 8 || (cols & 7) || ld < cols || (ld & 7)) return hipErrorInvalidValue;
    const long n = rows * (cols >> 3);
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    SLSMI_FMT_DISPATCH(fmt, hipLaunchKernelGGL(range_stat_kernel<F>, dim3(grid), dim3(256), 0, s, x, rows, cols, ld, grp_in, grp_valid, out2));
    return hipGetLastError();
}

// sink: blocks x 256 floats.  Returns the launch error; flops of one launch = blocks x 4 waves x iters x 16 MFMAs x 16384.
hipError_t launch_mfma_rate(float* sink, int blocks, int iters, int fmt, hipStream_t s) {
    if (!sink || blocks < 1 || iters < 1) return hipErrorInvalidValue;
    SLSMI_FMT_DISPATCH(fmt, hipLaunchKernelGGL(mfma_rate_kernel<F>, dim3(blocks), dim3(256), 0, s, sink, iters));
    return hipGetLastError();
}

}  // namespace slsmi
