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

}  // namespace

// sink: blocks x 256 floats.  Returns the launch error; flops of one launch = blocks x 4 waves x iters x 16 MFMAs x 16384.
hipError_t launch_mfma_rate(float* sink, int blocks, int iters, int fmt, hipStream_t s) {
    if (!sink || blocks < 1 || iters < 1) return hipErrorInvalidValue;
    SLSMI_FMT_DISPATCH(fmt, hipLaunchKernelGGL(mfma_rate_kernel<F>, dim3(blocks), dim3(256), 0, s, sink, iters));
    return hipGetLastError();
}

}  // namespace slsmi
