// Diagnostic: register-only MFMA loop -> achievable f16 MFMA rate and in-kernel clock on this device.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(256) void k(const _Float16* in, float* out, unsigned long long* clk, int iters) {
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = *(const f16x8*)(in + ((threadIdx.x * 8 + i) * 8) % 4096);
        b[i] = *(const f16x8*)(in + ((threadIdx.x * 8 + i + 4) * 8) % 4096);
    }
    f32x4 acc[4][4] = {};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}
__global__ __launch_bounds__(256) void k32(const _Float16* in, float* out, unsigned long long* clk, int iters) {
    f16x8 a[2], b[4];
    for (int i = 0; i < 2; ++i) a[i] = *(const f16x8*)(in + ((threadIdx.x * 8 + i) * 8) % 4096);
    for (int i = 0; i < 4; ++i) b[i] = *(const f16x8*)(in + ((threadIdx.x * 8 + i + 4) * 8) % 4096);
    f32x16 acc[2][4] = {};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}
int main(int argc, char** argv) {
    int wavesPerSimd = argc > 1 ? atoi(argv[1]) : 1;
    int use32 = argc > 2 ? atoi(argv[2]) : 0;
    int iters = 20000;
    int blocks = 256 * wavesPerSimd;  // 256-thread blocks: 1 wave per SIMD each
    _Float16* in; float* out; unsigned long long* clk;
    hipMalloc(&in, 4096 * 2 + 64); hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
    _Float16 h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0); hipLaunchKernelGGL(use32 ? k32 : k, dim3(blocks), dim3(256), 0, 0, in, out, clk, iters); hipEventRecord(e1);
        hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
        double fl = (double)blocks * 4 * iters * 16 * 16384.0;
        printf("%s waves/SIMD=%d: %.3f ms  %.0f TF  clock %.0f MHz (cycles %llu)\n", use32 ? "32x32x16" : "16x16x32", wavesPerSimd, ms, fl / ms / 1e9, 100.0 * c[0] / c[1], c[0]);
    }
    return 0;
}
