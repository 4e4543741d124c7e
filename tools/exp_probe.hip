// Issue cost of v_exp_f32 on gfx950, alone and beside independent packed-fp32 work, and of a polynomial exp2.
// Build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/exp_probe tools/exp_probe.hip ; run on the GPU box.
// Each kernel runs ITER iterations of a fixed instruction mix on WPS waves per SIMD of every CU; the figure printed is
// nanoseconds per iteration per SIMD (time / ITER / WPS ... i.e. issue cost of the mix when WPS waves share the SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096;

#define EXP8(x)                                                                                                     \
    asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"                       \
                 "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"                          \
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]))
#define PK8(y, c)                                                                                                   \
    asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n"         \
                 "v_pk_fma_f32 %3, %3, %8, %8\n v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n"         \
                 "v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8"                                          \
                 : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7])     \
                 : "v"(c))
#define FMA8(z, c)                                                                                                  \
    asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n"                  \
                 "v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n"                  \
                 "v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"                                                \
                 : "+v"(z[0]), "+v"(z[1]), "+v"(z[2]), "+v"(z[3]), "+v"(z[4]), "+v"(z[5]), "+v"(z[6]), "+v"(z[7])     \
                 : "v"(c))
// interleaved: one exp, then two packed fmas, eight times
#define MIX8(x, y, w, c)                                                                                            \
    asm volatile("v_exp_f32 %0, %0\n v_pk_fma_f32 %8, %8, %24, %24\n v_pk_fma_f32 %16, %16, %24, %24\n"              \
                 "v_exp_f32 %1, %1\n v_pk_fma_f32 %9, %9, %24, %24\n v_pk_fma_f32 %17, %17, %24, %24\n"              \
                 "v_exp_f32 %2, %2\n v_pk_fma_f32 %10, %10, %24, %24\n v_pk_fma_f32 %18, %18, %24, %24\n"            \
                 "v_exp_f32 %3, %3\n v_pk_fma_f32 %11, %11, %24, %24\n v_pk_fma_f32 %19, %19, %24, %24\n"            \
                 "v_exp_f32 %4, %4\n v_pk_fma_f32 %12, %12, %24, %24\n v_pk_fma_f32 %20, %20, %24, %24\n"            \
                 "v_exp_f32 %5, %5\n v_pk_fma_f32 %13, %13, %24, %24\n v_pk_fma_f32 %21, %21, %24, %24\n"            \
                 "v_exp_f32 %6, %6\n v_pk_fma_f32 %14, %14, %24, %24\n v_pk_fma_f32 %22, %22, %24, %24\n"            \
                 "v_exp_f32 %7, %7\n v_pk_fma_f32 %15, %15, %24, %24\n v_pk_fma_f32 %23, %23, %24, %24"              \
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),    \
                   "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]),    \
                   "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7])     \
                 : "v"(c))

// 2^t for t <= 0 without the transcendental unit: round to nearest integer through the 1.5 * 2^23 trick, degree-4
// polynomial on [-0.5, 0.5], exponent added as an integer.  Two values per call (packed operations where they exist).
__device__ inline f32x2 exp2_poly(f32x2 t) {
    const f32x2 magic = {12582912.f, 12582912.f};
    t = __builtin_elementwise_max(t, f32x2{-126.f, -126.f});
    const f32x2 r = t + magic;
    const f32x2 n = r - magic;
    const f32x2 f = t - n;
    f32x2 p = f32x2{0.0096181291f, 0.0096181291f};
    p = p * f + f32x2{0.0555041087f, 0.0555041087f};
    p = p * f + f32x2{0.2402265070f, 0.2402265070f};
    p = p * f + f32x2{0.6931471806f, 0.6931471806f};
    p = p * f + f32x2{1.f, 1.f};
    f32x2 o;
    o[0] = __builtin_bit_cast(float, __builtin_bit_cast(int, p[0]) + (__builtin_bit_cast(int, r[0]) << 23));
    o[1] = __builtin_bit_cast(float, __builtin_bit_cast(int, p[1]) + (__builtin_bit_cast(int, r[1]) << 23));
    return o;
}

template <int KIND>
__global__ __launch_bounds__(256) void probe(float* out, float seed) {
    float x[8], z[8];
    f32x2 y[8], w[8];
    const f32x2 c = {0.999f, 0.999f};
    for (int i = 0; i < 8; ++i) {
        x[i] = seed + threadIdx.x * 1e-3f + i;
        z[i] = x[i];
        y[i] = f32x2{x[i], x[i] + 0.5f};
        w[i] = y[i];
    }
    for (int it = 0; it < ITER; ++it) {
        if (KIND == 0) { EXP8(x); }                       // 8 v_exp_f32
        if (KIND == 1) { PK8(y, c); PK8(w, c); }          // 16 v_pk_fma_f32
        if (KIND == 2) { EXP8(x); PK8(y, c); PK8(w, c); } // 8 exp then 16 packed fmas
        if (KIND == 3) { MIX8(x, y, w, c); }              // the same, interleaved 1 : 2
        if (KIND == 4) { FMA8(z, c[0]); }                 // 8 v_fma_f32
        if (KIND == 5) {                                  // polynomial exp2 of 8 values (compiler's schedule)
            for (int i = 0; i < 4; ++i) {
                f32x2 t = {x[2 * i], x[2 * i + 1]};
                t = exp2_poly(t) - f32x2{1.5f, 1.5f};
                x[2 * i] = t[0];
                x[2 * i + 1] = t[1];
            }
        }
        if (KIND == 6) {                                  // half by polynomial, half by v_exp_f32
            for (int i = 0; i < 2; ++i) {
                f32x2 t = {x[2 * i], x[2 * i + 1]};
                t = exp2_poly(t) - f32x2{1.5f, 1.5f};
                x[2 * i] = t[0];
                x[2 * i + 1] = t[1];
            }
            for (int i = 4; i < 8; ++i) x[i] = __builtin_amdgcn_exp2f(x[i]) - 1.5f;
        }
        if (KIND == 7) {                                  // all eight by v_exp_f32, same surrounding arithmetic as 5
            for (int i = 0; i < 8; ++i) x[i] = __builtin_amdgcn_exp2f(x[i]) - 1.5f;
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += x[i] + z[i] + y[i][0] + y[i][1] + w[i][0] + w[i][1];
    if (s == 12345.678f) out[0] = s;
}

template <int KIND>
static void run(const char* name, int n_cu, float* out) {
    for (int wps = 1; wps <= 2; ++wps) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        dim3 grid(n_cu * wps), block(256);
        hipLaunchKernelGGL(probe<KIND>, grid, block, 0, 0, out, 0.25f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<KIND>, grid, block, 0, 0, out, 0.25f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s waves/SIMD %d: %7.2f ns per iteration per wave, %7.2f ns per iteration per SIMD\n", name, wps,
               ms * 1e6 / ITER, ms * 1e6 / ITER / wps);
    }
}

int main() {
    int n_cu = 0;
    hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0);
    float* out;
    hipMalloc(&out, 256);
    // accuracy of the polynomial first (host copy of the same arithmetic is not needed: compare on the device)
    run<0>("8 v_exp_f32", n_cu, out);
    run<4>("8 v_fma_f32", n_cu, out);
    run<1>("16 v_pk_fma_f32", n_cu, out);
    run<2>("8 v_exp_f32 then 16 v_pk_fma_f32", n_cu, out);
    run<3>("8 x (v_exp_f32, 2 v_pk_fma_f32)", n_cu, out);
    run<7>("8 values: exp2 by v_exp_f32, minus const", n_cu, out);
    run<5>("8 values: exp2 by polynomial, minus const", n_cu, out);
    run<6>("4 by polynomial + 4 by v_exp_f32", n_cu, out);
    hipFree(out);
    return 0;
}
