// Shared device helpers for the gfx950 kernels of libslsmi.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

namespace slsmi {

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

#define SLSMI_WAVE 64

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
    return __uint_as_float(((uint32_t)v) << 16);
}

// round-to-nearest-even f32 -> bf16; a plain cast lowers to v_cvt_pk_bf16_f32
// on gfx950 and keeps NaN a NaN.
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}

// 16-bit operand format of the MFMA GEMMs and of every stored activation:
// bfloat16 (8-bit mantissa) or IEEE half (11-bit mantissa).  Both run the
// 16x16x32 MFMA at the same rate and share every fragment layout, so kernels
// are templated on the format and move raw 16-bit words.
enum { FMT_BF16 = 0, FMT_F16 = 1 };

template <int FMT>
struct Op;

template <>
struct Op<FMT_BF16> {
    static __device__ __forceinline__ bf16_t pack(float f) { return f32_to_bf16(f); }
    static __device__ __forceinline__ float unpack(bf16_t v) { return bf16_to_f32(v); }
    // two values -> one 32-bit word (first value in the low half)
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
    }
    static __device__ __forceinline__ uint32_t pack2_unit(float lo, float hi) { return pack2(lo, hi); }
    static __device__ __forceinline__ f32x4 mfma(u16x8 a, u16x8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};

template <>
struct Op<FMT_F16> {
    // saturate instead of producing inf: half tops out at 65504
    static __device__ __forceinline__ bf16_t pack(float f) {
        const _Float16 h = (_Float16)fminf(fmaxf(f, -65504.0f), 65504.0f);
        return __builtin_bit_cast(bf16_t, h);
    }
    static __device__ __forceinline__ float unpack(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
        const f32x2 v = {__builtin_amdgcn_fmed3f(lo, -65504.0f, 65504.0f), __builtin_amdgcn_fmed3f(hi, -65504.0f, 65504.0f)};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
    }
    // values known to lie inside the format's range: no saturation needed
    static __device__ __forceinline__ uint32_t pack2_unit(float lo, float hi) {
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, f16x2));
    }
    static __device__ __forceinline__ f32x4 mfma(u16x8 a, u16x8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

// Raises a kernel's dynamic-LDS limit once per device (the attribute is per device; a process
// may drive several).  `state` is one zero-initialised array per kernel instantiation.
struct LdsAttrState {
    std::atomic<int> done[64];
};
inline hipError_t ensure_dynamic_lds(const void* kernel, int bytes, LdsAttrState& state) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (state.done[dev].load(std::memory_order_acquire)) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) state.done[dev].store(1, std::memory_order_release);
    return e;
}

// Runs `...` once with a constexpr int F bound to the runtime format value.
#define SLSMI_FMT_DISPATCH(fmt, ...)        \
    do {                                    \
        if ((fmt) == ::slsmi::FMT_F16) {    \
            constexpr int F = ::slsmi::FMT_F16;  \
            __VA_ARGS__;                    \
        } else {                            \
            constexpr int F = ::slsmi::FMT_BF16; \
            __VA_ARGS__;                    \
        }                                   \
    } while (0)

// Single fp32 operations that the compiler may not fuse with their neighbours.  The build runs with
// -ffp-contract=fast, under which the backend turns a * b + c into an fma wherever it likes -- differently in
// different kernels -- and no pragma stops it; a product that passes through an empty asm statement is opaque to it.
__device__ __forceinline__ float fmul_nc(float a, float b) {
    float r = a * b;
    asm("" : "+v"(r));
    return r;
}
__device__ __forceinline__ float fadd_nc(float a, float b) { return a + b; }   // (operands are never visible products)
__device__ __forceinline__ float fsub_nc(float a, float b) { return a - b; }
__device__ __forceinline__ float fma_nc(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// Folded LayerNorm (GemmArgs::ln_mode): the pieces every form of it computes with the same operations in the same
// order, so that the forms agree bit for bit.
__device__ __forceinline__ float ln_fold_u(float x, float m, float s) { return fmul_nc(fsub_nc(x, m), s); }
// running sums of u and u^2 over a lane's 16 values of a 64-column slice, in column-block order
__device__ __forceinline__ void ln_fold_acc(float u, float& s1, float& s2) {
    s1 = fadd_nc(s1, u);
    s2 = fma_nc(u, u, s2);
}
// from a row's totals S1 = sum u, S2 = sum u^2 (1024 values) and the (m, s) that defined u:
// alpha = 1 / sqrt(var u + eps s^2), nubar = -mean u, and x's own mean and 1 / sqrt(var + eps)
__device__ __forceinline__ void ln_fold_factors(float S1, float S2, float m, float s, float eps, float& alpha, float& nubar,
                                                float& mean, float& rstd) {
    const float ubar = fmul_nc(S1, 1.0f / 1024.0f);
    const float var = fmaxf(fma_nc(-ubar, ubar, fmul_nc(S2, 1.0f / 1024.0f)), 0.f);
    const float den = sqrtf(fma_nc(fmul_nc(eps, s), s, var));
    alpha = 1.0f / den;
    nubar = -ubar;
    float shift = ubar / s;
    asm("" : "+v"(shift));
    mean = fadd_nc(m, shift);
    rstd = fmul_nc(alpha, s);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// erf-GELU as torch.nn.GELU() / F.gelu default:  gelu(x) = max(x, 0) - |x| Phi(-|x|), with
// Phi(-a) = 2^P(a) for a degree-6 polynomial P fitted on [0, 8] (beyond 8 the term is < 1e-14).
// |error| <= 6e-8 in exact arithmetic, 3e-7 evaluated in fp32 over [-12, 12]; one v_exp_f32 is
// the only transcendental, and the polynomial runs as packed fp32 FMAs on value pairs
// (the GEMM and conv epilogues evaluate this 64 times per lane).
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
    const f32x2 a = {fminf(fabsf(x[0]), 8.0f), fminf(fabsf(x[1]), 8.0f)};
    f32x2 p = __builtin_elementwise_fma(f32x2{3.309304304e-05f, 3.309304304e-05f}, a,
                                        f32x2{-7.692224994e-04f, -7.692224994e-04f});
    p = __builtin_elementwise_fma(p, a, f32x2{8.080728349e-03f, 8.080728349e-03f});
    p = __builtin_elementwise_fma(p, a, f32x2{-5.341212655e-02f, -5.341212655e-02f});
    p = __builtin_elementwise_fma(p, a, f32x2{-4.587709536e-01f, -4.587709536e-01f});
    p = __builtin_elementwise_fma(p, a, f32x2{-1.151201710e+00f, -1.151201710e+00f});
    p = __builtin_elementwise_fma(p, a, f32x2{-9.999930601e-01f, -9.999930601e-01f});
    const f32x2 q = {__builtin_amdgcn_exp2f(p[0]), __builtin_amdgcn_exp2f(p[1])};
    const f32x2 r = {fmaxf(x[0], 0.0f), fmaxf(x[1], 0.0f)};
    return __builtin_elementwise_fma(-a, q, r);
}

__device__ __forceinline__ float gelu_erf(float x) {
    const float a = fminf(fabsf(x), 8.0f);
    float p = fmaf(3.309304304e-05f, a, -7.692224994e-04f);
    p = fmaf(p, a, 8.080728349e-03f);
    p = fmaf(p, a, -5.341212655e-02f);
    p = fmaf(p, a, -4.587709536e-01f);
    p = fmaf(p, a, -1.151201710e+00f);
    p = fmaf(p, a, -9.999930601e-01f);
    return fmaf(-a, __builtin_amdgcn_exp2f(p), fmaxf(x, 0.0f));
}

__device__ __forceinline__ f32x4 gelu_erf4(f32x4 v) {
    const f32x2 lo = gelu_erf2(f32x2{v[0], v[1]}), hi = gelu_erf2(f32x2{v[2], v[3]});
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}

// The same for results that are stored in the 16-bit operand format FMT: the positive part is limited to the
// format's largest finite value here (half: 65504; the result can only be smaller), so the store needs no
// saturation of its own (Op<FMT>::pack2_unit) -- same bits as gelu_erf4 followed by the saturating pack2.
template <int FMT>
__device__ __forceinline__ f32x2 gelu_erf2_out16(f32x2 x) {
    const f32x2 a = {fminf(fabsf(x[0]), 8.0f), fminf(fabsf(x[1]), 8.0f)};
    f32x2 p = __builtin_elementwise_fma(f32x2{3.309304304e-05f, 3.309304304e-05f}, a,
                                        f32x2{-7.692224994e-04f, -7.692224994e-04f});
    p = __builtin_elementwise_fma(p, a, f32x2{8.080728349e-03f, 8.080728349e-03f});
    p = __builtin_elementwise_fma(p, a, f32x2{-5.341212655e-02f, -5.341212655e-02f});
    p = __builtin_elementwise_fma(p, a, f32x2{-4.587709536e-01f, -4.587709536e-01f});
    p = __builtin_elementwise_fma(p, a, f32x2{-1.151201710e+00f, -1.151201710e+00f});
    p = __builtin_elementwise_fma(p, a, f32x2{-9.999930601e-01f, -9.999930601e-01f});
    const f32x2 q = {__builtin_amdgcn_exp2f(p[0]), __builtin_amdgcn_exp2f(p[1])};
    const f32x2 r = FMT == FMT_F16 ? f32x2{__builtin_amdgcn_fmed3f(x[0], 0.0f, 65504.0f), __builtin_amdgcn_fmed3f(x[1], 0.0f, 65504.0f)}
                                   : f32x2{fmaxf(x[0], 0.0f), fmaxf(x[1], 0.0f)};
    return __builtin_elementwise_fma(-a, q, r);
}

template <int FMT>
__device__ __forceinline__ f32x4 gelu_erf4_out16(f32x4 v) {
    const f32x2 lo = gelu_erf2_out16<FMT>(f32x2{v[0], v[1]}), hi = gelu_erf2_out16<FMT>(f32x2{v[2], v[3]});
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}

__device__ __forceinline__ float selu(float x) {
    const float alpha = 1.6732632423543772848170429916717f;
    const float scale = 1.0507009873554804934193349852946f;
    return scale * (x > 0.0f ? x : alpha * expm1f(x));
}

}  // namespace slsmi
