// 16-bit-operand (bf16 or f16) MFMA GEMM for gfx950:  C[m][n] (op)= sum_k A[m][k] * W[n][k] (+ bias[n])
//
// Block tile 128 x BN x 64 (BN = 128 or 64), 256 threads = 4 waves.  Both
// operand tiles are staged HBM/L2 -> LDS with 16-byte direct-to-LDS loads
// (global_load_lds_dwordx4): the LDS image is lane-linear ([row][64] bf16, 128 B
// rows), so the bank-conflict swizzle (16-byte chunk index XOR (row & 7)) is
// applied to the per-lane SOURCE address and again on the ds_read_b128 side.
// Two LDS stages: the loads of K-tile t+1 are issued before the MFMAs of tile t.
// MFMA: v_mfma_f32_16x16x32_{bf16,f16} with the weight fragment as the first operand,
// so a lane's four accumulator registers are four consecutive n of one row m:
// epilogue accesses are 8-byte (bf16) or 16-byte (fp32) per lane.
#include "common.h"
#include "kernels.h"

namespace slsmi {

namespace {

constexpr int BM = 128;
constexpr int BK = 64;
constexpr int ROW_BYTES = BK * 2;  // 128

struct GemmParams {
    const bf16_t* A;
    const bf16_t* W;
    const float* bias;
    void* C;
    long lda, ldw, ldc;
    int M, N, K;
    int mode;
    int nz;
    long zA;
    int zWmod;
    long zW, zBias;
    int zCdiv;
    long zC1, zC2;
    int splitk;
    int ktiles_per_split;
    int grp_in, grp_valid, grp_out;
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int BN, int FMT>
__global__ __launch_bounds__(256, 2) void gemm16_kernel(GemmParams p) {
    constexpr int A_BYTES = BM * ROW_BYTES;   // 16 KiB
    constexpr int B_BYTES = BN * ROW_BYTES;   // 16 or 8 KiB
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int WAVES_N = BN / 64;          // 2 or 1
    constexpr int WAVES_M = 4 / WAVES_N;      // 2 or 4
    constexpr int WM = BM / WAVES_M;          // 64 or 32
    constexpr int MI = WM / 16;               // 4 or 2
    constexpr int NJ = 4;
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N;
    const int wn = wave % WAVES_N;

    const int m0 = blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;
    int z = blockIdx.z;
    int kt_begin = 0;
    int kt_end = p.K / BK;
    const bf16_t* A = p.A;
    const bf16_t* W = p.W;
    const float* bias = p.bias;
    long c_off = 0;
    if (p.splitk > 1) {
        kt_begin = z * p.ktiles_per_split;
        kt_end = min(kt_end, kt_begin + p.ktiles_per_split);
        if (kt_begin >= kt_end) return;
    } else if (p.nz > 1) {
        A += (long)z * p.zA;
        const int zw = z % p.zWmod;
        W += (long)zw * p.zW;
        if (bias) bias += (long)zw * p.zBias;
        c_off = (long)(z / p.zCdiv) * p.zC1 + (long)(z % p.zCdiv) * p.zC2;
    }

    // ---- staging addresses: one wave-instruction moves 8 rows x 128 B
    const int srow = lane >> 3;                 // row within the 8-row piece (= row & 7)
    const int schunk = (lane & 7) ^ srow;       // swizzled source chunk
    const bf16_t* a_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int r = m0 + (wave * 4 + i) * 8 + srow;
        r = min(r, p.M - 1);
        a_src[i] = A + (long)r * p.lda + schunk * 8;
    }
    constexpr int B_PIECES = BN / 8 / 4;        // per wave: 4 or 2
    const bf16_t* b_src[B_PIECES];
#pragma unroll
    for (int i = 0; i < B_PIECES; ++i) {
        const int r = n0 + (wave * B_PIECES + i) * 8 + srow;
        b_src[i] = W + (long)r * p.ldw + schunk * 8;
    }

    auto stage = [&](int buf, int kt) {
        char* base = lds + buf * STAGE;
        const long koff = (long)kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + koff),
                                             (lptr_t)(base + (wave * 4 + i) * 8 * ROW_BYTES), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_PIECES; ++i) {
            __builtin_amdgcn_global_load_lds((gptr_t)(b_src[i] + koff),
                                             (lptr_t)(base + A_BYTES + (wave * B_PIECES + i) * 8 * ROW_BYTES), 16, 0, 0);
        }
    };

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- fragment read addresses
    const int fr = lane & 15;
    const int fg = lane >> 4;
    const int fx = lane & 7;
    const int a_row_off = (wm * WM + fr) * ROW_BYTES;
    const int b_row_off = A_BYTES + (wn * 64 + fr) * ROW_BYTES;
    const int c0 = ((0 + fg) ^ fx) << 4;
    const int c1 = ((4 + fg) ^ fx) << 4;

    stage(0, kt_begin);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int cur = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        if (kt + 1 < kt_end) stage(cur ^ 1, kt + 1);
        const char* base = lds + cur * STAGE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int co = s ? c1 : c0;
            u16x8 af[MI], bfr[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i)
                af[i] = *reinterpret_cast<const u16x8*>(base + a_row_off + i * 16 * ROW_BYTES + co);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                bfr[j] = *reinterpret_cast<const u16x8*>(base + b_row_off + j * 16 * ROW_BYTES + co);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = Op<FMT>::mfma(bfr[j], af[i], acc[i][j]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: lane holds C[m][n .. n+3], m = .. + (lane & 15), n = .. + 4*(lane >> 4)
    const int mode = p.mode;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * WM + i * 16 + fr;
        if (m >= p.M) continue;
        long orow = m;
        if (p.grp_in > 0) {
            const int g = m / p.grp_in;
            const int pos = m - g * p.grp_in;
            if (pos >= p.grp_valid) continue;
            orow = (long)g * p.grp_out + pos;
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + wn * 64 + j * 16 + fg * 4;
            f32x4 v = acc[i][j];
            if (bias && mode != GEMM_F32_ATOMIC) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(bias + n);
                v += b;
            }
            const long idx = c_off + orow * p.ldc + n;
            if (mode == GEMM_BF16 || mode == GEMM_BF16_GELU) {
                if (mode == GEMM_BF16_GELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
                }
                u16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = Op<FMT>::pack(v[r]);
                *reinterpret_cast<u16x4*>(reinterpret_cast<bf16_t*>(p.C) + idx) = o;
            } else if (mode == GEMM_F32_ATOMIC) {
                float* c = reinterpret_cast<float*>(p.C) + idx;
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(c + r, v[r]);
            } else {
                float* c = reinterpret_cast<float*>(p.C) + idx;
                if (mode == GEMM_F32_RESID_GELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
                }
                if (mode == GEMM_F32_RELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
                if (mode == GEMM_F32_RESID || mode == GEMM_F32_RESID_GELU) {
                    const f32x4 old = *reinterpret_cast<const f32x4*>(c);
                    v += old;
                }
                *reinterpret_cast<f32x4*>(c) = v;
            }
        }
    }
}

}  // namespace

hipError_t launch_gemm(const GemmArgs& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return hipErrorInvalidValue;
    if (a.bn != 128 && a.bn != 64) return hipErrorInvalidValue;
    if (a.fmt != FMT_BF16 && a.fmt != FMT_F16) return hipErrorInvalidValue;
    if (a.N % a.bn != 0 || a.K % BK != 0) return hipErrorInvalidValue;
    if ((a.lda % 8) || (a.ldw % 8) || (a.ldc % 4)) return hipErrorInvalidValue;
    if ((reinterpret_cast<uintptr_t>(a.A) & 15) || (reinterpret_cast<uintptr_t>(a.W) & 15) ||
        (reinterpret_cast<uintptr_t>(a.C) & 15))
        return hipErrorInvalidValue;
    if (a.bias && (reinterpret_cast<uintptr_t>(a.bias) & 15)) return hipErrorInvalidValue;
    if (a.splitk > 1 && (a.nz != 1 || a.mode != GEMM_F32_ATOMIC)) return hipErrorInvalidValue;
    if (a.mode == GEMM_F32_ATOMIC && a.splitk < 1) return hipErrorInvalidValue;
    if (a.nz < 1 || a.zWmod < 1 || a.zCdiv < 1) return hipErrorInvalidValue;
    if ((a.zA % 8) || (a.zW % 8) || (a.zBias % 4) || (a.zC1 % 4) || (a.zC2 % 4)) return hipErrorInvalidValue;
    GemmParams p;
    p.A = a.A; p.W = a.W; p.bias = a.bias; p.C = a.C;
    p.lda = a.lda; p.ldw = a.ldw; p.ldc = a.ldc;
    p.M = a.M; p.N = a.N; p.K = a.K; p.mode = a.mode;
    p.nz = a.nz; p.zA = a.zA; p.zWmod = a.zWmod; p.zW = a.zW; p.zBias = a.zBias;
    p.zCdiv = a.zCdiv; p.zC1 = a.zC1; p.zC2 = a.zC2;
    p.splitk = a.splitk;
    const int ktiles = a.K / BK;
    p.ktiles_per_split = (ktiles + a.splitk - 1) / a.splitk;
    p.grp_in = a.grp_in; p.grp_valid = a.grp_valid; p.grp_out = a.grp_out;
    const int gz = a.splitk > 1 ? a.splitk : a.nz;
    dim3 grid(a.N / a.bn, (a.M + BM - 1) / BM, gz);
    if (grid.y > 65535 || grid.z > 65535) return hipErrorInvalidValue;
    if (a.bn == 128 && a.fmt == FMT_BF16)
        hipLaunchKernelGGL((gemm16_kernel<128, FMT_BF16>), grid, dim3(256), 0, s, p);
    else if (a.bn == 128)
        hipLaunchKernelGGL((gemm16_kernel<128, FMT_F16>), grid, dim3(256), 0, s, p);
    else if (a.fmt == FMT_BF16)
        hipLaunchKernelGGL((gemm16_kernel<64, FMT_BF16>), grid, dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL((gemm16_kernel<64, FMT_F16>), grid, dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace slsmi
