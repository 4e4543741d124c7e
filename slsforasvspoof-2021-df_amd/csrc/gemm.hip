// 16-bit-operand (bf16 or f16) MFMA GEMM for gfx950:
//     C[m][n] (op)= sum_k A[m][k] * W[n][k] (+ bias[n])
//
// Two kernels share the fragment layout and the staging scheme; both start their
// accumulators from the bias and add the K-tiles in the same order, so a result does
// not depend on which of them a shape selects:
//
//  * gemm16_big_kernel: 256 x 128 x 64 block tile, 8 waves (2 per SIMD), three LDS
//    stages (144 KiB, one persistent workgroup per CU), loads two K-tiles ahead behind
//    a counted `s_waitcnt vmcnt`, fragments double-buffered in registers, and the two
//    waves of every SIMD one segment apart (one multiplies while the other fetches).
//    Used for every large GEMM of the forward.
//  * gemm16_kernel<BN>: 128 x BN x 64 (BN = 128 or 64), 4 waves, two stages filled through registers;
//    used for small M, the grouped positional conv (batched over blockIdx.z) and
//    split-K.
//
// Staging: 16-byte `global_load_lds_dwordx4`; the LDS image is lane-linear
// ([row][64] 16-bit, 128-B rows), so the bank-conflict swizzle (16-byte chunk
// index XOR (row & 7)) is applied to the per-lane SOURCE address and again on the
// ds_read_b128 side.  MFMA: v_mfma_f32_16x16x32_{bf16,f16} with the weight
// fragment as the first operand, so a lane's four accumulator registers are four
// consecutive n of one row m: epilogue accesses are 8 B (16-bit out; 16 B after a lane-pair
// exchange in the 256 x 128 kernel) or 16 B (fp32).
#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>

#include "common.h"
#include "kernels.h"
#include "gemm_params.h"

namespace slsmi {

namespace {


// Row dots of one wave's accumulator block with a column vector: out[n_base / 64][m] =
// sum over the block's 64 columns of v[m][n] * w[n].  Lane (fr, fg) holds 4 columns of row
// 16 i + fr per sub-tile; the four fg lanes of a row are summed with two shuffles.
template <int MI>
__device__ __forceinline__ void store_row_dots(const GemmParams& p, const f32x4 (&v)[MI][4], int m_base, int n_base,
                                               int lane) {
    const int fr = lane & 15, fg = lane >> 4;
    f32x4 w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const f32x4*>(p.dot_w + n_base + j * 16 + fg * 4);
    float* dst = p.dot_out + (long)(n_base >> 6) * p.M;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) d = fmaf(v[i][j][r], w[j][r], d);
        d += __shfl_xor(d, 16, 64);
        d += __shfl_xor(d, 32, 64);
        const int m = m_base + i * 16 + fr;
        if (fg == 0 && m < p.M) dst[m] = d;
    }
}

// One wave's 64 x 64 (MI = 4) or 32 x 64 (MI = 2) accumulator block (bias included) -> memory.
// Lane holds C[m][n .. n+3] with m = m_base + 16 i + (lane & 15), n = n_base + 16 j + 4 (lane >> 4).
template <int FMT, int MI>
__device__ __forceinline__ void store_block(const GemmParams& p, const f32x4 (&acc)[MI][4], long c_off, int m_base,
                                            int n_base, int lane) {
    const int fr = lane & 15, fg = lane >> 4;
    const int mode = p.mode;
    if (p.dot_w) {  // GEMM_F32_RESID only (launcher): the dots are taken of the updated C
        f32x4 v[MI][4];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = min(m_base + i * 16 + fr, p.M - 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long idx = c_off + (long)m * p.ldc + n_base + j * 16 + fg * 4;
                v[i][j] = acc[i][j] + *reinterpret_cast<const f32x4*>(p.R + idx);
                if (m_base + i * 16 + fr < p.M) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + idx) = v[i][j];
            }
        }
        store_row_dots<MI>(p, v, m_base, n_base, lane);
        return;
    }
    const bool ln_rows = p.ln_aff != nullptr;  // folded LayerNorm with the rows' factors given
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m_base + i * 16 + fr;
        if (m >= p.M) continue;
        f32x2 af = {1.f, 0.f};
        if (ln_rows) af = *reinterpret_cast<const f32x2*>(p.ln_aff + (long)m * 2);
        long orow = m;
        if (p.grp_in > 0) {
            const int g = m / p.grp_in;
            const int pos = m - g * p.grp_in;
            if (pos >= p.grp_valid) continue;
            orow = (long)g * p.grp_out + pos;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n_base + j * 16 + fg * 4;
            f32x4 v = acc[i][j];
            if (ln_rows) {  // the accumulators started from -ubar g (see the kernel): alpha acc + c
                const f32x4 cv = *reinterpret_cast<const f32x4*>(p.ln_c + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fma_nc(v[r], af[0], cv[r]);
            }
            const long idx = c_off + orow * p.ldc + n;
            if (mode == GEMM_BF16 || mode == GEMM_BF16_GELU) {
                if (mode == GEMM_BF16_GELU) {
                    if (p.pre_out) {
                        u16x4 q;
#pragma unroll
                        for (int r = 0; r < 4; ++r) q[r] = Op<FMT>::pack(v[r]);
                        *reinterpret_cast<u16x4*>(p.pre_out + idx) = q;
                    }
                    v = gelu_erf4(v);
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
                    v = gelu_erf4(v);
                }
                if (mode == GEMM_F32_RELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
                if (mode == GEMM_F32_RESID || mode == GEMM_F32_RESID_GELU) v += *reinterpret_cast<const f32x4*>(p.R + idx);
                *reinterpret_cast<f32x4*>(c) = v;
            }
        }
    }
}

// One K-tile (64 deep) of a wave's MI x 4 MFMA block from the staged tiles.
template <int FMT, int MI>
__device__ __forceinline__ void mma_ktile(f32x4 (&acc)[MI][4], const char* a_rows, const char* b_rows, int c0,
                                          int c1) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int co = s ? c1 : c0;
        u16x8 af[MI], bfr[4];
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const u16x8*>(a_rows + i * 16 * ROW_BYTES + co);
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const u16x8*>(b_rows + j * 16 * ROW_BYTES + co);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = Op<FMT>::mfma(bfr[j], af[i], acc[i][j]);
    }
}

// ------------------------------------------------------------------ 256 x 128 x 64, three stages, staggered waves
// Three LDS stages of one 64-deep K-tile each (A 256 rows + W 128 rows, 128-byte
// rows = whole cache lines per LDS-DMA row, 48 KiB per stage, 144 KiB in all, one
// workgroup per CU).  The workgroup is persistent (tiles b, b + grid, ...) and the
// K-tiles of all its output tiles form ONE stream through the ring of stages: the
// loads always run two K-tiles ahead, across output-tile boundaries too.
// A wave alternates
//   R(t): read the 16 fragments of K-tile t into one of two register sets; refill
//         the stage of K-tile t-1 with K-tile t+2 (W pieces + first A pieces); counted
//         `s_waitcnt vmcnt` = this wave's loads of K-tile t+1 have landed; barrier
//   M(t-1): rest of K-tile t+2's loads; the 32 MFMAs of K-tile t-1; barrier
// Waves 4-7 share their SIMDs with waves 0-3 and run the same program ONE SEGMENT
// LATER, so on every SIMD one wave multiplies while its partner fetches: the matrix
// pipe does not idle during LDS reads, waits and barriers.
//   Readiness: K-tile j is first read in R(j) of waves 0-3; every wave waited for its
//   pieces of K-tile j in its own R(j-1), which ends, at the latest (waves 4-7), with
//   the barrier right before that R(j).
//   Re-use: the stage of K-tile t-1 is refilled from R(t) on; every wave's fragment reads
//   of K-tile t-1 were waited for (lgkmcnt(0)) before the barrier that ended its R(t-1),
//   which for waves 4-7 is the barrier right before R(t) of waves 0-3.  This needs one
//   segment between R(t-1) and R(t) of a wave: true in the stream, arranged by one
//   load-only segment after the very first R.
// Output tile boundary (nt K-tiles per output tile): the operand pointers switch to the
// next output tile before R(nt-2), whose loads are K-tile 0 of that tile; after M(nt-1)
// the wave writes its 64 x 64 accumulator block straight from registers (no LDS, no
// barrier: lanes 16 apart exchange halves with v_permlane16_swap so that every lane
// stores 16 contiguous bytes and a wave-store covers 16 rows x 64 B), and re-initialises
// the accumulators with the next tile's bias, which an LDS-DMA load parked in LDS at
// the pointer switch.
constexpr int BIG_BM = 256;
constexpr int BIG_BN = 128;
constexpr int BIG_A_BYTES = BIG_BM * ROW_BYTES;  // 32 KiB
constexpr int BIG_B_BYTES = BIG_BN * ROW_BYTES;  // 16 KiB
constexpr int BIG_STAGE = BIG_A_BYTES + BIG_B_BYTES;
constexpr int BIG_STAGES = 3;
constexpr int BIG_BIAS_OFF = BIG_STAGES * BIG_STAGE;  // two 1 KiB bias images behind the stages
constexpr int BIG_LDS = BIG_BIAS_OFF + 2 * 1024;      // 149504

#ifdef SLSMI_STAMPS  // diagnostic build only: cycle stamps of workgroup 0, wave 0
__device__ unsigned long long g_stamps[256];
#define STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && stamp_n < 250) g_stamps[stamp_n++] = ((unsigned long long)(i) << 56) | (__builtin_amdgcn_s_memtime() & 0xFFFFFFFFFFFFFFull); } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

// A K-tile's six loads per wave are issued in two parts: the W pieces and the first
// BIG_NR A pieces in the R segment, the remaining A pieces in the following M segment
// (issuing an LDS-DMA load costs the issuing wave ~100 cycles, so the split balances
// the two segment kinds).
constexpr int BIG_NR = 2;

// LNROWS: the epilogue applies a folded LayerNorm whose per-row factors are given (GemmParams::ln_aff)
template <int FMT, int MODE, bool LNROWS = false>
__global__ __launch_bounds__(512, 2) void gemm16_big_kernel(GemmParams p) {
#ifdef SLSMI_STAMPS
    int stamp_n = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7
    const int wm = wave >> 1, wn = wave & 1;
    const int late = wave >> 2;  // 1: waves 4-7, one segment behind

    const int nt = p.K / BK;  // even, >= 4 (launcher)
    const int tiles_n = p.N / BIG_BN, tiles_m = (p.M + BIG_BM - 1) / BIG_BM;
    const int total_tiles = tiles_n * tiles_m;
    int tile = blockIdx.x;  // < total_tiles (launcher)
    int tile_m, tile_n;
    xcd_tile_map(tile, tiles_n, tiles_m, tile_m, tile_n);
    int m0 = tile_m * BIG_BM, n0 = tile_n * BIG_BN;

    // staging: one wave-instruction moves 8 rows x 128 B; 32 A pieces + 16 W pieces per K-tile
    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;
    const bf16_t* a_src[4];
    const bf16_t* b_src[2];
    auto set_sources = [&](int sm0, int sn0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = min(sm0 + (wave * 4 + i) * 8 + srow, p.M - 1);
            a_src[i] = p.A + (long)r * p.lda + schunk * 8;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = sn0 + (wave * 2 + i) * 8 + srow;
            b_src[i] = p.W + (long)r * p.ldw + schunk * 8;
        }
    };
    // every wave parks the 128 bias values of an output tile in LDS (same bytes from every
    // wave; lanes 32-63 repeat lanes 0-31 so the load stays inside the bias array)
    const bool has_bias = p.bias != nullptr;
    auto load_bias = [&](int sn0, int which) {
        if (has_bias)
            __builtin_amdgcn_global_load_lds((gptr_t)(p.bias + sn0 + (lane & 31) * 4),
                                             (lptr_t)(lds + BIG_BIAS_OFF + which * 1024), 16, 0, 0);
    };
    constexpr int NR = BIG_NR;
    auto a_koff = [&](long koff) {  // A repeats along K with period p.ka (= K: never wraps)
        while (koff >= p.ka) koff -= p.ka;
        return koff;
    };
    auto stage_r = [&](int buf, int kt) {
        char* base = lds + buf * BIG_STAGE;
        const long koff = (long)kt * BK;
        const long koff_a = a_koff(koff);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(b_src[i] + koff),
                                             (lptr_t)(base + BIG_A_BYTES + (wave * 2 + i) * 8 * ROW_BYTES), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < NR; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + koff_a),
                                             (lptr_t)(base + (wave * 4 + i) * 8 * ROW_BYTES), 16, 0, 0);
    };
    auto stage_m = [&](int buf, int kt) {
        char* base = lds + buf * BIG_STAGE;
        const long koff_a = a_koff((long)kt * BK);
#pragma unroll
        for (int i = NR; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + koff_a),
                                             (lptr_t)(base + (wave * 4 + i) * 8 * ROW_BYTES), 16, 0, 0);
    };

    f32x4 acc[4][4];

    const int fr = lane & 15, fg = lane >> 4, fx = lane & 7;
    const int a_row_off = (wm * 64 + fr) * ROW_BYTES;
    const int b_row_off = BIG_A_BYTES + (wn * 64 + fr) * ROW_BYTES;
    const int c0 = ((0 + fg) ^ fx) << 4;
    const int c1 = ((4 + fg) ^ fx) << 4;

    // fragment sets: [k-step][sub-tile] for A and W
    typedef u16x8 frag_t[2][4];
    auto read_frags = [&](int buf, frag_t& fa, frag_t& fb) {
        const char* base = lds + buf * BIG_STAGE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int co = s ? c1 : c0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                fa[s][i] = *reinterpret_cast<const u16x8*>(base + a_row_off + i * 16 * ROW_BYTES + co);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                fb[s][j] = *reinterpret_cast<const u16x8*>(base + b_row_off + j * 16 * ROW_BYTES + co);
        }
    };
    auto mma = [&](const frag_t& fa, const frag_t& fb) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = Op<FMT>::mfma(fb[s][j], fa[s][i], acc[i][j]);
    };
    // raw barrier: LDS-DMA loads stay in flight across it (a __syncthreads() would drain them)
    auto seg_barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    // ring positions: `bt` holds the K-tile the next R reads, `bprev` the one before it
    int bt = 0, bprev = BIG_STAGES - 1;
    auto advance = [&]() {
        const int nb = bt + 1 == BIG_STAGES ? 0 : bt + 1;
        bprev = bt;
        bt = nb;
    };
    // R: fragments of the K-tile in `bt`; first part of the loads of source K-tile `kt_load`
    // into `bprev`; the wait leaves only the NR + 2 loads just issued pending.
    auto seg_R = [&](frag_t& fa, frag_t& fb, int kt_load) {
        read_frags(bt, fa, fb);
        stage_r(bprev, kt_load);
        if (NR == 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        if (NR == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        if (NR == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if (NR == 3) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        if (NR == 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        seg_barrier();
    };
    // M: rest of the loads the preceding R began, then the 32 MFMAs of the other register set
    auto seg_M = [&](const frag_t& fa, const frag_t& fb, int kt_load) {
        __builtin_amdgcn_s_setprio(1);
        stage_m(bprev, kt_load);
        mma(fa, fb);
        __builtin_amdgcn_s_setprio(0);
        seg_barrier();
        advance();
    };
    auto init_acc = [&](int which) {
        float nm[4] = {1.f, 1.f, 1.f, 1.f};
        if (LNROWS) {  // the bias image holds g: start from -ubar[row] g[n], as the 208 x 256 kernel does
#pragma unroll
            for (int i = 0; i < 4; ++i) nm[i] = p.ln_aff[(long)min(m0 + wm * 64 + i * 16 + fr, p.M - 1) * 2 + 1];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
            if (has_bias)
                bv = *reinterpret_cast<const f32x4*>(lds + BIG_BIAS_OFF + which * 1024 + (wn * 64 + j * 16 + fg * 4) * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (LNROWS) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] = fmul_nc(bv[r], nm[i]);
                } else {
                    acc[i][j] = bv;
                }
            }
        }
    };

    // ---- epilogue of one wave: acc (bias included) -> activation -> C, from registers
    constexpr int mode = MODE;  // a template parameter: each epilogue is compiled alone, with its own register needs
    auto epilogue = [&](int em0, int en0) {
        const bool gelu = mode == GEMM_BF16_GELU || mode == GEMM_F32_RESID_GELU;
        if (LNROWS) {  // alpha[row] acc + c[n]
            float al[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) al[i] = p.ln_aff[(long)min(em0 + wm * 64 + i * 16 + fr, p.M - 1) * 2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 cv = *reinterpret_cast<const f32x4*>(p.ln_c + en0 + wn * 64 + j * 16 + fg * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] = fma_nc(acc[i][j][r], al[i], cv[r]);
            }
        }
        if (gelu) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = gelu_erf4(acc[i][j]);
        }
        if (mode == GEMM_F32_RELU) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] = fmaxf(acc[i][j][r], 0.f);
        }
        // output row of accumulator row block i (negative: not stored).  Without row compaction the
        // four rows are 16 apart; with it (one launch per forward) each row is mapped on its own.
        long orow[4];
        if (p.grp_in > 0) {  // workgroup-uniform
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = em0 + wm * 64 + i * 16 + fr;
                const int g = m / p.grp_in;
                const int pos = m - g * p.grp_in;
                orow[i] = (m < p.M && pos < p.grp_valid) ? (long)g * p.grp_out + pos : -1;
            }
        } else {
            const int m = em0 + wm * 64 + fr;
#pragma unroll
            for (int i = 0; i < 4; ++i) orow[i] = m + 16 * i < p.M ? (long)(m + 16 * i) : -1;
        }
        if (mode == GEMM_BF16 || mode == GEMM_BF16_GELU) {
            // lane (fr, fg) holds columns 16 j + 4 fg .. + 3 of row fr: 8 bytes.  Lanes fg and fg ^ 1
            // swap so that even fg gets columns 8 (fg >> 1) .. + 7 of sub-tile 2 jp and odd fg the
            // same columns of sub-tile 2 jp + 1: 16 bytes per lane, 64 contiguous bytes per row.
            bf16_t* cbase = reinterpret_cast<bf16_t*>(p.C) + en0 + wn * 64 + (fg & 1) * 16 + (fg >> 1) * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    const f32x4 v0 = acc[i][2 * jp], v1 = acc[i][2 * jp + 1];
                    const uint32_t e_lo = Op<FMT>::pack2(v0[0], v0[1]), e_hi = Op<FMT>::pack2(v0[2], v0[3]);
                    const uint32_t o_lo = Op<FMT>::pack2(v1[0], v1[1]), o_hi = Op<FMT>::pack2(v1[2], v1[3]);
                    const auto lo = __builtin_amdgcn_permlane16_swap(e_lo, o_lo, false, false);
                    const auto hi = __builtin_amdgcn_permlane16_swap(e_hi, o_hi, false, false);
                    u32x4 out = u32x4{lo[0], hi[0], lo[1], hi[1]};
                    if (orow[i] >= 0) *reinterpret_cast<u32x4*>(cbase + orow[i] * p.ldc + jp * 32) = out;
                }
            }
        } else {
            float* cbase = reinterpret_cast<float*>(p.C) + en0 + wn * 64 + fg * 4;
            const bool resid = mode == GEMM_F32_RESID || mode == GEMM_F32_RESID_GELU;
            if (resid) {
                const float* rbase = p.R + en0 + wn * 64 + fg * 4;
                f32x4 old[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const long lrow = orow[i] >= 0 ? orow[i] : 0;  // rows that are not stored read row 0
                        old[i][j] = *reinterpret_cast<const f32x4*>(rbase + lrow * p.ldc + j * 16);
                    }
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), visible to the compiler's own wait bookkeeping
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[i][j] += old[i][j];
                        if (orow[i] >= 0) *reinterpret_cast<f32x4*>(cbase + orow[i] * p.ldc + j * 16) = acc[i][j];
                    }
                if (p.dot_w) store_row_dots<4>(p, acc, em0 + wm * 64, en0 + wn * 64, lane);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (orow[i] >= 0) *reinterpret_cast<f32x4*>(cbase + orow[i] * p.ldc + j * 16) = acc[i][j];
            }
        }
    };

    // ---- start of the stream: K-tiles 0 and 1 of the first output tile, its bias
    set_sources(m0, n0);
    int bias_buf = 0;
    load_bias(n0, bias_buf);
    stage_r(0, 0);
    stage_m(0, 0);
    stage_r(1, 1);
    stage_m(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    seg_barrier();
    if (late) seg_barrier();
    init_acc(bias_buf);
    frag_t a0, b0, a1, b1;
    seg_R(a0, b0, 2);        // R(0); bt = 0, bprev = 2
    stage_m(bprev, 2);       // load-only segment (see Re-use above)
    seg_barrier();
    advance();               // bt = 1, bprev = 0

    // One output tile per iteration, entered after R(0) of it.  The epilogue of the previous tile is
    // deferred into this iteration and placed so that BOTH wave groups run theirs between the same two
    // barriers (waves 4-7 before their R(1), waves 0-3 after theirs): the two epilogues overlap each
    // other instead of each stalling the other group for a whole segment.
    // (The residual modes keep their epilogue right after the last M segment: its 64 registers of
    // old C values do not fit beside two live fragment sets.)
    // (Nor does the folded-LayerNorm epilogue: its vectors do not fit beside two live fragment sets either.)
    constexpr bool defer = !(MODE == GEMM_F32_RESID || MODE == GEMM_F32_RESID_GELU) && !LNROWS;
    bool pending = false;  // the previous tile's accumulators are still to be stored
    int prev_m0 = 0, prev_n0 = 0;
    for (;;) {
        STAMP(1);
        if (pending && (!defer || late)) {
            epilogue(prev_m0, prev_n0);
            init_acc(bias_buf);
        }
        seg_R(a1, b1, 3);  // R(1)
        if (defer && pending && !late) {
            epilogue(prev_m0, prev_n0);
            init_acc(bias_buf);
        }
        seg_M(a0, b0, 3);  // M(0)
        for (int t = 2; t + 3 < nt; t += 2) {
            seg_R(a0, b0, t + 2);
            seg_M(a1, b1, t + 2);
            seg_R(a1, b1, t + 3);
            seg_M(a0, b0, t + 3);
        }
        STAMP(2);
        // switch the operand pointers to the next output tile of this workgroup; the last tile
        // re-requests itself (the loads are harmless and keep the stream branch-free)
        prev_m0 = m0;
        prev_n0 = n0;
        const bool more = tile + (int)gridDim.x < total_tiles;
        if (more) tile += gridDim.x;
        xcd_tile_map(tile, tiles_n, tiles_m, tile_m, tile_n);
        m0 = tile_m * BIG_BM;
        n0 = tile_n * BIG_BN;
        set_sources(m0, n0);
        bias_buf ^= 1;
        load_bias(n0, bias_buf);
        seg_R(a0, b0, 0);    // R(nt-2)
        seg_M(a1, b1, 0);    // M(nt-3)
        seg_R(a1, b1, 1);    // R(nt-1)
        seg_M(a0, b0, 1);    // M(nt-2)
        seg_R(a0, b0, 2);    // R(0) of the next output tile
        seg_M(a1, b1, 2);    // M(nt-1)
        STAMP(3);
        pending = true;
        if (!more) {
            epilogue(prev_m0, prev_n0);
            STAMP(4);
            break;
        }
    }
    if (!late) seg_barrier();  // the last segment of waves 4-7
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the re-requested K-tiles land before the LDS is released
}

// ------------------------------------------------------------------ 128 x BN, 2 stages
constexpr int BM = 128;

template <int BN, int FMT>
__global__ __launch_bounds__(256, 2) void gemm16_kernel(GemmParams p) {
    constexpr int A_BYTES = BM * ROW_BYTES;   // 16 KiB
    constexpr int B_BYTES = BN * ROW_BYTES;   // 16 or 8 KiB
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int WAVES_N = BN / 64;          // 2 or 1
    constexpr int WAVES_M = 4 / WAVES_N;      // 2 or 4
    constexpr int WM = BM / WAVES_M;          // 64 or 32
    constexpr int MI = WM / 16;               // 4 or 2
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N;
    const int wn = wave % WAVES_N;

    int tile_m = blockIdx.y, tile_n = blockIdx.x;
    if (gridDim.z == 1) xcd_tile_map(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x, gridDim.y, tile_m, tile_n);
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
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
        if (p.mode != GEMM_F32_ATOMIC) {  // partial sums side by side: slice z of C, bias in slice 0
            c_off = (long)z * p.zC1;
            if (z > 0) bias = nullptr;
            kt_end = max(kt_end, kt_begin);  // an empty K range still writes its (zero) slice
        } else if (kt_begin >= kt_end) {
            return;
        }
    } else if (p.nz > 1) {
        A += (long)z * p.zA;
        const int zw = z % p.zWmod;
        W += (long)zw * p.zW;
        if (bias) bias += (long)zw * p.zBias;
        c_off = (long)(z / p.zCdiv) * p.zC1 + (long)(z % p.zCdiv) * p.zC2;
    }

    // Staging through registers (16-byte global loads, ds_write_b128), not LDS-DMA: a launch of this kernel has one
    // or two workgroups per CU, i.e. one or two waves per SIMD, and an LDS-DMA instruction holds its wave's issue
    // for ~100 cycles -- eight of them per K-tile cost more than the K-tile's 32 MFMAs.  Thread t takes the 16-byte
    // chunk t & 7 of rows (t >> 3) + 32 i; chunk c of row r lands at position c ^ (r & 7) of the row (the layout
    // the fragment reads expect).
    const int prow = tid >> 3, pch = tid & 7;
    constexpr int B_PIECES = BN / 32;           // per thread: 4 or 2
    const bf16_t* a_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a_src[i] = A + (long)min(m0 + prow + 32 * i, p.M - 1) * p.lda + pch * 8;
    const bf16_t* b_src[B_PIECES];
#pragma unroll
    for (int i = 0; i < B_PIECES; ++i) b_src[i] = W + (long)(n0 + prow + 32 * i) * p.ldw + pch * 8;
    const int lds_piece = prow * ROW_BYTES + ((pch ^ (prow & 7)) << 4);   // (+ 32 i rows: the same row & 7)
    u16x8 ra[4], rb[B_PIECES];
    auto fetch = [&](int kt) {
        const long koff = (long)kt * BK;
        long koff_a = koff;  // A repeats along K with period p.ka (= K: never wraps)
        while (koff_a >= p.ka) koff_a -= p.ka;
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const u16x8*>(a_src[i] + koff_a);
#pragma unroll
        for (int i = 0; i < B_PIECES; ++i) rb[i] = *reinterpret_cast<const u16x8*>(b_src[i] + koff);
    };
    auto park = [&](int buf) {
        char* base = lds + buf * STAGE + lds_piece;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u16x8*>(base + i * 32 * ROW_BYTES) = ra[i];
#pragma unroll
        for (int i = 0; i < B_PIECES; ++i) *reinterpret_cast<u16x8*>(base + A_BYTES + i * 32 * ROW_BYTES) = rb[i];
    };

    const int fr = lane & 15, fg = lane >> 4, fx = lane & 7;
    // the accumulators start from the bias (as in the 256 x 128 kernel, so both kernels
    // round identically and a result does not depend on which one a shape selects)
    f32x4 acc[MI][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (bias && p.mode != GEMM_F32_ATOMIC) bv = *reinterpret_cast<const f32x4*>(bias + n0 + wn * 64 + j * 16 + fg * 4);
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[i][j] = bv;
    }
    if (p.ln_aff) {  // folded LayerNorm, rows' factors given: start from -ubar[row] g[n], as the 208 x 256 kernel does
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const float nm = p.ln_aff[(long)min(m0 + wm * WM + i * 16 + fr, p.M - 1) * 2 + 1];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 gv = *reinterpret_cast<const f32x4*>(p.ln_vec + n0 + wn * 64 + j * 16 + fg * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = fmul_nc(gv[r], nm);
            }
        }
    }

    const int a_row_off = (wm * WM + fr) * ROW_BYTES;
    const int b_row_off = A_BYTES + (wn * 64 + fr) * ROW_BYTES;
    const int c0 = ((0 + fg) ^ fx) << 4;
    const int c1 = ((4 + fg) ^ fx) << 4;

    // K-tile kt in LDS stage cur, K-tile kt + 1 in registers (requested one iteration earlier, parked into the other
    // stage after this iteration's products), K-tile kt + 2 requested
    if (kt_begin < kt_end) {
        fetch(kt_begin);
        park(0);
        if (kt_begin + 1 < kt_end) fetch(kt_begin + 1);
    }
    __syncthreads();
    // The two 32-deep steps of a K-tile keep their fragments in two register sets, and each set is read while the
    // other one multiplies (with one wave per SIMD nothing else covers an LDS read's latency): step 1 of tile kt is
    // requested before the products of step 0, step 0 of tile kt + 1 -- behind the barrier that releases its stage --
    // before the products of step 1.  The order of the products is that of mma_ktile.
    u16x8 af0[MI], bf0[4], af1[MI], bf1[4];
    auto read_frags = [&](int buf, int co, u16x8 (&af)[MI], u16x8 (&bfr)[4]) {
        const char* base = lds + buf * STAGE;
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const u16x8*>(base + a_row_off + i * 16 * ROW_BYTES + co);
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const u16x8*>(base + b_row_off + j * 16 * ROW_BYTES + co);
    };
    auto products = [&](const u16x8 (&af)[MI], const u16x8 (&bfr)[4]) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = Op<FMT>::mfma(bfr[j], af[i], acc[i][j]);
    };
    int cur = 0;
    if (kt_begin < kt_end) read_frags(0, c0, af0, bf0);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        read_frags(cur, c1, af1, bf1);
        products(af0, bf0);
        if (kt + 1 < kt_end) {
            park(cur ^ 1);   // last read in iteration kt - 1: every wave took those fragments into its products before
                             // it reached the barrier of that iteration
            if (kt + 2 < kt_end) fetch(kt + 2);
        }
        // (not __syncthreads(): that would also wait for the global loads just requested)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + 1 < kt_end) read_frags(cur ^ 1, c0, af0, bf0);
        products(af1, bf1);
        cur ^= 1;
    }
    store_block<FMT, MI>(p, acc, c_off, m0 + wm * WM, n0 + wn * 64, lane);
}

// ------------------------------------------------------------------ epilogue of a K split
// part[z][m][n] (z < S, slice stride `slice` floats; the bias is already in slice 0) -> C by `mode`.
// One thread per four consecutive columns; the slices are added in index order.
template <int FMT>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const float* __restrict__ part, int S, long slice, int M,
                                                              int N, int mode, void* __restrict__ C, long ldc,
                                                              const float* __restrict__ R,
                                                              const float* __restrict__ dot_w,
                                                              float* __restrict__ dot_out) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int n4 = N >> 2;
    const long m = idx / n4;
    const int n = (int)(idx - m * n4) << 2;
    const bool live = m < M;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        const float* src = part + m * N + n;
        v = *reinterpret_cast<const f32x4*>(src);
        for (int z = 1; z < S; ++z) v += *reinterpret_cast<const f32x4*>(src + z * slice);
        const long o = m * ldc + n;
        if (mode == GEMM_BF16 || mode == GEMM_BF16_GELU) {
            if (mode == GEMM_BF16_GELU) v = gelu_erf4(v);
            u16x4 q;
#pragma unroll
            for (int r = 0; r < 4; ++r) q[r] = Op<FMT>::pack(v[r]);
            *reinterpret_cast<u16x4*>(reinterpret_cast<bf16_t*>(C) + o) = q;
        } else {
            if (mode == GEMM_F32_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (mode == GEMM_F32_RESID) v += *reinterpret_cast<const f32x4*>(R + o);
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(C) + o) = v;
        }
    }
    if (dot_w) {  // uniform; N % 64 == 0, so the 16 threads of a 64-column block are 16 aligned lanes of one row
        const f32x4 w = *reinterpret_cast<const f32x4*>(dot_w + n);
        float d = live ? v[0] * w[0] + v[1] * w[1] + v[2] * w[2] + v[3] * w[3] : 0.f;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) d += __shfl_xor(d, o, 64);
        if (live && (threadIdx.x & 15) == 0) dot_out[(long)(n >> 6) * M + m] = d;
    }
}

template <int FMT, int MODE, bool LNROWS = false>
hipError_t launch_big_mode(const GemmParams& p, dim3 grid, hipStream_t s) {
    static LdsAttrState lds_attr;
    const hipError_t attr =
        ensure_dynamic_lds(reinterpret_cast<const void*>(&gemm16_big_kernel<FMT, MODE, LNROWS>), BIG_LDS, lds_attr);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL((gemm16_big_kernel<FMT, MODE, LNROWS>), grid, dim3(512), BIG_LDS, s, p);
    return hipGetLastError();
}

template <int FMT>
hipError_t launch_big(const GemmParams& p, dim3 grid, hipStream_t s) {
    if (p.ln_aff)
        return p.mode == GEMM_BF16 ? launch_big_mode<FMT, GEMM_BF16, true>(p, grid, s)
                                   : launch_big_mode<FMT, GEMM_BF16_GELU, true>(p, grid, s);
    switch (p.mode) {
        case GEMM_BF16: return launch_big_mode<FMT, GEMM_BF16>(p, grid, s);
        case GEMM_BF16_GELU: return launch_big_mode<FMT, GEMM_BF16_GELU>(p, grid, s);
        case GEMM_F32: return launch_big_mode<FMT, GEMM_F32>(p, grid, s);
        case GEMM_F32_RESID: return launch_big_mode<FMT, GEMM_F32_RESID>(p, grid, s);
        case GEMM_F32_RELU: return launch_big_mode<FMT, GEMM_F32_RELU>(p, grid, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace

static int device_cu_count() {  // of the current device; 0 on failure
    static std::atomic<int> n_cu_of[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int n_cu = n_cu_of[dev].load(std::memory_order_relaxed);
    if (n_cu == 0) {
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) return 0;
        n_cu_of[dev].store(n_cu, std::memory_order_relaxed);
    }
    return n_cu;
}

// The K slices the launcher cuts a GEMM of few output tiles into when it is given a scratch (1: no cut).
static int own_split_slices(const GemmArgs& a, int bn) {
    if (!a.split_ws || a.splitk != 1 || a.nz != 1 || a.grp_in != 0 || a.N % 4 != 0 || a.K % BK != 0) return 1;
    if (a.ln_mode != GemmArgs::LN_NONE) return 1;  // the split's second kernel does not carry the folded LayerNorm
    if (a.pre_out) return 1;                       // nor the second output
    if (!(a.mode == GEMM_BF16 || a.mode == GEMM_BF16_GELU || a.mode == GEMM_F32 || a.mode == GEMM_F32_RESID ||
          a.mode == GEMM_F32_RELU))
        return 1;
    const int ktiles = a.K / BK;
    const long tiles = (long)(a.N / bn) * ((a.M + BM - 1) / BM);
    const int n_cu = device_cu_count();
    if (n_cu <= 0 || tiles * 2 > n_cu || ktiles < 4) return 1;
    long S = std::min<long>(ktiles / 2, n_cu / tiles);
    S = std::min<long>(S, 16);
    S = std::min<long>(S, (long)(a.split_ws_bytes / ((size_t)a.M * a.N * 4)));
    return S >= 2 ? (int)S : 1;
}

constexpr long WIDE_TILE_COST_X100 = 141;  // time of one 208 x 256 tile in units of a 256 x 128 tile, x 100

static bool takes_big_kernel(const GemmArgs& a) {
    const int n_cu_hint = device_cu_count();
    const long big_tiles = (long)(a.N / BIG_BN) * ((a.M + BIG_BM - 1) / BIG_BM);
    return (a.bn == 0 || a.bn == 1 || a.bn == 256) && a.nz == 1 && a.splitk == 1 && a.M >= 1024 && a.N % BIG_BN == 0 &&
           !a.pre_out &&
           (n_cu_hint <= 0 || big_tiles * 4 >= (long)n_cu_hint * 3) &&
           a.K % (2 * BK) == 0 && a.K >= 4 * BK &&  // the staggered schedule wants an even tile count
           a.ldc % 8 == 0 && a.mode != GEMM_F32_ATOMIC && a.mode != GEMM_F32_RESID_GELU;  // 16-byte row stores
}

// Column width of the 128-row kernel's tile for `a` when it does not take a persistent kernel: bn = 0 (library's
// choice), 1 (256 x 128 asked for) and 256 (208 x 256 asked for) all fall back to 128 or 64 by N.  One place, so that
// the launcher and gemm_split_scratch_bytes always see the same K split.
static int tile_width_of(const GemmArgs& a) {
    return a.bn == 0 || a.bn == 1 || a.bn == 256 ? (a.N % 128 == 0 ? 128 : 64) : a.bn;
}

size_t gemm_split_scratch_bytes(const GemmArgs& a, size_t limit) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0 || takes_big_kernel(a)) return 0;
    GemmArgs q = a;
    q.split_ws = reinterpret_cast<float*>(16);  // any non-null value: only asked about
    q.split_ws_bytes = limit;
    const int bn = tile_width_of(a);
    if (a.N % bn != 0) return 0;
    const int S = own_split_slices(q, bn);
    return S > 1 ? (size_t)S * a.M * a.N * 4 : 0;
}

// Whether the library itself (bn = 0) gives an M x N GEMM with enough rows for the 256 x 128 kernel to the 208 x 256
// kernel: its tiles must fill three quarters of the CUs and take fewer weighted rounds.
static bool wide_wins(int M, int N, int n_cu) {
    const int m_step = gemm_wide_row_step(M);
    const long wt = gemm_wide_tiles(M, N, m_step);
    const long bt = (long)(N / BIG_BN) * ((M + BIG_BM - 1) / BIG_BM);
    const long wr = (wt + n_cu - 1) / n_cu, br = (bt + n_cu - 1) / n_cu;
    // a 208 x 256 tile is 1.625 tiles of 256 x 128 in matrix work and runs its K loop faster
    return wt * 4 >= (long)n_cu * 3 && wr * WIDE_TILE_COST_X100 < br * 100;
}

bool gemm_ln_fold_fits(int M, int ffn, bool any_batch) {
    const int n_cu = device_cu_count();
    if (n_cu <= 0 || M < 1 || ffn % 256 != 0) return false;
    const int m_step = gemm_wide_row_step(M);
    if (gemm_wide_apply_cgroups(M, 3072, m_step, n_cu) < 1 || gemm_wide_apply_cgroups(M, ffn, m_step, n_cu) < 1) return false;
    if (any_batch) return true;
    return M >= 1024 && wide_wins(M, 1024, n_cu) && wide_wins(M, 3072, n_cu) && wide_wins(M, ffn, n_cu);
}

static hipError_t launch_gemm_checked(const GemmArgs& a, hipStream_t s) {
    if (a.deferred_slices) {
        *a.deferred_slices = 1;
        if (!(a.mode == GEMM_F32 || a.mode == GEMM_F32_RESID) || a.dot_w || a.ldc != a.N) return hipErrorInvalidValue;
    }
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return hipErrorInvalidValue;
    if (a.bn != 128 && a.bn != 64 && a.bn != 0 && a.bn != 256 && a.bn != 1) return hipErrorInvalidValue;
    if (a.fmt != FMT_BF16 && a.fmt != FMT_F16) return hipErrorInvalidValue;
    if ((a.lda % 8) || (a.ldw % 8) || (a.ldc % 4)) return hipErrorInvalidValue;
    if ((reinterpret_cast<uintptr_t>(a.A) & 15) || (reinterpret_cast<uintptr_t>(a.W) & 15) ||
        (reinterpret_cast<uintptr_t>(a.C) & 15))
        return hipErrorInvalidValue;
    if (a.bias && (reinterpret_cast<uintptr_t>(a.bias) & 15)) return hipErrorInvalidValue;
    // split-K: atomics into one C, or (GEMM_F32) partial sums written to splitk slices of C, zC1 elements apart
    if (a.splitk > 1 && (a.nz != 1 || (a.mode != GEMM_F32_ATOMIC && (a.mode != GEMM_F32 || a.zC1 < (long)a.M * a.ldc))))
        return hipErrorInvalidValue;
    if (a.mode == GEMM_F32_ATOMIC && a.splitk < 1) return hipErrorInvalidValue;
    if (a.nz < 1 || a.zWmod < 1 || a.zCdiv < 1) return hipErrorInvalidValue;
    if ((a.zA % 8) || (a.zW % 8) || (a.zBias % 4) || (a.zC1 % 4) || (a.zC2 % 4)) return hipErrorInvalidValue;
    // bn == 0: the library picks the tile (256 x 128 three-stage kernel for large plain GEMMs)
    // (a shape that would give the 256 x 128 kernel fewer tiles than three quarters of the CUs is better served
    // by 128-row tiles, cut along K as well if they are still few)
    const bool big = takes_big_kernel(a);
    const int bn = tile_width_of(a);
    if (a.N % bn != 0 || a.K % BK != 0) return hipErrorInvalidValue;
    GemmParams p;
    p.A = a.A; p.W = a.W; p.bias = a.bias; p.C = a.C;
    p.lda = a.lda; p.ldw = a.ldw; p.ldc = a.ldc;
    p.M = a.M; p.N = a.N; p.K = a.K; p.mode = a.mode;
    if (a.ka < 0 || (a.ka > 0 && (a.ka % BK != 0 || a.K % a.ka != 0))) return hipErrorInvalidValue;
    p.ka = a.ka > 0 ? a.ka : a.K;
    p.nz = a.nz; p.zA = a.zA; p.zWmod = a.zWmod; p.zW = a.zW; p.zBias = a.zBias;
    p.zCdiv = a.zCdiv; p.zC1 = a.zC1; p.zC2 = a.zC2;
    p.splitk = a.splitk;
    const int ktiles = a.K / BK;
    p.ktiles_per_split = (ktiles + a.splitk - 1) / a.splitk;
    p.grp_in = a.grp_in; p.grp_valid = a.grp_valid; p.grp_out = a.grp_out;
    p.dot_w = a.dot_w; p.dot_out = a.dot_out;
    p.pre_out = a.pre_out;
    if (a.pre_out && (a.mode != GEMM_BF16_GELU || a.nz != 1 || a.splitk != 1 || a.grp_in > 0 ||
                      (reinterpret_cast<uintptr_t>(a.pre_out) & 7)))
        return hipErrorInvalidValue;
    const bool resid_mode = a.mode == GEMM_F32_RESID || a.mode == GEMM_F32_RESID_GELU;
    if (a.resid && (!resid_mode || (reinterpret_cast<uintptr_t>(a.resid) & 15))) return hipErrorInvalidValue;
    p.R = a.resid ? a.resid : reinterpret_cast<const float*>(a.C);
    if ((a.dot_w != nullptr) != (a.dot_out != nullptr)) return hipErrorInvalidValue;
    if (a.dot_w && (a.mode != GEMM_F32_RESID || a.nz != 1 || a.splitk != 1 || a.grp_in > 0 || a.N % 64 != 0 ||
                    (reinterpret_cast<uintptr_t>(a.dot_w) & 15)))
        return hipErrorInvalidValue;
    p.ln_mode = a.ln_mode; p.ln_proxy = a.ln_proxy; p.ln_vec = a.ln_vec; p.ln_a = a.ln_a; p.ln_slots = a.ln_slots;
    p.ln_stat = a.ln_stat; p.ln_eps = a.ln_eps; p.ln_cgroups = 0; p.ln_aff = nullptr; p.ln_c = nullptr;
    const bool ln_rows = a.ln_mode == GemmArgs::LN_APPLY && a.ln_aff != nullptr;
    if (ln_rows) {
        // rows' factors given: the 128-row or the 256 x 128 kernel, whichever the shape selects; c arrives as the bias
        if (!(a.mode == GEMM_BF16 || a.mode == GEMM_BF16_GELU) || a.K != 1024 || a.nz != 1 || a.splitk != 1 || a.grp_in != 0 ||
            !a.bias || !a.ln_vec || a.bn == 256 || ((reinterpret_cast<uintptr_t>(a.ln_aff) | reinterpret_cast<uintptr_t>(a.ln_vec)) & 15))
            return hipErrorInvalidValue;
        p.ln_mode = GemmArgs::LN_NONE;  // nothing of the 208 x 256 kernel's variants
        p.ln_aff = a.ln_aff; p.ln_c = a.bias; p.bias = nullptr;  // (the 256 x 128 kernel takes g as its bias image, below)
    } else if (a.ln_mode != GemmArgs::LN_NONE) {  // the folded LayerNorm in the 208 x 256 kernel
        const int n_cu = device_cu_count();
        if (n_cu <= 0) return hipErrorUnknown;
        if (a.ln_mode != GemmArgs::LN_EMIT && a.ln_mode != GemmArgs::LN_APPLY) return hipErrorInvalidValue;
        if ((a.bn != 0 && a.bn != 256) || !gemm_wide_supported(p)) return hipErrorInvalidValue;
        const uintptr_t ptrs = reinterpret_cast<uintptr_t>(a.ln_proxy) | reinterpret_cast<uintptr_t>(a.ln_vec) |
                               reinterpret_cast<uintptr_t>(a.ln_a) | reinterpret_cast<uintptr_t>(a.ln_slots) |
                               reinterpret_cast<uintptr_t>(a.ln_stat);
        if (ptrs & 15) return hipErrorInvalidValue;
        return launch_gemm_wide(p, gemm_wide_row_step(a.M), a.fmt, n_cu, s);
    }
    // the 208 x 256 kernel: when it is asked for, or when its tiles fill the CUs in fewer (weighted) rounds
    if (!ln_rows && (a.bn == 256 || a.bn == 0) && gemm_wide_supported(p)) {
        const int n_cu = device_cu_count();
        if (n_cu <= 0) return hipErrorUnknown;
        const int m_step = gemm_wide_row_step(a.M);
        const bool wide = a.bn == 256 || (big && wide_wins(a.M, a.N, n_cu));
        if (wide) return launch_gemm_wide(p, m_step, a.fmt, n_cu, s);
    } else if (a.bn == 256) {
        return hipErrorInvalidValue;
    }
    if (big) {
        const long tiles = (long)(a.N / BIG_BN) * ((a.M + BIG_BM - 1) / BIG_BM);
        const int n_cu = device_cu_count();  // one workgroup per CU (146 KiB of LDS each)
        if (n_cu <= 0) return hipErrorUnknown;
        dim3 grid((unsigned)(tiles < n_cu ? tiles : n_cu), 1, 1);
        if (ln_rows) p.bias = p.ln_vec;
        return a.fmt == FMT_BF16 ? launch_big<FMT_BF16>(p, grid, s) : launch_big<FMT_F16>(p, grid, s);
    }
    // few output tiles and a scratch at hand: cut K into S slices (see GemmArgs::split_ws)
    const int own_split = big ? 1 : own_split_slices(a, bn);
    if (own_split > 1) {
        p.C = a.split_ws;
        p.ldc = a.N;
        p.mode = GEMM_F32;
        p.splitk = own_split;
        p.ktiles_per_split = (ktiles + own_split - 1) / own_split;
        p.zC1 = (long)a.M * a.N;
        p.dot_w = nullptr;
        p.dot_out = nullptr;
    }
    const int gz = own_split > 1 ? own_split : (a.splitk > 1 ? a.splitk : a.nz);
    dim3 grid(a.N / bn, (a.M + BM - 1) / BM, gz);
    if (grid.y > 65535 || grid.z > 65535) return hipErrorInvalidValue;
    if (bn == 128 && a.fmt == FMT_BF16)
        hipLaunchKernelGGL((gemm16_kernel<128, FMT_BF16>), grid, dim3(256), 0, s, p);
    else if (bn == 128)
        hipLaunchKernelGGL((gemm16_kernel<128, FMT_F16>), grid, dim3(256), 0, s, p);
    else if (a.fmt == FMT_BF16)
        hipLaunchKernelGGL((gemm16_kernel<64, FMT_BF16>), grid, dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL((gemm16_kernel<64, FMT_F16>), grid, dim3(256), 0, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || own_split == 1) return e;
    if (a.deferred_slices) {   // the caller's next kernel adds the slices
        *a.deferred_slices = own_split;
        return e;
    }
    const float* R = a.resid ? a.resid : reinterpret_cast<const float*>(a.C);
    const long threads = (long)a.M * (a.N / 4);
    const dim3 egrid((unsigned)((threads + 255) / 256));
    SLSMI_FMT_DISPATCH(a.fmt, hipLaunchKernelGGL(splitk_epilogue_kernel<F>, egrid, dim3(256), 0, s, a.split_ws, own_split,
                                                 (long)a.M * a.N, a.M, a.N, a.mode, a.C, a.ldc, R, a.dot_w, a.dot_out));
    return hipGetLastError();
}

// In-place timing of the GEMM launches (diagnostic): while a timing window is open, every launch_gemm call is
// bracketed by two HIP events recorded on the stream the kernels go to, so a caller can time the GEMM kernels of
// a whole forward where they run (cold operands, the neighbours' cache contents) instead of shape by shape.
namespace {
struct GemmTimer {
    std::mutex mu;
    std::atomic<bool> open{false};
    std::vector<hipEvent_t> ev;   // pairs: 2 i before, 2 i + 1 after launch i
    std::vector<double> flop;     // 2 M N K nz of launch i
    int cap = 0;                  // launches this window may time
    int device = -1;              // the window's device: launches on other devices run untimed and uncounted
    int used = 0;                 // launches recorded
    int dropped = 0;              // launches beyond the window's capacity (not timed)
};
GemmTimer g_gemm_timer;
}  // namespace

hipError_t launch_gemm(const GemmArgs& a, hipStream_t s) {
    GemmTimer& t = g_gemm_timer;
    if (!t.open.load(std::memory_order_acquire)) return launch_gemm_checked(a, s);
    std::lock_guard<std::mutex> lock(t.mu);
    int dev = -1;
    if (!t.open.load(std::memory_order_relaxed) || hipGetDevice(&dev) != hipSuccess || dev != t.device)
        return launch_gemm_checked(a, s);
    if (t.used >= t.cap) {
        ++t.dropped;
        return launch_gemm_checked(a, s);
    }
    hipError_t e = hipEventRecord(t.ev[2 * t.used], s);
    if (e != hipSuccess) return e;
    e = launch_gemm_checked(a, s);
    if (e != hipSuccess) return e;
    e = hipEventRecord(t.ev[2 * t.used + 1], s);
    if (e != hipSuccess) return e;
    // (algorithmic work: a launch on split operands counts the layer's own contraction length)
    t.flop[t.used] = 2.0 * a.M * a.N * (a.k_algo > 0 ? a.k_algo : (a.ka > 0 ? a.ka : a.K)) * (a.nz > 1 ? a.nz : 1);
    ++t.used;
    return hipSuccess;
}

hipError_t gemm_timing_begin(int max_launches) {
    GemmTimer& t = g_gemm_timer;
    std::lock_guard<std::mutex> lock(t.mu);
    if (t.open.load(std::memory_order_relaxed) || max_launches < 1 || max_launches > 65536) return hipErrorInvalidValue;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    if (!t.ev.empty() && dev != t.device) {  // events belong to the device they were made on
        for (hipEvent_t e : t.ev) (void)hipEventDestroy(e);
        t.ev.clear();
    }
    t.device = dev;
    while (t.ev.size() < (size_t)2 * max_launches) {
        hipEvent_t e;
        hipError_t rc = hipEventCreate(&e);
        if (rc != hipSuccess) return rc;
        t.ev.push_back(e);
    }
    t.flop.assign(max_launches, 0.0);
    t.cap = max_launches;
    t.used = 0;
    t.dropped = 0;
    t.open.store(true, std::memory_order_release);
    return hipSuccess;
}

hipError_t gemm_timing_end(int* launches, int* dropped, double* seconds, double* flops) {
    GemmTimer& t = g_gemm_timer;
    std::lock_guard<std::mutex> lock(t.mu);
    if (!t.open.load(std::memory_order_relaxed)) return hipErrorInvalidValue;
    t.open.store(false, std::memory_order_release);
    double sec = 0.0, fl = 0.0;
    for (int i = 0; i < t.used; ++i) {
        hipError_t rc = hipEventSynchronize(t.ev[2 * i + 1]);
        float ms = 0.f;
        if (rc == hipSuccess) rc = hipEventElapsedTime(&ms, t.ev[2 * i], t.ev[2 * i + 1]);
        if (rc != hipSuccess) return rc;
        sec += 1e-3 * ms;
        fl += t.flop[i];
    }
    *launches = t.used;
    *dropped = t.dropped;
    *seconds = sec;
    *flops = fl;
    return hipSuccess;
}

#ifdef SLSMI_STAMPS
extern "C" int slsmi_debug_read_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 256);
}
#endif

}  // namespace slsmi
