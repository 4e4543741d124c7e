// 16-bit-operand (bf16 or f16) MFMA GEMM for gfx950:
//     C[m][n] (op)= sum_k A[m][k] * W[n][k] (+ bias[n])
//
// Two kernels share the fragment layout, the staging scheme and the epilogue:
//
//  * gemm16_big_kernel: 256 x 128 x 64 block tile, 8 waves (2 per SIMD), three LDS
//    stages (144 KiB, one workgroup per CU), loads two K-tiles ahead behind a counted
//    `s_waitcnt vmcnt(6)`, fragments double-buffered in registers, and the two waves
//    of every SIMD half a phase apart (one multiplies while the other fetches).
//    Used for every large GEMM of the forward.
//  * gemm16_kernel<BN>: 128 x BN x 64 (BN = 128 or 64), 4 waves, two stages;
//    used for small M, the grouped positional conv (batched over blockIdx.z) and
//    split-K.
//
// Staging: 16-byte `global_load_lds_dwordx4`; the LDS image is lane-linear
// ([row][64] 16-bit, 128-B rows), so the bank-conflict swizzle (16-byte chunk
// index XOR (row & 7)) is applied to the per-lane SOURCE address and again on the
// ds_read_b128 side.  MFMA: v_mfma_f32_16x16x32_{bf16,f16} with the weight
// fragment as the first operand, so a lane's four accumulator registers are four
// consecutive n of one row m: epilogue accesses are 8 B (16-bit out) or 16 B (fp32).
#include "common.h"
#include "kernels.h"

namespace slsmi {

namespace {

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

// Workgroups are dealt round-robin over the 8 XCDs (ids b and b + 8 share an L2),
// so a plain (x, y) map spreads every operand panel over all eight 4 MiB L2s.
// Re-number so that each XCD walks ONE contiguous range of a tile order whose
// concurrently resident tiles form a compact patch (8 tile rows x all-needed
// tile columns, rows fastest).  Affinity only: any placement computes the same tiles.
__device__ __forceinline__ void xcd_tile_map(int lin, int tiles_n, int tiles_m, int& tile_m, int& tile_n) {
    const int total = tiles_n * tiles_m;
    const int q = total >> 3, r = total & 7;
    const int xcd = lin & 7, j = lin >> 3;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    constexpr int GM = 8;
    const int band = v / (GM * tiles_n);
    const int rem = v - band * (GM * tiles_n);
    const int band_rows = min(GM, tiles_m - band * GM);
    tile_n = rem / band_rows;
    tile_m = band * GM + (rem - tile_n * band_rows);
}

// One wave's 64 x 64 (MI = 4) or 32 x 64 (MI = 2) accumulator block -> memory.
// Lane holds C[m][n .. n+3] with m = m_base + 16 i + (lane & 15), n = n_base + 16 j + 4 (lane >> 4).
template <int FMT, int MI>
__device__ __forceinline__ void store_block(const GemmParams& p, const f32x4 (&acc)[MI][4], const float* bias,
                                            long c_off, int m_base, int n_base, int lane) {
    const int fr = lane & 15, fg = lane >> 4;
    const int mode = p.mode;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m_base + i * 16 + fr;
        if (m >= p.M) continue;
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
            if (bias && mode != GEMM_F32_ATOMIC) v += *reinterpret_cast<const f32x4*>(bias + n);
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
                if (mode == GEMM_F32_RESID || mode == GEMM_F32_RESID_GELU) v += *reinterpret_cast<const f32x4*>(c);
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
// workgroup per CU).  A wave alternates
//   R(t): read the 16 fragments of K-tile t into one of two register sets; refill
//         the stage of tile t-1 with tile t+2 (6 LDS-DMA loads); counted
//         `s_waitcnt vmcnt(6)` = this wave's loads of tile t+1 have landed; barrier
//   M(t): the 32 MFMAs of tile t; barrier
// Waves 4-7 share their SIMDs with waves 0-3 and run the same program ONE SEGMENT
// LATER, so on every SIMD one wave multiplies while its partner fetches: the matrix
// pipe does not idle during LDS reads, waits and barriers.
// Slots (barrier to barrier) of waves 0-3: R(0), -, R(1), M(0), R(2), M(1), ...; waves
// 4-7 run the same list one slot later.
//   Readiness: tile j is first read in R(j) of waves 0-3; every wave waited for its
//   pieces of tile j in its own R(j-1), which ends, at the latest (waves 4-7), with
//   the barrier right before that R(j).
//   Re-use: the stage of tile t-1 is refilled from R(t) on; every wave's fragment reads
//   of tile t-1 were waited for (lgkmcnt(0)) before the barrier that ended its R(t-1),
//   which for waves 4-7 is the barrier right before R(t) of waves 0-3.
constexpr int BIG_BM = 256;
constexpr int BIG_BN = 128;
constexpr int BIG_A_BYTES = BIG_BM * ROW_BYTES;  // 32 KiB
constexpr int BIG_B_BYTES = BIG_BN * ROW_BYTES;  // 16 KiB
constexpr int BIG_STAGE = BIG_A_BYTES + BIG_B_BYTES;
constexpr int BIG_STAGES = 3;
constexpr int BIG_LDS = BIG_STAGES * BIG_STAGE;  // 147456

#ifndef SLSMI_ABL
#define SLSMI_ABL 0  // diagnostic builds only: 1 = no LDS-DMA in the loop, 2 = no MFMA, 3 = no fragment reads
#endif

template <int FMT>
__global__ __launch_bounds__(512, 2) void gemm16_big_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7
    const int wm = wave >> 1, wn = wave & 1;
    const int late = wave >> 2;  // 1: waves 4-7, one segment behind

    const int nt = p.K / BK;
    const int tiles_n = p.N / BIG_BN, tiles_m = (p.M + BIG_BM - 1) / BIG_BM;
    const int total_tiles = tiles_n * tiles_m;
    // persistent: workgroup b takes tiles b, b + gridDim.x, ... (all on one XCD's L2, see
    // xcd_tile_map); the first two K-tiles of the NEXT output tile are requested before
    // the epilogue of the current one, so its load latency hides behind the stores.
    int tile = blockIdx.x;
    int tile_m, tile_n;
    xcd_tile_map(tile, tiles_n, tiles_m, tile_m, tile_n);
    int m0 = tile_m * BIG_BM, n0 = tile_n * BIG_BN;

    // staging: one wave-instruction moves 8 rows x 128 B; 32 A pieces + 16 W pieces per K-tile
    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;
    const bf16_t* a_src[4];
    const bf16_t* b_src[2];
    auto set_sources = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = min(m0 + (wave * 4 + i) * 8 + srow, p.M - 1);
            a_src[i] = p.A + (long)r * p.lda + schunk * 8;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = n0 + (wave * 2 + i) * 8 + srow;
            b_src[i] = p.W + (long)r * p.ldw + schunk * 8;
        }
    };
    set_sources();
    // A K-tile's six loads per wave are issued in two parts: the W pieces and the first
    // SLSMI_NR A pieces in the R segment, the remaining A pieces in the following M segment
    // (issuing an LDS-DMA load costs the issuing wave ~100 cycles, so the split balances
    // the two segment kinds).
#ifndef SLSMI_NR
#define SLSMI_NR 2
#endif
    constexpr int NR = SLSMI_NR;
    auto stage_r = [&](int buf, int kt) {
        char* base = lds + buf * BIG_STAGE;
        const long koff = (long)kt * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(b_src[i] + koff),
                                             (lptr_t)(base + BIG_A_BYTES + (wave * 2 + i) * 8 * ROW_BYTES), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < NR; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + koff),
                                             (lptr_t)(base + (wave * 4 + i) * 8 * ROW_BYTES), 16, 0, 0);
    };
    auto stage_m = [&](int buf, int kt) {
        char* base = lds + buf * BIG_STAGE;
        const long koff = (long)kt * BK;
#pragma unroll
        for (int i = NR; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + koff),
                                             (lptr_t)(base + (wave * 4 + i) * 8 * ROW_BYTES), 16, 0, 0);
    };
    auto stage = [&](int buf, int kt) {
        stage_r(buf, kt);
        stage_m(buf, kt);
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
    auto seg_barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    // R(t): `steady` = tiles t+1 and t+2 both exist (branch-free).  The wait leaves only the
    // NR + 2 loads of tile t+2 issued just above pending: tile t+1 has landed.
    auto seg_R = [&](int t, int bt, int bprev, frag_t& fa, frag_t& fb, bool steady) {
        if (SLSMI_ABL != 3) read_frags(bt, fa, fb);
        if (steady || t + 2 < nt) {
            if (SLSMI_ABL != 1) stage_r(bprev, t + 2);
            if (NR == 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            if (NR == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            if (NR == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            if (NR == 3) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            if (NR == 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        seg_barrier();
    };
    // M segment that follows R(t): MFMAs of tile t-1 and the rest of tile t+2's loads
    // (the stage of tile t-1... is `bprev` of that R: free since every wave's R(t-1) ended)
    auto seg_M = [&](int t, int bprev, const frag_t& fa, const frag_t& fb, bool steady) {
        __builtin_amdgcn_s_setprio(1);
        if ((steady || t + 2 < nt) && SLSMI_ABL != 1) stage_m(bprev, t + 2);
        if (SLSMI_ABL != 2) {
            mma(fa, fb);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(fa[0][i]), "v"(fb[0][i]), "v"(fa[1][i]), "v"(fb[1][i]));
        }
        __builtin_amdgcn_s_setprio(0);
        seg_barrier();
    };
    auto next_buf = [](int b) { return b + 1 == BIG_STAGES ? 0 : b + 1; };

    stage(0, 0);
    stage(1, 1);
  for (;;) {  // one output tile per iteration; K-tiles 0 and 1 of it are already requested
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    seg_barrier();
    if (late) seg_barrier();
    frag_t a0, b0, a1, b1;
    seg_R(0, 0, BIG_STAGES - 1, a0, b0, false);
    // one slot without an M segment: it carries the second part of tile 2's loads and makes
    // R(t) of waves 0-3 start one barrier after R(t-1) of waves 4-7 for t = 1 as for every t
    if (2 < nt && SLSMI_ABL != 1) stage_m(BIG_STAGES - 1, 2);
    seg_barrier();
    int t = 1, bt = 1, bprev = 0;
    for (; t + 3 < nt; t += 2) {  // t odd: tiles t (set 1) and t+1 (set 0), both steady
        seg_R(t, bt, bprev, a1, b1, true);
        seg_M(t, bprev, a0, b0, true);
        const int bn1 = next_buf(bt);
        seg_R(t + 1, bn1, bt, a0, b0, true);
        seg_M(t + 1, bt, a1, b1, true);
        bprev = bn1;
        bt = next_buf(bn1);
    }
    // tail (nt is even): remaining pairs with the conditional stage/wait, then the last odd tile;
    // register sets stay statically named so nothing is indexed at run time
    for (; t + 1 < nt; t += 2) {
        seg_R(t, bt, bprev, a1, b1, false);
        seg_M(t, bprev, a0, b0, false);
        const int bn1 = next_buf(bt);
        seg_R(t + 1, bn1, bt, a0, b0, false);
        seg_M(t + 1, bt, a1, b1, false);
        bprev = bn1;
        bt = next_buf(bn1);
    }
    seg_R(t, bt, bprev, a1, b1, false);  // t == nt - 1
    seg_M(t, bprev, a0, b0, false);
    mma(a1, b1);
    if (!late) seg_barrier();

    // ---- epilogue through LDS: per-lane accumulator fragments are 8-byte (16-bit out) or
    // 16-byte (fp32) pieces scattered over 16 rows, which makes a store-issue-bound tail;
    // instead each 64-row band of the tile is transposed through the third stage and
    // written as whole rows (256 B or 512 B per row, 16 B per lane).
    {
        // every wave is past its last fragment read: stages 0 and 1 may take the next tile
        seg_barrier();  // raw barriers from here on: LDS-DMA loads are in flight and must not be drained
        const int cur_m0 = m0, cur_n0 = n0;
        tile += gridDim.x;
        const bool more = tile < total_tiles;
        if (more) {
            xcd_tile_map(tile, tiles_n, tiles_m, tile_m, tile_n);
            m0 = tile_m * BIG_BM;
            n0 = tile_n * BIG_BN;
            set_sources();
            stage(0, 0);
            stage(1, 1);
        }
        const int mode = p.mode;
        const bool out16 = mode == GEMM_BF16 || mode == GEMM_BF16_GELU;
        const int pitch = out16 ? (BIG_BN * 2 + 16) : (BIG_BN * 4 + 16);
        char* cl = lds + 2 * BIG_STAGE;
        // bias + activation in registers
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + cur_n0 + wn * 64 + j * 16 + fg * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 v = acc[i][j] + bv;
                if (mode == GEMM_BF16_GELU || mode == GEMM_F32_RESID_GELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
                }
                if (mode == GEMM_F32_RELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
                acc[i][j] = v;
            }
        }
        const bool resid = mode == GEMM_F32_RESID || mode == GEMM_F32_RESID_GELU;
        for (int band = 0; band < 4; ++band) {
            if (band) seg_barrier();  // previous band's rows have been read back
            if (wm == band) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int row = i * 16 + fr, col = wn * 64 + j * 16 + fg * 4;
                        if (out16) {
                            u16x4 o;
#pragma unroll
                            for (int r = 0; r < 4; ++r) o[r] = Op<FMT>::pack(acc[i][j][r]);
                            *reinterpret_cast<u16x4*>(cl + row * pitch + col * 2) = o;
                        } else {
                            *reinterpret_cast<f32x4*>(cl + row * pitch + col * 4) = acc[i][j];
                        }
                    }
            }
            seg_barrier();
            // 8 rows per wave; a row is 16 (16-bit) or 32 (fp32) lanes of 16 B
            const int lanes_per_row = out16 ? 16 : 32;
            const int rows_per_inst = 64 / lanes_per_row;
            const int seg = lane & (lanes_per_row - 1);
            for (int it = 0; it < 8 / rows_per_inst; ++it) {
                const int r = wave * 8 + it * rows_per_inst + lane / lanes_per_row;
                const int m = cur_m0 + band * 64 + r;
                if (m >= p.M || SLSMI_ABL == 4) continue;
                long orow = m;
                if (p.grp_in > 0) {
                    const int g = m / p.grp_in;
                    const int pos = m - g * p.grp_in;
                    if (pos >= p.grp_valid) continue;
                    orow = (long)g * p.grp_out + pos;
                }
                const f32x4 v = *reinterpret_cast<const f32x4*>(cl + r * pitch + seg * 16);
                if (out16) {
                    bf16_t* dst = reinterpret_cast<bf16_t*>(p.C) + orow * p.ldc + cur_n0 + seg * 8;
                    *reinterpret_cast<f32x4*>(dst) = v;
                } else {
                    float* dst = reinterpret_cast<float*>(p.C) + orow * p.ldc + cur_n0 + seg * 4;
                    f32x4 o = v;
                    if (resid) o += *reinterpret_cast<const f32x4*>(dst);
                    *reinterpret_cast<f32x4*>(dst) = o;
                }
            }
        }
        if (!more) break;
        seg_barrier();  // the band buffer (third stage) is free again
    }
  }
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
        if (kt_begin >= kt_end) return;
    } else if (p.nz > 1) {
        A += (long)z * p.zA;
        const int zw = z % p.zWmod;
        W += (long)zw * p.zW;
        if (bias) bias += (long)zw * p.zBias;
        c_off = (long)(z / p.zCdiv) * p.zC1 + (long)(z % p.zCdiv) * p.zC2;
    }

    const int srow = lane >> 3;                 // row within the 8-row piece (= row & 7)
    const int schunk = (lane & 7) ^ srow;       // swizzled source chunk
    const bf16_t* a_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = min(m0 + (wave * 4 + i) * 8 + srow, p.M - 1);
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
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + koff),
                                             (lptr_t)(base + (wave * 4 + i) * 8 * ROW_BYTES), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < B_PIECES; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(b_src[i] + koff),
                                             (lptr_t)(base + A_BYTES + (wave * B_PIECES + i) * 8 * ROW_BYTES), 16, 0, 0);
    };

    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4, fx = lane & 7;
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
        mma_ktile<FMT, MI>(acc, base + a_row_off, base + b_row_off, c0, c1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
    store_block<FMT, MI>(p, acc, bias, c_off, m0 + wm * WM, n0 + wn * 64, lane);
}

template <int FMT>
hipError_t launch_big(const GemmParams& p, dim3 grid, hipStream_t s) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm16_big_kernel<FMT>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL(gemm16_big_kernel<FMT>, grid, dim3(512), BIG_LDS, s, p);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_gemm(const GemmArgs& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return hipErrorInvalidValue;
    if (a.bn != 128 && a.bn != 64 && a.bn != 0) return hipErrorInvalidValue;
    if (a.fmt != FMT_BF16 && a.fmt != FMT_F16) return hipErrorInvalidValue;
    if ((a.lda % 8) || (a.ldw % 8) || (a.ldc % 4)) return hipErrorInvalidValue;
    if ((reinterpret_cast<uintptr_t>(a.A) & 15) || (reinterpret_cast<uintptr_t>(a.W) & 15) ||
        (reinterpret_cast<uintptr_t>(a.C) & 15))
        return hipErrorInvalidValue;
    if (a.bias && (reinterpret_cast<uintptr_t>(a.bias) & 15)) return hipErrorInvalidValue;
    if (a.splitk > 1 && (a.nz != 1 || a.mode != GEMM_F32_ATOMIC)) return hipErrorInvalidValue;
    if (a.mode == GEMM_F32_ATOMIC && a.splitk < 1) return hipErrorInvalidValue;
    if (a.nz < 1 || a.zWmod < 1 || a.zCdiv < 1) return hipErrorInvalidValue;
    if ((a.zA % 8) || (a.zW % 8) || (a.zBias % 4) || (a.zC1 % 4) || (a.zC2 % 4)) return hipErrorInvalidValue;
    // bn == 0: the library picks the tile (256 x 128 three-stage kernel for large plain GEMMs)
    const bool big = a.bn == 0 && a.nz == 1 && a.splitk == 1 && a.M >= 1024 && a.N % BIG_BN == 0 &&
                     a.K % (2 * BK) == 0 && a.K >= 4 * BK &&  // the staggered schedule wants an even tile count
                     a.ldc % 8 == 0 && a.mode != GEMM_F32_ATOMIC;  // 16-byte row stores
    const int bn = a.bn == 0 ? (a.N % 128 == 0 ? 128 : 64) : a.bn;
    if (a.N % bn != 0 || a.K % BK != 0) return hipErrorInvalidValue;
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
    if (big) {
        const long tiles = (long)(a.N / BIG_BN) * ((a.M + BIG_BM - 1) / BIG_BM);
        static int n_cu = 0;  // one workgroup per CU (144 KiB of LDS each)
        if (n_cu == 0) {
            int dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
            n_cu = prop.multiProcessorCount;
        }
        dim3 grid((unsigned)(tiles < n_cu ? tiles : n_cu), 1, 1);
        return a.fmt == FMT_BF16 ? launch_big<FMT_BF16>(p, grid, s) : launch_big<FMT_F16>(p, grid, s);
    }
    const int gz = a.splitk > 1 ? a.splitk : a.nz;
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
    return hipGetLastError();
}

}  // namespace slsmi
