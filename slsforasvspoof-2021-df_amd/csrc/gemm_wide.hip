// 16-bit-operand MFMA GEMM for gfx950 with a 208 x 256 x 64 block tile:
//     C[m][n] (op)= sum_k A[m][k] * W[n][k] (+ bias[n])
//
// Why this tile.  The scoring path's GEMMs have M = 201 B rows (201 frames per 4 s clip).  A tile of 208 rows
// (13 MFMA row blocks) placed every 201 rows holds exactly one clip, so at B = 64 the N = 1024 / 3072 / 4096
// shapes give 256 / 768 / 1024 tiles: whole rounds of the 256 CUs, where 256-row tiles leave the last round
// 59 % full.  A 208 x 256 tile also moves 35 bytes of operands into the CU per clock of matrix work against 47
// for 256 x 128, and operand delivery is what bounds the K loop.  Rows 201 .. 207 of a tile belong to the next
// clip: they are multiplied and not stored.  (Other M: tiles every 208 rows.)
//
// Eight waves, 2 (M) x 4 (N): waves 0-3 own row blocks 0-6 (7 x 4 accumulator fragments), waves 4-7 row blocks
// 7-12 (6 x 4); the two waves of a SIMD are one of each.  A K-tile is worked in two phases, the first and the
// second half of the wave's rows (4 + 3 or 3 + 3 row blocks) times all four of its column blocks:
//     [ ds_read the new fragments | issue LDS-DMA pieces | counted vmcnt | lgkmcnt(0) ] barrier
//     [ 32 or 24 MFMAs ]                                                                barrier
// Waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave multiplies while its partner reads
// and issues loads.  The LDS holds two stages of four regions each: A rows of the first halves (AL) and of the
// second halves (AH), W rows of the first (BL) and second (BH) column halves.  Phase 1 of K-tile t reads AL, BL,
// BH of its stage and issues AH and AL of K-tile t + 1 into the other stage; phase 2 reads AH and issues BL and
// BH of K-tile t + 2 into its own stage (just read in phase 1; every wave's fragment reads are complete before
// the barrier that ends its read part, so one phase of distance is enough for both wave groups): four pieces per
// wave and phase.  Loads complete in order, so one counted wait per K-tile -- at the end of phase 2's read part,
// for AL of K-tile t + 1 with the four BL / BH pieces issued after it still in flight -- covers everything that
// K-tile t + 1 reads; the data is read one barrier later.  Never vmcnt(0) in the loop.
//
// Numerics: the accumulators start from the bias and add K-tiles in ascending order with
// v_mfma_f32_16x16x32, exactly like the kernels of gemm.hip: the bits do not depend on which kernel runs.
#include <type_traits>

#include "common.h"
#include "gemm_params.h"
#include "kernels.h"

namespace slsmi {

namespace {

constexpr int W_BM = 208;
constexpr int W_BN = 256;
constexpr int AL_ROWS = 112;  // first-half rows: tile rows 0-63 (waves 0-3) and 112-159 (waves 4-7)
constexpr int AH_ROWS = 96;   // second-half rows: tile rows 64-111 and 160-207
constexpr int BH_ROWS = 128;  // per column half: 32 W rows for each of the four wave columns
constexpr int AL_OFF = 0;
constexpr int AH_OFF = AL_OFF + AL_ROWS * ROW_BYTES;   // 14336
constexpr int BL_OFF = AH_OFF + AH_ROWS * ROW_BYTES;   // 26624
constexpr int BH_OFF = BL_OFF + BH_ROWS * ROW_BYTES;   // 43008
constexpr int W_STAGE = BH_OFF + BH_ROWS * ROW_BYTES;  // 59392
constexpr int W_BIAS_OFF = 2 * W_STAGE;                // 118784: the 256 bias values of the current tile
constexpr int W_LDS = W_BIAS_OFF + 1024;               // 119808
// LayerNorm-applying variant (GemmArgs::ln_mode = LN_APPLY): the image at W_BIAS_OFF holds g of the current tile;
// behind it two images of c (the tile being multiplied and the one whose deferred epilogue is still to come) and
// (alpha, -ubar) of the workgroup's 208 rows
constexpr int W_C_OFF = W_BIAS_OFF + 1024;
constexpr int W_AFF_OFF = W_C_OFF + 2048;
constexpr int W_LDS_APPLY = W_AFF_OFF + W_BM * 8;      // 123520
constexpr int LN_NONE = GemmArgs::LN_NONE, LN_EMIT = GemmArgs::LN_EMIT, LN_APPLY = GemmArgs::LN_APPLY;

template <int N>
__device__ __forceinline__ void vm_wait() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void seg_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");  // raw: LDS-DMA loads stay in flight across it
    __builtin_amdgcn_sched_barrier(0);
}

// One wave's share of the kernel.  MIA: row blocks in the first half of its rows (4 for waves 0-3, 3 for waves 4-7).
// PER: the A operand repeats along K with period p.ka (GemmArgs::ka); a template flag so that the other
// instantiations keep their register allocation.
template <int FMT, int MODE, int MIA, int LN, bool PER>
__device__ __forceinline__ void wide_run(const GemmParams& p, const int m_step, char* lds, const int wave, const int lane) {
    constexpr int MI = MIA + 3;
    constexpr int WR = MIA == 4 ? 0 : 1;
    constexpr int RB = WR ? 112 : 0;      // first tile row of this wave
    constexpr int NAH = MIA == 4 ? 2 : 1;  // AH pieces this wave loads per K-tile (12 pieces: waves 0-3 two, 4-7 one)
    const int nal = wave < 6 ? 2 : 1;      // AL pieces (14 pieces: waves 0-5 two, 6-7 one)
    const int wc = wave & 3;
    constexpr bool out16 = MODE == GEMM_BF16 || MODE == GEMM_BF16_GELU;
    constexpr bool resid = MODE == GEMM_F32_RESID;
    // stores one wave issues in the epilogue of a full tile (the residual mode's loads drain the queue themselves)
    constexpr int E_FULL = resid ? 0 : MI * (out16 ? 2 : 4);

    const int nt = p.K / BK;  // even, >= 4 (launcher)
    const int tiles_n = p.N / W_BN, tiles_m = (p.M + m_step - 1) / m_step;
    const int total_tiles = tiles_n * tiles_m;
    int tile = blockIdx.x;  // < total_tiles (launcher)
    int tile_m, tile_n;
    // LN_APPLY: a workgroup keeps ONE tile row (its rows' LayerNorm factors are derived once, below) and walks the
    // tile columns cg, cg + cgroups, ...; the grid is tiles_m x cgroups workgroups (launcher)
    const int cgroups = LN == LN_APPLY ? p.ln_cgroups : 0;
    if (LN == LN_APPLY)
        xcd_tile_map(tile, cgroups, tiles_m, tile_m, tile_n);
    else
        xcd_tile_map(tile, tiles_n, tiles_m, tile_m, tile_n);
    const bool stat_writer = LN == LN_APPLY && tile_n == 0 && p.ln_stat != nullptr;
    int m0 = tile_m * m_step, n0 = tile_n * W_BN;
    auto has_more = [&]() { return LN == LN_APPLY ? tile_n + cgroups < tiles_n : tile + (int)gridDim.x < total_tiles; };
    auto next_tile = [&]() {
        if (LN == LN_APPLY) {
            tile_n += cgroups;
        } else {
            tile += gridDim.x;
            xcd_tile_map(tile, tiles_n, tiles_m, tile_m, tile_n);
        }
        m0 = tile_m * m_step;
        n0 = tile_n * W_BN;
    };

    // ---- staging: one wave-instruction moves 8 rows x 128 B; the LDS image is lane-linear, the bank swizzle
    // (16-byte chunk index XOR (row & 7)) is applied to the per-lane source address and again on the reads
    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;
    // per-lane source positions as 32-bit element offsets from p.W / p.A (the launcher checks that they fit)
    unsigned bl_src[2], al_src[2], ah_src[2];
    const unsigned bh_step = 32u * (unsigned)p.ldw;  // BH rows are 32 W rows after the BL rows
    auto set_sources = [&](int sm0, int sn0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int R = 16 * wave + 8 * q + srow;            // row of the BL / BH region
            const int col = ((R >> 5) << 6) + (R & 31);        // -> column of the tile (BH: + 32)
            bl_src[q] = (unsigned)(sn0 + col) * (unsigned)p.ldw + schunk * 8;
            const int Ra = 8 * (wave + 8 * q) + srow;          // row of the AL / AH region (piece wave + 8 q)
            const int trow_l = Ra < 64 ? Ra : Ra + 48;         // AL rows 64-111 are tile rows 112-159
            const int trow_h = Ra < 48 ? Ra + 64 : Ra + 112;   // AH rows 0-47 are tile rows 64-111, 48-95 are 160-207
            al_src[q] = (unsigned)min(sm0 + min(trow_l, W_BM - 1), p.M - 1) * (unsigned)p.lda + schunk * 8;
            ah_src[q] = (unsigned)min(sm0 + min(trow_h, W_BM - 1), p.M - 1) * (unsigned)p.lda + schunk * 8;
        }
    };
#ifndef SLSMI_WIDE_KNOCK
#define SLSMI_WIDE_KNOCK 0  // diagnostic builds only (wrong results): 1 no loads in the K loop, 2 no fragment reads, 3 no MFMAs
#endif
    bool in_loop = false;
    auto glds = [&](const bf16_t* base, unsigned off, char* dst) {
        if (SLSMI_WIDE_KNOCK == 1 && in_loop) return;
        __builtin_amdgcn_global_load_lds((gptr_t)(base + off), (lptr_t)dst, 16, 0, 0);
    };
    // koff: first element of the K-tile (< K)
    auto ld_bl = [&](char* sb, unsigned koff) {
        glds(p.W, bl_src[0] + koff, sb + BL_OFF + (16 * wave) * ROW_BYTES);
        glds(p.W, bl_src[1] + koff, sb + BL_OFF + (16 * wave + 8) * ROW_BYTES);
    };
    auto ld_bh = [&](char* sb, unsigned koff) {
        glds(p.W, bl_src[0] + bh_step + koff, sb + BH_OFF + (16 * wave) * ROW_BYTES);
        glds(p.W, bl_src[1] + bh_step + koff, sb + BH_OFF + (16 * wave + 8) * ROW_BYTES);
    };
    auto ld_al = [&](char* sb, unsigned koff) {
        if (PER) while (koff >= (unsigned)p.ka) koff -= (unsigned)p.ka;
        glds(p.A, al_src[0] + koff, sb + AL_OFF + (8 * wave) * ROW_BYTES);
        if (nal == 2) glds(p.A, al_src[1] + koff, sb + AL_OFF + (8 * wave + 64) * ROW_BYTES);
    };
    auto ld_ah = [&](char* sb, unsigned koff) {
        if (PER) while (koff >= (unsigned)p.ka) koff -= (unsigned)p.ka;
        glds(p.A, ah_src[0] + koff, sb + AH_OFF + (8 * wave) * ROW_BYTES);
        if (NAH == 2) glds(p.A, ah_src[1] + koff, sb + AH_OFF + (8 * wave + 64) * ROW_BYTES);
    };
    const bool has_bias = p.bias != nullptr;  // LN_APPLY: always (launcher)
    constexpr int BIAS_LOADS = LN == LN_APPLY ? 2 : 1;
    int cbuf = 0;  // LN_APPLY: the image of c that belongs to the tile being multiplied
    auto ld_bias = [&](int sn0) {  // the same 1 KiB from every wave (a wave waits for its own copy only)
        if (LN == LN_APPLY) {
            __builtin_amdgcn_global_load_lds((gptr_t)(p.ln_vec + sn0 + lane * 4), (lptr_t)(lds + W_BIAS_OFF), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(p.bias + sn0 + lane * 4), (lptr_t)(lds + W_C_OFF + cbuf * 1024), 16, 0, 0);
        } else if (has_bias) {
            __builtin_amdgcn_global_load_lds((gptr_t)(p.bias + sn0 + lane * 4), (lptr_t)(lds + W_BIAS_OFF), 16, 0, 0);
        }
    };
    // ---- fragments
    const int fr = lane & 15, fg = lane >> 4, fx = lane & 7;
    const int c0 = ((0 + fg) ^ fx) << 4;
    const int c1 = ((4 + fg) ^ fx) << 4;
    const int al_rd = AL_OFF + ((WR ? 64 : 0) + fr) * ROW_BYTES;
    const int ah_rd = AH_OFF + ((WR ? 48 : 0) + fr) * ROW_BYTES;
    const int bl_rd = BL_OFF + (wc * 32 + fr) * ROW_BYTES;
    const int bh_rd = BH_OFF + (wc * 32 + fr) * ROW_BYTES;
    u16x8 fa[2][4];   // [k-step][row block of the half being multiplied]
    u16x8 fbl[2][2];  // W, first column half: [k-step][column block]
    u16x8 fbh[2][2];
    f32x4 acc[MI][4];
    auto rd_a = [&](const char* sb, int off, auto count) {
        if (SLSMI_WIDE_KNOCK == 2 && in_loop) return;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < decltype(count)::value; ++i)
                fa[s][i] = *reinterpret_cast<const u16x8*>(sb + off + i * 16 * ROW_BYTES + (s ? c1 : c0));
    };
    auto rd_b = [&](const char* sb, int off, u16x8 (&fb)[2][2]) {
        if (SLSMI_WIDE_KNOCK == 2 && in_loop) return;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                fb[s][j] = *reinterpret_cast<const u16x8*>(sb + off + j * 16 * ROW_BYTES + (s ? c1 : c0));
    };
    // rows I0 .. I0 + NI - 1 of the wave x its four column blocks, both k-steps
    auto mma = [&](auto i0, auto ni) {
        constexpr int I0 = decltype(i0)::value, NI = decltype(ni)::value;
        if (SLSMI_WIDE_KNOCK == 3) {
            asm volatile("" ::"v"(fa[0][0]), "v"(fbl[0][0]), "v"(fbh[0][0]), "v"(fa[1][NI - 1]), "v"(fbl[1][1]), "v"(fbh[1][1]));
            return;
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < NI; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[I0 + i][j] = Op<FMT>::mfma(fbl[s][j], fa[s][i], acc[I0 + i][j]);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[I0 + i][2 + j] = Op<FMT>::mfma(fbh[s][j], fa[s][i], acc[I0 + i][2 + j]);
            }
        __builtin_amdgcn_s_setprio(0);
    };
    using I0_ = std::integral_constant<int, 0>;
    using I3_ = std::integral_constant<int, 3>;
    using IA_ = std::integral_constant<int, MIA>;

    auto init_acc = [&]() {
        if (LN == LN_APPLY) {  // -ubar[row] g[n]: the mean's share of the product, subtracted up front
            float nm[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) nm[i] = *reinterpret_cast<const float*>(lds + W_AFF_OFF + (RB + 16 * i + fr) * 8 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 gv = *reinterpret_cast<const f32x4*>(lds + W_BIAS_OFF + (wc * 64 + j * 16 + fg * 4) * 4);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] = fmul_nc(gv[r], nm[i]);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
            if (has_bias) bv = *reinterpret_cast<const f32x4*>(lds + W_BIAS_OFF + (wc * 64 + j * 16 + fg * 4) * 4);
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[i][j] = bv;
        }
    };

    // ---- epilogue of this wave's MI x 4 fragments: acc (bias included) -> activation -> C, from registers.
    // rows_valid: rows of the tile that are stored (m_step, fewer in the last tile row).  Returns whether every
    // row block of the wave stored (then exactly E_FULL store instructions were issued).
    auto epilogue = [&](int em0, int en0, int ebuf) -> bool {
        const int rows_valid = min(m_step, p.M - em0);
        if (MODE == GEMM_BF16_GELU && LN != LN_APPLY) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = gelu_erf4_out16<FMT>(acc[i][j]);
        }
        if (MODE == GEMM_F32_RELU) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] = fmaxf(acc[i][j][r], 0.f);
        }
        if (out16) {
            // lane (fr, fg) holds columns 16 j + 4 fg .. + 3 of row fr: 8 bytes.  Lanes fg and fg ^ 1 swap so that
            // even fg gets columns 8 (fg >> 1) .. + 7 of sub-tile 2 jp and odd fg the same columns of sub-tile
            // 2 jp + 1: 16 bytes per lane, 64 contiguous bytes per row.
            bf16_t* cbase = reinterpret_cast<bf16_t*>(p.C) + en0 + wc * 64 + (fg & 1) * 16 + (fg >> 1) * 8;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                if (RB + 16 * i >= rows_valid) continue;  // wave-uniform
                const int trow = RB + 16 * i + fr;
                float al = 0.f;
                if (LN == LN_APPLY) al = *reinterpret_cast<const float*>(lds + W_AFF_OFF + trow * 8);
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    f32x4 v0 = acc[i][2 * jp], v1 = acc[i][2 * jp + 1];
                    if (LN == LN_APPLY) {  // alpha[row] acc + c[n], then the activation, one row block at a time
                        const char* cimg = lds + W_C_OFF + ebuf * 1024 + (wc * 64 + jp * 32 + fg * 4) * 4;
                        const f32x4 c0v = *reinterpret_cast<const f32x4*>(cimg), c1v = *reinterpret_cast<const f32x4*>(cimg + 64);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            v0[r] = fma_nc(v0[r], al, c0v[r]);
                            v1[r] = fma_nc(v1[r], al, c1v[r]);
                        }
                        if (MODE == GEMM_BF16_GELU) {
                            v0 = gelu_erf4_out16<FMT>(v0);
                            v1 = gelu_erf4_out16<FMT>(v1);
                        }
                    }
                    // (the GELU above has limited its results to the format's range already)
                    constexpr bool sat = MODE != GEMM_BF16_GELU;
                    const uint32_t e_lo = sat ? Op<FMT>::pack2(v0[0], v0[1]) : Op<FMT>::pack2_unit(v0[0], v0[1]);
                    const uint32_t e_hi = sat ? Op<FMT>::pack2(v0[2], v0[3]) : Op<FMT>::pack2_unit(v0[2], v0[3]);
                    const uint32_t o_lo = sat ? Op<FMT>::pack2(v1[0], v1[1]) : Op<FMT>::pack2_unit(v1[0], v1[1]);
                    const uint32_t o_hi = sat ? Op<FMT>::pack2(v1[2], v1[3]) : Op<FMT>::pack2_unit(v1[2], v1[3]);
                    const auto lo = __builtin_amdgcn_permlane16_swap(e_lo, o_lo, false, false);
                    const auto hi = __builtin_amdgcn_permlane16_swap(e_hi, o_hi, false, false);
                    const u32x4 out = u32x4{lo[0], hi[0], lo[1], hi[1]};
                    if (trow < rows_valid) *reinterpret_cast<u32x4*>(cbase + (long)(em0 + trow) * p.ldc + jp * 32) = out;
                }
            }
        } else {
            float* cbase = reinterpret_cast<float*>(p.C) + en0 + wc * 64 + fg * 4;
            const float* rbase = p.R + en0 + wc * 64 + fg * 4;
            f32x4 dw[4];
            if (resid && p.dot_w) {
#pragma unroll
                for (int j = 0; j < 4; ++j) dw[j] = *reinterpret_cast<const f32x4*>(p.dot_w + en0 + wc * 64 + j * 16 + fg * 4);
            }
            f32x4 gm[4];
            if (LN == LN_EMIT) {
#pragma unroll
                for (int j = 0; j < 4; ++j) gm[j] = *reinterpret_cast<const f32x4*>(p.ln_vec + en0 + wc * 64 + j * 16 + fg * 4);
            }
            // The old values (and the rows' (m, s)) of row block i + PF are requested before row block i is
            // worked on, so a wave keeps PF + 1 row blocks of loads in flight and never waits for its own stores:
            // taken one row block at a time, each block costs a full memory round trip behind the previous
            // block's stores, and the whole chip runs its epilogues at a fraction of the memory bandwidth.
            constexpr int PF = 2;
            f32x4 old[MI][4];
            f32x2 prx[MI];
            auto row_of = [&](int i) -> int {
                const int trow = RB + 16 * i + fr;
                return em0 + (trow < rows_valid ? trow : RB + 16 * i);  // lanes past the end read a row that exists
            };
            auto fetch = [&](int i) {  // unconditional (straight-line code keeps the waits counted): a row block
                                       // past the end reads the matrix's last row
                const int row_i = row_of(i);
                const long row = row_i < p.M ? row_i : p.M - 1;
                if (LN == LN_EMIT) prx[i] = *reinterpret_cast<const f32x2*>(p.ln_proxy + row * 2);
                if (resid) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) old[i][j] = *reinterpret_cast<const f32x4*>(rbase + row * p.ldc + j * 16);
                }
            };
            if (resid || LN == LN_EMIT) {
#pragma unroll
                for (int i = 0; i < PF && i < MI; ++i) fetch(i);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                if (resid || LN == LN_EMIT) {
                    if (i + PF < MI) fetch(i + PF);
                    __builtin_amdgcn_sched_barrier(0);  // the requests stay ahead of the work below
                } else if (RB + 16 * i >= rows_valid) {
                    continue;  // wave-uniform
                }
                // (with loads in flight a row block past the end is not skipped: it runs with no live lane,
                // so that every path consumes what it requested and the compiler can count its waits)
                const int trow = RB + 16 * i + fr;
                const bool live = trow < rows_valid;
                const long row = row_of(i);
                f32x2 pr = {0.f, 1.f};
                if (LN == LN_EMIT) pr = prx[i];
                if (resid) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] += old[i][j];
                }
                if (live) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(cbase + row * p.ldc + j * 16) = acc[i][j];
                }
                if (LN == LN_EMIT) {
                    // u = (x - m) s; the consumer's operand u gamma in 16 bits (16-byte stores after the lane-pair
                    // exchange of the 16-bit epilogue above); this wave's 64-column sums of u and u^2
                    bf16_t* abase = p.ln_a + en0 + wc * 64 + (fg & 1) * 16 + (fg >> 1) * 8;
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int jp = 0; jp < 2; ++jp) {
                        f32x4 u0, u1;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            u0[r] = ln_fold_u(acc[i][2 * jp][r], pr[0], pr[1]);
                            ln_fold_acc(u0[r], s1, s2);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            u1[r] = ln_fold_u(acc[i][2 * jp + 1][r], pr[0], pr[1]);
                            ln_fold_acc(u1[r], s1, s2);
                        }
                        f32x4 v0, v1;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            v0[r] = fmul_nc(u0[r], gm[2 * jp][r]);
                            v1[r] = fmul_nc(u1[r], gm[2 * jp + 1][r]);
                        }
                        const uint32_t e_lo = Op<FMT>::pack2(v0[0], v0[1]), e_hi = Op<FMT>::pack2(v0[2], v0[3]);
                        const uint32_t o_lo = Op<FMT>::pack2(v1[0], v1[1]), o_hi = Op<FMT>::pack2(v1[2], v1[3]);
                        const auto lo = __builtin_amdgcn_permlane16_swap(e_lo, o_lo, false, false);
                        const auto hi = __builtin_amdgcn_permlane16_swap(e_hi, o_hi, false, false);
                        const u32x4 out = u32x4{lo[0], hi[0], lo[1], hi[1]};
                        if (live) *reinterpret_cast<u32x4*>(abase + row * p.N + jp * 32) = out;
                    }
                    s1 = fadd_nc(s1, __shfl_xor(s1, 16, 64));   // (p0 + p1) + (p2 + p3) over the four column quarters
                    s1 = fadd_nc(s1, __shfl_xor(s1, 32, 64));
                    s2 = fadd_nc(s2, __shfl_xor(s2, 16, 64));
                    s2 = fadd_nc(s2, __shfl_xor(s2, 32, 64));
                    if (fg == 0 && live)
                        *reinterpret_cast<f32x2*>(p.ln_slots + ((long)((en0 + wc * 64) >> 6) * p.M + row) * 2) = f32x2{s1, s2};
                }
                if (resid && p.dot_w) {  // row dots of the updated C with dot_w over this wave's 64 columns
                    float d = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) d = fmaf(acc[i][j][r], dw[j][r], d);
                    d += __shfl_xor(d, 16, 64);
                    d += __shfl_xor(d, 32, 64);
                    if (fg == 0 && live) p.dot_out[(long)((en0 + wc * 64) >> 6) * p.M + row] = d;
                }
            }
        }
        return RB + 16 * (MI - 1) < rows_valid;
    };

    // ---- start of the stream: K-tile 0 whole, BL / BH of K-tile 1 (what the steady state has issued by then)
    char* const sb0 = lds;
    char* const sb1 = lds + W_STAGE;
    set_sources(m0, n0);
    ld_bias(n0);
    ld_bl(sb0, 0);
    ld_al(sb0, 0);
    ld_bh(sb0, 0);
    ld_ah(sb0, 0);
    ld_bl(sb1, BK);
    ld_bh(sb1, BK);
    if (LN == LN_APPLY) {
        // LayerNorm factors of the workgroup's 208 rows from the producer's 16 slice sums per row: two lanes per
        // row, eight slices each, added in slice order; both lanes then hold the same totals
        const int t2 = wave * 64 + lane;
        const int row = t2 >> 1, half = t2 & 1;
        if (row < W_BM) {  // wave-uniform (416 = 6.5 waves: lanes come in pairs, so the shuffle partner is active)
            const long gr = min(m0 + row, p.M - 1);
            const float* sl = p.ln_slots + ((long)(half * 8) * p.M + gr) * 2;  // slices 8 half .. 8 half + 7 of this row
            f32x2 q[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) q[e] = *reinterpret_cast<const f32x2*>(sl + (long)e * p.M * 2);
            const f32x2 pr = *reinterpret_cast<const f32x2*>(p.ln_proxy + gr * 2);
            float s1 = q[0][0], s2 = q[0][1];
#pragma unroll
            for (int e = 1; e < 8; ++e) {
                s1 = fadd_nc(s1, q[e][0]);
                s2 = fadd_nc(s2, q[e][1]);
            }
            const float o1 = __shfl_xor(s1, 1, 64), o2 = __shfl_xor(s2, 1, 64);
            const float S1 = half ? fadd_nc(o1, s1) : fadd_nc(s1, o1);   // (slices 0-7) + (slices 8-15)
            const float S2 = half ? fadd_nc(o2, s2) : fadd_nc(s2, o2);
            float alpha, nubar, mean, rstd;
            ln_fold_factors(S1, S2, pr[0], pr[1], p.ln_eps, alpha, nubar, mean, rstd);
            if (half == 0) {
                *reinterpret_cast<f32x2*>(lds + W_AFF_OFF + row * 8) = f32x2{alpha, nubar};
                if (stat_writer && row < m_step && m0 + row < p.M) *reinterpret_cast<f32x2*>(p.ln_stat + gr * 2) = f32x2{mean, rstd};
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    vm_wait<0>();
    seg_barrier();
    if (WR) seg_barrier();  // waves 4-7 run one barrier behind
    init_acc();

    if (SLSMI_WIDE_KNOCK == 2) {  // something defined in the fragment registers
        rd_b(sb0, bl_rd, fbl);
        rd_b(sb0, bh_rd, fbh);
        rd_a(sb0, al_rd, std::integral_constant<int, 4>());
    }
    in_loop = true;
    // end of a phase's read part: this wave's fragment reads are complete (the regions it read may be refilled
    // by anyone from the next phase on)
    auto end_reads = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        seg_barrier();
    };
    // Waves 0-3 store a finished tile at the start of the next tile's first multiply part, waves 4-7 right after
    // their last one: both groups then run their epilogues between the same two barriers, beside each other,
    // instead of one after the other with the partner waiting.
    // (The residual mode keeps its epilogue right after the last multiply part: the old values it loads do not fit
    // beside live fragment registers.)
    // (Nor does the LayerNorm-applying variant: its epilogue's extra values would push the K loop's load addresses
    // out of the register file.)
    constexpr bool defer = WR == 0 && !resid && LN != LN_APPLY;
    bool pending = false;   // waves 0-3: the previous tile's accumulators are still to be stored
    int prev_m0 = 0, prev_n0 = 0, prev_buf = 0;
    for (;;) {
        const bool more = has_more();
        const int cur_m0 = m0, cur_n0 = n0, cur_buf = cbuf;
        for (int t = 0; t < nt; t += 2) {
            // ================= K-tile t: stage 0 =================
            {
                const bool sw = t == nt - 2;  // the loads issued from its second phase on belong to the next tile
                // phase 1: first half of the rows
                rd_b(sb0, bl_rd, fbl);
                rd_b(sb0, bh_rd, fbh);
                rd_a(sb0, al_rd, IA_());
                ld_ah(sb1, (unsigned)(t + 1) * BK);
                ld_al(sb1, (unsigned)(t + 1) * BK);
                end_reads();
                bool stores_after_al = false;  // a whole epilogue's stores sit between AL and the wait below
                if (defer && pending) {  // t == 0
                    stores_after_al = epilogue(prev_m0, prev_n0, prev_buf) && E_FULL > 0;
                    init_acc();
                    pending = false;
                    __builtin_amdgcn_sched_barrier(0);
                }
                mma(I0_(), IA_());
                seg_barrier();
                // phase 2: second half of the rows (W fragments already in registers)
                rd_a(sb0, ah_rd, I3_());
                bool issue = true;
                unsigned k2 = (unsigned)(t + 2) * BK;
                bool with_bias = false;
                if (sw) {
                    issue = more;
                    k2 = 0;
                    if (more) {
                        next_tile();
                        set_sources(m0, n0);
                        cbuf ^= 1;
                        ld_bias(n0);
                        with_bias = has_bias;
                    }
                }
                if (issue) {
                    ld_bl(sb0, k2);
                    ld_bh(sb0, k2);
                }
                // AL of K-tile t + 1 (and all older loads) has landed; what was issued after it may be pending
                if (!issue) vm_wait<0>();
                else if (with_bias) vm_wait<4 + BIAS_LOADS>();
                else if (stores_after_al) vm_wait<4 + E_FULL>();
                else vm_wait<4>();
                end_reads();
                mma(IA_(), I3_());
                seg_barrier();
            }
            // ================= K-tile t + 1: stage 1 =================
            {
                const bool last = t + 2 == nt;
                const bool issue = !last || more;  // the last K-tile loads K-tiles 0 and 1 of the next tile
                rd_b(sb1, bl_rd, fbl);
                rd_b(sb1, bh_rd, fbh);
                rd_a(sb1, al_rd, IA_());
                if (issue) {
                    ld_ah(sb0, last ? 0u : (unsigned)(t + 2) * BK);
                    ld_al(sb0, last ? 0u : (unsigned)(t + 2) * BK);
                }
                end_reads();
                mma(I0_(), IA_());
                seg_barrier();
                rd_a(sb1, ah_rd, I3_());
                if (issue) {
                    const unsigned k2 = last ? (unsigned)BK : (unsigned)(t + 3) * BK;
                    ld_bl(sb1, k2);
                    ld_bh(sb1, k2);
                    vm_wait<4>();
                } else {
                    vm_wait<0>();
                }
                end_reads();
                mma(IA_(), I3_());
                seg_barrier();
            }
        }
        if (defer && more) {
            pending = true;
            prev_m0 = cur_m0;
            prev_n0 = cur_n0;
            prev_buf = cur_buf;
        } else {
            epilogue(cur_m0, cur_n0, cur_buf);
            if (more) init_acc();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!more) break;
    }
    if (!WR) seg_barrier();  // the last barrier of waves 4-7
}

template <int FMT, int MODE, int LN, bool PER = false>
__global__ __launch_bounds__(512, 2) void gemm16_wide_kernel(GemmParams p, int m_step) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // 0..7
    if (wave < 4)
        wide_run<FMT, MODE, 4, LN, PER>(p, m_step, lds, wave, lane);
    else
        wide_run<FMT, MODE, 3, LN, PER>(p, m_step, lds, wave, lane);
}

template <int FMT, int MODE, int LN = LN_NONE, bool PER = false>
hipError_t launch_wide_mode(const GemmParams& p, int m_step, dim3 grid, hipStream_t s) {
    static LdsAttrState lds_attr;
    constexpr int LDS_BYTES = LN == LN_APPLY ? W_LDS_APPLY : W_LDS;
    const hipError_t attr = ensure_dynamic_lds(reinterpret_cast<const void*>(&gemm16_wide_kernel<FMT, MODE, LN, PER>), LDS_BYTES, lds_attr);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL((gemm16_wide_kernel<FMT, MODE, LN, PER>), grid, dim3(512), LDS_BYTES, s, p, m_step);
    return hipGetLastError();
}

template <int FMT>
hipError_t launch_wide_fmt(const GemmParams& p, int m_step, dim3 grid, hipStream_t s) {
    if (p.ln_mode == LN_EMIT) return launch_wide_mode<FMT, GEMM_F32_RESID, LN_EMIT>(p, m_step, grid, s);
    if (p.ln_mode == LN_APPLY)
        return p.mode == GEMM_BF16 ? launch_wide_mode<FMT, GEMM_BF16, LN_APPLY>(p, m_step, grid, s)
                                   : launch_wide_mode<FMT, GEMM_BF16_GELU, LN_APPLY>(p, m_step, grid, s);
    if (p.ka != p.K) {  // periodic A: the two modes the conv layers use
        if (p.mode == GEMM_BF16) return launch_wide_mode<FMT, GEMM_BF16, LN_NONE, true>(p, m_step, grid, s);
        if (p.mode == GEMM_F32) return launch_wide_mode<FMT, GEMM_F32, LN_NONE, true>(p, m_step, grid, s);
        return hipErrorInvalidValue;
    }
    switch (p.mode) {
        case GEMM_BF16: return launch_wide_mode<FMT, GEMM_BF16>(p, m_step, grid, s);
        case GEMM_BF16_GELU: return launch_wide_mode<FMT, GEMM_BF16_GELU>(p, m_step, grid, s);
        case GEMM_F32: return launch_wide_mode<FMT, GEMM_F32>(p, m_step, grid, s);
        case GEMM_F32_RESID: return launch_wide_mode<FMT, GEMM_F32_RESID>(p, m_step, grid, s);
        case GEMM_F32_RELU: return launch_wide_mode<FMT, GEMM_F32_RELU>(p, m_step, grid, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace

bool gemm_wide_supported(const GemmParams& p) {
    const bool mode_ok = p.mode == GEMM_BF16 || p.mode == GEMM_BF16_GELU || p.mode == GEMM_F32 ||
                         p.mode == GEMM_F32_RESID || p.mode == GEMM_F32_RELU;
    const bool offsets_fit = p.lda > 0 && p.ldw > 0 && (long)p.M * p.lda + p.K < (1L << 32) && (long)p.N * p.ldw + p.K < (1L << 32) &&
                             p.ka >= BK && p.ka % BK == 0 &&
                             (p.ka == p.K || (p.ln_mode == LN_NONE && (p.mode == GEMM_BF16 || p.mode == GEMM_F32)));
    bool ln_ok = p.ln_mode == LN_NONE;
    if (p.ln_mode == LN_EMIT)
        ln_ok = p.mode == GEMM_F32_RESID && p.N == 1024 && p.ln_proxy && p.ln_vec && p.ln_a && p.ln_slots;
    if (p.ln_mode == LN_APPLY)
        ln_ok = (p.mode == GEMM_BF16 || p.mode == GEMM_BF16_GELU) && p.K == 1024 && p.bias && p.ln_proxy && p.ln_vec &&
                p.ln_slots;
    return mode_ok && ln_ok && offsets_fit && p.nz == 1 && p.splitk == 1 && p.grp_in == 0 && p.N % W_BN == 0 &&
           p.K % (2 * BK) == 0 && p.K >= 4 * BK && p.ldc % 8 == 0 && p.M >= 1 &&
           (p.dot_w == nullptr || p.mode == GEMM_F32_RESID);
}

int gemm_wide_row_step(int M) {
    // a step that divides M keeps tiles aligned with the clips of a batch (201 frames per 4 s clip)
    for (int step = W_BM; step >= 193; --step)
        if (M % step == 0) return step;
    return W_BM;
}

long gemm_wide_tiles(int M, int N, int m_step) { return (long)((M + m_step - 1) / m_step) * (N / W_BN); }

static std::atomic<int> g_wide_grid_cap{0};  // diagnostic: at most this many workgroups per launch (0: one per CU)

int gemm_wide_apply_cgroups(int M, int N, int m_step, int n_cu) {
    const int tiles_m = (M + m_step - 1) / m_step, tiles_n = N / W_BN;
    for (int c = tiles_n; c >= 1; --c)
        if (tiles_n % c == 0 && (long)tiles_m * c <= n_cu) return c;
    return 0;
}

hipError_t launch_gemm_wide(const GemmParams& p0, int m_step, int fmt, int n_cu, hipStream_t s) {
    if (!gemm_wide_supported(p0) || m_step < 1 || m_step > W_BM || n_cu <= 0) return hipErrorInvalidValue;
    GemmParams p = p0;
    const long tiles = gemm_wide_tiles(p.M, p.N, m_step);
    const int cap = g_wide_grid_cap.load(std::memory_order_relaxed);
    if (cap > 0 && cap < n_cu) n_cu = cap;
    dim3 grid((unsigned)(tiles < n_cu ? tiles : n_cu), 1, 1);
    if (p.ln_mode == LN_APPLY) {
        // every workgroup keeps one tile row: the largest divisor c of the tile columns with tiles_m x c <= CUs
        p.ln_cgroups = gemm_wide_apply_cgroups(p.M, p.N, m_step, n_cu);
        if (p.ln_cgroups < 1) return hipErrorInvalidValue;  // more tile rows than CUs
        grid.x = (unsigned)(((p.M + m_step - 1) / m_step) * p.ln_cgroups);
    }
    return fmt == FMT_BF16 ? launch_wide_fmt<FMT_BF16>(p, m_step, grid, s) : launch_wide_fmt<FMT_F16>(p, m_step, grid, s);
}

}  // namespace slsmi

// Diagnostic (tools/dual_stream_probe.py): caps the workgroups of every later 208 x 256 launch of this process.
extern "C" int slsmi_debug_wide_grid_cap(int cap) {
    slsmi::g_wide_grid_cap.store(cap < 0 ? 0 : cap, std::memory_order_relaxed);
    return 0;
}
