// Positional convolution of the encoder (wav2vec/wav2vec2.py:915-917): x += GELU(SamePad(Conv1d(k = 128,
// groups = 16)(x)) + bias), with the input slab of a (208 frames of a clip, group) RESIDENT in LDS.
//
// Why a kernel of its own.  Per group the conv is a Toeplitz product, out[t][co] = sum over tap < 128, ci < 64 of
// xg[t + tap][ci] * w[co][tap][ci] (xg = the group's 64 channels, zero rows around the frames): as a batched GEMM
// (N = 64, K = 8192, lda = 64) every 64-deep K-tile is one tap, and its A tile is the previous tap's moved down by
// one row.  The GEMM kernel stages that tile again for every tap -- 16 KB of LDS-DMA and two barriers per 64 MFMAs
// of a workgroup, matrix pipes busy 43 % of the time.  Here the slab's 336 rows x 64 channels (42 KB) are staged
// ONCE per (clip, group) and every tap reads its A fragments from the same image at a row offset; only the weights
// (8 KB per tap) stream through LDS.
//
// Workgroup = (pair of segments, group), a segment being 208 consecutive frames of a clip (a clip of up to 208 frames
// is one segment, a longer one several, each with the 336 slab rows its frames read): 2 x 208 output rows x 64
// channels, eight waves as 4 (M) x 2 (N): a wave owns 7 or 6 row blocks of one segment x 32 channels, so a tap's two weight fragments serve 7 row blocks and the tap
// costs the wave 18 or 16 fragment reads for 28 or 24 MFMAs.  A tap's second 32-deep step is read while its first
// multiplies and the next tap's first step while its second does; weights run seven taps ahead in eight 8 KB stages,
// one piece per wave and tap; one barrier per tap, no vmcnt(0) before the last taps.
//
// Numerics: accumulators start from the bias, taps ascend, the two 32-deep steps of a tap in order -- the K order
// of the GEMM form (k = 64 tap + ci), the same v_mfma_f32_16x16x32 per step, the same epilogue operations: the
// two forms give the same bits.
#include <type_traits>

#include "common.h"
#include "gemm_params.h"
#include "kernels.h"

namespace slsmi {

namespace {

constexpr int PC_ROWS = 208;                      // output rows per clip (13 row blocks)
constexpr int PC_SLAB = 336;                      // slab rows staged per clip: 208 + 127 rounded up to whole pieces
constexpr int PC_SLAB_BYTES = PC_SLAB * ROW_BYTES;    // 43008
constexpr int PC_W_OFF = 2 * PC_SLAB_BYTES;       // 86016: the weight stages behind the two slabs
constexpr int PC_NS = 8;                          // weight stages: requests run PC_NS - 1 taps ahead
constexpr int PC_W_STAGE = 64 * ROW_BYTES;        // 8192: 64 output channels x 64 input channels of one tap
constexpr int PC_LDS = PC_W_OFF + PC_NS * PC_W_STAGE; // 151552
constexpr int PC_TAPS = 128;

#ifndef SLSMI_PC_KNOCK
#define SLSMI_PC_KNOCK 0  // diagnostic builds only (wrong results): 1 no barrier in the loop, 2 no weight loads in the loop, 3 no MFMAs, 4 no fragment reads in the loop
#endif

// One wave's share.  MI: its row blocks (7: rows 0-111 of its clip, 6: rows 112-207).
template <int FMT, int MI>
__device__ __forceinline__ void posconv_run(const bf16_t* __restrict__ xg, const bf16_t* __restrict__ w,
                                            const float* __restrict__ bias, float* __restrict__ x, int B, int T, int R,
                                            int segs, char* lds, int wave, int lane, int pair, int g) {
    constexpr int RB = MI == 7 ? 0 : 112;
    // waves w and w + 4 share a SIMD: one with 7 row blocks, one with 6
    const int wn = wave & 1;
    const int clip_l = (wave >> 1) & 1;
    const int nseg = B * segs;                             // segment sg: frames (sg % segs) * 208 .. of clip sg / segs
    const int sg = min(2 * pair + clip_l, nseg - 1);       // an odd count: the last pair's second half repeats the first
    const int clip = sg / segs;
    const int t_off = (sg - clip * segs) * PC_ROWS;
    const bool stores = 2 * pair + clip_l < nseg;
    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;                  // (row & 7 == srow: pieces start at multiples of 8 rows)

    // ---- both slabs: 84 pieces of 8 rows x 128 B, dealt over the waves; rows past the slab's end repeat its last
    // row (they reach only output rows >= T, which are not stored)
#pragma unroll 1
    for (int pc = wave; pc < 2 * (PC_SLAB / 8); pc += 8) {
        const int c_l = pc >= PC_SLAB / 8 ? 1 : 0;
        const int prow = (pc - c_l * (PC_SLAB / 8)) * 8 + srow;
        const int s_g = min(2 * pair + c_l, nseg - 1);
        const int c_g = s_g / segs;
        const int r_g = (s_g - c_g * segs) * PC_ROWS + prow;   // slab row of the clip
        const bf16_t* src = xg + (((long)c_g * 16 + g) * R + min(r_g, R - 1)) * 64 + schunk * 8;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + c_l * PC_SLAB_BYTES + (pc - c_l * (PC_SLAB / 8)) * 1024), 16, 0, 0);
    }
    // ---- weights: piece `wave` (8 output channels x 128 B) of a tap
    const bf16_t* w_src = w + ((long)g * 64 + wave * 8 + srow) * (PC_TAPS * 64) + schunk * 8;
    auto ld_w = [&](int tap) {
        __builtin_amdgcn_global_load_lds((gptr_t)(w_src + tap * 64), (lptr_t)(lds + PC_W_OFF + (tap & (PC_NS - 1)) * PC_W_STAGE + wave * 1024),
                                         16, 0, 0);
    };
#pragma unroll
    for (int tap = 0; tap < PC_NS - 1; ++tap) ld_w(tap);

    // ---- fragments: lane (fr, fg) reads 16 bytes (chunk 4 s + fg) of row fr of a 16-row block
    const int fr = lane & 15, fg = lane >> 4;
    const char* slab = lds + clip_l * PC_SLAB_BYTES;
    const int a_row0 = RB + fr;                            // + 16 i + tap
    const int w_rd = PC_W_OFF + (wn * 32 + fr) * ROW_BYTES;
    const int wx = fr & 7;
    const int wc0 = ((0 + fg) ^ wx) << 4, wc1 = ((4 + fg) ^ wx) << 4;
    // the fragments of one 32-deep step (s = 0, 1) of a tap
    auto rd_frags = [&](int tap, int s, u16x8 (&fa)[MI], u16x8 (&fb)[2]) {
        const char* ws = lds + w_rd + (tap & (PC_NS - 1)) * PC_W_STAGE + (s ? wc1 : wc0);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const u16x8*>(ws + j * 16 * ROW_BYTES);
        const int row = a_row0 + tap;
        const int ax = row & 7;                            // rows 16 apart share it
        const char* as = slab + row * ROW_BYTES + ((((s ? 4 : 0) + fg) ^ ax) << 4);
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const u16x8*>(as + i * 16 * ROW_BYTES);
    };
    f32x4 acc[MI][2];
    {
        const float* bp = bias + g * 64 + wn * 32 + fg * 4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bp + j * 16);
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[i][j] = bv;
        }
    }
    auto mma = [&](const u16x8 (&fa)[MI], const u16x8 (&fb)[2]) {
        if (SLSMI_PC_KNOCK == 3) {
            asm volatile("" ::"v"(fa[0]), "v"(fb[0]), "v"(fa[MI - 1]), "v"(fb[1]));
            return;
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = Op<FMT>::mfma(fb[j], fa[i], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
    };
    bool in_loop = false;
    auto barrier = [&]() {
        if (SLSMI_PC_KNOCK == 1 && in_loop) return;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");  // raw: the weight loads stay in flight across it
        __builtin_amdgcn_sched_barrier(0);
    };

    u16x8 fa0[MI], fb0[2], fa1[MI], fb1[2];   // first steps of the tap being multiplied and of the next one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    barrier();
    rd_frags(0, 0, fa0, fb0);
    if (SLSMI_PC_KNOCK == 4) rd_frags(1, 0, fa1, fb1);
    in_loop = true;
    // One tap: read its second step's fragments, request the weights of tap + 7 into the stage that tap - 1 used
    // (every wave's reads of it ended before the barrier that closed tap - 1), multiply the first step, read the
    // next tap's first step, multiply the second step.  The wait leaves the five youngest requests in flight:
    // this wave's piece of tap + 2 has landed, and after the barrier everyone's has.
    auto step = [&](int tap, const u16x8 (&fa)[MI], const u16x8 (&fb)[2], u16x8 (&na)[MI], u16x8 (&nb)[2]) {
        u16x8 fa2[MI], fb2[2];
        if (SLSMI_PC_KNOCK == 4) {
#pragma unroll
            for (int i = 0; i < MI; ++i) fa2[i] = fa[i];
            fb2[0] = fb[0];
            fb2[1] = fb[1];
        } else {
            rd_frags(tap, 1, fa2, fb2);
        }
        const bool more_w = tap + PC_NS - 1 < PC_TAPS;
        if (more_w && SLSMI_PC_KNOCK != 2) ld_w(tap + PC_NS - 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(fa, fb);
        __builtin_amdgcn_sched_barrier(0);
        if (tap + 1 < PC_TAPS && SLSMI_PC_KNOCK != 4) rd_frags(tap + 1, 0, na, nb);
        __builtin_amdgcn_sched_barrier(0);
        mma(fa2, fb2);
        static_assert(PC_NS == 8, "the counted wait below is for seven taps of lookahead");
        if (more_w) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        barrier();
    };
#pragma unroll 1
    for (int tap = 0; tap < PC_TAPS; tap += 2) {
        step(tap, fa0, fb0, fa1, fb1);
        step(tap + 1, fa1, fb1, fa0, fb0);
    }

    // ---- epilogue: x[t][64 g + n] += gelu(acc), rows t < T
    if (!stores) return;
    float* xb = x + (long)clip * T * 1024 + g * 64 + wn * 32 + fg * 4;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int t = t_off + RB + 16 * i + fr;
        if (t >= T) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float* c = xb + (long)t * 1024 + j * 16;
            f32x4 v = gelu_erf4(acc[i][j]);
            v += *reinterpret_cast<const f32x4*>(c);
            *reinterpret_cast<f32x4*>(c) = v;
        }
    }
}

template <int FMT>
__global__ __launch_bounds__(512, 2) void posconv_slab_kernel(const bf16_t* __restrict__ xg, const bf16_t* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ x, int B, int T,
                                                              int R, int segs) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // workgroups b and b + 8 share an XCD: each XCD serves two groups, whose weights (1 MB each) stay in its L2
    const int bid = blockIdx.x;
    const int g = (bid & 7) * 2 + ((bid >> 3) & 1);
    const int pair = bid >> 4;
    if (wave >> 2)
        posconv_run<FMT, 6>(xg, w, bias, x, B, T, R, segs, lds, wave, lane, pair, g);
    else
        posconv_run<FMT, 7>(xg, w, bias, x, B, T, R, segs, lds, wave, lane, pair, g);
}

}  // namespace

bool posconv_slab_supported(int T, int slab_rows) { return T >= 1 && slab_rows >= T + PC_TAPS - 1 && (slab_rows % 8) == 0; }

hipError_t launch_posconv_slab(const bf16_t* xg, const bf16_t* w, const float* bias, float* x, int B, int T, int slab_rows,
                               int fmt, hipStream_t s) {
    if (B < 1 || !posconv_slab_supported(T, slab_rows)) return hipErrorInvalidValue;
    const int segs = (T + PC_ROWS - 1) / PC_ROWS;
    const long nseg = (long)B * segs;
    if (nseg > (1L << 26)) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(((nseg + 1) / 2) * 16));
    hipError_t rc = hipSuccess;
    SLSMI_FMT_DISPATCH(fmt, {
        static LdsAttrState lds_attr;
        rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&posconv_slab_kernel<F>), PC_LDS, lds_attr);
        if (rc == hipSuccess) {
            hipLaunchKernelGGL((posconv_slab_kernel<F>), grid, dim3(512), PC_LDS, s, xg, w, bias, x, B, T, slab_rows, segs);
            rc = hipGetLastError();
        }
    });
    return rc;
}

}  // namespace slsmi
