// Self-attention core for one (utterance, head) per workgroup: 16 heads x 64.
//
// The whole K and V of a head (T <= 512 frames) sit in LDS; there is no online
// softmax.  Scores are computed TRANSPOSED (S^T = K Q^T, keys on the MFMA row
// index) so that a lane ends up holding, for one query column, four consecutive
// keys per 16-key tile.  Two such tiles packed to 16 bits are exactly the B-operand
// fragment of the second product O^T = V^T P^T when the k index inside each
// 32-key step is enumerated as (lane group g, j) -> key 4g + j (j < 4) or
// 16 + 4g + (j - 4): P never leaves registers.  V stays row-major in LDS
// ([key][64], 160-byte pitch) and the matching A-operand fragment (V^T) comes from
// two `ds_read_b64_tr_b16` transposed reads; with that pitch the eight key rows
// a 32-lane half touches fall on eight disjoint 8-bank windows.
// The softmax runs in the base-2 domain: p = 2^(scale * (s - max s)), one fused
// multiply-add and one v_exp_f32 per score; scale = log2(e) gives the natural softmax.
// The head-averaged attention weights the reference also returns
// (wav2vec/wav2vec2.py:1046-1052, unused by every head) are not produced.
#include "common.h"
#include "kernels.h"

#ifndef SLSMI_ATT_KNOCK
#define SLSMI_ATT_KNOCK 0  // diagnostic builds only (wrong results): 1 no K / V loads, 2 staging and stores only
#endif

namespace slsmi {

namespace {

constexpr int MAX_KS = 16;        // 32-key steps a workgroup can hold: T <= 512 (K 64 KiB + V 80 KiB of LDS)
constexpr int V_PITCH = 160;      // bytes per V row (128 used)

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

// KS = ceil(T / 32) is a template parameter so that the whole kernel is straight-line
// code: the LDS fragment reads of many tiles are in flight together instead of one
// read-wait-multiply group per tile.
//
// VARLEN: utterance b attends to its first nframes[b] keys only (the key-padding mask the
// reference's encoder passes to MultiheadAttention, wav2vec/wav2vec2.py:1050, for frames past the
// utterance's own length, :567-586); T stays the row pitch of the batch.  Rows past nframes[b]
// hold finite values of the padded audio, so masking their scores is all that is needed.
//
// Waves per workgroup: 8 up to T = 448 (up to T = 256 two workgroups of 64 KiB fit a CU -> four waves per
// SIMD, beyond that one workgroup -> two; one wave's softmax -- a quarter-rate v_exp_f32 per score --
// overlaps other waves' LDS fragment reads and MFMAs), 4 beyond (the score registers of a long row need the room).
template <int KS>
struct AttnShape {
    static constexpr int NW = KS <= 14 ? 8 : 4;
    static constexpr int MIN_WAVES_PER_SIMD = KS <= 8 ? 4 : (KS <= 14 ? 2 : 1);  // what LDS and registers allow
};

//
// DROP (training forward): MultiheadAttention's dropout on the attention probabilities (attention_dropout,
// wav2vec2.py:1012): keep[(utterance, head)][query][key] bytes (row pitch `pitch` >= 32 KS), 1 = kept; a dropped
// probability leaves the row sum as it is and the product with V without its term; kept ones are scaled by
// drop_scale = 1 / (1 - p) through the row's normalisation.
template <int FMT, int KS, bool VARLEN, bool DROP = false>
__global__ __launch_bounds__(AttnShape<KS>::NW * 64, DROP ? 2 : AttnShape<KS>::MIN_WAVES_PER_SIMD) void attention_kernel(const bf16_t* __restrict__ qkv,
                                                        bf16_t* __restrict__ out, int T, float scale,
                                                        const int* __restrict__ nframes,
                                                        const uint8_t* __restrict__ keep, int pitch, float drop_scale) {
    constexpr int NW = AttnShape<KS>::NW;
    constexpr int NT = NW * 64;
    constexpr int KEYS = KS * 32;
    constexpr int NKT = KS * 2;   // 16-key tiles
    constexpr int KS_BYTES = KEYS * 128;
    extern __shared__ __attribute__((aligned(16))) char lds[];  // KS_BYTES + KEYS * V_PITCH
    char* Ks = lds;
    char* Vs = lds + KS_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x >> 4;
    const int h = blockIdx.x & 15;
    const bf16_t* base = qkv + (long)b * T * 3072 + h * 64;
    const int Tv = VARLEN ? min(max(nframes[b], 1), T) : T;  // keys this utterance attends to

    // ---- this wave's Q fragments for all its query tiles (qt = wave, wave + NW, ...): requested
    // first so that their latency hides behind the K/V staging
    const int fr = lane & 15;
    const int fg = lane >> 4;
    const int nqt = (T + 15) >> 4;
    constexpr int MAX_QT = (NKT + NW - 1) / NW;
    constexpr bool PREFETCH_Q = KS <= 7;  // beyond that the score registers need the room
    constexpr int QHOLD = PREFETCH_Q ? MAX_QT : 1;
    u16x8 qall[QHOLD][2];
    if (PREFETCH_Q) {
#pragma unroll
        for (int i = 0; i < QHOLD; ++i) {
            const int qrow = min((wave + NW * i) * 16 + fr, T - 1);
#pragma unroll
            for (int s = 0; s < 2; ++s)
                qall[i][s] = *reinterpret_cast<const u16x8*>(base + (long)qrow * 3072 + s * 32 + fg * 8);
        }
    }

    // ---- stage K (swizzled 128-byte rows) and V (160-byte rows); rows >= T are zero.
    // Every load is unconditional (row clamped) so that all of them are in flight at once.
    {
        constexpr int ITERS = (KEYS * 8 + NT - 1) / NT;  // 16-byte chunks per thread
        u16x8 kv[ITERS], vv[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int i = tid + it * NT;
            const int key = min(i >> 3, T - 1), c = i & 7;
            const bf16_t* row = base + (long)key * 3072 + c * 8;
#if SLSMI_ATT_KNOCK == 1
            kv[it] = u16x8{1, 2, 3, 4, 5, 6, 7, (unsigned short)i};
            vv[it] = kv[it];
#else
            kv[it] = *reinterpret_cast<const u16x8*>(row + 1024);
            vv[it] = *reinterpret_cast<const u16x8*>(row + 2048);
#endif
        }
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int i = tid + it * NT;
            const int key = i >> 3, c = i & 7;
            const u16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
            if ((KEYS * 8) % NT == 0 || key < KEYS) {
                *reinterpret_cast<u16x8*>(Ks + key * 128 + ((c ^ (key & 7)) << 4)) = key < T ? kv[it] : zero;
                *reinterpret_cast<u16x8*>(Vs + key * V_PITCH + (c << 4)) = key < T ? vv[it] : zero;
            }
        }
    }
    __syncthreads();

    // transposed-read address of this lane inside a (32-key step, 16-column tile) block:
    // lane 4q + p of a 16-lane group points at key row q, columns 4p .. 4p + 3
    const int v_lane_off = (4 * fg + (fr >> 2)) * V_PITCH + (fr & 3) * 8;
    const int k_lane_off = fr * 128;
    const int kc0 = ((0 + fg) ^ (fr & 7)) << 4, kc1 = ((4 + fg) ^ (fr & 7)) << 4;
#if SLSMI_ATT_KNOCK == 2
    {   // memory only: one 16-byte store per lane and query tile, nothing computed
        for (int qi = 0; qi < MAX_QT; ++qi) {
            const int qt = wave + NW * qi;
            if (qt >= nqt) break;
            const int q = qt * 16 + fr;
            bf16_t* orow = out + ((long)b * T + min(q, T - 1)) * 1024 + h * 64 + (fg & 1) * 16 + (fg >> 1) * 8;
            const u16x8 kk = *reinterpret_cast<const u16x8*>(Ks + (q & 127) * 128 + fg * 16);
            for (int jp = 0; jp < 2; ++jp)
                if (q < T) *reinterpret_cast<u16x8*>(orow + jp * 32) = kk;
        }
        return;
    }
#endif
#pragma unroll
    for (int qi = 0; qi < MAX_QT; ++qi) {
        const int qt = wave + NW * qi;
        if (qt >= nqt) break;
        // keep the K/V fragment reads of one query tile from being hoisted over the previous one
        asm volatile("" ::: "memory");
        const int q = qt * 16 + fr;
        u16x8 qf[2];
        if (PREFETCH_Q) {
            qf[0] = qall[qi][0];
            qf[1] = qall[qi][1];
        } else {
            const int qrow = min(q, T - 1);
#pragma unroll
            for (int s = 0; s < 2; ++s)
                qf[s] = *reinterpret_cast<const u16x8*>(base + (long)qrow * 3072 + s * 32 + fg * 8);
        }

        uint32_t km[DROP ? NKT : 1];  // this query's keep bytes of keys 16 kt + 4 fg .. + 3, requested before the products
        if (DROP) {
            const uint8_t* krow = keep + ((long)blockIdx.x * T + min(q, T - 1)) * pitch + fg * 4;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) km[kt] = *reinterpret_cast<const uint32_t*>(krow + kt * 16);
        }
        f32x4 st[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            const u16x8 k0 = *reinterpret_cast<const u16x8*>(Ks + kt * 16 * 128 + k_lane_off + kc0);
            const u16x8 k1 = *reinterpret_cast<const u16x8*>(Ks + kt * 16 * 128 + k_lane_off + kc1);
            st[kt] = Op<FMT>::mfma(k0, qf[0], f32x4{0.f, 0.f, 0.f, 0.f});
            st[kt] = Op<FMT>::mfma(k1, qf[1], st[kt]);
        }
        // ---- softmax over keys (column q = fr; keys spread over regs, tiles and lane groups);
        // keys >= T can only sit in the last TAIL tiles (KS is ceil(T / 32) rounded up to an
        // instantiated value, by at most one step above 8)
        constexpr int TAIL = VARLEN ? NKT : (KS <= 8 ? 2 : 4);
#pragma unroll
        for (int kt = NKT - TAIL; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (kt * 16 + fg * 4 + r >= Tv) st[kt][r] = -INFINITY;
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {  // two three-input maxima per four scores
            mx = fmaxf(fmaxf(mx, st[kt][0]), st[kt][1]);
            mx = fmaxf(fmaxf(mx, st[kt][2]), st[kt][3]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float nmx = -mx * scale;
        const f32x2 sc2 = {scale, scale}, nm2 = {nmx, nmx};
        f32x2 sum2 = {0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const f32x2 e = __builtin_elementwise_fma(f32x2{st[kt][2 * hh], st[kt][2 * hh + 1]}, sc2, nm2);
                const f32x2 p = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
                st[kt][2 * hh] = p[0];
                st[kt][2 * hh + 1] = p[1];
                sum2 += p;
            }
        }
        float sum = sum2[0] + sum2[1];
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = DROP ? drop_scale / sum : 1.0f / sum;
        if (DROP) {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (((km[kt] >> (8 * r)) & 0xFFu) == 0) st[kt][r] = 0.f;
        }

        // ---- O^T = V^T P^T
        f32x4 ot[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) ot[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const uint32_t p0 = Op<FMT>::pack2_unit(st[2 * ks][0], st[2 * ks][1]);
            const uint32_t p1 = Op<FMT>::pack2_unit(st[2 * ks][2], st[2 * ks][3]);
            const uint32_t p2 = Op<FMT>::pack2_unit(st[2 * ks + 1][0], st[2 * ks + 1][1]);
            const uint32_t p3 = Op<FMT>::pack2_unit(st[2 * ks + 1][2], st[2 * ks + 1][3]);
            const u16x8 pf = __builtin_bit_cast(u16x8, u32x4{p0, p1, p2, p3});
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const char* blk = Vs + ks * 32 * V_PITCH + nt * 32 + v_lane_off;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(blk));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(blk + 16 * V_PITCH));
                const u16x8 vv = __builtin_bit_cast(u16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                ot[nt] = Op<FMT>::mfma(vv, pf, ot[nt]);
            }
        }
        // lane (fr, fg) holds columns 16 nt + 4 fg .. + 3 of query fr: 8 bytes per column tile.  Lanes fg and
        // fg ^ 1 swap halves so that even fg gets columns 8 (fg >> 1) .. + 7 of tile 2 jp and odd fg the same
        // columns of tile 2 jp + 1: 16 bytes per lane, 64 contiguous bytes per row and store.
        bf16_t* orow = out + ((long)b * T + min(q, T - 1)) * 1024 + h * 64 + (fg & 1) * 16 + (fg >> 1) * 8;
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
            const f32x4 v0 = ot[2 * jp] * inv, v1 = ot[2 * jp + 1] * inv;
            // without dropout an output row is a convex combination of V rows, which are finite values of the
            // format: no saturation needed in the pack
            const uint32_t e_lo = DROP ? Op<FMT>::pack2(v0[0], v0[1]) : Op<FMT>::pack2_unit(v0[0], v0[1]);
            const uint32_t e_hi = DROP ? Op<FMT>::pack2(v0[2], v0[3]) : Op<FMT>::pack2_unit(v0[2], v0[3]);
            const uint32_t o_lo = DROP ? Op<FMT>::pack2(v1[0], v1[1]) : Op<FMT>::pack2_unit(v1[0], v1[1]);
            const uint32_t o_hi = DROP ? Op<FMT>::pack2(v1[2], v1[3]) : Op<FMT>::pack2_unit(v1[2], v1[3]);
            const auto lo = __builtin_amdgcn_permlane16_swap(e_lo, o_lo, false, false);
            const auto hi = __builtin_amdgcn_permlane16_swap(e_hi, o_hi, false, false);
            if (q < T) *reinterpret_cast<u32x4*>(orow + jp * 32) = u32x4{lo[0], hi[0], lo[1], hi[1]};
        }
    }
}

template <int FMT, int KS, bool VARLEN, bool DROP = false>
hipError_t launch_attention_one(dim3 grid, hipStream_t s, const bf16_t* qkv, bf16_t* out, int T, float scale,
                                const int* nframes, const uint8_t* keep = nullptr, int pitch = 0, float drop_scale = 1.f) {
    constexpr int LDS = KS * 32 * (128 + V_PITCH);
    static LdsAttrState lds_attr;
    const hipError_t attr =
        ensure_dynamic_lds(reinterpret_cast<const void*>(&attention_kernel<FMT, KS, VARLEN, DROP>), LDS, lds_attr);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL((attention_kernel<FMT, KS, VARLEN, DROP>), grid, dim3(AttnShape<KS>::NW * 64), LDS, s, qkv, out, T, scale,
                       nframes, keep, pitch, drop_scale);
    return hipGetLastError();
}

template <int FMT>
hipError_t launch_attention_drop_ks(int ks, dim3 grid, hipStream_t s, const bf16_t* qkv, bf16_t* out, int T, float scale,
                                    const uint8_t* keep, int pitch, float drop_scale) {
    switch (ks) {
        case 1: return launch_attention_one<FMT, 1, false, true>(grid, s, qkv, out, T, scale, nullptr, keep, pitch, drop_scale);
        case 2: return launch_attention_one<FMT, 2, false, true>(grid, s, qkv, out, T, scale, nullptr, keep, pitch, drop_scale);
        case 3: return launch_attention_one<FMT, 3, false, true>(grid, s, qkv, out, T, scale, nullptr, keep, pitch, drop_scale);
        case 4: return launch_attention_one<FMT, 4, false, true>(grid, s, qkv, out, T, scale, nullptr, keep, pitch, drop_scale);
        case 5: return launch_attention_one<FMT, 5, false, true>(grid, s, qkv, out, T, scale, nullptr, keep, pitch, drop_scale);
        case 6: return launch_attention_one<FMT, 6, false, true>(grid, s, qkv, out, T, scale, nullptr, keep, pitch, drop_scale);
        case 7: return launch_attention_one<FMT, 7, false, true>(grid, s, qkv, out, T, scale, nullptr, keep, pitch, drop_scale);
        case 8: return launch_attention_one<FMT, 8, false, true>(grid, s, qkv, out, T, scale, nullptr, keep, pitch, drop_scale);
        default: return hipErrorInvalidValue;
    }
}

template <int FMT, bool VARLEN>
hipError_t launch_attention_ks(int ks, dim3 grid, hipStream_t s, const bf16_t* qkv, bf16_t* out, int T, float scale,
                               const int* nframes) {
    switch (ks) {
        case 1: return launch_attention_one<FMT, 1, VARLEN>(grid, s, qkv, out, T, scale, nframes);
        case 2: return launch_attention_one<FMT, 2, VARLEN>(grid, s, qkv, out, T, scale, nframes);
        case 3: return launch_attention_one<FMT, 3, VARLEN>(grid, s, qkv, out, T, scale, nframes);
        case 4: return launch_attention_one<FMT, 4, VARLEN>(grid, s, qkv, out, T, scale, nframes);
        case 5: return launch_attention_one<FMT, 5, VARLEN>(grid, s, qkv, out, T, scale, nframes);
        case 6: return launch_attention_one<FMT, 6, VARLEN>(grid, s, qkv, out, T, scale, nframes);
        case 7: return launch_attention_one<FMT, 7, VARLEN>(grid, s, qkv, out, T, scale, nframes);
        case 8: return launch_attention_one<FMT, 8, VARLEN>(grid, s, qkv, out, T, scale, nframes);
        case 9: case 10: return launch_attention_one<FMT, 10, VARLEN>(grid, s, qkv, out, T, scale, nframes);
        case 11: case 12: return launch_attention_one<FMT, 12, VARLEN>(grid, s, qkv, out, T, scale, nframes);
        case 13: case 14: return launch_attention_one<FMT, 14, VARLEN>(grid, s, qkv, out, T, scale, nframes);
        default: return launch_attention_one<FMT, 16, VARLEN>(grid, s, qkv, out, T, scale, nframes);
    }
}

}  // namespace

// The training forward's attention with dropout on the probabilities: keep [B * 16][T][pitch] bytes, pitch a
// multiple of 32 and >= T, at most 256 frames.
hipError_t launch_attention_dropout(const bf16_t* qkv, bf16_t* out, int B, int T, int fmt, const uint8_t* keep, int pitch,
                                    float drop_scale, hipStream_t s) {
    if (B <= 0 || T <= 0 || T > 256 || !keep || pitch % 32 != 0 || pitch < T) return hipErrorInvalidValue;
    const float scale = 1.4426950408889634f;
    hipError_t rc = hipSuccess;
    SLSMI_FMT_DISPATCH(fmt, rc = (launch_attention_drop_ks<F>((T + 31) / 32, dim3(B * 16), s, qkv, out, T, scale, keep, pitch, drop_scale)));
    return rc;
}

hipError_t launch_attention(const bf16_t* qkv, bf16_t* out, int B, int T, int fmt, hipStream_t s, const int* nframes) {
    if (B <= 0 || T <= 0 || T > MAX_KS * 32) return hipErrorInvalidValue;
    const float scale = 1.4426950408889634f;  // log2(e): natural softmax
    hipError_t rc = hipSuccess;
    if (nframes) {
        SLSMI_FMT_DISPATCH(fmt, rc = (launch_attention_ks<F, true>((T + 31) / 32, dim3(B * 16), s, qkv, out, T, scale, nframes)));
    } else {
        SLSMI_FMT_DISPATCH(fmt, rc = (launch_attention_ks<F, false>((T + 31) / 32, dim3(B * 16), s, qkv, out, T, scale, nullptr)));
    }
    return rc;
}

}  // namespace slsmi
