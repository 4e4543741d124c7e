// Self-attention core for one (utterance, head) per workgroup: 16 heads x 64.
//
// The whole K and V of a head (T <= 224 frames) sit in LDS; there is no online
// softmax.  Scores are computed TRANSPOSED (S^T = K Q^T, keys on the MFMA row
// index) so that a lane ends up holding, for one query column, four consecutive
// keys per 16-key tile.  Two such tiles packed to bf16 are exactly the B-operand
// fragment of the second product O^T = V^T P^T when the k index inside each
// 32-key step is enumerated as (lane group g, j) -> key 4g + j (j < 4) or
// 16 + 4g + (j - 4): P never leaves registers.  V is staged transposed
// (Vt[d][key]) so the matching A-operand fragment is two 8-byte LDS reads.
// The head-averaged attention weights the reference also returns
// (wav2vec/wav2vec2.py:1046-1052, unused by every head) are not produced.
#include "common.h"
#include "kernels.h"

namespace slsmi {

namespace {

constexpr int KT2 = 7;            // 32-key steps
constexpr int KEYS = KT2 * 32;    // 224
constexpr int NKT = KT2 * 2;      // 16-key tiles
constexpr int VT_LD = 232;        // Vt row stride in elements (464 B: conflict-free b64 reads)
constexpr int KS_BYTES = KEYS * 128;
constexpr int VT_BYTES = 64 * VT_LD * 2;

template <int FMT>
__global__ __launch_bounds__(256, 2) void attention_kernel(const bf16_t* __restrict__ qkv,
                                                        bf16_t* __restrict__ out, int T) {
    __shared__ __attribute__((aligned(16))) char lds[KS_BYTES + VT_BYTES];
    char* Ks = lds;
    bf16_t* Vt = reinterpret_cast<bf16_t*>(lds + KS_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x >> 4;
    const int h = blockIdx.x & 15;
    const bf16_t* base = qkv + (long)b * T * 3072 + h * 64;

    // ---- stage K (swizzled rows) and V (transposed)
    for (int i = tid; i < KEYS * 8; i += 256) {
        const int key = i >> 3, c = i & 7;
        u16x8 kv = {0, 0, 0, 0, 0, 0, 0, 0};
        u16x8 vv = {0, 0, 0, 0, 0, 0, 0, 0};
        if (key < T) {
            const bf16_t* row = base + (long)key * 3072 + c * 8;
            kv = *reinterpret_cast<const u16x8*>(row + 1024);
            vv = *reinterpret_cast<const u16x8*>(row + 2048);
        }
        *reinterpret_cast<u16x8*>(Ks + key * 128 + ((c ^ (key & 7)) << 4)) = kv;
#pragma unroll
        for (int e = 0; e < 8; ++e) Vt[(c * 8 + e) * VT_LD + key] = vv[e];
    }
    __syncthreads();

    const int fr = lane & 15;
    const int fg = lane >> 4;
    const int nqt = (T + 15) >> 4;
    for (int qt = wave; qt < nqt; qt += 4) {
        // keep the K/V fragment reads inside the loop: hoisted, they cost ~230 registers
        asm volatile("" ::: "memory");
        const int q = qt * 16 + fr;
        const int qrow = min(q, T - 1);
        u16x8 qf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
            qf[s] = *reinterpret_cast<const u16x8*>(base + (long)qrow * 3072 + s * 32 + fg * 8);

        f32x4 st[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const u16x8 kf = *reinterpret_cast<const u16x8*>(
                    Ks + (kt * 16 + fr) * 128 + (((s * 4 + fg) ^ (fr & 7)) << 4));
                st[kt] = Op<FMT>::mfma(kf, qf[s], st[kt]);
            }
        }
        // ---- softmax over keys (column q = fr; keys spread over regs, tiles and lane groups)
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * 16 + fg * 4 + r;
                if (key >= T) st[kt][r] = -INFINITY;
                mx = fmaxf(mx, st[kt][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __expf(st[kt][r] - mx);
                st[kt][r] = p;
                sum += p;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;

        // ---- O^T = V^T P^T
        f32x4 ot[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) ot[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KT2; ++ks) {
            u16x8 pf;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                pf[j] = Op<FMT>::pack(st[2 * ks][j]);
                pf[4 + j] = Op<FMT>::pack(st[2 * ks + 1][j]);
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const bf16_t* vrow = Vt + (nt * 16 + fr) * VT_LD + ks * 32 + fg * 4;
                const u16x4 lo = *reinterpret_cast<const u16x4*>(vrow);
                const u16x4 hi = *reinterpret_cast<const u16x4*>(vrow + 16);
                u16x8 vv;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    vv[j] = lo[j];
                    vv[4 + j] = hi[j];
                }
                ot[nt] = Op<FMT>::mfma(vv, pf, ot[nt]);
            }
        }
        if (q < T) {
            bf16_t* orow = out + ((long)b * T + q) * 1024 + h * 64 + fg * 4;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                u16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = Op<FMT>::pack(ot[nt][r] * inv);
                *reinterpret_cast<u16x4*>(orow + nt * 16) = o;
            }
        }
    }
}

}  // namespace

hipError_t launch_attention(const bf16_t* qkv, bf16_t* out, int B, int T, int fmt, hipStream_t s) {
    if (B <= 0 || T <= 0 || T > KEYS) return hipErrorInvalidValue;
    SLSMI_FMT_DISPATCH(fmt, hipLaunchKernelGGL(attention_kernel<F>, dim3(B * 16), dim3(256), 0, s, qkv, out, T));
    return hipGetLastError();
}

}  // namespace slsmi
