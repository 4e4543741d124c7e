// Gradient kernels of the training path (reference main.py:201-258: loss.backward() through
// Model.forward).  Every contraction of the backward pass runs on the forward's MFMA GEMM
// (C = A W^T): for y = x W^T the input gradient is dy (W^T)^T with a transposed weight copy,
// the weight gradient is (dy^T)(x^T)^T with both activations transposed by transpose16_kernel.
// Gradient operands are bfloat16 (fp32 accumulation): gradients span more binades than half
// precision holds without loss scaling.  What is here are the pieces around those GEMMs.
#include <type_traits>
#include <vector>

#include "adam.h"
#include "common.h"
#include "kernels.h"

namespace slsmi {

namespace {

inline int grid_for(long n, int per_block, int cap = 65535) {
    long g = (n + per_block - 1) / per_block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// d/dz of the erf-GELU: Phi(z) + z phi(z)
__device__ __forceinline__ float gelu_grad(float z) {
    const float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * z * z);
    return fmaf(z, pdf, cdf);
}

// ---------------------------------------------------------------- transpose
// out[c][r] = in[r * ld + c] for r < R, c < C (ld < C: overlapping rows, the conv layers' operand
// view), converted from format SRC to DST; out[c][r] = 0 for R <= r < ldo.  64 x 64 tiles through LDS.
// tb.grp_in > 0: rows at position >= tb.grp_valid inside each group of tb.grp_in rows are read as zeros
// (the never-written rows of a padded per-utterance layout).  blockIdx.z = z1 * tb.zdiv + z2 moves the
// input by z1 * tb.in1 + z2 * tb.in2 elements and the output by z1 * tb.out1 + z2 * tb.out2.
template <int SRC, int DST>
__global__ __launch_bounds__(256) void transpose16_kernel(const bf16_t* __restrict__ in, long ld, int R, int C,
                                                          bf16_t* __restrict__ out, long ldo, long wrows,
                                                          TransposeBatch tb) {
    __shared__ bf16_t tile[64][66];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    {
        const int z1 = blockIdx.z / tb.zdiv, z2 = blockIdx.z % tb.zdiv;
        in += z1 * tb.in1 + z2 * tb.in2;
        out += z1 * tb.out1 + z2 * tb.out2;
    }
    const long r0 = (long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const long r = r0 + ty + 4 * i;
        const int c = c0 + tx;
        bf16_t v = 0;
        if (r < R && c < C && (tb.grp_in <= 0 || (int)(r % tb.grp_in) < tb.grp_valid)) {
            v = in[r * ld + c];
            if (SRC != DST) v = Op<DST>::pack(Op<SRC>::unpack(v));
        }
        tile[ty + 4 * i][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty + 4 * i;
        const long r = r0 + tx;
        if (c < C && r < wrows) out[(long)c * ldo + r] = tile[tx][ty + 4 * i];
    }
}

// fp32 in[r][c] (row stride C) -> bfloat16 copy[r][c] and its transpose outT[c][r], outT[c][r] = 0 for
// R <= r < ldo: the 16-bit operand of an input-gradient GEMM and of a weight-gradient GEMM in one pass.
__global__ __launch_bounds__(256) void cvt_transpose_kernel(const float* __restrict__ in, int R, int C,
                                                            bf16_t* __restrict__ copy, bf16_t* __restrict__ outT,
                                                            long ldo) {
    __shared__ bf16_t tile[64][66];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long r0 = (long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const long r = r0 + ty + 4 * i;
        const int c = c0 + tx;
        bf16_t v = 0;
        if (r < R && c < C) {
            v = Op<FMT_BF16>::pack(in[r * C + c]);
            copy[r * C + c] = v;
        }
        tile[ty + 4 * i][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty + 4 * i;
        const long r = r0 + tx;
        if (c < C && r < ldo) outT[(long)c * ldo + r] = tile[tx][ty + 4 * i];
    }
}

// A weight matrix for the training step in one pass: fp32 in[r][c] * scale -> copy[r][c] in format CF (the forward's
// operand) and outT[c * ldo + r] in bfloat16 (the input-gradient GEMM's operand), rounded from the copy's value.
template <int CF>
__device__ __forceinline__ void cvt_transpose_w_tile(const float* __restrict__ in, int R, int C, float scale,
                                                     bf16_t* __restrict__ copy, bf16_t* __restrict__ outT, long ldo) {
    // 64 x 64 tiles; 16 bytes read, 8 bytes written per lane on both sides (C % 4 == 0, ldo % 4 == 0)
    __shared__ __attribute__((aligned(8))) bf16_t tile[64][68];
    const int t = threadIdx.x;
    const int quad = (t & 15) * 4, line = t >> 4;
    const long r0 = (long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long r = r0 + line + 16 * i;
        const int c = c0 + quad;
        u16x4 q = {0, 0, 0, 0};
        if (r < R && c < C) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(in + r * C + c);
            u16x4 cp;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                cp[e] = Op<CF>::pack(v[e] * scale);
                q[e] = CF == FMT_BF16 ? cp[e] : Op<FMT_BF16>::pack(Op<CF>::unpack(cp[e]));
            }
            *reinterpret_cast<u16x4*>(copy + r * C + c) = cp;
        }
        *reinterpret_cast<u16x4*>(&tile[line + 16 * i][quad]) = q;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + line + 16 * i;
        const long r = r0 + quad;
        if (c >= C || r >= R) continue;
        u16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = tile[quad + e][line + 16 * i];
        bf16_t* dst = outT + (long)c * ldo + r;
        if (r + 3 < R) {
            *reinterpret_cast<u16x4*>(dst) = o;
        } else {
            for (int e = 0; e < 4 && r + e < R; ++e) dst[e] = o[e];
        }
    }
}

template <int CF>
__global__ __launch_bounds__(256) void cvt_transpose_w_kernel(const float* __restrict__ in, int R, int C, float scale,
                                                              bf16_t* __restrict__ copy, bf16_t* __restrict__ outT,
                                                              long ldo) {
    cvt_transpose_w_tile<CF>(in, R, C, scale, copy, outT, ldo);
}

// The same for up to CVT_W_BATCH matrices of one shape in one launch (blockIdx.z = matrix): the weight reload of a
// training step converts ~150 matrices, each too small to fill the device alone.
struct CvtTransposeWTable {
    CvtTransposeW item[CVT_W_BATCH];
};
template <int CF>
__global__ __launch_bounds__(256) void cvt_transpose_w_batch_kernel(int R, int C, CvtTransposeWTable t) {
    const CvtTransposeW& j = t.item[blockIdx.z];
    cvt_transpose_w_tile<CF>(j.src, R, C, j.scale, j.copy, j.outT, j.ldo);
}

// ---------------------------------------------------------------- column sums (bias gradients)
// Stage 1: partial[p][n] = sum of in[m][n] over the rows m of chunk p.  Stage 2 adds the chunks in
// index order (reproducible bits).  T16: 16-bit input of format FMT, else fp32.
template <bool T16, int FMT>
__global__ __launch_bounds__(256) void colsum_part_kernel(const void* __restrict__ in, long ld, long M, int N,
                                                          long rows_per_part, float* __restrict__ partial) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const long m0 = (long)blockIdx.y * rows_per_part;
    const long m1 = m0 + rows_per_part < M ? m0 + rows_per_part : M;
    float s = 0.f;
    for (long m = m0; m < m1; ++m)
        s += T16 ? Op<FMT>::unpack(reinterpret_cast<const bf16_t*>(in)[m * ld + n])
                 : reinterpret_cast<const float*>(in)[m * ld + n];
    partial[(long)blockIdx.y * N + n] = s;
}


// Where the column sums go: columns [i seg, (i + 1) seg) to out[i], times scale[i]; at most three segments.
struct ColsumOut {
    float* out[3];
    float scale[3];
    int seg;
};
static ColsumOut colsum_single(float* out, int N, float scale) { return ColsumOut{{out, nullptr, nullptr}, {scale, 0.f, 0.f}, N}; }

// Both stages in one launch for a few thousand rows (the training step: M = batch x 201): a workgroup owns 64
// columns, its 16 waves take every 16th row each with four interleaved running sums, and the 16 wave sums are
// added in wave order.
template <bool T16, int FMT>
__global__ __launch_bounds__(1024) void colsum_one_kernel(const void* __restrict__ in, long ld, long M, int N,
                                                          int accumulate, ColsumOut o) {
    __shared__ float red[16][64];
    const int col = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + col;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < N) {
        auto at = [&](long m) {
            return T16 ? Op<FMT>::unpack(reinterpret_cast<const bf16_t*>(in)[m * ld + n])
                       : reinterpret_cast<const float*>(in)[m * ld + n];
        };
        long m = rg;
        for (; m + 48 < M; m += 64) {
            acc[0] += at(m);
            acc[1] += at(m + 16);
            acc[2] += at(m + 32);
            acc[3] += at(m + 48);
        }
        for (; m < M; m += 16) acc[0] += at(m);
    }
    red[rg][col] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (rg == 0 && n < N) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += red[r][col];
        // segment n / seg of the columns goes to its own array with its own scale (a null array: not wanted)
        const int sg = n / o.seg;
        float* base = sg == 0 ? o.out[0] : (sg == 1 ? o.out[1] : o.out[2]);
        const float scale = sg == 0 ? o.scale[0] : (sg == 1 ? o.scale[1] : o.scale[2]);
        if (base) {
            float* dst = base + (n - sg * o.seg);
            *dst = accumulate ? fmaf(scale, t, *dst) : scale * t;
        }
    }
}

// The same sums for many matrices of partial sums at once (blockIdx.y = matrix z, z_in elements apart): columns
// n < C of matrix z go to dg[z], the others to db[z].
constexpr int LN_PART_BATCH = 64;
struct LnPartOut {
    float* dg[LN_PART_BATCH];
    float* db[LN_PART_BATCH];
};
__global__ __launch_bounds__(1024) void ln_part_sums_kernel(const float* __restrict__ in, long z_in, long M, int C,
                                                            LnPartOut o) {
    __shared__ float red[16][64];
    const int col = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + col;    // 2 C is a multiple of 64
    const float* src = in + blockIdx.y * z_in + n;
    const long ld = 2L * C;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    long m = rg;
    for (; m + 48 < M; m += 64) {
        acc[0] += src[m * ld];
        acc[1] += src[(m + 16) * ld];
        acc[2] += src[(m + 32) * ld];
        acc[3] += src[(m + 48) * ld];
    }
    for (; m < M; m += 16) acc[0] += src[m * ld];
    red[rg][col] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (rg == 0) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += red[r][col];
        float* base = n < C ? o.dg[blockIdx.y] : o.db[blockIdx.y];
        if (base) base[n < C ? n : n - C] = t;
    }
}

// ---------------------------------------------------------------- LayerNorm (+ GELU) backward
// y = LN(x) gamma + beta over C channels; optional a = gelu(y) (the conv blocks, wav2vec2.py:803-813).
// Input: x (the LayerNorm's input: fp32, or 16-bit of format XF), dy = gradient of the loss with
// respect to y (or to a, with GELU), fp32.  Output: dx = resid + the gradient with respect to x
// (fp32, resid nullable), or its bfloat16 rounding (OUT16).  One wave per row; the lane owns columns
// 4 lane + 256 i .. + 3; a workgroup accumulates its rows' contributions to dgamma / dbeta in registers
// and leaves them in part[workgroup][2][C]; their column sums over the workgroups are dgamma and dbeta.
// Rows of a padded per-utterance layout (grp_in > 0) at positions >= grp_valid produce zero and
// contribute nothing.
template <int C, bool X16, int XF, bool GELU, bool OUT16>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ xin, const float* __restrict__ dy,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float eps, const float* __restrict__ resid, void* __restrict__ out,
                                                     float* __restrict__ part, long rows, int grp_in, int grp_valid,
                                                     const float* __restrict__ resid2, bf16_t* __restrict__ copy16,
                                                     int dy_slices, long dy_slice) {
    constexpr int NV = C / 256;  // f32x4 pieces per lane
    __shared__ float red[4][2][C];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 dg[NV], db[NV], g4[NV], b4[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        dg[i] = db[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        g4[i] = *reinterpret_cast<const f32x4*>(gamma + i * 256 + lane * 4);
        b4[i] = GELU ? *reinterpret_cast<const f32x4*>(beta + i * 256 + lane * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const bool valid = grp_in <= 0 || (int)(row % grp_in) < grp_valid;
        f32x4 x[NV], d[NV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const long off = row * C + i * 256 + lane * 4;
            if (X16) {
                const u16x4 raw = *reinterpret_cast<const u16x4*>(reinterpret_cast<const bf16_t*>(xin) + off);
#pragma unroll
                for (int e = 0; e < 4; ++e) x[i][e] = Op<XF>::unpack(raw[e]);
            } else {
                x[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(xin) + off);
            }
            d[i] = *reinterpret_cast<const f32x4*>(dy + off);
            // (a GEMM's K slices, added in index order as its summing kernel would)
            for (int z = 1; z < dy_slices; ++z) d[i] += *reinterpret_cast<const f32x4*>(dy + z * dy_slice + off);
            s += (x[i][0] + x[i][1]) + (x[i][2] + x[i][3]);
        }
        const float mean = wave_sum(s) * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                x[i][e] -= mean;
                q = fmaf(x[i][e], x[i][e], q);
            }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / C) + eps);
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = x[i][e] * rstd;
                float dz = d[i][e];
                if (GELU) dz *= gelu_grad(fmaf(xh, g4[i][e], b4[i][e]));
                if (valid) {
                    dg[i][e] = fmaf(dz, xh, dg[i][e]);
                    db[i][e] += dz;
                }
                const float g = dz * g4[i][e];
                x[i][e] = xh;
                d[i][e] = g;
                m1 += g;
                m2 = fmaf(g, xh, m2);
            }
        m1 = wave_sum(m1) * (1.0f / C);
        m2 = wave_sum(m2) * (1.0f / C);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const long off = row * C + i * 256 + lane * 4;
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = valid ? rstd * (d[i][e] - m1 - x[i][e] * m2) : 0.f;
            if (OUT16) {
                u16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = Op<FMT_BF16>::pack(r[e]);
                *reinterpret_cast<u16x4*>(reinterpret_cast<bf16_t*>(out) + off) = o;
            } else {
                if (resid) r += *reinterpret_cast<const f32x4*>(resid + off);
                if (resid2) r += *reinterpret_cast<const f32x4*>(resid2 + off);
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + off) = r;
                if (copy16) {  // the bfloat16 operand of the GEMM that takes this gradient next
                    u16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = Op<FMT_BF16>::pack(r[e]);
                    *reinterpret_cast<u16x4*>(copy16 + off) = o;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        *reinterpret_cast<f32x4*>(&red[wave][0][i * 256 + lane * 4]) = dg[i];
        *reinterpret_cast<f32x4*>(&red[wave][1][i * 256 + lane * 4]) = db[i];
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 2 * C; j += 256) {
        const int which = j / C, c = j - which * C;
        part[((long)blockIdx.x * 2 + which) * C + c] =
            (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
    }
}

// ---------------------------------------------------------------- GELU forward / backward on 16-bit rows
template <int FMT>
__global__ __launch_bounds__(256) void gelu16_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, long n8) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const u16x8 v = *reinterpret_cast<const u16x8*>(in + i * 8);
        u16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = Op<FMT>::pack(gelu_erf(Op<FMT>::unpack(v[e])));
        *reinterpret_cast<u16x8*>(out + i * 8) = o;
    }
}

// g[R][C] (bfloat16, in place) *= gelu'(pre) with pre in format FMT, 64 x 64 tiles, the result also written transposed:
// outT[c][r] (row stride ldo >= R, zeros for R <= r < ldo), the weight-gradient GEMM's operand.
template <int FMT>
__global__ __launch_bounds__(256) void gelu_bwd16_transpose_kernel(const bf16_t* __restrict__ pre, bf16_t* __restrict__ g,
                                                                   int R, int C, bf16_t* __restrict__ outT, long ldo) {
    __shared__ bf16_t tile[64][66];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long r0 = (long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const long r = r0 + ty + 4 * i;
        const int c = c0 + tx;
        bf16_t v = 0;
        if (r < R && c < C) {
            const long o = r * C + c;
            v = Op<FMT_BF16>::pack(Op<FMT_BF16>::unpack(g[o]) * gelu_grad(Op<FMT>::unpack(pre[o])));
            g[o] = v;
        }
        tile[ty + 4 * i][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty + 4 * i;
        const long r = r0 + tx;
        if (c < C && r < ldo) outT[(long)c * ldo + r] = tile[tx][ty + 4 * i];
    }
}

// fp32 -> x (u: fp32 pre-activation): out = x + gelu(u)  /  backward: du16 = dx * gelu'(u)
__global__ __launch_bounds__(256) void add_gelu_kernel(const float* __restrict__ x, const float* __restrict__ u,
                                                       float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = x[i] + gelu_erf(u[i]);
}

__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ x, long n, float scale) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) x[i] *= scale;
}

// ---------------------------------------------------------------- attention backward
// One workgroup (AB_WAVES waves) per (utterance, head), T <= 256 frames, all products on MFMA with bfloat16
// operands.  Forward: S = Q K^T (q pre-scaled), P = softmax(S), O = P V.  Given dO:
//   D_i = dO_i . O_i,  dP = dO V^T,  dS = P o (dP - D),  dQ = dS K,  dK = dS^T Q,  dV = P^T dO.
// Q, K, V (converted to bfloat16) and dO of the head sit in LDS as rows of 144 bytes; the T x T
// matrices are never stored: every wave re-computes the tiles it needs from the operands.
//   Phase A, a wave per 16-query tile: scores TRANSPOSED (S^T = K Q^T, as the forward kernel), so a lane
//   holds one query column: row maximum / sum by two shuffles, P^T and dS^T stay in registers and are,
//   packed, the B operand of dQ^T = K^T dS^T (K^T fragments by transposed LDS reads).  The row
//   statistics go to LDS for phase B.
//   Phase B, a wave per 16-key tile, looping over the query tiles: scores NOT transposed (S = Q K^T),
//   so a lane holds one key column and four queries -- exactly the B operand of the 16x16x16 products
//   dV^T += dO^T P and dK^T += Q^T dS over the 16 queries (dO^T, Q^T by transposed LDS reads).
// A small batch has few (utterance, head) pairs: gridDim.y = S workgroups then share one pair.  Each stages the
// whole head, takes the row statistics of ALL query tiles (the score product alone, a third of phase A's matrix
// work), and does the rest -- dP, dQ in phase A, a key tile in phase B -- only for tiles t with t % S == blockIdx.y.
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
// ---------------------------------------------------------------- weight gradient without transposed copies
// C_z[n][k] = alpha * sum over m < M of A_z[m][n] * B_z[m][k]: the weight gradient dW = dY^T X straight from the
// row-major output gradient A (bfloat16) and the row-major saved activation B (format BFMT, converted to bfloat16 on
// the way into LDS).  A 128 x 128 tile per workgroup, four waves of 64 x 64 (NC = 128), or -- for an A of 64 columns,
// the positional conv's groups -- a 64 x 256 tile, four waves side by side (NC = 64); a K-tile is 64 rows m of both
// operands, staged as they lie (pitch = row bytes + 32), and the MFMA fragments -- eight m values of one column per
// lane -- come out of LDS by transposed reads (ds_read_b64_tr_b16: four rows of one column each).  A lane group fg
// takes rows 4 fg .. + 3 and 16 + 4 fg .. + 3 of a 32-row step, so one read touches 16 consecutive rows: with these
// pitches their pieces fall on all banks evenly.  One LDS stage; the next K-tile's global loads are in flight (in
// registers) during the products.
// Rows in groups (p.grp > 0): row m is row m % grp of group m / grp, and a group's first row lies gA / gB elements
// after the previous group's -- the clips of a batch whose rows are not one arithmetic sequence (the positional
// conv's input slabs).
struct TnParams {
    const bf16_t* A; long lda, zA;
    const bf16_t* B; long ldb, zB;
    float* C; long ldc, zC;
    int M;
    float alpha;
    int grp; long gA, gB;
};

// ADAM form: the product is not stored; it is the gradient g of an Adam step that the epilogue takes on the spot, on
// the parameter tensor the product has the shape of (C[n][k] <-> p[n][k], row pitch ldc) and its two moments: p, m, v
// are read and written where the plain form would only have written g, and the separate optimizer kernel would have
// read g, p, m, v and written p, m, v -- 8 bytes per parameter less traffic, and the 24 bytes that remain move while
// other workgroups of the launch multiply.  Batch entry z works on operand slices zsrc[z] (the layer's index: skipped
// layers leave gaps) and on the tensors of table row z.  adam.h's arithmetic: the same bits as storing alpha * acc and
// running adam_kernel over it.
struct TnAdamTable {
    float* p[TN_ADAM_BATCH];
    float* m[TN_ADAM_BATCH];
    float* v[TN_ADAM_BATCH];
    long long step[TN_ADAM_BATCH];
    int zsrc[TN_ADAM_BATCH];
    // the library's own 16-bit copies of the tensor (null: none kept): what cvt_transpose_w_tile makes of the new
    // values -- copy[n][k] = pack(p scale) in format copy_fmt, outT[k][n] (row pitch ldo) = the same value as bfloat16
    bf16_t* copy[TN_ADAM_BATCH];
    bf16_t* outT[TN_ADAM_BATCH];
    long ldo;
    float scale;
    int copy_fmt;
    AdamHyper h;
};
struct TnNoAdam {};
// Column sums of A on the way (the bias gradient that belongs to a weight gradient dY^T X is the column sum of dY):
// db[z] (null: none) receives alpha * sum over m of A_z[m][n] for the workgroups' n, written by the workgroups of
// tile column 0.  128-column form without row groups only.
struct TnBias {
    float* db[TN_ADAM_BATCH];
    int n;   // entries of db that count (batch entries at or beyond n have none: a launch may have more than the table holds)
};

// (GRP: a template flag, so that the plain form keeps its affine row addresses)
template <int BFMT, int NC, bool GRP, bool ADAM = false>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(TnParams p, std::conditional_t<ADAM, TnAdamTable, TnNoAdam> adam,
                                                         TnBias bias) {
    constexpr int KC = NC == 128 ? 128 : 256;          // columns of B (of C) per workgroup
    constexpr int A_CH = NC / 8, B_CH = KC / 8;         // 16-byte chunks per staged row
    // pitch = row bytes + 32: consecutive rows start 8 banks apart (mod 64 and mod 32), so the 32-byte pieces that a
    // transposed read takes from 8 (or 4) consecutive rows fall on disjoint bank windows (+ 16 left neighbouring rows
    // half a window apart: a third of the LDS cycles were bank conflicts)
    constexpr int A_PITCH = NC * 2 + 32, B_PITCH = KC * 2 + 32;
    constexpr int A_TILE = 64 * A_PITCH;
    constexpr int A_PIECES = 64 * A_CH / 256, B_PIECES = 64 * B_CH / 256;
    // ADAM: halves of the finished fp32 tile are parked in the same memory (64 rows of 512 bytes, the 16-byte chunk
    // index XORed with the row's low five bits against bank conflicts) so that the step runs over whole 512-byte rows
    constexpr int G_PITCH = 128 * 4;
    constexpr int LDS_BYTES = ADAM && 64 * G_PITCH > A_TILE + 64 * B_PITCH ? 64 * G_PITCH : A_TILE + 64 * B_PITCH;
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = NC == 128 ? wave >> 1 : 0, wn = NC == 128 ? wave & 1 : wave;
    const int fr = lane & 15, fg = lane >> 4;
    const int n0 = blockIdx.x * NC, k0 = blockIdx.y * KC;
    long zs = blockIdx.z;
    if constexpr (ADAM) zs = adam.zsrc[blockIdx.z];
    const bf16_t* A = p.A + zs * p.zA + n0;
    const bf16_t* B = p.B + zs * p.zB + k0;
    float* C = ADAM ? nullptr : p.C + blockIdx.z * p.zC;
    u16x8 ra[A_PIECES], rb[B_PIECES];
    // piece q = tid + 256 i of a tile: row q / chunks, 16-byte chunk q % chunks
    auto row_off = [&](long m, long ld, long gs) -> long {
        if (GRP) {
            const int g = (int)m / p.grp;   // (M < 2^31: launcher)
            return g * gs + ((int)m - g * p.grp) * ld;
        }
        return m * ld;
    };
    auto fetch = [&](int kt) {
#pragma unroll
        for (int i = 0; i < A_PIECES; ++i) {
            const int q = tid + 256 * i;
            const long m = (long)kt * 64 + q / A_CH;
            ra[i] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (m < p.M) ra[i] = *reinterpret_cast<const u16x8*>(A + row_off(m, p.lda, p.gA) + (q % A_CH) * 8);
        }
#pragma unroll
        for (int i = 0; i < B_PIECES; ++i) {
            const int q = tid + 256 * i;
            const long m = (long)kt * 64 + q / B_CH;
            rb[i] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (m < p.M) rb[i] = *reinterpret_cast<const u16x8*>(B + row_off(m, p.ldb, p.gB) + (q % B_CH) * 8);
        }
    };
    // (a thread's A pieces are always chunk tid % A_CH of rows tid / A_CH + (256 / A_CH) i: eight columns of its own)
    float* const db = NC == 128 && !GRP && blockIdx.y == 0 && (int)blockIdx.z < bias.n ? bias.db[blockIdx.z] : nullptr;   // workgroup-uniform
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto park = [&]() {
#pragma unroll
        for (int i = 0; i < A_PIECES; ++i) {
            const int q = tid + 256 * i;
            *reinterpret_cast<u16x8*>(lds + (q / A_CH) * A_PITCH + (q % A_CH) * 16) = ra[i];
            if (db) {   // (rows past M were fetched as zeros)
#pragma unroll
                for (int e = 0; e < 8; ++e) cs[e] += Op<FMT_BF16>::unpack(ra[i][e]);
            }
        }
#pragma unroll
        for (int i = 0; i < B_PIECES; ++i) {
            const int q = tid + 256 * i;
            u16x8 b = rb[i];
            if (BFMT != FMT_BF16) {
#pragma unroll
                for (int e = 0; e < 8; ++e) b[e] = Op<FMT_BF16>::pack(Op<BFMT>::unpack(b[e]));
            }
            *reinterpret_cast<u16x8*>(lds + A_TILE + (q / B_CH) * B_PITCH + (q % B_CH) * 16) = b;
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // this lane's address inside a 16-row, 16-column block for the transposed read
    const char* a_base = lds + (4 * fg + (fr >> 2)) * A_PITCH + (fr & 3) * 8 + wm * 128;   // columns wm * 64 .. of the A tile
    const char* b_base = lds + A_TILE + (4 * fg + (fr >> 2)) * B_PITCH + (fr & 3) * 8 + wn * 128;
    auto frag = [&](const char* base, int pitch, int step, int blk) {
        const char* q = base + step * 32 * pitch + blk * 32;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(q));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(q + 16 * pitch));
        return __builtin_bit_cast(u16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    const int nk = (p.M + 63) / 64;
    fetch(0);
    // (the barriers of the loop are raw ones behind a wait for the LDS traffic only: __syncthreads() would also wait
    // for the global loads in flight)
    for (int kt = 0; kt < nk; ++kt) {
        park();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + 1 < nk) fetch(kt + 1);
#pragma unroll
        for (int step = 0; step < 2; ++step) {
            u16x8 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = frag(a_base, A_PITCH, step, i);
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = frag(b_base, B_PITCH, step, j);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = Op<FMT_BF16>::mfma(bf[j], af[i], acc[i][j]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (db) {   // the 16 row groups' sums of a column, added in row-group order (the loop's last barrier is behind all LDS reads)
        float* red = reinterpret_cast<float*>(lds);
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(tid >> 4) * 128 + (tid & 15) * 8 + e] = cs[e];
        __syncthreads();
        if (tid < 128) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) t += red[r * 128 + tid];
            db[n0 + tid] = p.alpha * t;
        }
        __syncthreads();
    }
    // lane (fr, fg) holds C[n0 + wm 64 + 16 i + fr][k0 + wn 64 + 16 j + 4 fg .. + 3]
    if constexpr (ADAM) {
        static_assert(!ADAM || NC == 128, "the fused step is built for the 128 x 128 tile");
        const AdamConsts c = adam_consts(adam.h, adam.step[blockIdx.z]);
        // The tile in two halves of 64 rows: the two waves that own a half park it in LDS (32 KiB: inside the operand
        // tiles' memory; two workgroups share a CU and one's step overlaps the other's products), then all four waves
        // take its step.  Piece q = tid + 256 it of a half: row q / 32, 16-byte chunk q % 32 -- a wave instruction covers
        // two whole 512-byte rows of the tensor.  Four pieces per thread and array in flight: a slot's next piece is
        // requested as soon as its present one is stored; every value is touched once per step, hence the non-temporal
        // hints (as in adam_kernel).
        const bool copies = adam.copy[blockIdx.z] != nullptr;
        // pieces per thread and array in flight: piece k lives in slot k % NS.  (Measured on the training step, with
        // the copies: two workgroups per CU and 4 / 5 / 6 / 7 / 8 slots 10.00 / 10.02 / 10.03 / 10.04 / 10.45 ms; three
        // workgroups per CU -- 168 registers, spills -- and 4 / 6 slots 10.20 / 10.66 ms.)
        constexpr int NS = 4;
        f32x4 pp[NS], mm[NS], vv[NS];
        auto off_of = [&](int k) -> long {   // piece k of the tile: half k / 8, piece k % 8 of it
            const int q = tid + 256 * (k & 7);
            return (long)(n0 + 64 * (k >> 3) + (q >> 5)) * p.ldc + k0 + (q & 31) * 4;
        };
        auto get = [&](int k) {
            const long off = off_of(k);
            pp[k % NS] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(adam.p[blockIdx.z] + off));
            mm[k % NS] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(adam.m[blockIdx.z] + off));
            vv[k % NS] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(adam.v[blockIdx.z] + off));
        };
#pragma unroll
        for (int k = 0; k < NS; ++k) get(k);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            // (half 0: the loop's last barrier is behind every wave's fragment reads, the operand tiles may be overwritten)
            if (half == 1) __syncthreads();
            if (wm == half) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int row = i * 16 + fr, chunk = wn * 16 + j * 4 + fg;
                        *reinterpret_cast<f32x4*>(lds + row * G_PITCH + ((chunk ^ (row & 31)) << 4)) = acc[i][j] * p.alpha;
                    }
            }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int k = 8 * half + kk;
                const int q = tid + 256 * kk;
                const f32x4 g4 = *reinterpret_cast<const f32x4*>(lds + (q >> 5) * G_PITCH + (((q & 31) ^ ((q >> 5) & 31)) << 4));
                f32x4 pq = pp[k % NS], mq = mm[k % NS], vq = vv[k % NS];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float pr = pq[r], mr = mq[r], vr = vq[r];
                    adam_one(pr, g4[r], mr, vr, c);
                    pq[r] = pr;
                    mq[r] = mr;
                    vq[r] = vr;
                }
                const long off = off_of(k);
                __builtin_nontemporal_store(pq, reinterpret_cast<f32x4*>(adam.p[blockIdx.z] + off));
                __builtin_nontemporal_store(mq, reinterpret_cast<f32x4*>(adam.m[blockIdx.z] + off));
                __builtin_nontemporal_store(vq, reinterpret_cast<f32x4*>(adam.v[blockIdx.z] + off));
                if (k + NS < 16) get(k + NS);   // the slot's next piece
                if (copies) {   // (workgroup-uniform) the forward's operand now, and its bfloat16 form parked where g was
                    u16x4 cp, tq;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float sv = pq[r] * adam.scale;
                        if (adam.copy_fmt == FMT_F16) {
                            cp[r] = Op<FMT_F16>::pack(sv);
                            tq[r] = Op<FMT_BF16>::pack(Op<FMT_F16>::unpack(cp[r]));
                        } else {
                            cp[r] = tq[r] = Op<FMT_BF16>::pack(sv);
                        }
                    }
                    *reinterpret_cast<u16x4*>(adam.copy[blockIdx.z] + off) = cp;
                    *reinterpret_cast<u16x4*>(lds + (q >> 5) * G_PITCH + (((q & 31) ^ ((q >> 5) & 31)) << 4)) = tq;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (copies) {
                // the half's 64 (n) x 128 (k) bfloat16 values, transposed: a thread gathers eight consecutive n of one
                // k (eight lanes cover the half's 128 bytes of a row of outT, a wave instruction eight such rows)
                __syncthreads();
                const int nch = tid & 7;
                bf16_t* const dstT = adam.outT[blockIdx.z] + (long)k0 * adam.ldo + n0 + 64 * half + nch * 8;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int kk2 = (tid >> 3) + 32 * it;
                    u16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int n = nch * 8 + e;
                        o[e] = *reinterpret_cast<const bf16_t*>(lds + n * G_PITCH + ((((kk2 >> 2) ^ (n & 31))) << 4) + (kk2 & 3) * 2);
                    }
                    *reinterpret_cast<u16x8*>(dstT + (long)kk2 * adam.ldo) = o;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float* row = C + (long)(n0 + wm * 64 + i * 16 + fr) * p.ldc + k0 + wn * 64 + 4 * fg;
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(row + j * 16) = acc[i][j] * p.alpha;
    }
}

// Column sums of bfloat16 matrices [M][ld] that lie z_in elements apart (blockIdx.y = matrix): dst[z][n] = scale * sum
// over m of in_z[m][n], n < N; 16 waves per 64 columns, added in a fixed order (colsum_one_kernel's).
struct ColsumLayersOut {
    float* p[COLSUM_LAYERS_PER_LAUNCH];
};
__global__ __launch_bounds__(1024) void colsum_layers_kernel(const bf16_t* __restrict__ in, long z_in, long ld, long M,
                                                             float scale, ColsumLayersOut o) {
    __shared__ float red[16][64];
    float* const out = o.p[blockIdx.y];
    if (!out) return;
    const int col = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + col;
    const bf16_t* src = in + blockIdx.y * z_in + n;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    long m = rg;
    for (; m + 48 < M; m += 64) {
        acc[0] += Op<FMT_BF16>::unpack(src[m * ld]);
        acc[1] += Op<FMT_BF16>::unpack(src[(m + 16) * ld]);
        acc[2] += Op<FMT_BF16>::unpack(src[(m + 32) * ld]);
        acc[3] += Op<FMT_BF16>::unpack(src[(m + 48) * ld]);
    }
    for (; m < M; m += 16) acc[0] += Op<FMT_BF16>::unpack(src[m * ld]);
    red[rg][col] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (rg == 0) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += red[r][col];
        out[n] = scale * t;
    }
}

// g (bfloat16, in place) *= gelu'(pre), pre of format FMT; eight values per thread
template <int FMT>
__global__ __launch_bounds__(256) void gelu_bwd16_kernel(const bf16_t* __restrict__ pre, bf16_t* __restrict__ g, long n8) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const u16x8 pv = *reinterpret_cast<const u16x8*>(pre + i * 8);
        u16x8 gv = *reinterpret_cast<const u16x8*>(g + i * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            gv[e] = Op<FMT_BF16>::pack(Op<FMT_BF16>::unpack(gv[e]) * gelu_grad(Op<FMT>::unpack(pv[e])));
        *reinterpret_cast<u16x8*>(g + i * 8) = gv;
    }
}

// AB_PITCH (template argument of the kernel): bytes per LDS row (128 used).  160 where the four matrices fit (<= 224
// rows): eight consecutive rows then start on eight disjoint 8-bank windows for the transposed reads; 144 up to 256 rows.
constexpr int AB_WAVES = 8;    // per workgroup: the 14 tiles of 201 frames (padded to 224) take two rounds per phase; 16 waves spill

__device__ __forceinline__ f32x4 mfma16_bf16(s16x4 a, s16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}

template <int FMT, int AB_PITCH>
__global__ __launch_bounds__(AB_WAVES * 64) void attn_bwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                       const bf16_t* __restrict__ dout, bf16_t* __restrict__ dqkv,
                                                       int T, const uint8_t* __restrict__ keep_qk,
                                                       const uint8_t* __restrict__ keep_kq, int pitch, float drop_scale) {
    // keep_qk / keep_kq (null: no dropout): the forward's dropout on the attention probabilities, 1 = kept, as
    // [(utterance, head)][query][key] and [(utterance, head)][key][query] bytes (row pitch `pitch`), kept ones scaled
    // by drop_scale: O = (P o M) V, so dV = (P o M)^T dO and dP = M o (dO V^T); D = dO . O holds as it is.
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int Tp = (T + 31) & ~31;  // rows held (zero beyond T): whole 32-key steps
    char* Qs = lds;
    char* Ks = Qs + Tp * AB_PITCH;
    char* Vs = Ks + Tp * AB_PITCH;
    char* Ds = Vs + Tp * AB_PITCH;  // dO
    float* st_m = reinterpret_cast<float*>(Ds + Tp * AB_PITCH);  // row maximum
    float* st_i = st_m + Tp;                                     // 1 / row sum
    float* st_d = st_i + Tp;                                     // D
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x >> 4, h = blockIdx.x & 15;
    const int S = gridDim.y, z = blockIdx.y;
    const long row0 = (long)b * T;

    // ---- stage the head's rows (bfloat16), zero rows beyond T; D_i.  Tp <= 256 rows of eight 16-byte pieces are at
    // most four rounds of the workgroup: every round's loads are issued before the first is used.
    {
        u16x8 qv[4], kv[4], vv[4], dv[4], ov[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int i = tid + it * AB_WAVES * 64;
            const int r = min(i >> 3, T - 1), c = i & 7;   // clamped: the loads stay unconditional
            const bf16_t* src = qkv + (row0 + r) * 3072 + h * 64 + c * 8;
            qv[it] = *reinterpret_cast<const u16x8*>(src);
            kv[it] = *reinterpret_cast<const u16x8*>(src + 1024);
            vv[it] = *reinterpret_cast<const u16x8*>(src + 2048);
            dv[it] = *reinterpret_cast<const u16x8*>(dout + (row0 + r) * 1024 + h * 64 + c * 8);
            ov[it] = *reinterpret_cast<const u16x8*>(o + (row0 + r) * 1024 + h * 64 + c * 8);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int i = tid + it * AB_WAVES * 64;
            if (i >= Tp * 8) break;   // whole waves leave together (Tp * 8 is a multiple of the wave size)
            const int r = i >> 3, c = i & 7;
            u16x8 q = {0, 0, 0, 0, 0, 0, 0, 0}, k = q, v = q, d = q;
            float s = 0.f;
            if (r < T) {
                q = qv[it]; k = kv[it]; v = vv[it]; d = dv[it];
                if (FMT != FMT_BF16) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        q[e] = Op<FMT_BF16>::pack(Op<FMT>::unpack(q[e]));
                        k[e] = Op<FMT_BF16>::pack(Op<FMT>::unpack(k[e]));
                        v[e] = Op<FMT_BF16>::pack(Op<FMT>::unpack(v[e]));
                    }
                }
                // D_r = dO_r . O_r: the eight lanes of a row hold eight columns each
#pragma unroll
                for (int e = 0; e < 8; ++e) s = fmaf(Op<FMT>::unpack(ov[it][e]), Op<FMT_BF16>::unpack(d[e]), s);
            }
            *reinterpret_cast<u16x8*>(Qs + r * AB_PITCH + c * 16) = q;
            *reinterpret_cast<u16x8*>(Ks + r * AB_PITCH + c * 16) = k;
            *reinterpret_cast<u16x8*>(Vs + r * AB_PITCH + c * 16) = v;
            *reinterpret_cast<u16x8*>(Ds + r * AB_PITCH + c * 16) = d;
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 4, 64);
            if (c == 0) st_d[r] = s;
        }
    }
    __syncthreads();

    const int fr = lane & 15, fg = lane >> 4;
    const int NT = Tp >> 4;  // 16-row tiles (even)
    // fragment of rows [tile*16, +16) for the 16x16x32 product: 8 consecutive d of row fr at d = 32 s + 8 fg
    auto frag = [&](const char* base, int tile, int s) {
        return *reinterpret_cast<const u16x8*>(base + (tile * 16 + fr) * AB_PITCH + s * 64 + fg * 16);
    };
    // transposed fragment: the four rows tile16*16 + 4 fg .. + 3 of column dt*16 + fr
    auto frag_t = [&](const char* base, int tile16, int dt) {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (lds_s16x4_ptr)(base + (tile16 * 16 + 4 * fg + (fr >> 2)) * AB_PITCH + dt * 32 + (fr & 3) * 8));
    };

    // ================= phase A: statistics and dQ, one 16-query tile per wave iteration
    for (int qt = wave; qt < NT; qt += AB_WAVES) {
        if (qt * 16 >= T) break;
        const bool own = qt % S == z;  // wave-uniform
        const u16x8 q0 = frag(Qs, qt, 0), q1 = frag(Qs, qt, 1);
        const u16x8 g0 = frag(Ds, qt, 0), g1 = frag(Ds, qt, 1);
        f32x4 sc[16], dp[16];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 16; ++kt) {
            if (kt < NT) {
                f32x4 a = Op<FMT_BF16>::mfma(frag(Ks, kt, 0), q0, f32x4{0.f, 0.f, 0.f, 0.f});
                a = Op<FMT_BF16>::mfma(frag(Ks, kt, 1), q1, a);
                f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
                if (own) {
                    d = Op<FMT_BF16>::mfma(frag(Vs, kt, 0), g0, d);
                    d = Op<FMT_BF16>::mfma(frag(Vs, kt, 1), g1, d);
                    if (keep_qk) {  // lane: query qt*16 + fr, keys kt*16 + 4 fg + r
                        const uint32_t m = *reinterpret_cast<const uint32_t*>(
                            keep_qk + ((long)blockIdx.x * T + min(qt * 16 + fr, T - 1)) * pitch + kt * 16 + fg * 4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) d[r] = ((m >> (8 * r)) & 0xFFu) ? d[r] * drop_scale : 0.f;
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (kt * 16 + fg * 4 + r >= T) a[r] = -INFINITY;
                    mx = fmaxf(mx, a[r]);
                }
                sc[kt] = a;
                dp[kt] = d;
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 16; ++kt)
            if (kt < NT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sc[kt][r] = __expf(sc[kt][r] - mx);
                    sum += sc[kt][r];
                }
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        const int q = qt * 16 + fr;
        const float dq_d = st_d[q];
        if (fg == 0) {
            st_m[q] = mx;
            st_i[q] = inv;
        }
        if (!own) continue;  // another workgroup of this head writes this tile's dQ
        f32x4 acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            if (2 * ks < NT) {
                f32x4 e0, e1;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    e0[r] = sc[2 * ks][r] * inv * (dp[2 * ks][r] - dq_d);
                    e1[r] = sc[2 * ks + 1][r] * inv * (dp[2 * ks + 1][r] - dq_d);
                }
                const u16x8 pf = __builtin_bit_cast(
                    u16x8, u32x4{Op<FMT_BF16>::pack2(e0[0], e0[1]), Op<FMT_BF16>::pack2(e0[2], e0[3]),
                                 Op<FMT_BF16>::pack2(e1[0], e1[1]), Op<FMT_BF16>::pack2(e1[2], e1[3])});
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const s16x4 lo = frag_t(Ks, 2 * ks, nt), hi = frag_t(Ks, 2 * ks + 1, nt);
                    const u16x8 kk = __builtin_bit_cast(u16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                    acc[nt] = Op<FMT_BF16>::mfma(kk, pf, acc[nt]);
                }
            }
        if (q < T) {
            bf16_t* dst = dqkv + (row0 + q) * 3072 + h * 64 + fg * 4;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                u16x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = Op<FMT_BF16>::pack(acc[nt][r]);
                *reinterpret_cast<u16x4*>(dst + nt * 16) = v;
            }
        }
    }
    __syncthreads();

    // ================= phase B: dK and dV, one 16-key tile per wave iteration
    for (int kt = z + S * wave; kt < NT; kt += S * AB_WAVES) {
        if (kt * 16 >= T) break;
        const u16x8 k0 = frag(Ks, kt, 0), k1 = frag(Ks, kt, 1);
        const u16x8 v0 = frag(Vs, kt, 0), v1 = frag(Vs, kt, 1);
        f32x4 ak[4], av[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) ak[nt] = av[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int qt = 0; qt < NT; ++qt) {
            if (qt * 16 >= T) break;
            f32x4 a = Op<FMT_BF16>::mfma(frag(Qs, qt, 0), k0, f32x4{0.f, 0.f, 0.f, 0.f});
            a = Op<FMT_BF16>::mfma(frag(Qs, qt, 1), k1, a);
            f32x4 d = Op<FMT_BF16>::mfma(frag(Ds, qt, 0), v0, f32x4{0.f, 0.f, 0.f, 0.f});
            d = Op<FMT_BF16>::mfma(frag(Ds, qt, 1), v1, d);
            // lane: key kt*16 + fr, queries qt*16 + 4 fg + r
            const int qb = qt * 16 + fg * 4;
            const f32x4 m4 = *reinterpret_cast<const f32x4*>(st_m + qb);
            const f32x4 i4 = *reinterpret_cast<const f32x4*>(st_i + qb);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(st_d + qb);
            float p[4], ds[4];
            uint32_t km = 0x01010101u;
            if (keep_kq)  // lane: key kt*16 + fr, queries qb .. qb + 3
                km = *reinterpret_cast<const uint32_t*>(keep_kq + ((long)blockIdx.x * T + min(kt * 16 + fr, T - 1)) * pitch + qb);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p[r] = qb + r < T ? __expf(a[r] - m4[r]) * i4[r] : 0.f;
                const float mk = ((km >> (8 * r)) & 0xFFu) ? (keep_kq ? drop_scale : 1.f) : 0.f;
                ds[r] = p[r] * (mk * d[r] - d4[r]);
                p[r] *= mk;   // what multiplied V in the forward
            }
            const s16x4 pb = __builtin_bit_cast(s16x4, u32x2{Op<FMT_BF16>::pack2(p[0], p[1]), Op<FMT_BF16>::pack2(p[2], p[3])});
            const s16x4 sb = __builtin_bit_cast(s16x4, u32x2{Op<FMT_BF16>::pack2(ds[0], ds[1]), Op<FMT_BF16>::pack2(ds[2], ds[3])});
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                av[nt] = mfma16_bf16(frag_t(Ds, qt, nt), pb, av[nt]);
                ak[nt] = mfma16_bf16(frag_t(Qs, qt, nt), sb, ak[nt]);
            }
        }
        const int key = kt * 16 + fr;
        if (key < T) {
            bf16_t* dst = dqkv + (row0 + key) * 3072 + h * 64 + fg * 4;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                u16x4 vk, vv;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    vk[r] = Op<FMT_BF16>::pack(ak[nt][r]);
                    vv[r] = Op<FMT_BF16>::pack(av[nt][r]);
                }
                *reinterpret_cast<u16x4*>(dst + 1024 + nt * 16) = vk;
                *reinterpret_cast<u16x4*>(dst + 2048 + nt * 16) = vv;
            }
        }
    }
}

// ---------------------------------------------------------------- in-tree head, backward
// z = W2 (mask o relu(W1 LN(pooled) + b1)) + b2, logp = log_softmax(z)  (model.py:183-189, :246).
// (0) statistics of the classifier's LayerNorm per utterance: stats[b] = (mean, rstd)
__global__ __launch_bounds__(256) void cls_stats_kernel(const float* __restrict__ pooled, int dict,
                                                        float* __restrict__ stats) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const float* row = pooled + (long)b * dict;
    float s = 0.f;
    for (int j = threadIdx.x; j < dict; j += 256) s += row[j];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)dict;
    __syncthreads();
    float q = 0.f;
    for (int j = threadIdx.x; j < dict; j += 256) {
        const float d = row[j] - mean;
        q = fmaf(d, d, q);
    }
    q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
    __syncthreads();
    if (threadIdx.x == 0) {
        stats[b * 2] = mean;
        stats[b * 2 + 1] = 1.0f / sqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)dict + 1e-5f);
    }
}

// (1) one workgroup, thread n = hidden unit: gradients of the last Linear and at the hidden layer.
// hidden = relu output before the dropout mask; mask (nullable) holds 0 or 1 / (1 - p).
__global__ __launch_bounds__(256) void cls_bwd_out_kernel(const float* __restrict__ dlogp, const float* __restrict__ logp,
                                                          const float* __restrict__ hidden, const float* __restrict__ mask,
                                                          const float* __restrict__ w2, int B, float* __restrict__ dw2,
                                                          float* __restrict__ db2, float* __restrict__ db1,
                                                          float* __restrict__ dh1) {
    const int n = threadIdx.x;
    float a0 = 0.f, a1 = 0.f, s0 = 0.f, s1 = 0.f, sb = 0.f;
    for (int b = 0; b < B; ++b) {
        const float g0 = dlogp[b * 2], g1 = dlogp[b * 2 + 1];
        const float dz0 = g0 - __expf(logp[b * 2]) * (g0 + g1), dz1 = g1 - __expf(logp[b * 2 + 1]) * (g0 + g1);
        const float h = hidden[b * 256 + n], mk = mask ? mask[b * 256 + n] : 1.0f;
        a0 = fmaf(dz0, h * mk, a0);
        a1 = fmaf(dz1, h * mk, a1);
        s0 += dz0;
        s1 += dz1;
        const float d = h > 0.f ? (dz0 * w2[n] + dz1 * w2[256 + n]) * mk : 0.f;
        dh1[b * 256 + n] = d;
        sb += d;
    }
    if (dw2) {
        dw2[n] = a0;
        dw2[256 + n] = a1;
    }
    if (db2 && n < 2) db2[n] = n ? s1 : s0;
    if (db1) db1[n] = sb;
}

// (2) thread j = input feature: dW1[n][j] = sum_b dh1[b][n] hb[b][j], dhb[b][j] = sum_n dh1[b][n] W1[n][j],
// with hb = LN(pooled) re-computed from the statistics; also the LayerNorm's dgamma / dbeta.  B <= 64.
__global__ __launch_bounds__(256) void cls_bwd_fc1_kernel(const float* __restrict__ pooled, const float* __restrict__ stats,
                                                          const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                          const float* __restrict__ w1, const float* __restrict__ dh1,
                                                          int B, int dict, float* __restrict__ dw1,
                                                          float* __restrict__ dhb, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;  // one wave per workgroup: 64 workgroups for 4096 columns
    if (j >= dict) return;
    const float g = ln_g[j], be = ln_b[j];
    float sg = 0.f, sb = 0.f;
    // eight utterances at a time in registers; dw1 collects over the passes
    for (int b0 = 0; b0 < B; b0 += 8) {
        float xh[8], acc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int b = b0 + u < B ? b0 + u : B - 1;
            xh[u] = (pooled[(long)b * dict + j] - stats[b * 2]) * stats[b * 2 + 1];
            acc[u] = 0.f;
        }
#pragma unroll 4
        for (int n = 0; n < 256; ++n) {
            const float w = w1[(long)n * dict + j];
            float d = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float h = b0 + u < B ? dh1[(b0 + u) * 256 + n] : 0.f;
                d = fmaf(h, fmaf(xh[u], g, be), d);
                acc[u] = fmaf(h, w, acc[u]);
            }
            if (dw1) {
                float* o = dw1 + (long)n * dict + j;
                *o = b0 == 0 ? d : *o + d;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (b0 + u < B) {
                dhb[(long)(b0 + u) * dict + j] = acc[u];
                sg = fmaf(acc[u], xh[u], sg);
                sb += acc[u];
            }
        }
    }
    if (dgamma) dgamma[j] = sg;
    if (dbeta) dbeta[j] = sb;
}

// (3) one workgroup per utterance: LayerNorm backward over the pooled row -> dpool[b][:]
__global__ __launch_bounds__(256) void cls_bwd_ln_kernel(const float* __restrict__ pooled, const float* __restrict__ stats,
                                                         const float* __restrict__ ln_g, const float* __restrict__ dhb,
                                                         int dict, float* __restrict__ dpool) {
    __shared__ float red[2][4];
    const int b = blockIdx.x;
    const float mean = stats[b * 2], rstd = stats[b * 2 + 1];
    float m1 = 0.f, m2 = 0.f;
    for (int j = threadIdx.x; j < dict; j += 256) {
        const float g = dhb[(long)b * dict + j] * ln_g[j];
        m1 += g;
        m2 = fmaf(g, (pooled[(long)b * dict + j] - mean) * rstd, m2);
    }
    m1 = wave_sum(m1);
    m2 = wave_sum(m2);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = m1;
        red[1][threadIdx.x >> 6] = m2;
    }
    __syncthreads();
    m1 = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (float)dict;
    m2 = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (float)dict;
    for (int j = threadIdx.x; j < dict; j += 256) {
        const float g = dhb[(long)b * dict + j] * ln_g[j];
        const float xh = (pooled[(long)b * dict + j] - mean) * rstd;
        dpool[(long)b * dict + j] = rstd * (g - m1 - xh * m2);
    }
}

// Gradient at the encoder's pre-activation: the time mean spreads dpool / T over the frames (dpool nullable: the
// classifier does not read the code), the decoder path adds dcode (nullable); ReLU and TopK -- per frame or by
// window votes -- pass it where the code is non-zero (model.py:70-79, model_window_topk.py:196).
__global__ __launch_bounds__(256) void sae_code_grad_kernel(const float* __restrict__ code, const float* __restrict__ dpool,
                                                            const float* __restrict__ dcode, bf16_t* __restrict__ denc,
                                                            long M, int T, int dict) {
    const long n = M * dict;
    const float invT = 1.0f / (float)T;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long m = i / dict;
        const int d = (int)(i - m * dict);
        float g = 0.f;
        if (code[i] != 0.f) g = (dpool ? dpool[(m / T) * dict + d] * invT : 0.f) + (dcode ? dcode[i] : 0.f);
        denc[i] = Op<FMT_BF16>::pack(g);
    }
}

// loss = mean((recon + b_dec - x)^2) (model.py:224-225): drecon = dloss * 2 (recon + b_dec - x) / n as
// bfloat16, and dx = -drecon as fp32 (the target of the loss is not detached in the reference)
__global__ __launch_bounds__(256) void mse_grad_kernel(const float* __restrict__ recon, const float* __restrict__ b_dec,
                                                       const float* __restrict__ x, const float* __restrict__ dloss,
                                                       long n, bf16_t* __restrict__ drecon, float* __restrict__ dx) {
    const float coef = dloss[0] * 2.0f / (float)n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float g = coef * (recon[i] + b_dec[i & 1023] - x[i]);
        drecon[i] = Op<FMT_BF16>::pack(g);
        dx[i] = -g;
    }
}

// The reconstruction as the classifier's input (use_sparse_features=False, model.py:228-233): the gradient at
// recon + b_dec is the time mean's share dpool[b] / T plus, with the loss (dloss nullable), its
// dloss * 2 (recon + b_dec - x) / n.  drecon: bfloat16, total: fp32 (for b_dec's column sum), dx = minus the loss's
// share (the loss's target is not detached in the reference).
__global__ __launch_bounds__(256) void recon_grad_kernel(const float* __restrict__ recon, const float* __restrict__ b_dec,
                                                         const float* __restrict__ x, const float* __restrict__ dloss,
                                                         const float* __restrict__ dpool, long n, int T,
                                                         bf16_t* __restrict__ drecon, float* __restrict__ dx,
                                                         float* __restrict__ total) {
    const float coef = dloss ? dloss[0] * 2.0f / (float)n : 0.f;
    const float invT = 1.0f / (float)T;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i & 1023);
        const float g = dloss ? coef * (recon[i] + b_dec[c] - x[i]) : 0.f;
        const float t = fmaf(dpool[((i >> 10) / T) * 1024 + c], invT, g);
        drecon[i] = Op<FMT_BF16>::pack(t);
        total[i] = t;
        dx[i] = -g;
    }
}

// out[m][c] = dpool[m / T][c] / T: back through a mean over the T frames of every clip (rows of 1024).
__global__ __launch_bounds__(256) void spread_rows_kernel(const float* __restrict__ dpool, long n, int T, float* __restrict__ out) {
    const float invT = 1.0f / (float)T;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        out[i] = dpool[((i >> 10) / T) * 1024 + (i & 1023)] * invT;
}

__global__ __launch_bounds__(256) void add_f32_kernel(float* __restrict__ a, const float* __restrict__ b, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) a[i] += b[i];
}

// out = (resid ? resid : 0) + a * m: a dropout mask's multipliers applied on the way into (or back out of) a residual add
__global__ __launch_bounds__(256) void mul_f32_kernel(float* out, const float* a, const float* __restrict__ m,
                                                      const float* resid, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        out[i] = (resid ? resid[i] : 0.f) + a[i] * m[i];
}

template <int FMT>
__global__ __launch_bounds__(256) void mul16_f32_kernel(bf16_t* __restrict__ x, const float* __restrict__ m, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        x[i] = Op<FMT>::pack(Op<FMT>::unpack(x[i]) * m[i]);
}

__global__ __launch_bounds__(256) void cvt_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        out[i] = Op<FMT_BF16>::pack(in[i]);
}

// ---------------------------------------------------------------- positional conv, backward
// x1 = x0 + gelu(u), u = grouped conv of x0 (wav2vec2.py:915-917).  du = dx1 o gelu'(u), written twice:
// compact bfloat16 [B T][1024] and packed like the forward's input, [B][16][R][64] with du of frame t
// in slab row 64 + t and zeros around, for the transposed conv.
__global__ __launch_bounds__(256) void posconv_grad_pack_kernel(const float* __restrict__ dx, const float* __restrict__ u,
                                                                bf16_t* __restrict__ du, bf16_t* __restrict__ dug,
                                                                int B, int T, int R) {
    const long total = (long)B * 16 * R * 8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int piece = (int)(i & 7);
        long rest = i >> 3;
        const int r = (int)(rest % R);
        rest /= R;
        const int g = (int)(rest & 15);
        const int b = (int)(rest >> 4);
        u16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
        const int t = r - 64;
        if (t >= 0 && t < T) {
            const long off = ((long)b * T + t) * 1024 + g * 64 + piece * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = Op<FMT_BF16>::pack(dx[off + e] * gelu_grad(u[off + e]));
            *reinterpret_cast<u16x8*>(du + off) = o;
        }
        *reinterpret_cast<u16x8*>(dug + i * 8) = o;
    }
}

// Weight of the transposed conv: wf[g][ci][127 - j][co] = w[g][co][j][ci] (w: the forward's packed
// weight of format FMT, [16][64][128][64]) as bfloat16.
template <int FMT>
__global__ __launch_bounds__(256) void posconv_flip_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ wf) {
    const long n = 16L * 64 * 128 * 64;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int co = (int)(i & 63), jf = (int)((i >> 6) & 127), ci = (int)((i >> 13) & 63), g = (int)(i >> 19);
        wf[i] = Op<FMT_BF16>::pack(Op<FMT>::unpack(w[(((long)g * 64 + co) * 128 + (127 - jf)) * 64 + ci]));
    }
}

// weight_norm(dim=2) backward (wav2vec2.py:874): w[co][ci][j] = g[j] v[co][ci][j] / n[j], n[j] = ||v[:, :, j]||.
// dwf: gradient at the folded weight in the library's layout [co][j][ci] (co over all 1024 outputs); v and dv in
// the reference's layout [co][ci][j].  dg[j] = sum dw v / n,  dv = (g / n) (dw - v (sum dw v) / n^2).
// Three launches, every global access contiguous: per output channel co the [128 j][64 ci] slice of dwf goes
// through LDS so that a thread owns one tap j on both sides; (1) per-co partial sums of dw v and v v for every
// tap, (2) their sums over co in index order, (3) dv.
constexpr int WN_TAPS = 128, WN_CI = 64, WN_CO = 1024;

__device__ __forceinline__ void wn_load_dw_tile(const float* __restrict__ dwf, int co, float (*dwt)[WN_CI + 1]) {
    const float* src = dwf + (long)co * WN_TAPS * WN_CI;
    for (int e = threadIdx.x; e < WN_TAPS * WN_CI; e += 256) dwt[e >> 6][e & 63] = src[e];
}

__global__ __launch_bounds__(256) void weightnorm_partial_kernel(const float* __restrict__ dwf, const float* __restrict__ v,
                                                                 float* __restrict__ partial) {
    __shared__ float dwt[WN_TAPS][WN_CI + 1];
    __shared__ float red[2][WN_TAPS];
    const int co = blockIdx.x;
    wn_load_dw_tile(dwf, co, dwt);
    __syncthreads();
    const int j = threadIdx.x & 127, half = threadIdx.x >> 7;
    const float* vrow = v + (long)co * WN_CI * WN_TAPS + j;
    float sdv = 0.f, svv = 0.f;
#pragma unroll 8
    for (int ci = half * 32; ci < half * 32 + 32; ++ci) {
        const float vv = vrow[ci * WN_TAPS];
        sdv = fmaf(dwt[j][ci], vv, sdv);
        svv = fmaf(vv, vv, svv);
    }
    if (half == 1) {
        red[0][j] = sdv;
        red[1][j] = svv;
    }
    __syncthreads();
    if (half == 0) {
        partial[(long)co * 2 * WN_TAPS + j] = sdv + red[0][j];
        partial[(long)co * 2 * WN_TAPS + WN_TAPS + j] = svv + red[1][j];
    }
}

// stat[j] = sum over co of partial[co][0][j], stat[128 + j] = the same of partial[co][1][j]; dg
__global__ __launch_bounds__(1024) void weightnorm_reduce_kernel(const float* __restrict__ partial, float* __restrict__ stat,
                                                                 float* __restrict__ dg) {
    __shared__ float red[4][2 * WN_TAPS];
    const int col = threadIdx.x & 255, grp = threadIdx.x >> 8;  // four groups of 256 output channels
    float a4[4] = {0.f, 0.f, 0.f, 0.f};  // four interleaved running sums: sixteen loads in flight
#pragma unroll 4
    for (int co = grp * 256; co < grp * 256 + 256; co += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) a4[u] += partial[(long)(co + u) * 2 * WN_TAPS + col];
    }
    red[grp][col] = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    __syncthreads();
    if (grp == 0) {
        const float t = ((red[0][col] + red[1][col]) + red[2][col]) + red[3][col];
        stat[col] = t;
        red[0][col] = t;
    }
    __syncthreads();
    if (threadIdx.x < WN_TAPS && dg) dg[threadIdx.x] = red[0][threadIdx.x] / sqrtf(red[0][WN_TAPS + threadIdx.x]);
}

__global__ __launch_bounds__(256) void weightnorm_apply_kernel(const float* __restrict__ dwf, const float* __restrict__ v,
                                                               const float* __restrict__ gv, const float* __restrict__ stat,
                                                               float* __restrict__ dv) {
    __shared__ float dwt[WN_TAPS][WN_CI + 1];
    const int co = blockIdx.x;
    wn_load_dw_tile(dwf, co, dwt);
    __syncthreads();
    const int j = threadIdx.x & 127, half = threadIdx.x >> 7;
    const float sdv = stat[j], svv = stat[WN_TAPS + j];
    const float a = gv[j] / sqrtf(svv), c = sdv / svv;
    const long base = (long)co * WN_CI * WN_TAPS + j;
#pragma unroll 8
    for (int ci = half * 32; ci < half * 32 + 32; ++ci)
        dv[base + ci * WN_TAPS] = a * (dwt[j][ci] - v[base + ci * WN_TAPS] * c);
}


// ---------------------------------------------------------------- conv feature extractor, backward
// out[(b P + t)][c] = t < T ? in[(b T + t)][c] : 0   (compact frames -> padded per-utterance rows)
__global__ __launch_bounds__(256) void expand_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int B,
                                                          int T, int P, int C4) {
    const long n = (long)B * P * C4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C4);
        const long row = i / C4;
        const int b = (int)(row / P), t = (int)(row - (long)b * P);
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t < T) v = reinterpret_cast<const f32x4*>(in)[((long)b * T + t) * C4 + c];
        reinterpret_cast<f32x4*>(out)[i] = v;
    }
}

// Conv1d(512, 512, k, stride) as the flat GEMM y[m] = sum_j a[stride m + j] W_j: the GEMM dcol = dy W gives
// the gradient at every (m, j) operand slice; the gradient at input row r adds the slices that read it:
// da[r][c] = sum over j < k with (r - j) >= 0 and (r - j) % stride == 0 of dcol[(r - j) / stride][j 512 + c],
// inside one utterance (Pin = stride * Pout rows each).
__global__ __launch_bounds__(256) void col2im_kernel(const bf16_t* __restrict__ dcol, float* __restrict__ da, int B,
                                                     int Pin, int Pout, int k, int stride) {
    const long n = (long)B * Pin * 64;  // 8-channel pieces
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int piece = (int)(i & 63);
        const long row = i >> 6;
        const int b = (int)(row / Pin), r = (int)(row - (long)b * Pin);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int j = 0; j < k; ++j) {
            const int d = r - j;
            if (d < 0 || d % stride) continue;
            const int m = d / stride;
            if (m >= Pout) continue;
            const u16x8 v = *reinterpret_cast<const u16x8*>(dcol + (((long)b * Pout + m) * k + j) * 512 + piece * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += Op<FMT_BF16>::unpack(v[e]);
        }
        float* dst = da + row * 512 + piece * 8;
        *reinterpret_cast<f32x4*>(dst) = f32x4{acc[0], acc[1], acc[2], acc[3]};
        *reinterpret_cast<f32x4*>(dst + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
    }
}

// gradient at the packed conv weight [co][k][ci] -> the reference's Conv1d layout [co][ci][k]; the packed gradient
// may come as `slices` partial sums (n elements apart), added here in index order
__global__ __launch_bounds__(256) void conv_grad_unpack_kernel(const float* __restrict__ packed, float* __restrict__ dst,
                                                               int co, int ci, int k, int slices) {
    const long n = (long)co * ci * k;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int j = (int)(i % k);
        const long rest = i / k;
        const int c = (int)(rest % ci), o = (int)(rest / ci);
        const float* src = packed + ((long)o * k + j) * ci + c;
        float v = src[0];
        for (int z = 1; z < slices; ++z) v += src[(long)z * n];
        dst[i] = v;
    }
}

// First conv block (Conv1d(1, 512, k = 10, stride 5) + LayerNorm + GELU), backward: the block's output is
// re-computed per frame from the waveform (as the forward kernel does), so nothing of it is stored.
// da: gradient at the block's output, fp32 [B][P][512].  A wave owns one frame at a time, the lane
// channels 8 lane .. 8 lane + 7; every workgroup leaves its sums in part[workgroup][6656]:
// dW [512][10], dbias [512], dgamma [512], dbeta [512].
constexpr int CONV0_GRAD_N = 512 * 13;
__global__ __launch_bounds__(256) void conv0_bwd_kernel(const float* __restrict__ wav, int L, int T, long P,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, const float* __restrict__ da, int B,
                                                        float* __restrict__ part) {
    __shared__ float red[4][104 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float wr[8][10], br[8], gr[8], be[8];
    float dw[8][10], dbs[8], dgm[8], dbt[8];
    const int c0 = lane * 8;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            wr[c][j] = w[(c0 + c) * 10 + j];
            dw[c][j] = 0.f;
        }
        br[c] = bias[c0 + c];
        gr[c] = gamma[c0 + c];
        be[c] = beta[c0 + c];
        dbs[c] = dgm[c] = dbt[c] = 0.f;
    }
    const long frames = (long)B * T;
    const long fstep = (long)gridDim.x * 4;
    // a frame's ten samples and its sixteen gradient values per lane are requested one frame ahead: a SIMD holds one
    // wave of this kernel (the 104 running sums per lane), so nothing else would cover the loads
    float xn[10];
    f32x4 gn0 = {0.f, 0.f, 0.f, 0.f}, gn1 = gn0;
    auto request = [&](long f) {
        if (f >= frames) return;   // wave-uniform
        const int b = (int)(f / T), t = (int)(f - (long)b * T);
        const float* src = wav + (long)b * L + t * 5;
#pragma unroll
        for (int j = 0; j < 10; ++j) xn[j] = src[j];
        const float* g = da + ((long)b * P + t) * 512 + c0;
        gn0 = *reinterpret_cast<const f32x4*>(g);
        gn1 = *reinterpret_cast<const f32x4*>(g + 4);
    };
#pragma unroll
    for (int j = 0; j < 10; ++j) xn[j] = 0.f;
    request((long)blockIdx.x * 4 + wave);
    for (long fidx = (long)blockIdx.x * 4 + wave; fidx < frames; fidx += fstep) {
        float x[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) x[j] = xn[j];
        const f32x4 g0 = gn0, g1 = gn1;
        request(fidx + fstep);
        float y[8];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float a = br[c];
#pragma unroll
            for (int j = 0; j < 10; ++j) a = fmaf(wr[c][j], x[j], a);
            y[c] = a;
            s += a;
        }
        const float mean = wave_sum(s) * (1.0f / 512.0f);
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            y[c] -= mean;
            q = fmaf(y[c], y[c], q);
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / 512.0f) + eps);
        float gz[8];
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float xh = y[c] * rstd;
            const float dz = (c < 4 ? g0[c] : g1[c - 4]) * gelu_grad(fmaf(xh, gr[c], be[c]));
            dgm[c] = fmaf(dz, xh, dgm[c]);
            dbt[c] += dz;
            y[c] = xh;
            gz[c] = dz * gr[c];
            m1 += gz[c];
            m2 = fmaf(gz[c], xh, m2);
        }
        m1 = wave_sum(m1) * (1.0f / 512.0f);
        m2 = wave_sum(m2) * (1.0f / 512.0f);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float dy = rstd * (gz[c] - m1 - y[c] * m2);
            dbs[c] += dy;
#pragma unroll
            for (int j = 0; j < 10; ++j) dw[c][j] = fmaf(dy, x[j], dw[c][j]);
        }
    }
    // lane's 104 values -> LDS [wave][value][lane]; the workgroup's sum goes to part in the output layout
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
        for (int j = 0; j < 10; ++j) red[wave][(c * 10 + j) * 64 + lane] = dw[c][j];
        red[wave][(80 + c) * 64 + lane] = dbs[c];
        red[wave][(88 + c) * 64 + lane] = dgm[c];
        red[wave][(96 + c) * 64 + lane] = dbt[c];
    }
    __syncthreads();
    float* dst = part + (long)blockIdx.x * CONV0_GRAD_N;
    for (int i = threadIdx.x; i < 104 * 64; i += 256) {
        const int v = i >> 6, ln = i & 63;
        const float sum = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
        int o;
        if (v < 80) o = (ln * 8 + v / 10) * 10 + v % 10;             // dW[channel][tap]
        else o = 5120 + ((v - 80) >> 3) * 512 + ln * 8 + ((v - 80) & 7);  // dbias / dgamma / dbeta [channel]
        dst[o] = sum;
    }
}

}  // namespace

// ================================================================ launchers
hipError_t launch_transpose16(const bf16_t* in, long ld, long R, int C, bf16_t* out, long ldo, int src_fmt,
                              hipStream_t s, const TransposeBatch& tb) {
    if (R <= 0 || C <= 0 || ld <= 0 || !in || !out || tb.nz < 1 || tb.nz > 65535 || tb.zdiv < 1)
        return hipErrorInvalidValue;
    if (tb.nz == 1 && ldo < R) return hipErrorInvalidValue;
    // a single matrix is written with its zero padding up to ldo; batch entries write their R rows only
    // (they sit side by side along the output rows; the caller zeroes what lies behind the last one)
    const long wrows = tb.nz == 1 ? ldo : R;
    dim3 grid((unsigned)((wrows + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)tb.nz);
    if (grid.y > 65535) return hipErrorInvalidValue;
    if (src_fmt == FMT_F16)
        hipLaunchKernelGGL((transpose16_kernel<FMT_F16, FMT_BF16>), grid, dim3(256), 0, s, in, ld, (int)R, C, out, ldo,
                           wrows, tb);
    else
        hipLaunchKernelGGL((transpose16_kernel<FMT_BF16, FMT_BF16>), grid, dim3(256), 0, s, in, ld, (int)R, C, out, ldo,
                           wrows, tb);
    return hipGetLastError();
}

hipError_t launch_cvt_transpose(const float* in, long R, int C, bf16_t* copy, bf16_t* outT, long ldo, hipStream_t s) {
    if (R <= 0 || C <= 0 || !in || !copy || !outT || ldo < R) return hipErrorInvalidValue;
    dim3 grid((unsigned)((ldo + 63) / 64), (unsigned)((C + 63) / 64));
    if (grid.y > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(cvt_transpose_kernel, grid, dim3(256), 0, s, in, (int)R, C, copy, outT, ldo);
    return hipGetLastError();
}

hipError_t launch_cvt_transpose_w(const float* in, long R, int C, float scale, int copy_fmt, bf16_t* copy, bf16_t* outT,
                                  long ldo, hipStream_t s) {
    if (R <= 0 || C <= 0 || (C % 4) || (ldo % 4) || !in || !copy || !outT || ldo < R) return hipErrorInvalidValue;
    if ((reinterpret_cast<uintptr_t>(in) & 15) || (reinterpret_cast<uintptr_t>(copy) & 7) ||
        (reinterpret_cast<uintptr_t>(outT) & 7))
        return hipErrorInvalidValue;
    dim3 grid((unsigned)((R + 63) / 64), (unsigned)((C + 63) / 64));
    if (grid.y > 65535) return hipErrorInvalidValue;
    SLSMI_FMT_DISPATCH(copy_fmt, hipLaunchKernelGGL(cvt_transpose_w_kernel<F>, grid, dim3(256), 0, s, in, (int)R, C, scale,
                                                    copy, outT, ldo));
    return hipGetLastError();
}

hipError_t launch_cvt_transpose_w_batch(const CvtTransposeW* items, int count, int copy_fmt, hipStream_t s) {
    if (count == 0) return hipSuccess;
    if (!items || count < 0) return hipErrorInvalidValue;
    std::vector<char> done(count, 0);
    for (int first = 0; first < count; ++first) {
        if (done[first]) continue;
        // the matrices of this one's shape, CVT_W_BATCH to a launch
        const int R = items[first].R, C = items[first].C;
        if (R <= 0 || C <= 0 || (C % 4)) return hipErrorInvalidValue;
        CvtTransposeWTable t;
        int m = 0;
        auto flush = [&]() -> hipError_t {
            if (!m) return hipSuccess;
            const dim3 grid((unsigned)((R + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)m);
            if (grid.y > 65535) return hipErrorInvalidValue;
            SLSMI_FMT_DISPATCH(copy_fmt, hipLaunchKernelGGL(cvt_transpose_w_batch_kernel<F>, grid, dim3(256), 0, s, R, C, t));
            m = 0;
            return hipGetLastError();
        };
        for (int i = first; i < count; ++i) {
            const CvtTransposeW& j = items[i];
            if (done[i] || j.R != R || j.C != C) continue;
            if (!j.src || !j.copy || !j.outT || j.ldo < R || (j.ldo % 4) || (reinterpret_cast<uintptr_t>(j.src) & 15) ||
                (reinterpret_cast<uintptr_t>(j.copy) & 7) || (reinterpret_cast<uintptr_t>(j.outT) & 7))
                return hipErrorInvalidValue;
            done[i] = 1;
            t.item[m++] = j;
            if (m == CVT_W_BATCH) {
                const hipError_t e = flush();
                if (e != hipSuccess) return e;
            }
        }
        const hipError_t e = flush();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

int colsum_parts(long M) {
    const long p = (M + 31) / 32;  // 32 rows per workgroup row: enough workgroups for a few hundred rows too
    return (int)(p < 1 ? 1 : (p > COLSUM_MAX_PARTS ? COLSUM_MAX_PARTS : p));
}

hipError_t launch_colsum(const void* in, int in_fmt, long ld, long M, int N, float scale, int accumulate,
                         float* partial, float* out, hipStream_t s) {
    if (M <= 0 || N <= 0 || !in || !partial || !out) return hipErrorInvalidValue;
    if (M <= 4096) {
        const dim3 g1((N + 63) / 64);
        if (in_fmt < 0)
            hipLaunchKernelGGL((colsum_one_kernel<false, 0>), g1, dim3(1024), 0, s, in, ld, M, N, accumulate,
                               colsum_single(out, N, scale));
        else if (in_fmt == FMT_F16)
            hipLaunchKernelGGL((colsum_one_kernel<true, FMT_F16>), g1, dim3(1024), 0, s, in, ld, M, N, accumulate,
                               colsum_single(out, N, scale));
        else
            hipLaunchKernelGGL((colsum_one_kernel<true, FMT_BF16>), g1, dim3(1024), 0, s, in, ld, M, N, accumulate,
                               colsum_single(out, N, scale));
        return hipGetLastError();
    }
    const int parts = colsum_parts(M);
    const long rpp = (M + parts - 1) / parts;
    dim3 grid((N + 255) / 256, parts);
    if (in_fmt < 0)
        hipLaunchKernelGGL((colsum_part_kernel<false, 0>), grid, dim3(256), 0, s, in, ld, M, N, rpp, partial);
    else if (in_fmt == FMT_F16)
        hipLaunchKernelGGL((colsum_part_kernel<true, FMT_F16>), grid, dim3(256), 0, s, in, ld, M, N, rpp, partial);
    else
        hipLaunchKernelGGL((colsum_part_kernel<true, FMT_BF16>), grid, dim3(256), 0, s, in, ld, M, N, rpp, partial);
    // the chunks' sums (parts <= COLSUM_MAX_PARTS rows of N) through the one-launch kernel: 16 waves per 64 columns
    hipLaunchKernelGGL((colsum_one_kernel<false, 0>), dim3((N + 63) / 64), dim3(1024), 0, s, partial, (long)N, (long)parts,
                       N, accumulate, colsum_single(out, N, scale));
    return hipGetLastError();
}

hipError_t launch_gemm_tn(const bf16_t* A, long lda, long zA, const bf16_t* B, long ldb, long zB, int b_fmt, float* C,
                          long ldc, long zC, long M, int N, int K, int nz, float alpha, hipStream_t s, int grp, long gA,
                          long gB, float* const* db) {
    const bool narrow = N % 128 != 0;   // 64 columns of A per workgroup against 256 of B
    if (!A || !B || !C || M <= 0 || M > 0x7fffffffL || N <= 0 || K <= 0 || (N % 64) || (K % (narrow ? 256 : 128)) || nz < 1 ||
        nz > 65535 || (lda % 8) || (ldb % 8) || (ldc % 4) || (zA % 8) || (zB % 8) || (zC % 4) || lda < N || ldb < 8 ||
        ldc < K)  // (ldb < K: rows of B that overlap, a conv's view)
        return hipErrorInvalidValue;
    if (grp < 0 || (grp > 0 && ((gA % 8) || (gB % 8)))) return hipErrorInvalidValue;
    if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15) || (reinterpret_cast<uintptr_t>(C) & 15))
        return hipErrorInvalidValue;
    TnParams p{A, lda, zA, B, ldb, zB, C, ldc, zC, (int)M, alpha, grp, gA, gB};
    if (narrow && grp == 0) p.grp = (int)M;   // one group: the narrow form is built with grouped addresses only
    TnBias bias{};
    if (db) {   // column sums of A beside the product: the plain 128-column form, one table's worth of batch entries
        if (narrow || grp > 0 || nz > TN_ADAM_BATCH) return hipErrorInvalidValue;
        for (int z = 0; z < nz; ++z) {
            if (db[z] && (reinterpret_cast<uintptr_t>(db[z]) & 3)) return hipErrorInvalidValue;
            bias.db[z] = db[z];
        }
        bias.n = nz;
    }
    const dim3 grid(narrow ? N / 64 : N / 128, narrow ? K / 256 : K / 128, nz);
    if (grid.y > 65535) return hipErrorInvalidValue;
    if (narrow) {
        SLSMI_FMT_DISPATCH(b_fmt, hipLaunchKernelGGL((gemm_tn_kernel<F, 64, true>), grid, dim3(256), 0, s, p, TnNoAdam{}, bias));
    } else if (grp > 0) {
        SLSMI_FMT_DISPATCH(b_fmt, hipLaunchKernelGGL((gemm_tn_kernel<F, 128, true>), grid, dim3(256), 0, s, p, TnNoAdam{}, bias));
    } else {
        SLSMI_FMT_DISPATCH(b_fmt, hipLaunchKernelGGL((gemm_tn_kernel<F, 128, false>), grid, dim3(256), 0, s, p, TnNoAdam{}, bias));
    }
    return hipGetLastError();
}

hipError_t launch_gemm_tn_adam(const bf16_t* A, long lda, long zA, const bf16_t* B, long ldb, long zB, int b_fmt, long M, int N,
                               int K, float alpha, const TnAdamEntry* entries, int count, double lr, double beta1, double beta2,
                               double eps, double weight_decay, long ldo, float copy_scale, int copy_fmt, hipStream_t s) {
    if (!A || !B || !entries || count < 1 || M <= 0 || M > 0x7fffffffL || N <= 0 || K <= 0 || (N % 128) || (K % 128) ||
        (lda % 8) || (ldb % 8) || (zA % 8) || (zB % 8) || lda < N || ldb < 8 || K / 128 > 65535)
        return hipErrorInvalidValue;
    if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15)) return hipErrorInvalidValue;
    for (int e = 0; e < count; ++e) {
        const TnAdamEntry& q = entries[e];
        if (!q.p || !q.m || !q.v || q.step < 1 || q.zsrc < 0 ||
            ((reinterpret_cast<uintptr_t>(q.p) | reinterpret_cast<uintptr_t>(q.m) | reinterpret_cast<uintptr_t>(q.v)) & 15))
            return hipErrorInvalidValue;
        if ((q.copy != nullptr) != (q.outT != nullptr)) return hipErrorInvalidValue;
        if (q.copy && (ldo < N || (ldo % 8) || (copy_fmt != FMT_BF16 && copy_fmt != FMT_F16) ||
                       ((reinterpret_cast<uintptr_t>(q.copy) | reinterpret_cast<uintptr_t>(q.outT)) & 15)))
            return hipErrorInvalidValue;
    }
    const AdamHyper h = adam_hyper(lr, beta1, beta2, eps, weight_decay);
    for (int e0 = 0; e0 < count; e0 += TN_ADAM_BATCH) {
        const int n = count - e0 < TN_ADAM_BATCH ? count - e0 : TN_ADAM_BATCH;
        TnAdamTable t{};
        TnBias bias{};
        bias.n = n;
        for (int e = 0; e < n; ++e) {
            bias.db[e] = entries[e0 + e].db;
            t.p[e] = entries[e0 + e].p;
            t.m[e] = entries[e0 + e].m;
            t.v[e] = entries[e0 + e].v;
            t.step[e] = entries[e0 + e].step;
            t.zsrc[e] = entries[e0 + e].zsrc;
            t.copy[e] = entries[e0 + e].copy;
            t.outT[e] = entries[e0 + e].outT;
        }
        t.ldo = ldo;
        t.scale = copy_scale;
        t.copy_fmt = copy_fmt;
        t.h = h;
        TnParams p{A, lda, zA, B, ldb, zB, nullptr, (long)K, 0, (int)M, alpha, 0, 0, 0};
        const dim3 grid(N / 128, K / 128, n);
        SLSMI_FMT_DISPATCH(b_fmt, hipLaunchKernelGGL((gemm_tn_kernel<F, 128, false, true>), grid, dim3(256), 0, s, p, t, bias));
        const hipError_t err = hipGetLastError();
        if (err != hipSuccess) return err;
    }
    return hipSuccess;
}

hipError_t launch_colsum_layers(const bf16_t* in, long z_in, long ld, long M, int N, float scale, float* const* dst,
                                int layers, hipStream_t s) {
    if (!in || !dst || M <= 0 || N <= 0 || (N % 64) || ld < N || layers <= 0) return hipErrorInvalidValue;
    for (int l0 = 0; l0 < layers; l0 += COLSUM_LAYERS_PER_LAUNCH) {
        const int n = layers - l0 < COLSUM_LAYERS_PER_LAUNCH ? layers - l0 : COLSUM_LAYERS_PER_LAUNCH;
        ColsumLayersOut o{};
        bool any = false;
        for (int i = 0; i < n; ++i) any = (o.p[i] = dst[l0 + i]) || any;
        if (!any) continue;
        hipLaunchKernelGGL(colsum_layers_kernel, dim3(N / 64, n), dim3(1024), 0, s, in + l0 * z_in, z_in, ld, M, scale, o);
    }
    return hipGetLastError();
}

hipError_t launch_gelu_bwd16(const bf16_t* pre, int pre_fmt, bf16_t* g, long n, hipStream_t s) {
    if (!pre || !g || n <= 0 || (n % 8)) return hipErrorInvalidValue;
    SLSMI_FMT_DISPATCH(pre_fmt, hipLaunchKernelGGL(gelu_bwd16_kernel<F>, dim3(grid_for(n / 8, 256, 8192)), dim3(256), 0, s, pre,
                                                   g, n / 8));
    return hipGetLastError();
}

hipError_t launch_colsum3(const void* in, int in_fmt, long ld, long M, int seg, float* const out[3], const float scale[3],
                          float* partial, hipStream_t s) {
    if (M <= 0 || seg <= 0 || (seg % 64) || !in || !partial) return hipErrorInvalidValue;
    if (M > 4096) {  // the two-stage path, one segment at a time
        for (int i = 0; i < 3; ++i) {
            if (!out[i]) continue;
            const char* src = reinterpret_cast<const char*>(in) + (size_t)i * seg * (in_fmt < 0 ? 4 : 2);
            const hipError_t e = launch_colsum(src, in_fmt, ld, M, seg, scale[i], 0, partial, out[i], s);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    const ColsumOut o{{out[0], out[1], out[2]}, {scale[0], scale[1], scale[2]}, seg};
    const dim3 g1(3 * seg / 64);
    if (in_fmt < 0)
        hipLaunchKernelGGL((colsum_one_kernel<false, 0>), g1, dim3(1024), 0, s, in, ld, M, 3 * seg, 0, o);
    else if (in_fmt == FMT_F16)
        hipLaunchKernelGGL((colsum_one_kernel<true, FMT_F16>), g1, dim3(1024), 0, s, in, ld, M, 3 * seg, 0, o);
    else
        hipLaunchKernelGGL((colsum_one_kernel<true, FMT_BF16>), g1, dim3(1024), 0, s, in, ld, M, 3 * seg, 0, o);
    return hipGetLastError();
}

hipError_t launch_ln_part_sums(const float* part, long z_part, int groups, int C, float* const* dgamma, float* const* dbeta,
                               int count, hipStream_t s) {
    if (!part || !dgamma || !dbeta || groups <= 0 || C <= 0 || (C % 32) || count < 0) return hipErrorInvalidValue;
    for (int at = 0; at < count; at += LN_PART_BATCH) {
        const int m = count - at < LN_PART_BATCH ? count - at : LN_PART_BATCH;
        LnPartOut o{};
        bool any = false;
        for (int i = 0; i < m; ++i) {
            o.dg[i] = dgamma[at + i];
            o.db[i] = dbeta[at + i];
            any = any || o.dg[i] || o.db[i];
        }
        if (!any) continue;
        hipLaunchKernelGGL(ln_part_sums_kernel, dim3(2 * C / 64, m), dim3(1024), 0, s, part + at * z_part, z_part, (long)groups,
                           C, o);
    }
    return hipGetLastError();
}

int ln_bwd_groups(long rows) {
    const long g = (rows + 3) / 4;  // one row per wave up to 4096 rows: a thousand rows spread over 252 workgroups
    return (int)(g < 1 ? 1 : (g > LN_BWD_MAX_GROUPS ? LN_BWD_MAX_GROUPS : g));
}

hipError_t launch_ln_bwd(int C, const void* x, int x_fmt, const float* dy, const float* gamma, const float* beta,
                         float eps, int gelu, const float* resid, void* out, int out16, float* part, float* dgamma,
                         float* dbeta, long rows, int grp_in, int grp_valid, hipStream_t s, const float* resid2,
                         bf16_t* copy16, int dy_slices, long dy_slice) {
    if (rows <= 0 || !x || !dy || !gamma || !out || !part || (gelu && !beta) || (out16 && (resid || resid2 || copy16)) ||
        dy_slices < 1)
        return hipErrorInvalidValue;
    const int groups = ln_bwd_groups(rows);
    dim3 grid(groups), block(256);
#define LN_BWD(CC, X16, XF, G, O16)                                                                                   \
    hipLaunchKernelGGL((ln_bwd_kernel<CC, X16, XF, G, O16>), grid, block, 0, s, x, dy, gamma, beta, eps, resid, out, \
                       part, rows, grp_in, grp_valid, resid2, copy16, dy_slices, dy_slice)
    if (C == 1024 && x_fmt < 0 && !gelu && !out16) LN_BWD(1024, false, 0, false, false);
    else if (C == 512 && x_fmt == FMT_F16 && gelu && out16) LN_BWD(512, true, FMT_F16, true, true);
    else if (C == 512 && x_fmt == FMT_BF16 && gelu && out16) LN_BWD(512, true, FMT_BF16, true, true);
    else if (C == 512 && x_fmt == FMT_F16 && !gelu && !out16) LN_BWD(512, true, FMT_F16, false, false);
    else if (C == 512 && x_fmt == FMT_BF16 && !gelu && !out16) LN_BWD(512, true, FMT_BF16, false, false);
    else return hipErrorInvalidValue;
#undef LN_BWD
    if (dgamma || dbeta)  // part is [groups][2 C]: its column sums are dgamma (first C) and dbeta
        hipLaunchKernelGGL((colsum_one_kernel<false, 0>), dim3(2 * C / 64), dim3(1024), 0, s, part, 2L * C, (long)groups,
                           2 * C, 0, ColsumOut{{dgamma, dbeta, nullptr}, {1.0f, 1.0f, 0.f}, C});
    return hipGetLastError();
}

hipError_t launch_gelu16(const bf16_t* in, bf16_t* out, long n, int fmt, hipStream_t s) {
    if (n <= 0 || (n % 8)) return hipErrorInvalidValue;
    SLSMI_FMT_DISPATCH(fmt, hipLaunchKernelGGL(gelu16_kernel<F>, dim3(grid_for(n / 8, 256, 8192)), dim3(256), 0, s, in, out, n / 8));
    return hipGetLastError();
}

hipError_t launch_gelu_bwd16_transpose(const bf16_t* pre, int pre_fmt, bf16_t* g, long R, int C, bf16_t* outT, long ldo,
                                       hipStream_t s) {
    if (R <= 0 || C <= 0 || !pre || !g || !outT || ldo < R) return hipErrorInvalidValue;
    dim3 grid((unsigned)((ldo + 63) / 64), (unsigned)((C + 63) / 64));
    if (grid.y > 65535) return hipErrorInvalidValue;
    SLSMI_FMT_DISPATCH(pre_fmt, hipLaunchKernelGGL(gelu_bwd16_transpose_kernel<F>, grid, dim3(256), 0, s, pre, g, (int)R, C,
                                                   outT, ldo));
    return hipGetLastError();
}

hipError_t launch_scale_f32(float* x, long n, float scale, hipStream_t s) {
    if (n <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, s, x, n, scale);
    return hipGetLastError();
}

hipError_t launch_attention_bwd(const bf16_t* qkv, const bf16_t* o, const bf16_t* dout, bf16_t* dqkv, int B, int T,
                                int fmt, hipStream_t s, const uint8_t* keep_qk, const uint8_t* keep_kq, int pitch,
                                float drop_scale) {
    if (B <= 0 || T <= 0 || T > 256) return hipErrorInvalidValue;
    if ((keep_qk != nullptr) != (keep_kq != nullptr) || (keep_qk && (pitch % 32 != 0 || pitch < T))) return hipErrorInvalidValue;
    const int Tp = (T + 31) & ~31;
    const int row_pitch = Tp <= 224 ? 160 : 144;
    const int lds = Tp * (4 * row_pitch + 12);
    constexpr int LDS_MAX = 256 * (4 * 144 + 12);   // (224 rows of 160 bytes are fewer)
    // workgroups per (utterance, head): as many as fit the CUs in one round, at most four (each repeats the
    // staging and the score statistics) and not more than there are 16-row tiles
    int n_cu = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
        n_cu = 256;
    int share = n_cu / (B * 16);
    share = share < 1 ? 1 : (share > 4 ? 4 : share);
    if (share > Tp / 16) share = Tp / 16;
    const dim3 grid(B * 16, share);
    static LdsAttrState st[2][2];
    hipError_t e = hipSuccess;
    auto go = [&](auto kernel, LdsAttrState& a) {
        e = ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), LDS_MAX, a);
        if (e != hipSuccess) return;
        hipLaunchKernelGGL(kernel, grid, dim3(AB_WAVES * 64), lds, s, qkv, o, dout, dqkv, T, keep_qk, keep_kq, pitch, drop_scale);
    };
    if (fmt == FMT_F16) {
        if (row_pitch == 160) go(&attn_bwd_kernel<FMT_F16, 160>, st[1][1]);
        else go(&attn_bwd_kernel<FMT_F16, 144>, st[1][0]);
    } else {
        if (row_pitch == 160) go(&attn_bwd_kernel<FMT_BF16, 160>, st[0][1]);
        else go(&attn_bwd_kernel<FMT_BF16, 144>, st[0][0]);
    }
    if (e != hipSuccess) return e;
    return hipGetLastError();
}


hipError_t launch_cls_bwd(const float* dlogp, const float* logp, const float* pooled, const float* hidden,
                          const float* mask, const float* ln_g, const float* ln_b, const float* w1, const float* w2,
                          int B, int dict, float* stats, float* dh1, float* dhb, float* dpool, float* dln_g,
                          float* dln_b, float* dw1, float* db1, float* dw2, float* db2, hipStream_t s) {
    if (B <= 0 || B > 64 || dict <= 0 || dict % 256 != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(cls_stats_kernel, dim3(B), dim3(256), 0, s, pooled, dict, stats);
    hipLaunchKernelGGL(cls_bwd_out_kernel, dim3(1), dim3(256), 0, s, dlogp, logp, hidden, mask, w2, B, dw2, db2, db1, dh1);
    hipLaunchKernelGGL(cls_bwd_fc1_kernel, dim3((dict + 63) / 64), dim3(64), 0, s, pooled, stats, ln_g, ln_b, w1, dh1, B, dict,
                       dw1, dhb, dln_g, dln_b);
    hipLaunchKernelGGL(cls_bwd_ln_kernel, dim3(B), dim3(256), 0, s, pooled, stats, ln_g, dhb, dict, dpool);
    return hipGetLastError();
}

hipError_t launch_sae_code_grad(const float* code, const float* dpool, const float* dcode, bf16_t* denc, long M, int T,
                                int dict, hipStream_t s) {
    if (M <= 0 || T <= 0 || dict <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sae_code_grad_kernel, dim3(grid_for(M * dict, 256, 16384)), dim3(256), 0, s, code, dpool, dcode,
                       denc, M, T, dict);
    return hipGetLastError();
}

hipError_t launch_mse_grad(const float* recon, const float* b_dec, const float* x, const float* dloss, long rows,
                           bf16_t* drecon, float* dx, hipStream_t s) {
    if (rows <= 0 || !dloss) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mse_grad_kernel, dim3(grid_for(rows * 1024, 256, 8192)), dim3(256), 0, s, recon, b_dec, x, dloss,
                       rows * 1024, drecon, dx);
    return hipGetLastError();
}

hipError_t launch_recon_grad(const float* recon, const float* b_dec, const float* x, const float* dloss, const float* dpool,
                             long rows, int T, bf16_t* drecon, float* dx, float* total, hipStream_t s) {
    if (rows <= 0 || T <= 0 || !dpool) return hipErrorInvalidValue;
    hipLaunchKernelGGL(recon_grad_kernel, dim3(grid_for(rows * 1024, 256, 8192)), dim3(256), 0, s, recon, b_dec, x, dloss, dpool,
                       rows * 1024, T, drecon, dx, total);
    return hipGetLastError();
}

hipError_t launch_spread_rows(const float* dpool, long rows, int T, float* out, hipStream_t s) {
    if (rows <= 0 || T <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(spread_rows_kernel, dim3(grid_for(rows * 1024, 256, 8192)), dim3(256), 0, s, dpool, rows * 1024, T, out);
    return hipGetLastError();
}

hipError_t launch_add_f32(float* a, const float* b, long n, hipStream_t s) {
    if (n <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(add_f32_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, s, a, b, n);
    return hipGetLastError();
}

hipError_t launch_mul_f32(float* out, const float* a, const float* m, const float* resid, long n, hipStream_t s) {
    if (n <= 0 || !out || !a || !m) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mul_f32_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, s, out, a, m, resid, n);
    return hipGetLastError();
}

hipError_t launch_mul16_f32(bf16_t* x, const float* m, long n, int fmt, hipStream_t s) {
    if (n <= 0 || !x || !m) return hipErrorInvalidValue;
    SLSMI_FMT_DISPATCH(fmt, hipLaunchKernelGGL(mul16_f32_kernel<F>, dim3(grid_for(n, 256, 8192)), dim3(256), 0, s, x, m, n));
    return hipGetLastError();
}

hipError_t launch_cvt_bf16(const float* in, bf16_t* out, long n, hipStream_t s) {
    if (n <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(cvt_bf16_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, s, in, out, n);
    return hipGetLastError();
}

hipError_t launch_add_gelu(const float* x, const float* u, float* out, long n, hipStream_t s) {
    if (n <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(add_gelu_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, s, x, u, out, n);
    return hipGetLastError();
}

hipError_t launch_posconv_grad_pack(const float* dx, const float* u, bf16_t* du, bf16_t* dug, int B, int T,
                                    int rows_per_slab, hipStream_t s) {
    if (B <= 0 || T <= 0 || rows_per_slab < T + 128) return hipErrorInvalidValue;
    hipLaunchKernelGGL(posconv_grad_pack_kernel, dim3(grid_for((long)B * 16 * rows_per_slab * 8, 256)), dim3(256), 0, s, dx,
                       u, du, dug, B, T, rows_per_slab);
    return hipGetLastError();
}

hipError_t launch_posconv_flip(const bf16_t* w, int fmt, bf16_t* wf, hipStream_t s) {
    SLSMI_FMT_DISPATCH(fmt, hipLaunchKernelGGL(posconv_flip_kernel<F>, dim3(8192), dim3(256), 0, s, w, wf));
    return hipGetLastError();
}

hipError_t launch_weightnorm_bwd(const float* dwf, const float* v, const float* g, float* dv, float* dg, float* scratch,
                                 hipStream_t s) {
    if (!dwf || !v || !g || !scratch) return hipErrorInvalidValue;
    float* partial = scratch;                                   // [1024][2][128]
    float* stat = scratch + (size_t)WN_CO * 2 * WN_TAPS;        // [2][128]
    hipLaunchKernelGGL(weightnorm_partial_kernel, dim3(WN_CO), dim3(256), 0, s, dwf, v, partial);
    hipLaunchKernelGGL(weightnorm_reduce_kernel, dim3(1), dim3(1024), 0, s, partial, stat, dg);
    if (dv) hipLaunchKernelGGL(weightnorm_apply_kernel, dim3(WN_CO), dim3(256), 0, s, dwf, v, g, stat, dv);
    return hipGetLastError();
}


hipError_t launch_expand_rows(const float* in, float* out, int B, int T, int P, int C, hipStream_t s) {
    if (B <= 0 || T <= 0 || P < T || C % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(expand_rows_kernel, dim3(grid_for((long)B * P * (C / 4), 256, 16384)), dim3(256), 0, s, in, out, B,
                       T, P, C / 4);
    return hipGetLastError();
}

hipError_t launch_col2im(const bf16_t* dcol, float* da, int B, int Pin, int Pout, int k, int stride, hipStream_t s) {
    if (B <= 0 || Pout <= 0 || Pin != Pout * stride || k < 1 || stride < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(col2im_kernel, dim3(grid_for((long)B * Pin * 64, 256, 32768)), dim3(256), 0, s, dcol, da, B, Pin,
                       Pout, k, stride);
    return hipGetLastError();
}

hipError_t launch_conv_grad_unpack(const float* packed, float* dst, int co, int ci, int k, int slices, hipStream_t s) {
    if (!packed || !dst || co < 1 || ci < 1 || k < 1 || slices < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(conv_grad_unpack_kernel, dim3(grid_for((long)co * ci * k, 256, 8192)), dim3(256), 0, s, packed, dst,
                       co, ci, k, slices);
    return hipGetLastError();
}

int conv0_bwd_groups(long frames) {
    const long g = (frames + 63) / 64;
    return (int)(g < 1 ? 1 : (g > COLSUM_MAX_PARTS ? COLSUM_MAX_PARTS : g));
}

hipError_t launch_conv0_bwd(const float* wav, int B, int L, int T, long P, const float* w, const float* bias,
                            const float* gamma, const float* beta, float eps, const float* da, float* part,
                            float* sums, hipStream_t s) {
    if (B <= 0 || T <= 0 || P < T || !part || !sums) return hipErrorInvalidValue;
    const int groups = conv0_bwd_groups((long)B * T);
    hipLaunchKernelGGL(conv0_bwd_kernel, dim3(groups), dim3(256), 0, s, wav, L, T, P, w, bias, gamma, beta, eps, da, B,
                       part);
    hipLaunchKernelGGL((colsum_one_kernel<false, 0>), dim3((CONV0_GRAD_N + 63) / 64), dim3(1024), 0, s, part,
                       (long)CONV0_GRAD_N, (long)groups, CONV0_GRAD_N, 0, colsum_single(sums, CONV0_GRAD_N, 1.0f));
    return hipGetLastError();
}

}  // namespace slsmi
