// Utterance-level heads.
//
// SLS head (upstream model; only getAttenF survives in the reference tree,
// model_backup.py:186-202): fullfeature = sum_l sigmoid(fc0 . mean_t x_l) x_l, then
// BatchNorm2d(1) -> SELU -> max_pool2d(3, 3) -> fc1 -> SELU -> fc3 -> SELU -> log_softmax.
// The encoder keeps every layer output x_l (fp32) in its own buffer and the FFN's second
// GEMM leaves the row dots fc0 . x_l[row] beside it, so the head is two kernels after the
// last layer: all 24 B weights at once, then ONE pass that reads each x_l once, forms the
// weighted sum in registers and pools it - the [B, 24, T, 1024] tensor the reference
// concatenates (19.8 MB per utterance) and a running fp32 sum are never written.
//
// SAE head (model.py:68-89, :245-247): per-row top-k by a 4-pass radix select
// on the (non-negative) activations, mean over time, LayerNorm + 2-layer MLP.
#include "common.h"
#include "kernels.h"

namespace slsmi {

namespace {

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// w[l][b] = sigmoid(b0 + (1/T) sum_s sum_t dots_l[s][b T + t]): the layer weights from the row
// dots the residual GEMMs wrote (dots_l = dots + l * layer_stride is [slices][M], M = B T).
// One workgroup per (utterance, layer); fixed summation order.
__global__ __launch_bounds__(256) void sls_weight_kernel(const float* __restrict__ dots, long layer_stride,
                                                         const float* __restrict__ fc0_b, float* __restrict__ w,
                                                         int T, long M, int slices) {
    __shared__ float red[4];
    const int b = blockIdx.x, l = blockIdx.y;
    // thread t owns frame t (+256, ...) of every slice: independent loads, no index division
    float s = 0.f;
    for (int t = threadIdx.x; t < T; t += 256) {
        const float* p = dots + (long)l * layer_stride + (long)b * T + t;
        float part = 0.f;
#pragma unroll 16
        for (int sl = 0; sl < slices; ++sl) part += p[(long)sl * M];
        s += part;
    }
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) w[(long)l * gridDim.x + b] = 1.0f / (1.0f + expf(-(s / (float)T + fc0_b[0])));
}

// pooled[b][pi * 341 + pj] = max over the 3 x 3 window of selu(scale * f + shift),
// f[t][c] = sum_l w[l][b] * x_l[b T + t][c]   (BatchNorm2d(1) in eval is the scalar affine).
// One workgroup per (window row pi, utterance): thread i sums channels 4 i .. 4 i + 3 of the three
// frames over the layers (16-byte loads, a wave reads 1 KiB of a row), takes the max over the frames
// after the SELU, and the channel triples are joined through LDS.  Layers are added in index order.
template <int FMT>
__global__ __launch_bounds__(256) void sls_sum_pool_kernel(const float* __restrict__ xs, long layer_stride, int layers,
                                                           const float* __restrict__ w, bf16_t* __restrict__ pooled,
                                                           long ldp, int T, float scale, float shift,
                                                           const float* __restrict__ bn4, float bn_eps, long seg) {
    __shared__ float colmax[1024];
    __shared__ float wl[64];
    if (bn4) {  // the BatchNorm's own four values (weight, bias, running mean, running variance) instead of the affine
        const float inv = 1.0f / sqrtf(bn4[3] + bn_eps);
        scale = bn4[0] * inv;
        shift = bn4[1] - bn4[2] * scale;
    }
    const int pi = blockIdx.x, b = blockIdx.y, B = gridDim.y;
    const int PW = 1024 / 3;
    for (int l = threadIdx.x; l < layers; l += 256) wl[l] = w[(long)l * B + b];
    __syncthreads();
    const float* base = xs + ((long)b * T + 3 * pi) * 1024 + threadIdx.x * 4;
    f32x4 f[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll 4
    for (int l = 0; l < layers; ++l) {
        const float* p = base + (long)l * layer_stride;
        const float wg = wl[l];
#pragma unroll
        for (int d = 0; d < 3; ++d) f[d] += *reinterpret_cast<const f32x4*>(p + d * 1024) * wg;
    }
    f32x4 m;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        m[e] = selu(fmaf(f[0][e], scale, shift));
        m[e] = fmaxf(m[e], selu(fmaf(f[1][e], scale, shift)));
        m[e] = fmaxf(m[e], selu(fmaf(f[2][e], scale, shift)));
    }
    *reinterpret_cast<f32x4*>(colmax + threadIdx.x * 4) = m;
    __syncthreads();
    const long n = (long)(T / 3) * PW;
    bf16_t* out = pooled + (long)b * ldp;
    // seg > 0: the row is [hi | lo | hi], three segments of seg values (ldp = 3 seg): the rounded feature, the
    // remainder of that rounding, the rounded feature again -- for fc1 weights laid out [W_hi | W_hi | W_lo]
    const long width = seg > 0 ? seg : ldp;
    for (int pj = threadIdx.x; pj < PW; pj += 256) {
        const float v = fmaxf(fmaxf(colmax[3 * pj], colmax[3 * pj + 1]), colmax[3 * pj + 2]);
        const bf16_t hi = Op<FMT>::pack(v);
        out[(long)pi * PW + pj] = hi;
        if (seg > 0) {
            out[seg + (long)pi * PW + pj] = Op<FMT>::pack(v - Op<FMT>::unpack(hi));
            out[2 * seg + (long)pi * PW + pj] = hi;
        }
    }
    // the padding columns n .. width - 1 of the row (the fc1 GEMM reads whole 64-wide K-tiles)
    if (pi == 0)
        for (long i = n + threadIdx.x; i < width; i += 256) {
            out[i] = 0;
            if (seg > 0) out[seg + i] = out[2 * seg + i] = 0;
        }
}

__global__ __launch_bounds__(256) void sls_tail_kernel(const float* __restrict__ fc1out, int parts, long part_stride,
                                                       const float* __restrict__ b1, const float* __restrict__ w3,
                                                       const float* __restrict__ b3, float* __restrict__ logp) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = i * 256 + threadIdx.x;
        float a = 0.f;
        for (int q = 0; q < parts; ++q) a += fc1out[q * part_stride + (long)b * 1024 + n];  // fixed order
        const float h = selu(a + b1[n]);
        s0 = fmaf(w3[n], h, s0);
        s1 = fmaf(w3[1024 + n], h, s1);
    }
    s0 = block_sum_256(s0, red);
    s1 = block_sum_256(s1, red);
    if (threadIdx.x == 0) {
        const float z0 = selu(s0 + b3[0]), z1 = selu(s1 + b3[1]);
        const float m = fmaxf(z0, z1);
        const float lse = m + logf(expf(z0 - m) + expf(z1 - m));
        logp[b * 2 + 0] = z0 - lse;
        logp[b * 2 + 1] = z1 - lse;
    }
}

// Exact k-th largest of VPT * 256 non-negative values held as IEEE bit patterns u[] (0 for values
// <= 0; monotone for values >= 0), by a most-significant-byte-first radix select.  Returns the k-th largest
// pattern; *need = how many of the entries equal to it belong to the k largest.  With at most k positive
// entries the answer is 0 and *need the number of zero entries that fill the k up.
// Zero entries take no part in the histograms (after a ReLU half of a row would pile onto one counter), a bin's
// count of higher bins comes from a suffix sum inside each wave and the four wave totals: three barriers a pass;
// the first byte (the exponent) is settled without a histogram where it can be.
// hist: 256 ints; scan: 8 ints of scratch; sel: 2 ints.
template <int VPT>
__device__ __forceinline__ uint32_t radix_select_kth(const uint32_t (&u)[VPT], int k, int* hist, int* scan, int* sel,
                                                     int* need) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int pos = 0;
#pragma unroll
    for (int i = 0; i < VPT; ++i) pos += u[i] != 0u ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pos += __shfl_xor(pos, o, 64);
    if (lane == 0) scan[4 + wave] = pos;
    hist[threadIdx.x] = 0;
    __syncthreads();
    const int positives = scan[4] + scan[5] + scan[6] + scan[7];
    if (positives <= k) {   // the same for every thread
        *need = k - positives;
        return 0u;
    }
    uint32_t prefix = 0, mask = 0;
    int remaining = k;
    int first_pass = 0;
    {
        // The first byte is the exponent: activations of one layer share a handful of exponents, and a histogram by
        // LDS atomics would pile thousands of increments onto two or three counters.  Count from the largest
        // exponent downwards instead (one barrier per exponent); the k-th largest usually sits in the first few.
        // After TOP_TRIES exponents the histogram takes over.
        constexpr int TOP_TRIES = 6;
        uint32_t top = 0;
#pragma unroll
        for (int i = 0; i < VPT; ++i) top = max(top, u[i] >> 24);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) top = max(top, (uint32_t)__shfl_xor((int)top, o, 64));
        if (lane == 0) scan[wave] = (int)top;
        __syncthreads();
        top = (uint32_t)max(max(scan[0], scan[1]), max(scan[2], scan[3]));
        int left = k;
        for (int tr = 0; tr < TOP_TRIES && top >= (uint32_t)tr; ++tr) {
            const uint32_t e = top - tr;
            int c = 0;
#pragma unroll
            for (int i = 0; i < VPT; ++i) c += (u[i] != 0u && (u[i] >> 24) == e) ? 1 : 0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
            int* slot = scan + ((tr & 1) ? 0 : 4);     // (scan[0..3] held the maxima: the first counts go to scan[4..7])
            if (lane == 0) slot[wave] = c;
            __syncthreads();
            c = slot[0] + slot[1] + slot[2] + slot[3];
            if (c >= left) {
                prefix = e << 24;
                mask = 255u << 24;
                remaining = left;
                first_pass = 1;
                break;
            }
            left -= c;
        }
    }
    for (int pass = first_pass; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
#pragma unroll
        for (int i = 0; i < VPT; ++i)
            if (u[i] != 0u && (u[i] & mask) == prefix) atomicAdd(&hist[(u[i] >> shift) & 255], 1);
        __syncthreads();
        const int h = hist[threadIdx.x];
        hist[threadIdx.x] = 0;   // for the next pass: nobody else reads this bin
        int v = h;               // suffix sum over the wave's bins t .. 64 wave + 63
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_down(v, o, 64);
            if (lane + o < 64) v += up;
        }
        if (lane == 0) scan[wave] = v;
        __syncthreads();
        int incl = v;            // entries in bins >= t
        for (int q = wave + 1; q < 4; ++q) incl += scan[q];
        const int above = incl - h;
        if (above < remaining && remaining <= incl) {
            sel[0] = threadIdx.x;
            sel[1] = remaining - above;
        }
        __syncthreads();
        prefix |= (uint32_t)sel[0] << shift;
        mask |= 255u << shift;
        remaining = sel[1];
    }
    *need = remaining;
    return prefix;
}

// ---------------------------------------------------------------- SAE head
// Keep the k largest entries of a row of non-negative fp32 activations (exact k-th largest: radix_select_kth).
template <int VPT, int FMT>
__global__ __launch_bounds__(256) void topk_rows_kernel(float* __restrict__ enc, bf16_t* __restrict__ enc_bf16,
                                                        int dict, int k) {
    __shared__ int hist[256];
    __shared__ int scan[8];
    __shared__ int sel[2];
    __shared__ int tie_count;
    const long row = blockIdx.x;
    float* p = enc + row * dict;
    uint32_t u[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const float v = p[i * 256 + threadIdx.x];
        u[i] = v > 0.f ? __float_as_uint(v) : 0u;
    }
    if (threadIdx.x == 0) tie_count = 0;
    int remaining;
    const uint32_t thr = radix_select_kth<VPT>(u, k, hist, scan, sel, &remaining);
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        bool keep = u[i] > thr;
        if (u[i] == thr && thr != 0u) keep = atomicAdd(&tie_count, 1) < remaining;
        const float v = keep ? __uint_as_float(u[i]) : 0.f;
        p[i * 256 + threadIdx.x] = v;
        if (enc_bf16) enc_bf16[row * dict + i * 256 + threadIdx.x] = Op<FMT>::pack(v);
    }
}

// ---------------------------------------------------------------- window TopK (model_window_topk.py:118-203)
// Step 1, one workgroup per (window, utterance): the window's activation sums over its `w` frames
// and the k features with the largest sums (model_window_topk.py:151-165) -> keep[b][window][d] = 1 / 0.
// A feature whose sum is zero is never marked: its votes would be zero either way.
template <int VPT>
__global__ __launch_bounds__(256) void window_keep_kernel(const float* __restrict__ enc, uint8_t* __restrict__ keep,
                                                          int T, int dict, int k, int w, int stride, int nW) {
    __shared__ int hist[256];
    __shared__ int scan[256];
    __shared__ int sel[2];
    __shared__ int tie_count;
    const int win = blockIdx.x, b = blockIdx.y;
    float sum[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) sum[i] = 0.f;
    for (int t = win * stride; t < win * stride + w; ++t) {
        const float* row = enc + ((long)b * T + t) * dict;
#pragma unroll
        for (int i = 0; i < VPT; ++i) sum[i] += row[i * 256 + threadIdx.x];
    }
    uint32_t u[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) u[i] = sum[i] > 0.f ? __float_as_uint(sum[i]) : 0u;
    if (threadIdx.x == 0) tie_count = 0;
    int need;
    const uint32_t thr = radix_select_kth<VPT>(u, k, hist, scan, sel, &need);
    uint8_t* out = keep + ((long)b * nW + win) * dict;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        bool kp = u[i] > thr;
        if (u[i] == thr && thr != 0u) kp = atomicAdd(&tie_count, 1) < need;
        out[i * 256 + threadIdx.x] = kp ? 1 : 0;
    }
}

// Step 2, one workgroup per (frame, utterance): vote[d] = activation * (number of windows covering the
// frame that kept d) (model_window_topk.py:171-185); the frame keeps the activations of its k largest
// votes (:189-197).  With fewer than k positive votes the reference's topk also returns zero-vote
// features, which ones being left to its implementation; here they are the lowest-numbered ones.
template <int VPT, int FMT>
__global__ __launch_bounds__(256) void window_vote_topk_kernel(float* __restrict__ enc, bf16_t* __restrict__ enc_bf16,
                                                               const uint8_t* __restrict__ keep, int T, int dict,
                                                               int k, int w, int stride, int nW) {
    __shared__ int hist[256];
    __shared__ int scan[256];
    __shared__ int sel[2];
    __shared__ int tie_count;
    __shared__ int wave_total[4];
    const int t = blockIdx.x, b = blockIdx.y;
    const long row = (long)b * T + t;
    float* p = enc + row * dict;
    // windows i with i * stride <= t < i * stride + w
    const int hi = min(t / stride, nW - 1);
    const int lo = max(0, (t - w + stride) / stride);
    float x[VPT];
    uint32_t u[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        x[i] = p[i * 256 + threadIdx.x];
        int cnt = 0;
        for (int win = lo; win <= hi; ++win) cnt += keep[((long)b * nW + win) * dict + i * 256 + threadIdx.x];
        const float vote = x[i] * (float)cnt;
        u[i] = vote > 0.f ? __float_as_uint(vote) : 0u;
    }
    if (threadIdx.x == 0) tie_count = 0;
    int need;
    const uint32_t thr = radix_select_kth<VPT>(u, k, hist, scan, sel, &need);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int base = 0;  // zero-vote features seen in lower-numbered chunks
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        bool kp = u[i] > thr;
        if (thr != 0u) {
            if (u[i] == thr) kp = atomicAdd(&tie_count, 1) < need;
        } else {
            // workgroup-uniform branch: rank the zero-vote features by feature number
            const bool z = u[i] == 0u;
            const unsigned long long bal = __ballot(z);
            if (lane == 0) wave_total[wave] = __popcll(bal);
            __syncthreads();
            int before = 0, total = 0;
            for (int q = 0; q < 4; ++q) {
                if (q < wave) before += wave_total[q];
                total += wave_total[q];
            }
            const int rank = base + before + __popcll(bal & ((1ull << lane) - 1ull));
            if (z) kp = rank < need;
            base += total;
            __syncthreads();
        }
        const float v = kp ? x[i] : 0.f;
        p[i * 256 + threadIdx.x] = v;
        if (enc_bf16) enc_bf16[row * dict + i * 256 + threadIdx.x] = Op<FMT>::pack(v);
    }
}

// mean over the frames of utterance b (its first nframes[b] of T when nframes is given)
__global__ __launch_bounds__(256) void col_mean_kernel(const float* __restrict__ enc, const float* __restrict__ add,
                                                       float* __restrict__ pooled, int T, int dict,
                                                       const int* __restrict__ nframes) {
    const int b = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= dict) return;
    const int Tv = nframes ? min(max(nframes[b], 1), T) : T;
    float s = 0.f;
    const float* col = enc + (long)b * T * dict + j;
    int t = 0;
    for (; t + 8 <= Tv; t += 8) {   // eight loads in flight; the frames are added in order
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = col[(long)(t + e) * dict];
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[e];
    }
    for (; t < Tv; ++t) s += col[(long)t * dict];
    pooled[(long)b * dict + j] = s / (float)Tv + (add ? add[j] : 0.f);
}

// Classifier of the in-tree head (model.py:183-189, :246): LayerNorm(dict) -> Linear(dict, 256) + ReLU
// (-> Dropout, identity in eval) -> Linear(256, 2) -> log_softmax, in two kernels.
// (1) One workgroup per (utterance, 32 hidden units): every workgroup normalises the pooled row into
// LDS (the statistics are cheap to repeat), then each wave takes 8 hidden units, lanes striding the
// weight row with 16-byte loads and four independent accumulators so the loads pipeline.
constexpr int SAE_HID = 256;
constexpr int SAE_HID_PER_WG = 32;
__global__ __launch_bounds__(256) void sae_fc1_kernel(const float* __restrict__ pooled, const float* __restrict__ ln_g,
                                                      const float* __restrict__ ln_b, const float* __restrict__ w1,
                                                      const float* __restrict__ b1, float* __restrict__ h1, int dict) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // dict + 4
    float* hbuf = sm;
    float* red = sm + dict;
    const int b = blockIdx.y;
    const float* row = pooled + (long)b * dict;
    float s = 0.f;
    for (int j = threadIdx.x; j < dict; j += 256) s += row[j];
    const float mean = block_sum_256(s, red) / (float)dict;
    float q = 0.f;
    for (int j = threadIdx.x; j < dict; j += 256) {
        const float d = row[j] - mean;
        q = fmaf(d, d, q);
    }
    const float rstd = 1.0f / sqrtf(block_sum_256(q, red) / (float)dict + 1e-5f);
    for (int j = threadIdx.x; j < dict; j += 256) hbuf[j] = fmaf((row[j] - mean) * rstd, ln_g[j], ln_b[j]);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * SAE_HID_PER_WG + wave * (SAE_HID_PER_WG / 4);
    // the wave's 8 hidden units together: a pass over the row has 8 independent weight loads in flight (one unit at
    // a time made the kernel a chain of 128 load latencies); per unit the same sums in the same order as before
    constexpr int UNITS = SAE_HID_PER_WG / 4;
    f32x4 a[UNITS];
#pragma unroll
    for (int q = 0; q < UNITS; ++q) a[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* w = w1 + (long)n0 * dict;
    for (int j = lane * 4; j < dict; j += 256) {  // dict is a multiple of 256
        const f32x4 hv = *reinterpret_cast<const f32x4*>(hbuf + j);
        f32x4 wv[UNITS];
#pragma unroll
        for (int q = 0; q < UNITS; ++q) wv[q] = *reinterpret_cast<const f32x4*>(w + (long)q * dict + j);
#pragma unroll
        for (int q = 0; q < UNITS; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) a[q][e] = fmaf(wv[q][e], hv[e], a[q][e]);
    }
#pragma unroll
    for (int q = 0; q < UNITS; ++q) {
        const float d = wave_sum((a[q][0] + a[q][1]) + (a[q][2] + a[q][3]));
        if (lane == 0) h1[(long)b * SAE_HID + n0 + q] = fmaxf(d + b1[n0 + q], 0.f);
    }
}

// (2) One workgroup per utterance: the two logits and their log-softmax.
__global__ __launch_bounds__(256) void sae_fc2_kernel(const float* __restrict__ h1, const float* __restrict__ w2,
                                                      const float* __restrict__ b2, float* __restrict__ logp,
                                                      const float* __restrict__ mask) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    float h = h1[(long)b * SAE_HID + threadIdx.x];
    if (mask) h *= mask[(long)b * SAE_HID + threadIdx.x];
    float z0 = block_sum_256(w2[threadIdx.x] * h, red);
    float z1 = block_sum_256(w2[SAE_HID + threadIdx.x] * h, red);
    if (threadIdx.x == 0) {
        z0 += b2[0];
        z1 += b2[1];
        const float m = fmaxf(z0, z1);
        const float lse = m + logf(expf(z0 - m) + expf(z1 - m));
        logp[b * 2 + 0] = z0 - lse;
        logp[b * 2 + 1] = z1 - lse;
    }
}

__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ recon, const float* __restrict__ b_dec,
                                                  const float* __restrict__ x, long n, float inv_n,
                                                  float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float d = recon[i] + b_dec[i & 1023] - x[i];
        s = fmaf(d, d, s);
    }
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = s * inv_n;  // one partial per workgroup
}

// out[0] = sum of n partials, added in index order by one workgroup (reproducible bits)
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) out[0] = s;
}

}  // namespace

hipError_t launch_sls_weights(const float* dots, long layer_stride, int layers, const float* fc0_b, float* w, int B,
                              int T, int slices, hipStream_t s) {
    if (B <= 0 || T <= 0 || slices <= 0 || layers <= 0 || layers > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sls_weight_kernel, dim3(B, layers), dim3(256), 0, s, dots, layer_stride, fc0_b, w, T, (long)B * T,
                       slices);
    return hipGetLastError();
}

hipError_t launch_sls_sum_pool(const float* xs, long layer_stride, int layers, const float* w, bf16_t* pooled, long ldp,
                               int B, int T, float bn_scale, float bn_shift, int fmt, hipStream_t s, const float* bn4,
                               float bn_eps, long seg) {
    if (B <= 0 || B > 65535 || T < 3 || layers < 1 || layers > 64 || ldp < (long)(T / 3) * (1024 / 3) ||
        (layer_stride % 4) || (reinterpret_cast<uintptr_t>(xs) & 15))
        return hipErrorInvalidValue;
    if (seg < 0 || (seg > 0 && (seg < (long)(T / 3) * (1024 / 3) || ldp < 3 * seg))) return hipErrorInvalidValue;
    SLSMI_FMT_DISPATCH(fmt, hipLaunchKernelGGL(sls_sum_pool_kernel<F>, dim3(T / 3, B), dim3(256), 0, s, xs, layer_stride,
                                               layers, w, pooled, ldp, T, bn_scale, bn_shift, bn4, bn_eps, seg));
    return hipGetLastError();
}

hipError_t launch_sls_tail(const float* fc1out, int parts, long part_stride, const float* b1, const float* w3,
                           const float* b3, float* logp, int B, hipStream_t s) {
    if (B <= 0 || parts < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sls_tail_kernel, dim3(B), dim3(256), 0, s, fc1out, parts, part_stride, b1, w3, b3, logp);
    return hipGetLastError();
}

hipError_t launch_topk_rows(float* enc, bf16_t* enc_bf16, long rows, int dict, int k, int fmt, hipStream_t s) {
    if (rows <= 0 || k <= 0 || k > dict) return hipErrorInvalidValue;
    dim3 grid((unsigned)rows);
    switch (dict) {
        case 1024: SLSMI_FMT_DISPATCH(fmt, hipLaunchKernelGGL((topk_rows_kernel<4, F>), grid, dim3(256), 0, s, enc, enc_bf16, dict, k)); break;
        case 2048: SLSMI_FMT_DISPATCH(fmt, hipLaunchKernelGGL((topk_rows_kernel<8, F>), grid, dim3(256), 0, s, enc, enc_bf16, dict, k)); break;
        case 4096: SLSMI_FMT_DISPATCH(fmt, hipLaunchKernelGGL((topk_rows_kernel<16, F>), grid, dim3(256), 0, s, enc, enc_bf16, dict, k)); break;
        case 8192: SLSMI_FMT_DISPATCH(fmt, hipLaunchKernelGGL((topk_rows_kernel<32, F>), grid, dim3(256), 0, s, enc, enc_bf16, dict, k)); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_window_topk(float* enc, bf16_t* enc_bf16, uint8_t* keep, int B, int T, int dict, int k, int window,
                              int fmt, hipStream_t s) {
    const int stride = window / 2 > 1 ? window / 2 : 1;
    if (B <= 0 || B > 65535 || window < 2 || T < window || k <= 0 || k > dict || !enc || !keep)
        return hipErrorInvalidValue;
    const int nW = (T - window) / stride + 1;
    dim3 g1(nW, B), g2(T, B);
#define SLSMI_WINDOW_CASE(V)                                                                                           \
    hipLaunchKernelGGL((window_keep_kernel<V>), g1, dim3(256), 0, s, enc, keep, T, dict, k, window, stride, nW);       \
    SLSMI_FMT_DISPATCH(fmt, hipLaunchKernelGGL((window_vote_topk_kernel<V, F>), g2, dim3(256), 0, s, enc, enc_bf16,   \
                                               keep, T, dict, k, window, stride, nW));                                 \
    break
    switch (dict) {
        case 256: SLSMI_WINDOW_CASE(1);
        case 512: SLSMI_WINDOW_CASE(2);
        case 1024: SLSMI_WINDOW_CASE(4);
        case 2048: SLSMI_WINDOW_CASE(8);
        case 4096: SLSMI_WINDOW_CASE(16);
        case 8192: SLSMI_WINDOW_CASE(32);
        default: return hipErrorInvalidValue;
    }
#undef SLSMI_WINDOW_CASE
    return hipGetLastError();
}

long window_topk_scratch_bytes(int B, int T, int dict, int window) {
    const int stride = window / 2 > 1 ? window / 2 : 1;
    if (window < 2 || T < window) return 0;
    return (long)B * ((T - window) / stride + 1) * dict;
}

hipError_t launch_col_mean(const float* enc, const float* add, float* pooled, int B, int T, int dict, hipStream_t s,
                           const int* nframes) {
    if (B <= 0 || T <= 0 || dict <= 0 || !enc) return hipErrorInvalidValue;
    hipLaunchKernelGGL(col_mean_kernel, dim3((dict + 255) / 256, B), dim3(256), 0, s, enc, add, pooled, T, dict, nframes);
    return hipGetLastError();
}

hipError_t launch_sae_tail(const float* pooled, const float* ln_g, const float* ln_b, const float* w1, const float* b1,
                           const float* w2, const float* b2, float* hidden, float* logp, int B, int dict, hipStream_t s,
                           const float* mask) {
    if (B <= 0 || B > 65535 || dict <= 0 || dict > 8192 || dict % 256 != 0 || !hidden) return hipErrorInvalidValue;
    const size_t shm = (size_t)(dict + 4) * sizeof(float);
    hipLaunchKernelGGL(sae_fc1_kernel, dim3(SAE_HID / SAE_HID_PER_WG, B), dim3(256), shm, s, pooled, ln_g, ln_b, w1, b1,
                       hidden, dict);
    hipLaunchKernelGGL(sae_fc2_kernel, dim3(B), dim3(256), 0, s, hidden, w2, b2, logp, mask);
    return hipGetLastError();
}

hipError_t launch_mse(const float* recon, const float* b_dec, const float* x, long rows, float* partial, float* out,
                      hipStream_t s) {
    if (rows <= 0 || !partial) return hipErrorInvalidValue;
    const long n = rows * 1024;
    hipLaunchKernelGGL(mse_kernel, dim3(MSE_PARTS), dim3(256), 0, s, recon, b_dec, x, n, 1.0f / (float)n, partial);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, partial, MSE_PARTS, out);
    return hipGetLastError();
}

}  // namespace slsmi
