// The upstream SLS head in training form, forward and backward (README.md:46 trains Model(args, device); of the
// head only getAttenF survives in the reference tree, model_backup.py:186-202):
//     y0[b][l]   = mean_t x_l[b][t][:]                       (adaptive_avg_pool1d over the frames)
//     w[b][l]    = sigmoid(fc0 . y0[b][l] + b0)
//     full[b]    = sum_l w[b][l] x_l[b]                      [T][1024]
//     z          = BatchNorm2d(1)(full) on BATCH statistics  (train mode), SELU, max_pool2d(3, 3), flatten
//     h          = SELU(fc1 z + b1);  o = SELU(fc3 h + b3);  log_softmax(o)
// Everything here is fp32 on the vector ALUs: the head is 0.04 % of the step's arithmetic and its two large
// operands (the layer outputs, fc1's 1024 x 22847 weight) are each read once per pass, at memory speed.
// Every reduction has a fixed order (per-workgroup partials, combined by index), so a step is reproducible.
#include "common.h"
#include "kernels.h"

namespace slsmi {

namespace {

constexpr float SELU_ALPHA = 1.6732632423543772848170429916717f;
constexpr float SELU_SCALE = 1.0507009873554804934193349852946f;
constexpr int PW = 1024 / 3;  // pooled columns

__device__ __forceinline__ float selu_grad(float x) { return SELU_SCALE * (x > 0.0f ? 1.0f : SELU_ALPHA * expf(x)); }

__device__ __forceinline__ float block_sum(float v, float* red) {  // 256 threads; the same value in every thread
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ double block_sum_d(double v, double* red) {  // 256 threads, fixed tree
    __syncthreads();
    red[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    return red[0];
}

// y0[b][i][c] = mean over the T frames of x_{kept[i]}[b T + t][c].  grid (4, kept, B): 256 columns per workgroup.
__global__ __launch_bounds__(256) void sls_layer_means_kernel(const float* __restrict__ xs, long layer_stride, SlsKept kept,
                                                              int T, float* __restrict__ y0) {
    const int c = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y, b = blockIdx.z;
    const float* p = xs + (long)kept.idx[i] * layer_stride + (long)b * T * 1024 + c;
    // four partial sums over t mod 4, eight loads in flight; combined in a fixed order
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int t = 0;
#pragma unroll 2
    for (; t + 4 <= T; t += 4) {
        s0 += p[(long)t * 1024];
        s1 += p[(long)(t + 1) * 1024];
        s2 += p[(long)(t + 2) * 1024];
        s3 += p[(long)(t + 3) * 1024];
    }
    for (; t < T; ++t) s0 += p[(long)t * 1024];
    y0[((long)b * kept.n + i) * 1024 + c] = ((s0 + s1) + (s2 + s3)) / (float)T;
}

// wl[b][i] = sigmoid(fc0 . y0[b][i] + b0).  One workgroup per (b, i).
__global__ __launch_bounds__(256) void sls_layer_weights_kernel(const float* __restrict__ y0, const float* __restrict__ fc0_w,
                                                                const float* __restrict__ fc0_b, float* __restrict__ wl) {
    __shared__ float red[4];
    const long r = blockIdx.x;
    const f32x4 y = *reinterpret_cast<const f32x4*>(y0 + r * 1024 + threadIdx.x * 4);
    const f32x4 w = *reinterpret_cast<const f32x4*>(fc0_w + threadIdx.x * 4);
    const float z = block_sum((y[0] * w[0] + y[1] * w[1]) + (y[2] * w[2] + y[3] * w[3]), red) + fc0_b[0];
    if (threadIdx.x == 0) wl[r] = 1.0f / (1.0f + expf(-z));
}

// full[m][:] = sum_i wl[b][i] x_{kept[i]}[m][:], and the row's mean and sum of squared deviations (for the
// BatchNorm's batch statistics: rows have equal counts, so the row results combine exactly).  One workgroup per row.
__global__ __launch_bounds__(256) void sls_weighted_sum_kernel(const float* __restrict__ xs, long layer_stride, SlsKept kept,
                                                               const float* __restrict__ wl, int T, float* __restrict__ full,
                                                               float* __restrict__ part) {
    __shared__ float red[4];
    __shared__ float w[64];
    const long m = blockIdx.x;
    const int b = (int)(m / T);
    if ((int)threadIdx.x < kept.n) w[threadIdx.x] = wl[(long)b * kept.n + threadIdx.x];
    __syncthreads();
    f32x4 f = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < kept.n; ++i)
        f += *reinterpret_cast<const f32x4*>(xs + (long)kept.idx[i] * layer_stride + m * 1024 + threadIdx.x * 4) * w[i];
    *reinterpret_cast<f32x4*>(full + m * 1024 + threadIdx.x * 4) = f;
    const float mean = block_sum((f[0] + f[1]) + (f[2] + f[3]), red) * (1.0f / 1024.0f);
    const f32x4 d = f - mean;
    const float m2 = block_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]), red);
    if (threadIdx.x == 0) {
        part[m * 2] = mean;
        part[m * 2 + 1] = m2;
    }
}

// Batch statistics over all rows x 1024 values from the rows' (mean, sum of squared deviations), in double:
// stat = {mean, 1 / sqrt(var + eps), biased var}; batch_out (may be null) = {mean, unbiased var}, what
// nn.BatchNorm2d folds into its running statistics.  One workgroup.
__global__ __launch_bounds__(256) void sls_bn_stats_kernel(const float* __restrict__ part, long rows, float eps,
                                                           float* __restrict__ stat, float* __restrict__ batch_out) {
    __shared__ double red[256];
    const long per = (rows + 255) / 256;
    const long lo = threadIdx.x * per, hi = lo + per < rows ? lo + per : rows;
    double s = 0.0;
    for (long r = lo; r < hi; ++r) s += (double)part[r * 2];
    const double mean = block_sum_d(s, red) / (double)rows;
    double q = 0.0;
    for (long r = lo; r < hi; ++r) {
        const double d = (double)part[r * 2] - mean;
        q += (double)part[r * 2 + 1] + 1024.0 * d * d;
    }
    const double M2 = block_sum_d(q, red);
    if (threadIdx.x == 0) {
        const double n = (double)rows * 1024.0;
        const double var = M2 / n;
        stat[0] = (float)mean;
        stat[1] = (float)(1.0 / sqrt(var + (double)eps));
        stat[2] = (float)var;
        if (batch_out) {
            batch_out[0] = (float)mean;
            batch_out[1] = (float)(M2 / (n - 1.0));
        }
    }
}

// pooled[b][pi PW + pj] = max over the 3 x 3 window of selu(gamma (full - mean) rstd + beta) and which of the nine
// it was (3 di + dj; the first in that order among equals, as max_pool2d's scan).  grid (T / 3, B).
__global__ __launch_bounds__(256) void sls_bn_pool_kernel(const float* __restrict__ full, const float* __restrict__ stat,
                                                          const float* __restrict__ bn, int T, long P,
                                                          float* __restrict__ pooled, uint8_t* __restrict__ amax) {
    __shared__ float colmax[1024];
    __shared__ uint8_t colarg[1024];
    const int pi = blockIdx.x, b = blockIdx.y;
    const float mean = stat[0], rstd = stat[1], gamma = bn[0], beta = bn[1];
    const float* base = full + ((long)b * T + 3 * pi) * 1024 + threadIdx.x * 4;
    f32x4 best;
    int arg[4];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const f32x4 f = *reinterpret_cast<const f32x4*>(base + d * 1024);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float y = selu(fmaf((f[e] - mean) * rstd, gamma, beta));
            if (d == 0 || y > best[e]) {
                best[e] = y;
                arg[e] = d;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        colmax[threadIdx.x * 4 + e] = best[e];
        colarg[threadIdx.x * 4 + e] = (uint8_t)arg[e];
    }
    __syncthreads();
    for (int pj = threadIdx.x; pj < PW; pj += 256) {
        float v = colmax[3 * pj];
        int a = 3 * colarg[3 * pj];
#pragma unroll
        for (int dj = 1; dj < 3; ++dj) {
            const float u = colmax[3 * pj + dj];
            const int ua = 3 * colarg[3 * pj + dj] + dj;
            if (u > v || (u == v && ua < a)) {
                v = u;
                a = ua;
            }
        }
        pooled[(long)b * P + (long)pi * PW + pj] = v;
        amax[(long)b * P + (long)pi * PW + pj] = (uint8_t)a;
    }
}

// h1[b][n] = b1[n] + sum_k pooled[b][k] W1[n][k].  One workgroup per four output rows n (each pooled value read is used
// four times); clips in passes of eight.
constexpr int FC1_ROWS = 4;
__global__ __launch_bounds__(256) void sls_fc1_fwd_kernel(const float* __restrict__ pooled, const float* __restrict__ W1,
                                                          const float* __restrict__ b1, int B, long P, float* __restrict__ h1) {
    __shared__ float red[4];
    const int n0 = blockIdx.x * FC1_ROWS;
    const float* wrow = W1 + (long)n0 * P;
    for (int b0 = 0; b0 < B; b0 += 8) {
        float acc[FC1_ROWS][8];
#pragma unroll
        for (int r = 0; r < FC1_ROWS; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[r][j] = 0.f;
        const int nb = B - b0 < 8 ? B - b0 : 8;
#pragma unroll 2
        for (long k = threadIdx.x; k < P; k += 256) {
            float w[FC1_ROWS], x[8];
#pragma unroll
            for (int r = 0; r < FC1_ROWS; ++r) w[r] = wrow[(long)r * P + k];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = j < nb ? pooled[(long)(b0 + j) * P + k] : 0.f;
#pragma unroll
            for (int r = 0; r < FC1_ROWS; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[r][j] = fmaf(w[r], x[j], acc[r][j]);
        }
#pragma unroll
        for (int r = 0; r < FC1_ROWS; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j >= nb) break;  // uniform
                const float v = block_sum(acc[r][j], red);
                if (threadIdx.x == 0) h1[(long)(b0 + j) * 1024 + n0 + r] = v + b1[n0 + r];
            }
    }
}

// a3[b][j] = b3[j] + sum_n selu(h1[b][n]) W3[j][n];  logp[b] = log_softmax(selu(a3[b])).  One workgroup per clip.
__global__ __launch_bounds__(256) void sls_tail_fwd_kernel(const float* __restrict__ h1, const float* __restrict__ w3,
                                                           const float* __restrict__ b3, float* __restrict__ a3,
                                                           float* __restrict__ logp) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = i * 256 + threadIdx.x;
        const float h = selu(h1[(long)b * 1024 + n]);
        s0 = fmaf(w3[n], h, s0);
        s1 = fmaf(w3[1024 + n], h, s1);
    }
    s0 = block_sum(s0, red);
    s1 = block_sum(s1, red);
    if (threadIdx.x == 0) {
        const float p0 = s0 + b3[0], p1 = s1 + b3[1];
        a3[b * 2] = p0;
        a3[b * 2 + 1] = p1;
        const float z0 = selu(p0), z1 = selu(p1);
        const float mx = fmaxf(z0, z1);
        const float lse = mx + logf(expf(z0 - mx) + expf(z1 - mx));
        logp[b * 2] = z0 - lse;
        logp[b * 2 + 1] = z1 - lse;
    }
}

// Back through log_softmax, SELU, fc3, SELU: dh1[b][n] (gradient at fc1's output), and the gradients of fc3's
// weight and bias and fc1's bias (each may be null).  One workgroup; clips are added in index order.
__global__ __launch_bounds__(256) void sls_tail_bwd_kernel(const float* __restrict__ dlogp, const float* __restrict__ logp,
                                                           const float* __restrict__ a3, const float* __restrict__ h1,
                                                           const float* __restrict__ w3, int B, float* __restrict__ dh1,
                                                           float* __restrict__ dW3, float* __restrict__ db3,
                                                           float* __restrict__ db1) {
    __shared__ float da3[64][2];
    for (int b = threadIdx.x; b < B; b += 256) {
        const float g0 = dlogp[b * 2], g1 = dlogp[b * 2 + 1];
        const float sum = g0 + g1;
        da3[b][0] = (g0 - expf(logp[b * 2]) * sum) * selu_grad(a3[b * 2]);
        da3[b][1] = (g1 - expf(logp[b * 2 + 1]) * sum) * selu_grad(a3[b * 2 + 1]);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = i * 256 + threadIdx.x;
        const float w0 = w3[n], w1 = w3[1024 + n];
        float g0 = 0.f, g1 = 0.f, gb = 0.f;
        for (int b = 0; b < B; ++b) {
            const float pre = h1[(long)b * 1024 + n];
            const float h = selu(pre);
            g0 = fmaf(da3[b][0], h, g0);
            g1 = fmaf(da3[b][1], h, g1);
            const float d = fmaf(da3[b][0], w0, da3[b][1] * w1) * selu_grad(pre);
            dh1[(long)b * 1024 + n] = d;
            gb += d;
        }
        if (dW3) {
            dW3[n] = g0;
            dW3[1024 + n] = g1;
        }
        if (db1) db1[n] = gb;
    }
    if (db3 && threadIdx.x < 2) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += da3[b][threadIdx.x];
        db3[threadIdx.x] = s;
    }
}

// fc1 backward in one pass over W1: dW1[n][k] = sum_b dh1[b][n] pooled[b][k] (null: not wanted) and, per chunk of 64
// output rows, dpart[chunk][b][k] = sum_{n in chunk} dh1[b][n] W1[n][k].  grid (ceil(P / 256), 16); a thread owns one k.
constexpr int FC1_CHUNK = 64, FC1_CHUNKS = 1024 / FC1_CHUNK, FC1_PASS = 16;
__global__ __launch_bounds__(256) void sls_fc1_bwd_kernel(const float* __restrict__ dh1, const float* __restrict__ pooled,
                                                          const float* __restrict__ W1, int B, long P,
                                                          float* __restrict__ dW1, float* __restrict__ dpart) {
    __shared__ float g[FC1_CHUNK][FC1_PASS];
    const long k = (long)blockIdx.x * 256 + threadIdx.x;
    const int n0 = blockIdx.y * FC1_CHUNK;
    const bool live = k < P;
    for (int b0 = 0; b0 < B; b0 += FC1_PASS) {
        const int nb = B - b0 < FC1_PASS ? B - b0 : FC1_PASS;
        __syncthreads();
        for (int i = threadIdx.x; i < FC1_CHUNK * FC1_PASS; i += 256) {
            const int nn = i / FC1_PASS, j = i % FC1_PASS;
            g[nn][j] = j < nb ? dh1[(long)(b0 + j) * 1024 + n0 + nn] : 0.f;
        }
        __syncthreads();
        float pk[FC1_PASS], dp[FC1_PASS];
#pragma unroll
        for (int j = 0; j < FC1_PASS; ++j) {
            pk[j] = live && j < nb ? pooled[(long)(b0 + j) * P + k] : 0.f;
            dp[j] = 0.f;
        }
        if (live) {
            for (int nn = 0; nn < FC1_CHUNK; ++nn) {
                const long at = (long)(n0 + nn) * P + k;
                const float w = W1[at];
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < FC1_PASS; ++j) {
                    acc = fmaf(g[nn][j], pk[j], acc);
                    dp[j] = fmaf(g[nn][j], w, dp[j]);
                }
                if (dW1) dW1[at] = b0 == 0 ? acc : dW1[at] + acc;
            }
#pragma unroll
            for (int j = 0; j < FC1_PASS; ++j)
                if (j < nb) dpart[((long)blockIdx.y * B + b0 + j) * P + k] = dp[j];
        }
    }
}

// Back through the max pool and the SELU: the pooled cell's gradient (the chunks' partial sums added in index
// order) goes to the window's largest entry, times selu'(BatchNorm output there).  dbn (cleared by the caller)
// receives it; part2[(b, pi)] = the workgroup's sums of g and of g xhat.  grid (T / 3, B).
__global__ __launch_bounds__(256) void sls_pool_bwd_kernel(const float* __restrict__ dpart, int B, const uint8_t* __restrict__ amax,
                                                           const float* __restrict__ full, const float* __restrict__ stat,
                                                           const float* __restrict__ bn, int T, long P, float* __restrict__ dbn,
                                                           float* __restrict__ part2) {
    __shared__ float red[4];
    const int pi = blockIdx.x, b = blockIdx.y;
    const float mean = stat[0], rstd = stat[1], gamma = bn[0], beta = bn[1];
    float s1 = 0.f, s2 = 0.f;
    for (int pj = threadIdx.x; pj < PW; pj += 256) {
        const long cell = (long)pi * PW + pj;
        float dp = 0.f;
        for (int ch = 0; ch < FC1_CHUNKS; ++ch) dp += dpart[((long)ch * B + b) * P + cell];
        const int a = amax[(long)b * P + cell];
        const long at = ((long)b * T + 3 * pi + a / 3) * 1024 + 3 * pj + a % 3;
        const float xh = (full[at] - mean) * rstd;
        const float gv = dp * selu_grad(fmaf(xh, gamma, beta));
        dbn[at] = gv;
        s1 += gv;
        s2 = fmaf(gv, xh, s2);
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        part2[((long)b * gridDim.x + pi) * 2] = s1;
        part2[((long)b * gridDim.x + pi) * 2 + 1] = s2;
    }
}

// S1 = sum g, S2 = sum g xhat over the batch (double, fixed order): the BatchNorm's bias and weight gradients, and
// stat2 = {S1 / N, S2 / N}, the two means its input gradient subtracts.  One workgroup.
__global__ __launch_bounds__(256) void sls_bn_bwd_stats_kernel(const float* __restrict__ part2, int n, double N,
                                                               float* __restrict__ stat2, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta) {
    __shared__ double red[256];
    const int per = (n + 255) / 256;
    const int lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    double a = 0.0, q = 0.0;
    for (int r = lo; r < hi; ++r) {
        a += (double)part2[r * 2];
        q += (double)part2[r * 2 + 1];
    }
    const double S1 = block_sum_d(a, red);
    const double S2 = block_sum_d(q, red);
    if (threadIdx.x == 0) {
        stat2[0] = (float)(S1 / N);
        stat2[1] = (float)(S2 / N);
        if (dgamma) dgamma[0] = (float)S2;
        if (dbeta) dbeta[0] = (float)S1;
    }
}

// BatchNorm input gradient on batch statistics, in place: d[m][c] <- gamma rstd (d - mean(g) - xhat mean(g xhat)).
__global__ __launch_bounds__(256) void sls_dfull_kernel(float* __restrict__ d, const float* __restrict__ full,
                                                        const float* __restrict__ stat, const float* __restrict__ stat2,
                                                        const float* __restrict__ bn, long n4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float mean = stat[0], rstd = stat[1], k1 = stat2[0], k2 = stat2[1], sc = bn[0] * rstd;
    const f32x4 f = *reinterpret_cast<const f32x4*>(full + i * 4);
    f32x4 g = *reinterpret_cast<const f32x4*>(d + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = sc * ((g[e] - k1) - (f[e] - mean) * rstd * k2);
    *reinterpret_cast<f32x4*>(d + i * 4) = g;
}

// dwl_part[chunk][b][i] = sum over 32 of the clip's frames and all channels of dfull x_{kept[i]} (the chunks are added
// in index order by sls_fc0_bwd_kernel).  grid (kept, B, ceil(T / 32)).
constexpr int DW_FRAMES = 32;
__global__ __launch_bounds__(256) void sls_dweights_kernel(const float* __restrict__ xs, long layer_stride, SlsKept kept,
                                                           const float* __restrict__ dfull, int T, float* __restrict__ dwl_part) {
    __shared__ float red[4];
    const int i = blockIdx.x, b = blockIdx.y, t0 = blockIdx.z * DW_FRAMES;
    const int t1 = t0 + DW_FRAMES < T ? t0 + DW_FRAMES : T;
    const float* x = xs + (long)kept.idx[i] * layer_stride + (long)b * T * 1024 + threadIdx.x * 4;
    const float* d = dfull + (long)b * T * 1024 + threadIdx.x * 4;
    float s = 0.f;
#pragma unroll 4
    for (int t = t0; t < t1; ++t) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (long)t * 1024);
        const f32x4 dv = *reinterpret_cast<const f32x4*>(d + (long)t * 1024);
        s += (xv[0] * dv[0] + xv[1] * dv[1]) + (xv[2] * dv[2] + xv[3] * dv[3]);
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) dwl_part[((long)blockIdx.z * gridDim.y + b) * kept.n + i] = s;
}

// Back through the sigmoid and fc0: dz[b][i] = dwl w (1 - w), dwl = the frame chunks' sums in index order; fc0's weight gradient sum_(b, i) dz y0[b][i][:] and bias
// gradient sum dz (each may be null).  One workgroup; pairs in index order.
__global__ __launch_bounds__(256) void sls_fc0_bwd_kernel(const float* __restrict__ dwl_part, int chunks,
                                                          const float* __restrict__ wl, const float* __restrict__ y0, int pairs,
                                                          float* __restrict__ dz, float* __restrict__ dfc0_w,
                                                          float* __restrict__ dfc0_b) {
    for (int r = threadIdx.x; r < pairs; r += 256) {
        float dw = 0.f;
        for (int ch = 0; ch < chunks; ++ch) dw += dwl_part[(long)ch * pairs + r];
        const float w = wl[r];
        dz[r] = dw * w * (1.0f - w);
    }
    __threadfence_block();
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float sb = 0.f;
#pragma unroll 8
    for (int r = 0; r < pairs; ++r) {
        const float z = dz[r];
        acc += *reinterpret_cast<const f32x4*>(y0 + (long)r * 1024 + threadIdx.x * 4) * z;
        sb += z;
    }
    if (dfc0_w) *reinterpret_cast<f32x4*>(dfc0_w + threadIdx.x * 4) = acc;
    if (dfc0_b && threadIdx.x == 0) dfc0_b[0] = sb;
}

// The loss gradient at the kept layers' outputs: dlayers[kept[i]][m][:] = wl[b][i] dfull[m][:] + dz[b][i] fc0_w / T
// (the weighted sum's share and the time mean's).  grid (rows, kept).
__global__ __launch_bounds__(256) void sls_dlayers_kernel(const float* __restrict__ dfull, const float* __restrict__ wl,
                                                          const float* __restrict__ dz, const float* __restrict__ fc0_w,
                                                          SlsKept kept, int T, float* __restrict__ dlayers, long dl_stride) {
    const long m = blockIdx.x;
    const int i = blockIdx.y, b = (int)(m / T);
    const float w = wl[(long)b * kept.n + i], zt = dz[(long)b * kept.n + i] / (float)T;
    const f32x4 d = *reinterpret_cast<const f32x4*>(dfull + m * 1024 + threadIdx.x * 4);
    const f32x4 f0 = *reinterpret_cast<const f32x4*>(fc0_w + threadIdx.x * 4);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = fmaf(w, d[e], zt * f0[e]);
    *reinterpret_cast<f32x4*>(dlayers + (long)kept.idx[i] * dl_stride + m * 1024 + threadIdx.x * 4) = o;
}

}  // namespace

int sls_fc1_chunks() { return FC1_CHUNKS; }

hipError_t launch_sls_layer_means(const float* xs, long layer_stride, const SlsKept& kept, int B, int T, float* y0, hipStream_t s) {
    hipLaunchKernelGGL(sls_layer_means_kernel, dim3(4, kept.n, B), dim3(256), 0, s, xs, layer_stride, kept, T, y0);
    return hipGetLastError();
}

hipError_t launch_sls_layer_weights(const float* y0, const float* fc0_w, const float* fc0_b, int pairs, float* wl, hipStream_t s) {
    hipLaunchKernelGGL(sls_layer_weights_kernel, dim3(pairs), dim3(256), 0, s, y0, fc0_w, fc0_b, wl);
    return hipGetLastError();
}

hipError_t launch_sls_weighted_sum(const float* xs, long layer_stride, const SlsKept& kept, const float* wl, int B, int T,
                                   float* full, float* part, hipStream_t s) {
    hipLaunchKernelGGL(sls_weighted_sum_kernel, dim3((unsigned)((long)B * T)), dim3(256), 0, s, xs, layer_stride, kept, wl, T, full, part);
    return hipGetLastError();
}

hipError_t launch_sls_bn_stats(const float* part, long rows, float eps, float* stat, float* batch_out, hipStream_t s) {
    hipLaunchKernelGGL(sls_bn_stats_kernel, dim3(1), dim3(256), 0, s, part, rows, eps, stat, batch_out);
    return hipGetLastError();
}

hipError_t launch_sls_bn_pool(const float* full, const float* stat, const float* bn, int B, int T, float* pooled, uint8_t* amax,
                              hipStream_t s) {
    const long P = (long)(T / 3) * PW;
    hipLaunchKernelGGL(sls_bn_pool_kernel, dim3(T / 3, B), dim3(256), 0, s, full, stat, bn, T, P, pooled, amax);
    return hipGetLastError();
}

hipError_t launch_sls_fc1_fwd(const float* pooled, const float* W1, const float* b1, int B, long P, float* h1, hipStream_t s) {
    hipLaunchKernelGGL(sls_fc1_fwd_kernel, dim3(1024 / FC1_ROWS), dim3(256), 0, s, pooled, W1, b1, B, P, h1);
    return hipGetLastError();
}

hipError_t launch_sls_tail_fwd(const float* h1, const float* w3, const float* b3, int B, float* a3, float* logp, hipStream_t s) {
    hipLaunchKernelGGL(sls_tail_fwd_kernel, dim3(B), dim3(256), 0, s, h1, w3, b3, a3, logp);
    return hipGetLastError();
}

hipError_t launch_sls_tail_bwd(const float* dlogp, const float* logp, const float* a3, const float* h1, const float* w3, int B,
                               float* dh1, float* dW3, float* db3, float* db1, hipStream_t s) {
    if (B > 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sls_tail_bwd_kernel, dim3(1), dim3(256), 0, s, dlogp, logp, a3, h1, w3, B, dh1, dW3, db3, db1);
    return hipGetLastError();
}

hipError_t launch_sls_fc1_bwd(const float* dh1, const float* pooled, const float* W1, int B, long P, float* dW1, float* dpart,
                              hipStream_t s) {
    hipLaunchKernelGGL(sls_fc1_bwd_kernel, dim3((unsigned)((P + 255) / 256), FC1_CHUNKS), dim3(256), 0, s, dh1, pooled, W1, B, P,
                       dW1, dpart);
    return hipGetLastError();
}

hipError_t launch_sls_pool_bwd(const float* dpart, const uint8_t* amax, const float* full, const float* stat, const float* bn,
                               int B, int T, float* dbn, float* part2, hipStream_t s) {
    const long P = (long)(T / 3) * PW;
    hipLaunchKernelGGL(sls_pool_bwd_kernel, dim3(T / 3, B), dim3(256), 0, s, dpart, B, amax, full, stat, bn, T, P, dbn, part2);
    return hipGetLastError();
}

hipError_t launch_sls_bn_bwd_stats(const float* part2, int n, long count, float* stat2, float* dgamma, float* dbeta, hipStream_t s) {
    hipLaunchKernelGGL(sls_bn_bwd_stats_kernel, dim3(1), dim3(256), 0, s, part2, n, (double)count, stat2, dgamma, dbeta);
    return hipGetLastError();
}

hipError_t launch_sls_dfull(float* d, const float* full, const float* stat, const float* stat2, const float* bn, long rows,
                            hipStream_t s) {
    const long n4 = rows * 256;
    hipLaunchKernelGGL(sls_dfull_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, d, full, stat, stat2, bn, n4);
    return hipGetLastError();
}

int sls_dweight_chunks(int T) { return (T + DW_FRAMES - 1) / DW_FRAMES; }

hipError_t launch_sls_dweights(const float* xs, long layer_stride, const SlsKept& kept, const float* dfull, int B, int T,
                               float* dwl_part, hipStream_t s) {
    hipLaunchKernelGGL(sls_dweights_kernel, dim3(kept.n, B, sls_dweight_chunks(T)), dim3(256), 0, s, xs, layer_stride, kept, dfull,
                       T, dwl_part);
    return hipGetLastError();
}

hipError_t launch_sls_fc0_bwd(const float* dwl_part, int chunks, const float* wl, const float* y0, int pairs, float* dz,
                              float* dfc0_w, float* dfc0_b, hipStream_t s) {
    hipLaunchKernelGGL(sls_fc0_bwd_kernel, dim3(1), dim3(256), 0, s, dwl_part, chunks, wl, y0, pairs, dz, dfc0_w, dfc0_b);
    return hipGetLastError();
}

hipError_t launch_sls_dlayers(const float* dfull, const float* wl, const float* dz, const float* fc0_w, const SlsKept& kept,
                              int B, int T, float* dlayers, long dl_stride, hipStream_t s) {
    hipLaunchKernelGGL(sls_dlayers_kernel, dim3((unsigned)((long)B * T), kept.n), dim3(256), 0, s, dfull, wl, dz, fc0_w, kept, T,
                       dlayers, dl_stride);
    return hipGetLastError();
}

}  // namespace slsmi
