// Adam step of the training recipe (main.py:526 builds torch.optim.Adam(lr, weight_decay), main.py:238 steps it)
// over all parameter tensors in ONE launch: a table of tensors, a list of (tensor, chunk) work items, one
// workgroup per item.  HBM-bound: 16 bytes read and 12 written per parameter (p, g, m, v -> p, m, v), 16-byte
// accesses, nothing staged.  The arithmetic is torch's single-tensor Adam in fp32, in its order:
//   g' = g + wd p;  m += (1 - b1) (g' - m);  v = b2 v + (1 - b2) g' g';
//   p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include <cmath>

#include "common.h"
#include "kernels.h"

namespace slsmi {

namespace {

constexpr int ADAM_THREADS = 256;

struct AdamConsts {
    float one_minus_b1, b2, one_minus_b2, wd, step_size, inv_sqrt_bc2, eps;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamConsts& c) {
    g = __builtin_fmaf(c.wd, p, g);
    m = __builtin_fmaf(c.one_minus_b1, g - m, m);
    v = __builtin_fmaf(c.one_minus_b2 * g, g, c.b2 * v);
    const float denom = __builtin_fmaf(__builtin_sqrtf(v), c.inv_sqrt_bc2, c.eps);
    p = __builtin_fmaf(-c.step_size, m / denom, p);
}

// tab[t] = {p, g, m, v, n} as five 64-bit words; items[i] = {tensor, chunk}: elements chunk * chunk_elems ...
__global__ __launch_bounds__(ADAM_THREADS) void adam_kernel(const long long* __restrict__ tab,
                                                            const int2* __restrict__ items, int chunk_elems,
                                                            AdamConsts c) {
    const int2 it = items[blockIdx.x];
    const long long* e = tab + 5 * (long)it.x;
    float* __restrict__ p = reinterpret_cast<float*>(e[0]);
    const float* __restrict__ g = reinterpret_cast<const float*>(e[1]);
    float* __restrict__ m = reinterpret_cast<float*>(e[2]);
    float* __restrict__ v = reinterpret_cast<float*>(e[3]);
    const long n = e[4];
    const long first = (long)it.y * chunk_elems;
    const long last = first + chunk_elems < n ? first + chunk_elems : n;
    const bool vec = ((e[0] | e[1] | e[2] | e[3]) & 15) == 0;  // chunk_elems is a multiple of 4
    if (vec) {
        const long nv = (last - first) >> 2;
        f32x4* p4 = reinterpret_cast<f32x4*>(p + first);
        const f32x4* g4 = reinterpret_cast<const f32x4*>(g + first);
        f32x4* m4 = reinterpret_cast<f32x4*>(m + first);
        f32x4* v4 = reinterpret_cast<f32x4*>(v + first);
        auto four = [&](f32x4& pp, const f32x4& gg, f32x4& mm, f32x4& vv) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float pr = pp[r], mr = mm[r], vr = vv[r];
                adam_one(pr, gg[r], mr, vr, c);
                pp[r] = pr;
                mm[r] = mr;
                vv[r] = vr;
            }
        };
        long i = threadIdx.x;
        for (; i + ADAM_THREADS < nv; i += 2 * ADAM_THREADS) {  // two 16-byte pieces of every array in flight per lane
            const long j = i + ADAM_THREADS;
            f32x4 pa = p4[i], pb = p4[j], ma = m4[i], mb = m4[j], va = v4[i], vb = v4[j];
            const f32x4 ga = g4[i], gb = g4[j];
            four(pa, ga, ma, va);
            four(pb, gb, mb, vb);
            p4[i] = pa;
            p4[j] = pb;
            m4[i] = ma;
            m4[j] = mb;
            v4[i] = va;
            v4[j] = vb;
        }
        for (; i < nv; i += ADAM_THREADS) {
            f32x4 pp = p4[i], mm = m4[i], vv = v4[i];
            const f32x4 gg = g4[i];
            four(pp, gg, mm, vv);
            p4[i] = pp;
            m4[i] = mm;
            v4[i] = vv;
        }
        for (long i = first + (nv << 2) + threadIdx.x; i < last; i += ADAM_THREADS) adam_one(p[i], g[i], m[i], v[i], c);
    } else {
        for (long i = first + threadIdx.x; i < last; i += ADAM_THREADS) adam_one(p[i], g[i], m[i], v[i], c);
    }
}

}  // namespace

hipError_t launch_adam(const long long* tab, const int2* items, long n_items, int chunk_elems, float lr, float beta1,
                       float beta2, float eps, float weight_decay, long step, hipStream_t s) {
    if (!tab || !items || n_items < 1 || n_items > 0x7fffffffL || chunk_elems < 4 || (chunk_elems & 3) || step < 1)
        return hipErrorInvalidValue;
    AdamConsts c;
    const double bc1 = 1.0 - std::pow((double)beta1, (double)step);
    const double bc2 = 1.0 - std::pow((double)beta2, (double)step);
    c.one_minus_b1 = 1.0f - beta1;
    c.b2 = beta2;
    c.one_minus_b2 = 1.0f - beta2;
    c.wd = weight_decay;
    c.step_size = (float)((double)lr / bc1);
    c.inv_sqrt_bc2 = (float)(1.0 / std::sqrt(bc2));
    c.eps = eps;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)n_items), dim3(ADAM_THREADS), 0, s, tab, items, chunk_elems, c);
    return hipGetLastError();
}

}  // namespace slsmi
