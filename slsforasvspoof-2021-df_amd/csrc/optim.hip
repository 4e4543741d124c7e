// Adam step of the training recipe (main.py:526 builds torch.optim.Adam(lr, weight_decay), main.py:238 steps it)
// over all parameter tensors in ONE launch: a table of tensors, a list of (tensor, chunk) work items, one
// workgroup per item.  HBM-bound: 16 bytes read and 12 written per parameter (p, g, m, v -> p, m, v), 16-byte
// accesses, nothing staged; every stream is touched once per step, so loads and stores carry the non-temporal hint
// (1.27 GB per array: nothing would survive in a cache until the next step; measured 5 % off the kernel's time).  The arithmetic is torch's single-tensor Adam in fp32, in its order:
//   g' = g + wd p;  m += (1 - b1) (g' - m);  v = b2 v + (1 - b2) g' g';
//   p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// with t the TENSOR's own step count (torch keeps one per parameter: a parameter that had no gradient in some step --
// a layer skipped by layerdrop -- falls behind the others) and every constant formed in double from the double
// hyper-parameters and rounded once, as torch's Python scalars are (1 - 0.999 in float is 0.00099998713).
#include "adam.h"
#include "kernels.h"

namespace slsmi {

namespace {

constexpr int ADAM_THREADS = 256;

// tab[t] = {p, g, m, v, n, step} as six 64-bit words; items[i] = {tensor, chunk}: elements chunk * chunk_elems ...
__global__ __launch_bounds__(ADAM_THREADS) void adam_kernel(const long long* __restrict__ tab,
                                                            const int2* __restrict__ items, int chunk_elems,
                                                            AdamHyper h) {
    const int2 it = items[blockIdx.x];
    const long long* e = tab + 6 * (long)it.x;
    const AdamConsts c = adam_consts(h, e[5]);
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
            f32x4 pa = __builtin_nontemporal_load(p4 + i), pb = __builtin_nontemporal_load(p4 + j),
                  ma = __builtin_nontemporal_load(m4 + i), mb = __builtin_nontemporal_load(m4 + j),
                  va = __builtin_nontemporal_load(v4 + i), vb = __builtin_nontemporal_load(v4 + j);
            const f32x4 ga = __builtin_nontemporal_load(g4 + i), gb = __builtin_nontemporal_load(g4 + j);
            four(pa, ga, ma, va);
            four(pb, gb, mb, vb);
            __builtin_nontemporal_store(pa, p4 + i);
            __builtin_nontemporal_store(pb, p4 + j);
            __builtin_nontemporal_store(ma, m4 + i);
            __builtin_nontemporal_store(mb, m4 + j);
            __builtin_nontemporal_store(va, v4 + i);
            __builtin_nontemporal_store(vb, v4 + j);
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

hipError_t launch_adam(const long long* tab, const int2* items, long n_items, int chunk_elems, double lr, double beta1,
                       double beta2, double eps, double weight_decay, hipStream_t s) {
    if (!tab || !items || n_items < 1 || n_items > 0x7fffffffL || chunk_elems < 4 || (chunk_elems & 3))
        return hipErrorInvalidValue;
    const AdamHyper h = adam_hyper(lr, beta1, beta2, eps, weight_decay);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)n_items), dim3(ADAM_THREADS), 0, s, tab, items, chunk_elems, h);
    return hipGetLastError();
}

}  // namespace slsmi
