// torch.optim.Adam's single-tensor arithmetic (L2 weight decay, no amsgrad) as device functions, shared by the
// optimizer kernel (optim.hip) and by the weight-gradient kernel that takes the step in its epilogue (backward.hip):
// one definition, so both forms produce the same bits.
//   g' = g + wd p;  m += (1 - b1) (g' - m);  v = b2 v + (1 - b2) g' g';
//   p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// with t the TENSOR's own step count and every constant formed in double from the double hyper-parameters and
// rounded once, as torch's Python scalars are (1 - 0.999 in float is 0.00099998713).
#pragma once
#include <cmath>

#include "common.h"

namespace slsmi {

struct AdamHyper {
    double lr, b1, b2;
    float one_minus_b1, b2f, one_minus_b2, wd, eps;
};

struct AdamConsts {
    float one_minus_b1, b2, one_minus_b2, wd, step_size, inv_sqrt_bc2, eps;
};

inline AdamHyper adam_hyper(double lr, double beta1, double beta2, double eps, double weight_decay) {
    AdamHyper h;
    h.lr = lr;
    h.b1 = beta1;
    h.b2 = beta2;
    h.one_minus_b1 = (float)(1.0 - beta1);
    h.b2f = (float)beta2;
    h.one_minus_b2 = (float)(1.0 - beta2);
    h.wd = (float)weight_decay;
    h.eps = (float)eps;
    return h;
}

// the tensor's bias corrections at its step count t, in double (uniform over a workgroup: a few hundred scalar-like
// instructions against thousands of elements of memory traffic)
__device__ __forceinline__ AdamConsts adam_consts(const AdamHyper& h, long long step) {
    AdamConsts c;
    const double t = (double)step;
    const double bc1 = 1.0 - pow(h.b1, t), bc2 = 1.0 - pow(h.b2, t);
    c.one_minus_b1 = h.one_minus_b1;
    c.b2 = h.b2f;
    c.one_minus_b2 = h.one_minus_b2;
    c.wd = h.wd;
    c.eps = h.eps;
    c.step_size = (float)(h.lr / bc1);
    c.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    return c;
}

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamConsts& c) {
    g = __builtin_fmaf(c.wd, p, g);
    m = __builtin_fmaf(c.one_minus_b1, g - m, m);
    v = __builtin_fmaf(c.one_minus_b2 * g, g, c.b2 * v);
    const float denom = __builtin_fmaf(__builtin_sqrtf(v), c.inv_sqrt_bc2, c.eps);
    p = __builtin_fmaf(-c.step_size, m / denom, p);
}

}  // namespace slsmi
