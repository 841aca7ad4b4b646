// torch.optim.Adam's element update as the reference configures it (algorithms/algorithms.py:474-478: L2 weight decay folded into the
// gradient, bias-corrected, no amsgrad) -- device side, shared by the optimizer kernels (optim.hip) and by the kernels that apply it to a
// gradient element the moment they have reduced it (astgcnn.hip: ast_tail_kernel).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rulgnn {

__device__ __forceinline__ void adam_value(float& pi, float g, float& mi, float& vi, float lr_over_bc1, float inv_sqrt_bc2, float beta1,
                                           float beta2, float eps, float wd, float gscale) {
    const float gi = fmaf(wd, pi, g * gscale);
    mi = fmaf(beta1, mi, (1.f - beta1) * gi);
    vi = fmaf(beta2, vi, (1.f - beta2) * gi * gi);
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    pi = pi - lr_over_bc1 * (mi / denom);
}

// Adam applied where a gradient element is produced: every element of a small model's gradient is finalised by exactly one thread of the
// step's last reduction kernel, which then owns that parameter's update too -- no optimizer launch.  p == nullptr: none.
// (gbase: the gradient buffer the destinations point into; the parameter / moment of a destination sits at the same offset.)
struct AdamFuse {
    float* p;
    float* m;
    float* v;
    const float* gbase;
    float lr_over_bc1, inv_sqrt_bc2, beta1, beta2, eps, wd;
    __device__ __forceinline__ void operator()(float* dst, float g) const {
        if (!p) return;
        const int64_t i = dst - gbase;
        float pi = p[i], mi = m[i], vi = v[i];
        adam_value(pi, g, mi, vi, lr_over_bc1, inv_sqrt_bc2, beta1, beta2, eps, wd, 1.0f);
        m[i] = mi;
        v[i] = vi;
        p[i] = pi;
    }
};
struct NoEpilogue {
    __device__ __forceinline__ void operator()(float*, float) const {}
};

}  // namespace rulgnn
