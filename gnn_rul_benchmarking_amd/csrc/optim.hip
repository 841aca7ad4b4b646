// Optimizer-side kernels over the flat parameter buffer.
//   adam_step:          torch.optim.Adam as the reference configures it (algorithms/algorithms.py:474-478):
//                       L2 weight decay folded into the gradient, bias-corrected, no amsgrad.
//   bn_running_update:  nn.BatchNorm1d running statistics (momentum 0.1, unbiased running variance).
#include "stgcn_host.hpp"
#include "adam_device.hpp"

namespace rulgnn {

__device__ __forceinline__ void adam_element(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                             float* __restrict__ v, int64_t i, float lr_over_bc1, float inv_sqrt_bc2, float beta1,
                                             float beta2, float eps, float wd, float gscale) {
    float pi = p[i], mi = m[i], vi = v[i];
    adam_value(pi, g[i], mi, vi, lr_over_bc1, inv_sqrt_bc2, beta1, beta2, eps, wd, gscale);
    m[i] = mi;
    v[i] = vi;
    p[i] = pi;
}

// elements [0, n) among `stride` threads, this one being `tid`: 16-byte accesses where the four buffers allow (3.1 M parameters of the
// XJTU-SY wiring: 16 us with 4-byte accesses), the same arithmetic per element either way
__device__ __forceinline__ void adam_range(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                           int64_t n, int64_t tid, int64_t stride, float lr_over_bc1, float inv_sqrt_bc2, float beta1,
                                           float beta2, float eps, float wd, float gscale) {
    int64_t done = 0;
    if (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0) {
        const int64_t n4 = n >> 2;
        float4* p4 = reinterpret_cast<float4*>(p);
        const float4* g4 = reinterpret_cast<const float4*>(g);
        float4* m4 = reinterpret_cast<float4*>(m);
        float4* v4 = reinterpret_cast<float4*>(v);
        for (int64_t i = tid; i < n4; i += stride) {
            float4 pp = p4[i], mm = m4[i], vv = v4[i];
            const float4 gg = g4[i];
            adam_value(pp.x, gg.x, mm.x, vv.x, lr_over_bc1, inv_sqrt_bc2, beta1, beta2, eps, wd, gscale);
            adam_value(pp.y, gg.y, mm.y, vv.y, lr_over_bc1, inv_sqrt_bc2, beta1, beta2, eps, wd, gscale);
            adam_value(pp.z, gg.z, mm.z, vv.z, lr_over_bc1, inv_sqrt_bc2, beta1, beta2, eps, wd, gscale);
            adam_value(pp.w, gg.w, mm.w, vv.w, lr_over_bc1, inv_sqrt_bc2, beta1, beta2, eps, wd, gscale);
            p4[i] = pp; m4[i] = mm; v4[i] = vv;
        }
        done = n4 << 2;
    }
    for (int64_t i = done + tid; i < n; i += stride) adam_element(p, g, m, v, i, lr_over_bc1, inv_sqrt_bc2, beta1, beta2, eps, wd, gscale);
}

__global__ void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, int64_t n, float lr_over_bc1, float inv_sqrt_bc2, float beta1,
                                 float beta2, float eps, float wd, float gscale, const StepState* __restrict__ st,
                                 const float* __restrict__ guard) {
    // guarded step (rulgnn_adam_step_guarded_f32): a non-finite *guard -- the loss behind the gradient in the bucket, NaN when the
    // matrix-core training chain raised its f16 range status (stgcn_train_mx.hip) -- leaves parameters and moments untouched
    if (guard && !(fabsf(*guard) <= 3.0e38f)) return;
    if (st) {                      // device step state: bias corrections of the step the prepare kernel just advanced to
        lr_over_bc1 = st->lr_over_bc1;
        inv_sqrt_bc2 = st->inv_sqrt_bc2;
    }
    adam_range(p, g, m, v, n, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x, lr_over_bc1, inv_sqrt_bc2, beta1,
               beta2, eps, wd, gscale);
}

__device__ __forceinline__ void bn_running_entry(float* __restrict__ bn, const float* __restrict__ batch, int i, float momentum,
                                                 float unbias, int from_moments) {
    // layout [n_bn][2 (mean, var)][F]
    const bool is_var = (i / F) % 2 == 1;
    float b = batch[i];
    if (is_var) {
        if (from_moments) {
            const float m = batch[i - F];
            b = fmaxf(b - m * m, 0.f);
        }
        b *= unbias;
    }
    bn[i] = (1.f - momentum) * bn[i] + momentum * b;
}

__global__ void bn_running_update_kernel(float* __restrict__ bn, const float* __restrict__ batch, int n_bn, float momentum,
                                         float unbias, int from_moments, const float* __restrict__ guard) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_bn * 2 * F) return;
    if (guard && !(fabsf(*guard) <= 3.0e38f)) return;
    bn_running_entry(bn, batch, i, momentum, unbias, from_moments);
}

// The optimizer step and the running-statistics update behind a data-parallel bucket all-reduce in ONE launch (rulgnn_adam_bn_step_f32):
// the last workgroup carries the BatchNorm entries, the others the Adam elements -- each the arithmetic of its own kernel above.
__global__ void adam_bn_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                    int64_t n, float lr_over_bc1, float inv_sqrt_bc2, float beta1, float beta2, float eps, float wd,
                                    float gscale, float* __restrict__ bn, const float* __restrict__ batch, int n_bn, float momentum,
                                    float unbias, int from_moments, const float* __restrict__ guard) {
    if (guard && !(fabsf(*guard) <= 3.0e38f)) return;
    if (blockIdx.x == gridDim.x - 1) {
        for (int i = threadIdx.x; i < n_bn * 2 * F; i += blockDim.x) bn_running_entry(bn, batch, i, momentum, unbias, from_moments);
        return;
    }
    adam_range(p, g, m, v, n, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)(gridDim.x - 1) * blockDim.x, lr_over_bc1, inv_sqrt_bc2,
               beta1, beta2, eps, wd, gscale);
}

__global__ void step_state_set_kernel(StepState* s, uint64_t dropout_step, int64_t adam_step) {
    s->dropout_step = dropout_step;
    s->adam_step = adam_step;
    s->lr_over_bc1 = s->inv_sqrt_bc2 = 0.f;
    for (int l = 0; l < 8; ++l) s->drop_key[l] = 0u;
    s->pad[0] = s->pad[1] = 0u;
}

__global__ void step_prepare_dropout_kernel(StepState* s, uint64_t seed, int num_layers) {
    const uint64_t step = s->dropout_step + 1;
    s->dropout_step = step;
    for (int l = 0; l < 8; ++l) s->drop_key[l] = l < num_layers ? dropout_layer_key(seed, step, l) : 0u;
}

__global__ void step_prepare_adam_kernel(StepState* s, float lr, float beta1, float beta2) {
    const int64_t step = s->adam_step + 1;
    s->adam_step = step;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    s->lr_over_bc1 = (float)((double)lr / bc1);
    s->inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
}

int step_state_set(void* state, uint64_t dropout_step, int64_t adam_step, hipStream_t stream) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(step_state_set_kernel, dim3(1), dim3(1), 0, stream, static_cast<StepState*>(state), dropout_step, adam_step);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

int step_prepare_dropout(void* state, uint64_t seed, int num_layers, hipStream_t stream) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(step_prepare_dropout_kernel, dim3(1), dim3(1), 0, stream, static_cast<StepState*>(state), seed, num_layers);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

int step_prepare_adam(void* state, float lr, float beta1, float beta2, hipStream_t stream) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(step_prepare_adam_kernel, dim3(1), dim3(1), 0, stream, static_cast<StepState*>(state), lr, beta1, beta2);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

int adam_step(float* p, const float* g, float* m, float* v, int64_t n, int64_t step, float lr, float beta1, float beta2,
              float eps, float wd, float gscale, hipStream_t stream, void* step_state, const float* guard) {
    if (n <= 0) return RULGNN_OK;
    if (step_state) {
        const int rc = step_prepare_adam(step_state, lr, beta1, beta2, stream);
        if (rc != RULGNN_OK) return rc;
        step = 1;                  // placeholder: the kernel takes the corrections from the state
    }
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const int block = 256;
    int64_t grid = (n + block - 1) / block;
    if (grid > 1024) grid = 1024;
    (void)hipGetLastError();   // drop any stale error of the caller's earlier HIP calls
    hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)grid), dim3(block), 0, stream, p, g, m, v, n,
                       (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)), beta1, beta2, eps, wd, gscale,
                       static_cast<const StepState*>(step_state), guard);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

int bn_running_update(float* bn, const float* batch, int num_layers, int64_t count, float momentum, int from_moments,
                      hipStream_t stream, const float* guard) {
    const int n_bn = num_layers * 2;
    const float unbias = count > 1 ? (float)((double)count / (double)(count - 1)) : 1.f;
    const int total = n_bn * 2 * F;
    (void)hipGetLastError();   // drop any stale error of the caller's earlier HIP calls
    hipLaunchKernelGGL(bn_running_update_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, bn, batch, n_bn,
                       momentum, unbias, from_moments, guard);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

// the constants adam_step passes to its kernel, for a step kernel that applies the update itself (adam_device.hpp: AdamFuse)
void adam_fuse_args(AdamFuse* t, float* p, float* m, float* v, const float* gbase, int64_t step, float lr, float beta1, float beta2, float eps,
                    float wd) {
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    t->p = p; t->m = m; t->v = v; t->gbase = gbase;
    t->lr_over_bc1 = (float)((double)lr / bc1);
    t->inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    t->beta1 = beta1; t->beta2 = beta2; t->eps = eps; t->wd = wd;
}

int adam_bn_step(float* p, const float* g, float* m, float* v, int64_t n, int64_t step, float lr, float beta1, float beta2, float eps,
                 float wd, float gscale, float* bn, const float* batch, int num_layers, int64_t count, float momentum, int from_moments,
                 const float* guard, hipStream_t stream) {
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float unbias = count > 1 ? (float)((double)count / (double)(count - 1)) : 1.f;
    const int block = 256;
    int64_t grid = (n + block - 1) / block;
    if (grid > 1024) grid = 1024;
    if (grid < 1) grid = 1;
    (void)hipGetLastError();   // drop any stale error of the caller's earlier HIP calls
    hipLaunchKernelGGL(adam_bn_step_kernel, dim3((unsigned)grid + 1), dim3(block), 0, stream, p, g, m, v, n, (float)((double)lr / bc1),
                       (float)(1.0 / sqrt(bc2)), beta1, beta2, eps, wd, gscale, bn, batch, num_layers * 2, momentum, unbias,
                       from_moments, guard);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

}  // namespace rulgnn
