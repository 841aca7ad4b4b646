// "Tiled" ST_GCN path for num_patch > 64 (PHM2012 Condition_2: 160 patches, XJTU-SY: 1024 / 2048).
//
// At these sizes theta and fc1 are real [N, N] matrices (XJTU: 1M parameters each) and
// theta(A.X) = [batch*10, N] x [N, N] is a dense contraction (353 FLOP/B, MFMA-bound), so
// the layer is no longer fused into one register-resident kernel: activations live in HBM as
// [batch][10][N] tensors (the reference's own layout), the contraction is an LDS-tiled
// v_mfma_f32_16x16x4_f32 GEMM, and everything else is position-parallel: one thread per (sample, patch)
// holding the ten channels in registers; the causal taps t-1 / t-2 are read from the neighbour's
// column in memory (coalesced along t).  Reference: models/ST_GCN/Model.py:7-222.
#include "stgcn_device.hpp"
#include "stgcn_host.hpp"

namespace rulgnn {

typedef float f32x4t __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// SGEMM on the matrix cores: C[m][n] (+)= sum_k A(m,k) * B(n,k), generic strides, fp32 MFMA 16x16x4.
// 64x64 block tile, K step 16, 4 wavefronts each owning a 32x32 quadrant (2x2 MFMA tiles).
// ------------------------------------------------------------------------------------------------
struct GemmArgs {
    const float* A; int64_t sAm, sAk;
    const float* B; int64_t sBn, sBk;
    float* C; int64_t ldc;
    int M, N, K;
    int accumulate;      // C += instead of C =
};

__global__ __launch_bounds__(256) void sgemm_mfma_kernel(GemmArgs g) {
    __shared__ float As[16][64 + 4];      // [k][m]
    __shared__ float Bs[16][64 + 4];      // [k][n]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    f32x4t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4t){0.f, 0.f, 0.f, 0.f};
    const int li = lane & 15, kq = lane >> 4;
    for (int k0 = 0; k0 < g.K; k0 += 16) {
        // cooperative load: 64 x 16 elements of A and of B (4 + 4 per thread)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + e * 256, mm = idx & 63, kk = idx >> 6;
            const int gm = m0 + mm, gn = n0 + mm, gk = k0 + kk;
            As[kk][mm] = (gm < g.M && gk < g.K) ? g.A[gm * g.sAm + gk * g.sAk] : 0.f;
            Bs[kk][mm] = (gn < g.N && gk < g.K) ? g.B[gn * g.sBn + gk * g.sBk] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = As[4 * ks + kq][wm + 16 * i + li];
                b[i] = Bs[4 * ks + kq][wn + 16 * i + li];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // D layout: lane (g4 = lane>>4, col = lane&15), reg r -> row 4*g4 + r
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gm = m0 + wm + 16 * i + 4 * kq + r, gn = n0 + wn + 16 * j + li;
                if (gm < g.M && gn < g.N) {
                    float* c = g.C + (int64_t)gm * g.ldc + gn;
                    *c = g.accumulate ? *c + acc[i][j][r] : acc[i][j][r];
                }
            }
}

static int sgemm(const float* A, int64_t sAm, int64_t sAk, const float* B, int64_t sBn, int64_t sBk, float* C, int64_t ldc,
                 int M, int N, int K, bool accumulate, hipStream_t st) {
    if (M <= 0 || N <= 0) return RULGNN_OK;
    GemmArgs g{A, sAm, sAk, B, sBn, sBk, C, ldc, M, N, K, accumulate ? 1 : 0};
    (void)hipGetLastError();
    hipLaunchKernelGGL(sgemm_mfma_kernel, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, st, g);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

// ------------------------------------------------------------------------------------------------
// position-parallel kernels: thread = (sample b, patch t); tensors are [B][10][N]
// ------------------------------------------------------------------------------------------------
struct TArgs {
    int64_t B;
    int N, P, L;
};

// patch statistics -- Model.py:7-52.  x: [B][N][P]  ->  X0: [B][10][N]
__global__ void t_stats_kernel(const float* __restrict__ x, float* __restrict__ X0, TArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.B * a.N) return;
    const int64_t b = i / a.N;
    const int t = (int)(i % a.N);
    float st[F];
    patch_statistics(x + i * a.P, a.P, st);
#pragma unroll
    for (int c = 0; c < F; ++c) X0[(b * F + c) * a.N + t] = st[c];
}

// Pearson adjacency -- Model.py:53-71.  One block per sample.  A: [B][10][10]
__global__ __launch_bounds__(256) void t_gram_kernel(const float* __restrict__ X0, float* __restrict__ A, TArgs a) {
    __shared__ float red[NPAIR][4];
    __shared__ float mean[F];
    __shared__ float dots[NPAIR];
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xb = X0 + b * F * a.N;
    float s[F];
#pragma unroll
    for (int c = 0; c < F; ++c) s[c] = 0.f;
    for (int t = tid; t < a.N; t += 256)
#pragma unroll
        for (int c = 0; c < F; ++c) s[c] += xb[c * a.N + t];
#pragma unroll
    for (int c = 0; c < F; ++c) {
        float v = s[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[c][wave] = v;
    }
    __syncthreads();
    if (tid < F) mean[tid] = (red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3]) / (float)a.N;
    __syncthreads();
    float d[NPAIR];
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) d[i] = 0.f;
    for (int t = tid; t < a.N; t += 256) {
        float cx[F];
#pragma unroll
        for (int c = 0; c < F; ++c) cx[c] = xb[c * a.N + t] - mean[c];
#pragma unroll
        for (int p = 0; p < F; ++p)
#pragma unroll
            for (int q = p; q < F; ++q) d[sym(p, q)] = fmaf(cx[p], cx[q], d[sym(p, q)]);
    }
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) {
        float v = d[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[i][wave] = v;
    }
    __syncthreads();
    if (tid < NPAIR) dots[tid] = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
    __syncthreads();
    if (tid < F * F) {
        const int p = tid / F, q = tid % F;
        A[b * F * F + tid] = dots[sym(p, q)] / (sqrtf(dots[sym(p, p)]) * sqrtf(dots[sym(q, q)]));   // 0/0 -> NaN as the reference
    }
}

// out[b][c][t] = sum_c' A[b][c][c'] in[b][c'][t] (+ add[b][c][t])          (torch.bmm(A, X), Model.py:87)
__global__ void t_aggregate_kernel(const float* __restrict__ A, const float* __restrict__ in, const float* __restrict__ add,
                                   float* __restrict__ out, TArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.B * a.N) return;
    const int64_t b = i / a.N;
    const int t = (int)(i % a.N);
    const float* Ab = A + b * F * F;
    float x[F];
#pragma unroll
    for (int c = 0; c < F; ++c) x[c] = in[(b * F + c) * a.N + t];
#pragma unroll
    for (int c = 0; c < F; ++c) {
        float acc = add ? add[(b * F + c) * a.N + t] : 0.f;
#pragma unroll
        for (int q = 0; q < F; ++q) acc = fmaf(Ab[c * F + q], x[q], acc);
        out[(b * F + c) * a.N + t] = acc;
    }
}

// eval-mode TCN block of one layer after the theta GEMM (BatchNorm folded), Model.py:134-170,187-195:
//   H = leaky(Hpre + bias[t]);  o0 = relu(relu(bn1(conv1(H))) + H);  o1 = relu(relu(bn2(conv2(o0))) + o0);  Xn = o1 + X
// The causal taps need H at t-1 and o0 at t-2, t-3: recomputed from Hpre of the neighbours (cheap, no halo exchange).
__device__ __forceinline__ void t_load_H(const float* __restrict__ Hpre, const float* __restrict__ tb, int64_t b, int t, int N,
                                         float (&H)[F]) {
    if (t < 0) {
#pragma unroll
        for (int c = 0; c < F; ++c) H[c] = 0.f;
        return;
    }
    const float bias = tb[t];
#pragma unroll
    for (int c = 0; c < F; ++c) H[c] = leaky(Hpre[(b * F + c) * N + t] + bias);
}

__device__ __forceinline__ void t_conv_point(const float (&h0)[F], const float (&h1)[F], const float* __restrict__ w, float (&z)[F]) {
    // z[co] = sum_ci w[co][ci][0] h0[ci] + w[co][ci][1] h1[ci]   (h0 = tap at t - d, h1 = tap at t)
#pragma unroll
    for (int co = 0; co < F; ++co) {
        float acc = 0.f;
#pragma unroll
        for (int ci = 0; ci < F; ++ci) {
            acc = fmaf(w[(co * F + ci) * 2 + 0], h0[ci], acc);
            acc = fmaf(w[(co * F + ci) * 2 + 1], h1[ci], acc);
        }
        z[co] = acc;
    }
}

__device__ __forceinline__ void t_o0_at(const float* __restrict__ Hpre, const float* __restrict__ tb, const float* __restrict__ w1,
                                        const float* __restrict__ sc1, const float* __restrict__ sh1, int64_t b, int t, int N,
                                        float (&o0)[F]) {
    if (t < 0) {
#pragma unroll
        for (int c = 0; c < F; ++c) o0[c] = 0.f;
        return;
    }
    float H[F], Hm[F], z[F];
    t_load_H(Hpre, tb, b, t, N, H);
    t_load_H(Hpre, tb, b, t - 1, N, Hm);
    t_conv_point(Hm, H, w1, z);
#pragma unroll
    for (int c = 0; c < F; ++c) o0[c] = relu(relu(fmaf(z[c], sc1[c], sh1[c])) + H[c]);
}

__global__ void t_tcn_eval_kernel(const float* __restrict__ Hpre, const float* __restrict__ Xin, const float* __restrict__ prm_l,
                                  const float* __restrict__ bnf, float* __restrict__ Xout, TArgs a) {
    // prm_l: this layer's parameters (flat layout); bnf: [2][2][F] folded scale/shift of this layer
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.B * a.N) return;
    const int64_t b = i / a.N;
    const int t = (int)(i % a.N), N = a.N;
    const float* tb = prm_l + off_theta_b(N);
    const float* w1 = prm_l + off_conv_w(N, 0);
    const float* w2 = prm_l + off_conv_w(N, 1);
    float o0[F], o0m[F], z[F];
    t_o0_at(Hpre, tb, w1, bnf, bnf + F, b, t, N, o0);
    t_o0_at(Hpre, tb, w1, bnf, bnf + F, b, t - 2, N, o0m);
    t_conv_point(o0m, o0, w2, z);
#pragma unroll
    for (int c = 0; c < F; ++c) {
        const float o1 = relu(relu(fmaf(z[c], bnf[2 * F + c], bnf[3 * F + c])) + o0[c]);
        Xout[(b * F + c) * N + t] = o1 + Xin[(b * F + c) * N + t];
    }
}

__global__ void t_bnfold_kernel(const float* __restrict__ prm, const float* __restrict__ bn, float* __restrict__ bnf, int N, int L) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L * 2 * F) return;
    const int l = i / (2 * F), blk = (i / F) % 2, c = i % F, LS = layer_stride(N);
    const float mean = bn[((l * 2 + blk) * 2 + 0) * F + c], var = bn[((l * 2 + blk) * 2 + 1) * F + c];
    const float g = prm[l * LS + off_bn_g(N, blk) + c], be = prm[l * LS + off_bn_b(N, blk) + c];
    const float sc = g / sqrtf(var + BN_EPS);
    bnf[((l * 2 + blk) * 2 + 0) * F + c] = sc;
    bnf[((l * 2 + blk) * 2 + 1) * F + c] = be - mean * sc;
}

// channel max-pool (NaN-propagating), Model.py:218-219.  pooled: [B][N]
__global__ void t_pool_kernel(const float* __restrict__ X, float* __restrict__ pooled, TArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.B * a.N) return;
    const int64_t b = i / a.N;
    const int t = (int)(i % a.N);
    float m = X[(b * F) * a.N + t];
#pragma unroll
    for (int c = 1; c < F; ++c) {
        const float v = X[(b * F + c) * a.N + t];
        m = (v > m || v != v) ? v : m;
    }
    pooled[i] = m;
}

// head after the fc1 GEMM: pred[b] = fc2.b + sum_j fc2.w[j] relu(y1pre[b][j] + fc1.b[j]).  One block per sample.
__global__ __launch_bounds__(256) void t_head_kernel(const float* __restrict__ y1pre, const float* __restrict__ prm,
                                                     float* __restrict__ pred, TArgs a) {
    __shared__ float red[4];
    const int64_t b = blockIdx.x;
    const int N = a.N, L = a.L, tid = threadIdx.x;
    const float* b1 = prm + off_fc1_b(N, L);
    const float* w2 = prm + off_fc2_w(N, L);
    float s = 0.f;
    for (int j = tid; j < N; j += 256) s = fmaf(relu(y1pre[b * N + j] + b1[j]), w2[j], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) pred[b] = red[0] + red[1] + red[2] + red[3] + prm[off_fc2_b(N, L)];
}

// ------------------------------------------------------------------------------------------------
// host: eval forward
// ------------------------------------------------------------------------------------------------
static inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

size_t stgcn_tiled_forward_workspace_bytes(const rulgnn_stgcn_shape* s) {
    const size_t T = (size_t)s->batch * F * s->num_patch * sizeof(float);
    // X (ping), X (pong), AX, Hpre  +  A, pooled, y1pre, bnfold
    return 4 * al256(T) + al256((size_t)s->batch * F * F * 4) + 2 * al256((size_t)s->batch * s->num_patch * 4) +
           al256((size_t)s->num_layers * 4 * F * 4);
}

#define T_LAUNCH(kern, n, ...)                                                                              \
    do {                                                                                                    \
        (void)hipGetLastError();                                                                            \
        hipLaunchKernelGGL(kern, dim3((unsigned)(((n) + 255) / 256)), dim3(256), 0, stream, __VA_ARGS__);   \
        if (hipGetLastError() != hipSuccess) return RULGNN_EHIP;                                            \
    } while (0)

int stgcn_tiled_forward_eval(const rulgnn_stgcn_shape* s, const float* x, const float* prm, const float* bn, float* pred,
                             void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (!workspace || workspace_bytes < stgcn_tiled_forward_workspace_bytes(s)) return RULGNN_EWORKSPACE;
    const int N = s->num_patch, L = s->num_layers, LS = layer_stride(N);
    const int64_t B = s->batch, BN_ = B * N;
    TArgs a{B, N, s->patch_size, L};
    char* w = static_cast<char*>(workspace);
    const size_t T = al256((size_t)B * F * N * sizeof(float));
    float* Xa = reinterpret_cast<float*>(w); w += T;
    float* Xb = reinterpret_cast<float*>(w); w += T;
    float* AX = reinterpret_cast<float*>(w); w += T;
    float* Hpre = reinterpret_cast<float*>(w); w += T;
    float* A = reinterpret_cast<float*>(w); w += al256((size_t)B * F * F * 4);
    float* pooled = reinterpret_cast<float*>(w); w += al256((size_t)B * N * 4);
    float* y1pre = reinterpret_cast<float*>(w); w += al256((size_t)B * N * 4);
    float* bnf = reinterpret_cast<float*>(w);

    T_LAUNCH(t_bnfold_kernel, L * 2 * F, prm, bn, bnf, N, L);
    T_LAUNCH(t_stats_kernel, BN_, x, Xa, a);
    (void)hipGetLastError();
    hipLaunchKernelGGL(t_gram_kernel, dim3((unsigned)B), dim3(256), 0, stream, Xa, A, a);
    if (hipGetLastError() != hipSuccess) return RULGNN_EHIP;
    float* Xin = Xa;
    float* Xout = Xb;
    for (int l = 0; l < L; ++l) {
        const float* pl = prm + l * LS;
        T_LAUNCH(t_aggregate_kernel, BN_, A, Xin, (const float*)nullptr, AX, a);
        int rc = sgemm(AX, N, 1, pl + off_theta_w(N), N, 1, Hpre, N, (int)(B * F), N, N, false, stream);   // (A.X) theta^T
        if (rc != RULGNN_OK) return rc;
        T_LAUNCH(t_tcn_eval_kernel, BN_, Hpre, Xin, pl, bnf + l * 4 * F, Xout, a);
        float* tmp = Xin; Xin = Xout; Xout = tmp;
    }
    T_LAUNCH(t_pool_kernel, BN_, Xin, pooled, a);
    int rc = sgemm(pooled, N, 1, prm + off_fc1_w(N, L), N, 1, y1pre, N, (int)B, N, N, false, stream);          // fc1
    if (rc != RULGNN_OK) return rc;
    (void)hipGetLastError();
    hipLaunchKernelGGL(t_head_kernel, dim3((unsigned)B), dim3(256), 0, stream, y1pre, prm, pred, a);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

}  // namespace rulgnn
