// SGEMM on the gfx950 matrix cores, shared by the tiled ST_GCN path and the ASTGCNN path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rulgnn.h"

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
    int kchunk;          // split-K: blockIdx.z owns k in [z*kchunk, (z+1)*kchunk) and writes slice z of C (stride M*ldc)
};

static __global__ __launch_bounds__(256) void sgemm_mfma_kernel(GemmArgs g) {
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
    const int kbeg = blockIdx.z * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    g.C += (int64_t)blockIdx.z * g.M * g.ldc;
    // Cooperative tile load: 64 x 16 elements of A and of B (4 + 4 per thread).  The lane -> element mapping follows the
    // operand's contiguous dimension (k-fastest when the k stride is 1, else m-fastest) so that wavefront loads coalesce, and
    // the next k-step's elements are fetched into registers while the matrix cores work on the current one.
    const bool a_kfast = g.sAk == 1, b_kfast = g.sBk == 1;
    int am[4], ak[4], bm[4], bk[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int idx = tid + e * 256;
        am[e] = a_kfast ? idx >> 4 : idx & 63;
        ak[e] = a_kfast ? idx & 15 : idx >> 6;
        bm[e] = b_kfast ? idx >> 4 : idx & 63;
        bk[e] = b_kfast ? idx & 15 : idx >> 6;
    }
    float ra[4], rb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int gm = m0 + am[e], gka = k0 + ak[e], gn = n0 + bm[e], gkb = k0 + bk[e];
            ra[e] = (gm < g.M && gka < kend) ? g.A[gm * g.sAm + gka * g.sAk] : 0.f;
            rb[e] = (gn < g.N && gkb < kend) ? g.B[gn * g.sBn + gkb * g.sBk] : 0.f;
        }
    };
    if (kbeg < kend) fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            As[ak[e]][am[e]] = ra[e];
            Bs[bk[e]][bm[e]] = rb[e];
        }
        __syncthreads();
        if (k0 + 16 < kend) fetch(k0 + 16);
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
    GemmArgs g{A, sAm, sAk, B, sBn, sBk, C, ldc, M, N, K, accumulate ? 1 : 0, K > 0 ? K : 1};
    (void)hipGetLastError();
    hipLaunchKernelGGL(sgemm_mfma_kernel, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, st, g);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

// Split-K for reductions over a long K (weight gradients: K = batch * nodes) with few output tiles: `slices` partial
// products into `partial` ([slices][M][N], caller-provided), then a fixed-order sum -- deterministic, no atomics.
static __global__ __launch_bounds__(256) void sgemm_reduce_slices_kernel(const float* __restrict__ partial, float* __restrict__ C,
                                                                         int64_t ldc, int M, int N, int slices, int accumulate) {
    // 64 outputs per workgroup, four threads per output each summing every fourth slice, combined in a fixed order
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    float a = 0.f;
    if (e < M * N)
        for (int z = q; z < slices; z += 4) a += partial[(int64_t)z * M * N + e];
    part[q][lane] = a;
    __syncthreads();
    if (q == 0 && e < M * N) {
        const float v = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        float* c = C + (int64_t)(e / N) * ldc + (e % N);
        *c = accumulate ? *c + v : v;
    }
}

static inline int sgemm_splitk_slices(int M, int N, int K) {
    const int tiles = ((M + 63) / 64) * ((N + 63) / 64);
    int s = 1024 / (tiles > 0 ? tiles : 1);          // aim at ~1024 workgroups
    const int maxs = (K + 255) / 256;                // at least 256 k per slice
    if (s > maxs) s = maxs;
    if (s > 256) s = 256;
    return s < 1 ? 1 : s;
}

static int sgemm_splitk(const float* A, int64_t sAm, int64_t sAk, const float* B, int64_t sBn, int64_t sBk, float* C, int64_t ldc,
                        int M, int N, int K, bool accumulate, float* partial, hipStream_t st) {
    if (M <= 0 || N <= 0) return RULGNN_OK;
    const int slices = sgemm_splitk_slices(M, N, K);
    if (slices <= 1) return sgemm(A, sAm, sAk, B, sBn, sBk, C, ldc, M, N, K, accumulate, st);
    int kchunk = (K + slices - 1) / slices;
    kchunk = (kchunk + 15) & ~15;
    const int used = (K + kchunk - 1) / kchunk;
    GemmArgs g{A, sAm, sAk, B, sBn, sBk, partial, N, M, N, K, 0, kchunk};
    (void)hipGetLastError();
    hipLaunchKernelGGL(sgemm_mfma_kernel, dim3((N + 63) / 64, (M + 63) / 64, used), dim3(256), 0, st, g);
    hipLaunchKernelGGL(sgemm_reduce_slices_kernel, dim3((M * N + 63) / 64), dim3(256), 0, st, partial, C, ldc, M, N, used,
                       accumulate ? 1 : 0);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

// out[0] = sum(v[0..n)) with one workgroup: strided partial sums, then a fixed-order tree (deterministic).
static __global__ __launch_bounds__(1024) void block_sum_kernel(const float* __restrict__ v, int64_t n, float* __restrict__ out) {
    __shared__ float red[1024];
    float a = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) a += v[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int m = 512; m > 0; m >>= 1) {
        if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

// floats the caller must provide as `partial` for sgemm_splitk(M, N, any K)
static inline size_t sgemm_splitk_partial_floats(int M, int N) { return (size_t)256 * M * N; }

}  // namespace rulgnn
