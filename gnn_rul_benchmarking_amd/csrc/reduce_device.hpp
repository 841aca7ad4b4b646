// Device bodies of the fixed-order reductions behind the split-K products (csrc/sgemm.hip) -- shared with the family kernels that run them
// inside their own last launch (csrc/astgcnn.hip: ast_tail_kernel).  1024 threads per workgroup; every thread of it must call (a barrier).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rulgnn {

constexpr int GEMM_BATCH_MAX = 10;

// up to three sums out[e] = sum_r part[r][e] over per-workgroup partial rows (a job may write its result twice: out and out2)
struct RowsSumJobs {
    const float* part[3];
    float* out[3];
    float* out2[3];
    int rows[3], n[3];
    int64_t ld[3];
};
// columns [32 bx, 32 bx + 32) of job `job`: 32 columns x 32 row slices, four loads in flight per thread, the slices combined through LDS
// in a fixed order (deterministic)
// (epi(dst, value): called by the thread that stored `value` at `dst` -- e.g. the optimizer update of that element, adam_device.hpp)
template <class Epi>
__device__ __forceinline__ void rows_sum_job_body(const RowsSumJobs& jb, int job, int bx, float (&red)[32][33], const Epi& epi) {
    const float* part = jb.part[job];
    const int rows = jb.rows[job], n = jb.n[job];
    const int64_t ld = jb.ld[job];
    const int lane = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int e = bx * 32 + lane;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (e < n) {
        const float* p = part + e;
        int r = sl;
        for (; r + 96 < rows; r += 128) {
            a0 += p[(int64_t)r * ld];
            a1 += p[(int64_t)(r + 32) * ld];
            a2 += p[(int64_t)(r + 64) * ld];
            a3 += p[(int64_t)(r + 96) * ld];
        }
        for (; r < rows; r += 32) a0 += p[(int64_t)r * ld];
    }
    red[sl][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (sl == 0 && e < n) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 32; ++q) v += red[q][lane];
        jb.out[job][e] = v;
        epi(jb.out[job] + e, v);
        if (jb.out2[job]) {
            jb.out2[job][e] = v;
            epi(jb.out2[job] + e, v);
        }
    }
}
struct ReduceNoEpilogue {
    __device__ __forceinline__ void operator()(float*, float) const {}
};
__device__ __forceinline__ void rows_sum_job_body(const RowsSumJobs& jb, int job, int bx, float (&red)[32][33]) {
    rows_sum_job_body(jb, job, bx, red, ReduceNoEpilogue{});
}

// the slice sums of up to GEMM_BATCH_MAX split-K products (sgemm_splitk_batch): job j owns workgroups [first[j], first[j + 1])
struct ReduceBatch {
    const float* partial[GEMM_BATCH_MAX];
    float* C[GEMM_BATCH_MAX];
    int64_t ldc[GEMM_BATCH_MAX];
    int M[GEMM_BATCH_MAX], N[GEMM_BATCH_MAX], slices[GEMM_BATCH_MAX];
    int first[GEMM_BATCH_MAX + 1];
    int n;
};
// workgroup `block` of the batch: 64 outputs, sixteen threads per output each summing every sixteenth slice, combined in a fixed order
template <class Epi>
__device__ __forceinline__ void reduce_slices_batch_body(const ReduceBatch& b, int block, float (&part)[16][64], const Epi& epi) {
    int j = 0;
#pragma unroll
    for (int q = 1; q < GEMM_BATCH_MAX; ++q)
        if (q < b.n && block >= b.first[q]) j = q;
    const float* __restrict__ partial = b.partial[j];
    const int M = b.M[j], N = b.N[j], slices = b.slices[j];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int e = (block - b.first[j]) * 64 + lane;
    float a = 0.f;
    if (e < M * N)
        for (int z = q; z < slices; z += 16) a += partial[(int64_t)z * M * N + e];
    part[q][lane] = a;
    __syncthreads();
    if (q == 0 && e < M * N) {
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r += 4) v += (part[r][lane] + part[r + 1][lane]) + (part[r + 2][lane] + part[r + 3][lane]);
        float* dst = b.C[j] + (int64_t)(e / N) * b.ldc[j] + (e % N);
        *dst = v;
        epi(dst, v);
    }
}
__device__ __forceinline__ void reduce_slices_batch_body(const ReduceBatch& b, int block, float (&part)[16][64]) {
    reduce_slices_batch_body(b, block, part, ReduceNoEpilogue{});
}

}  // namespace rulgnn
