// Interface of the shared matrix-core SGEMM (csrc/sgemm.hip) and the small deterministic reductions next to it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rulgnn.h"

namespace rulgnn {

typedef float f32x4t __attribute__((ext_vector_type(4)));

// process-wide arithmetic of the big-tile GEMM: 1 = bf16 x 3 (default), 0 = fp32 matrix instructions (bit-compatible with the 64x64
// kernel); defined in rulgnn_api.hip (rulgnn_sgemm_mode)
int& sgemm_big_mode();

// C[m][n] (+)= sum_k A(m,k) * B(n,k) with element strides (sAm, sAk), (sBn, sBk); picks the tile / tall-and-skinny / big-tile kernel.
// `bf16` != 0: operands rounded to bf16 on the matrix cores where the shape qualifies, fp32 paths otherwise.
int sgemm(const float* A, int64_t sAm, int64_t sAk, const float* B, int64_t sBn, int64_t sBk, float* C, int64_t ldc,
          int M, int N, int K, bool accumulate, hipStream_t st, int bf16 = 0, const float* amax_a = nullptr, int amax_na = 0,
          const float* amax_b = nullptr, int amax_nb = 0, void* plane_ws = nullptr, size_t plane_ws_bytes = 0);
// (plane_ws: optional scratch of sgemm_planes_ws_bytes(M, N, K) bytes.  With it -- and with the maxima -- products whose shape
// fits run on PRE-SPLIT operands: a split pass writes each operand once as two k-contiguous f16 planes, the product kernel copies
// them HBM -> LDS by DMA (csrc/sgemm_planes.hip); everything else stays on the kernels of csrc/sgemm.hip.)
size_t sgemm_planes_ws_bytes(int M, int N, int K);
int sgemm_planes_slices(int M, int N, int K, bool want_split);       // 0: this shape does not run on the pre-split kernel
bool sgemm_planes_ok(const float* A, int64_t sAm, int64_t sAk, const float* B, int64_t sBn, int64_t sBk, int M, int N, int K, int slices);
int sgemm_planes(const float* A, int64_t sAm, int64_t sAk, const float* B, int64_t sBn, int64_t sBk, float* C, int64_t ldc, int M, int N, int K,
                 bool accumulate, int slices, const float* amax_a, int amax_na, const float* amax_b, int amax_nb, void* ws, size_t ws_bytes,
                 hipStream_t st, bool a_presplit = false);
// operand A split by the kernel that PRODUCES it (no split pass, no fp32 copy): where it writes inside `ws` -- f16 planes [M][K], k
// contiguous, hi = the top 11 significant bits of a s, lo = f16(a s - hi), s = *scale a power of two with max |a s| < 2^15
void sgemm_planes_a_slots(void* ws, int M, int K, void** hi, void** lo, float** scale);
// (amax_a / amax_b, both or neither: amax_na / amax_nb floats each whose maximum is max |A| / max |B| over the FINITE elements -- one partial
// maximum per workgroup of the kernels that produced the operands.  With them, outputs that fill the 256 x 256 tiles run the two-plane f16
// split (three matrix instructions per product instead of six; csrc/sgemm.hip: sgemm_f16x2v_kernel).)

// nparts partial maxima of |x| over the finite elements of x[0..n) (one per workgroup): an operand's scale row for the arguments above
int absmax_partials(const float* x, int64_t n, float* part, int nparts, hipStream_t st);

// The same product as a deterministic split-K reduction (weight gradients: K = all rows of the batch); `partial` is caller-provided
// scratch of sgemm_splitk_need_floats(M, N, K) floats.
int sgemm_splitk(const float* A, int64_t sAm, int64_t sAk, const float* B, int64_t sBn, int64_t sBk, float* C, int64_t ldc,
                 int M, int N, int K, bool accumulate, float* partial, hipStream_t st, const float* amax_a = nullptr, int amax_na = 0,
                 const float* amax_b = nullptr, int amax_nb = 0, void* plane_ws = nullptr, size_t plane_ws_bytes = 0);

// ... and colsum[m] = sum_k A(m, k) from the same pass (weight gradient + the bias gradient over the same rows); `ones`: K ones for
// the shapes that take two calls; `partial`: sgemm_splitk_need_floats(M, N + 1, K) floats
int sgemm_splitk_colsum(const float* A, int64_t sAm, int64_t sAk, const float* B, int64_t sBn, int64_t sBk, float* C, int64_t ldc,
                        int M, int N, int K, float* colsum, const float* ones, float* partial, hipStream_t st);

// ... for several (A, B) pairs at once (the same results; two launches for all of them where every pair is a long-k product of one tile
// shape); `partial`: sgemm_splitk_colsum_batch_floats(jobs, n) floats
struct SplitKColsumJob {
    const float* A; int64_t sAm, sAk;
    const float* B; int64_t sBn, sBk;
    float* C; int64_t ldc;
    int M, N, K;
    float* colsum;
};
size_t sgemm_splitk_colsum_batch_floats(const SplitKColsumJob* jobs, int n);
int sgemm_splitk_colsum_batch(const SplitKColsumJob* jobs, int n, const float* ones, float* partial, size_t partial_floats, hipStream_t st);

// Up to ten products C_j = A_j B_j^T (element strides as in sgemm) as one split-K launch of the 64 x 64 tile kernel + one reduction launch:
// the parameter gradients of a small model, each of which is a 5-8 us launch pair at its latency floor.  Deterministic (fixed slices,
// fixed-order sums); `partial`: sgemm_splitk_batch_floats(jobs, n) floats.
struct SplitKJob {
    const float* A; int64_t sAm, sAk;
    const float* B; int64_t sBn, sBk;
    float* C; int64_t ldc;
    int M, N, K;
};
size_t sgemm_splitk_batch_floats(const SplitKJob* jobs, int n);
int sgemm_splitk_batch(const SplitKJob* jobs, int n, float* partial, size_t partial_floats, hipStream_t st);
// ... the product launch alone; *reduce (reduce_device.hpp: ReduceBatch, reduce->first[n] workgroups of 1024 threads) describes the slice sums
// for a kernel of the caller that runs reduce_slices_batch_body beside its own last reductions
struct ReduceBatch;
int sgemm_splitk_batch_products(const SplitKJob* jobs, int n, float* partial, size_t partial_floats, hipStream_t st, ReduceBatch* reduce);

// out[e] = sum_r part[r * ld + e] over `rows` partial rows (fixed order); out[0] = sum(v[0..n)) with one workgroup (fixed order)
int rows_sum(const float* part, int rows, int64_t ld, int n, float* out, hipStream_t st);
int rows_sum2(const float* partA, float* outA, const float* partB, float* outB, int rows, int64_t ld, int n, hipStream_t st);
// ... plus a third sum of another shape whose result is written twice (outC and, if given, outC2)
int rows_sum3(const float* partA, float* outA, const float* partB, float* outB, int rows, int64_t ld, int n, const float* partC, float* outC,
              float* outC2, int rowsC, int64_t ldC, int nC, hipStream_t st);
// three sums of different shapes in one launch
int rows_sum_three(const float* const part[3], float* const out[3], const int rows[3], const int64_t ld[3], const int n[3], hipStream_t st);
// dst[c] = sum_r src[r][c] of a short row-major matrix with ONE workgroup.  A thread walks rows * C / 1024 elements alone: beyond
// ~64 of them the split-K pair wins again (10 240 rows x 50 columns: > 100 us)
inline bool cols_sum_small_ok(int64_t rows, int C) { return C >= 1 && C <= 64 && rows * C <= 65536; }
int cols_sum_small(const float* src, int rows, int C, float* dst, hipStream_t st);
int block_sum(const float* v, int64_t n, float* out, hipStream_t st);

// ---- scratch sizing (host) ----------------------------------------------------------------------------------------------------
constexpr int SKT_ROWS = 128;
static inline int sgemm_splitk_slices(int M, int N, int K) {
    // outputs that fill 128x128 tiles run on the big kernel (sgemm_big_ok): count its tiles, two workgroups per CU
    const bool big = M > 96 && N > 96;
    if (M > 192 && N > 192) {
        // outputs that fill 256 x 256 tiles (sgemm_wide_ok): one workgroup per CU (144 KB of LDS), so tiles x slices should be a whole
        // number of rounds of the 256 CUs -- [1024 x 1024] over K = 10 240 ran as 16 x 12 = 192 workgroups (a quarter of the chip idle),
        // as 16 x 16 it fills it
        const int wt = ((M + 255) / 256) * ((N + 255) / 256);
        int s = wt <= 256 ? (256 + wt / 2) / wt : 1;
        const int maxw = (K + 255) / 256;
        s = s > maxw ? maxw : s;
        if (s >= 1 && wt * s >= 160) return s;
    }
    const int t = big ? 128 : 64;
    const int tiles = ((M + t - 1) / t) * ((N + t - 1) / t);
    int s = (big ? 768 : 1024) / (tiles > 0 ? tiles : 1);          // aim at ~768 / ~1024 workgroups
    // at least 64 k per slice for the 64 x 64 tile kernel: a slice costs ~1 us per 16-k step (one tile load in flight), so a [50 x 50] output
    // over 10 240 rows took 20 us in 40 slices of 256 and takes a third of that in 160 of 64; 256 for the big tiles
    const int maxs = big ? (K + 255) / 256 : (K + 63) / 64;
    if (s > maxs) s = maxs;
    if (s > 256) s = 256;
    return s < 1 ? 1 : s;
}

// workgroups of the long-k path for this shape, 0 if it does not apply
static inline int sgemm_longk_blocks(int M, int N, int K) {
    if (!(M * N <= 1024 && K >= 512 && ((size_t)SKT_ROWS * (M + N) + 256) * sizeof(float) <= 48 * 1024)) return 0;
    const int nblk = (K + SKT_ROWS - 1) / SKT_ROWS;     // one k tile per workgroup while the partial buffer (1024 rows) allows
    return nblk > 1024 ? 1024 : nblk;
}

// floats the caller must provide as `partial` for sgemm_splitk(M, N, any K)
static inline size_t sgemm_splitk_partial_floats(int M, int N) { return (size_t)(M * N <= 1024 ? 1024 : 256) * M * N; }
// ... and the exact need for one known K
static inline size_t sgemm_splitk_need_floats(int M, int N, int K) {
    const int lb = sgemm_longk_blocks(M, N, K);
    return (size_t)(lb > 0 ? lb : sgemm_splitk_slices(M, N, K)) * M * N;
}
// ... and for every output of at most M x Nmax values (the two paths do not grow monotonically with the output size)
static inline size_t sgemm_splitk_bound_floats(int M, int Nmax, int K) {
    size_t v = sgemm_splitk_need_floats(M, Nmax, K);
    const size_t small = (size_t)sgemm_longk_blocks(1, 1, K) * (M * Nmax < 1024 ? M * Nmax : 1024);
    return small > v ? small : v;
}

}  // namespace rulgnn
