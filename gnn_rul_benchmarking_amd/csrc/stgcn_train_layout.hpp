// Reduction cells and step scratch of the ST_GCN training step, shared by the row-mapped fp32 phase chain (stgcn_train.hip) and the
// matrix-core chain (stgcn_train_mx.hip): both chains are bracketed by the same prepare / finalize kernels.
#pragma once
#include "stgcn_device.hpp"

namespace rulgnn {

// Per-step scalars that change from step to step live in the workspace right behind the loss cell (8 doubles), written by
// stgcn_prepare_kernel at the head of the step -- not in the kernel arguments: that keeps the argument block small (the
// phase kernels are SGPR-bound) and lets a captured hipGraph replay with fresh values.
struct StepScratch {
    uint32_t drop_key[8];
    float lr_over_bc1, inv_sqrt_bc2;
    double bn_count;      // values per channel behind the BatchNorm cells: batch * N of this shard, or of the GLOBAL batch when the
                          // cells are all-reduced between the phases (synchronised BatchNorm, stgcn_train_fwdbwd_syncbn)
    uint32_t pad[4];
};
static_assert(sizeof(StepScratch) == 64, "step scratch layout");

// Reduction cells (fp64): per BatchNorm the forward pair (sum z, sum z^2) and the backward pair (sum dy, sum dy*xhat), then
// the loss.  Every block adds its partial sums with one atomic per cell; 1280 blocks hitting the same 20 addresses serialise
// (measured: ~10 us per phase kernel), so the cells exist CELL_REPLICAS times, block b adds into replica b % CELL_REPLICAS and
// the consumers sum the replicas in a fixed order.
constexpr int CELL_REPLICAS = 16;
__host__ __device__ constexpr int cell_fwd(int L) { (void)L; return 0; }
__host__ __device__ constexpr int cell_bwd(int L) { return 2 * L * 2 * F; }
__host__ __device__ constexpr int cell_loss(int L) { return 2 * (2 * L * 2 * F); }
__host__ __device__ constexpr int cell_stride(int L) { return 2 * (2 * L * 2 * F) + 8; }
__host__ __device__ __forceinline__ StepScratch* step_scratch(double* cells, int L) {
    return reinterpret_cast<StepScratch*>(cells + CELL_REPLICAS * cell_stride(L));
}
__device__ __forceinline__ double cell_sum(const double* cells, int L, int i) {
    double v = 0.0;
#pragma unroll
    for (int r = 0; r < CELL_REPLICAS; ++r) v += cells[r * cell_stride(L) + i];
    return v;
}

// ---- BatchNorm table (matrix-core chain) -----------------------------------------------------------------------------------------------
// 2048 wavefronts each summing 16 replicas of every cell and finishing the BatchNorm constants in fp64 cost ~10 us of L2 traffic per
// phase kernel (50 MB against the same 170 lines).  In the matrix-core chain a reduction pair is finished ONCE, by workgroup 0 of the
// NEXT phase kernel (the pair is final when that kernel starts -- also under synchronised BatchNorm, where the all-reduce sits between the
// two kernels): it sums the pair's replicas, writes the constants into a table of 7 x F floats per BatchNorm behind the step scratch and
// then publishes the kernel's sequence number in StepScratch::pad[1]; the other workgroups build the operands that do not depend on a
// BatchNorm meanwhile and wait for the number.  Workgroup 0 is dispatched first, so whoever waits waits for a resident wavefront.
// (A last-arriver ticket at the END of the producing kernel was tried first: its device-scope release fence writes the L2 back --
// the XCDs' L2s are not coherent with each other -- and cost ~40 us per kernel.)  Table and flag travel by device-scope atomic
// stores / loads, which bypass the per-XCD L2.
constexpr int BN_TABLE_ROWS = 7;        // mean, istd, gamma, beta, gamma istd, mean(dy), mean(dy xhat)
constexpr int BN_TABLE_BYTES = 4608;    // 2 * 8 layers * 7 * F floats, rounded up
__host__ __device__ __forceinline__ float* bn_table(double* cells, int L) {
    return reinterpret_cast<float*>(reinterpret_cast<char*>(step_scratch(cells, L)) + sizeof(StepScratch) + 64);
}
__device__ __forceinline__ void table_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float table_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// One wavefront finishes BatchNorm b's forward (rows 0..4 from sum z, sum z^2) or backward (rows 5, 6) constants: lane j < 2 F sums the
// replicas of cell j of the pair (same fixed order as cell_sum()), lane c < F then holds both sums of channel c.
__device__ __forceinline__ void bn_pair_finish(double* cells, const float* prm, int L, int N, bool fwd, int b, int lane) {
    const int CS = cell_stride(L);
    const int base = (fwd ? cell_fwd(L) : cell_bwd(L)) + b * 2 * F;
    double v = 0.0;
    if (lane < 2 * F) {
#pragma unroll
        for (int r = 0; r < CELL_REPLICAS; ++r) v += __hip_atomic_load(cells + r * CS + base + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const double s1 = v, s2 = __shfl(v, (lane + F) & 63, 64);
    if (lane < F) {
        const int c = lane;
        const double cnt = step_scratch(cells, L)->bn_count;
        float* o = bn_table(cells, L) + b * BN_TABLE_ROWS * F;
        if (fwd) {
            const double mean = s1 / cnt;
            double var = s2 / cnt - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            const double istd = 1.0 / sqrt(var + (double)BN_EPS);
            const int LS = layer_stride(N);
            const double g = prm[(b / 2) * LS + off_bn_g(N, b % 2) + c];
            table_store(o + 0 * F + c, (float)mean);
            table_store(o + 1 * F + c, (float)istd);
            table_store(o + 2 * F + c, (float)g);
            table_store(o + 3 * F + c, prm[(b / 2) * LS + off_bn_b(N, b % 2) + c]);
            table_store(o + 4 * F + c, (float)(g * istd));
        } else {
            table_store(o + 5 * F + c, (float)(s1 / cnt));
            table_store(o + 6 * F + c, (float)(s2 / cnt));
        }
    }
}
// prologue of a phase kernel: the leader (wavefront 0 of workgroup 0) finishes the pair its predecessor completed and publishes `seq`;
// every other wavefront waits for it
__device__ __forceinline__ void bn_table_sync(double* cells, const float* prm, int L, int N, bool fwd, int b, unsigned seq, bool leader, int lane) {
    unsigned* flag = &step_scratch(cells, L)->pad[1];
    if (leader) {
        bn_pair_finish(cells, prm, L, N, fwd, b, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the table's stores have been acknowledged
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != seq) __builtin_amdgcn_s_sleep(8);
    }
    __builtin_amdgcn_wave_barrier();
}

}  // namespace rulgnn
