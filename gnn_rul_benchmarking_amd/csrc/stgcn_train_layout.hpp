// Reduction cells and step scratch of the ST_GCN training step, shared by the row-mapped fp32 phase chain (stgcn_train.hip) and the
// matrix-core chain (stgcn_train_mx.hip): both chains are bracketed by the same prepare / finalize kernels.
#pragma once
#include "stgcn_device.hpp"
#include "stgcn_host.hpp"

namespace rulgnn {

// Per-step scalars that change from step to step live in the workspace right behind the loss cell (8 doubles), written by
// stgcn_prepare_kernel at the head of the step -- not in the kernel arguments: that keeps the argument block small (the
// phase kernels are SGPR-bound) and lets a captured hipGraph replay with fresh values.
struct StepScratch {
    uint32_t drop_key[8];
    float lr_over_bc1, inv_sqrt_bc2;
    double bn_count;      // values per channel behind the BatchNorm cells: batch * N of this shard, or of the GLOBAL batch when the
                          // cells are all-reduced between the phases (synchronised BatchNorm, stgcn_train_fwdbwd_syncbn)
    uint32_t pad[4];      // [0] status word of the matrix-core chain (f16 range guard; bit 1: a step claimed clean cells that were not),
                          // [1] clean token: the finalize kernel of a matrix-core step left every cell and the status word zero,
                          // [2] ticket of that finalize kernel's workgroups,
                          // [3] STICKY count of the steps the range guard rejected (finalize_stats adds, no kernel ever clears it:
                          //     rulgnn_stgcn_train_guard_counter_offset; the caller zeroes it when it allocates the workspace)
};
static_assert(sizeof(StepScratch) == 64, "step scratch layout");
constexpr uint32_t WS_CLEAN_TOKEN = 0x52554C43u;

// The head-of-step scalars (dropout keys, Adam bias corrections, the BatchNorm count) for a step that SKIPS its prepare launch
// (RULGNN_TRAIN_WS_CLEAN: the previous matrix-core step's finalize kernel left the cells zero): F_0's workgroup 0 writes them -- nothing
// reads them before the next kernel -- and consumes the clean token; a missing token raises the status word, so that the step ends like
// one the range guard rejected (NaN loss, state untouched) instead of running on stale sums.
struct HeadScalars {
    StepScratch* sc;      // nullptr: the prepare kernel ran
    uint64_t seed, step;
    int L, has_adam;
    int64_t adam_step;
    float lr, beta1, beta2;
    double bn_count;
};
__device__ __forceinline__ void head_scalars(const HeadScalars& h, int tid) {
    StepScratch* sc = h.sc;
    if (tid == 0) {
        sc->bn_count = h.bn_count;
        if (sc->pad[1] != WS_CLEAN_TOKEN) {            // not clean: reject the step; the finalize kernel's ticket starts from zero, so that
            sc->pad[0] = 2u;                           // it can leave the workspace clean for the next one
            sc->pad[2] = 0u;
        }
        sc->pad[1] = 0u;
    }
    if (tid >= 64 && tid < 72) sc->drop_key[tid - 64] = tid - 64 < h.L ? dropout_layer_key(h.seed, h.step, tid - 64) : 0u;
    if (h.has_adam && tid == 128) {
        const double bc1 = 1.0 - pow((double)h.beta1, (double)h.adam_step);
        const double bc2 = 1.0 - pow((double)h.beta2, (double)h.adam_step);
        sc->lr_over_bc1 = (float)((double)h.lr / bc1);
        sc->inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    }
}

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
// the arrival counter of the single-launch step forms (one 32-bit word behind the step scratch; zero between steps: the prepare kernel
// and the finalize kernel of a matrix-core step leave it so)
__host__ __device__ __forceinline__ unsigned* step_barrier(double* cells, int L) {
    return reinterpret_cast<unsigned*>(reinterpret_cast<char*>(step_scratch(cells, L)) + sizeof(StepScratch));
}
__device__ __forceinline__ double cell_sum(const double* cells, int L, int i) {
    double v = 0.0;
#pragma unroll
    for (int r = 0; r < CELL_REPLICAS; ++r) v += cells[r * cell_stride(L) + i];
    return v;
}

// ---- BatchNorm constants of a phase (matrix-core chain) ------------------------------------------------------------------------------------
// One row block of BN_TABLE_ROWS x F floats per BatchNorm in the workgroup's LDS: mean, istd, gamma, beta, gamma istd (from the forward
// pair sum z, sum z^2) and mean(dy), mean(dy xhat) (from the backward pair).  A phase needs at most three reduction pairs; wavefront w of
// the workgroup finishes pair w: lane j < 2 F sums the 16 replicas of cell j (sixteen independent loads: ONE memory round trip; the
// same fixed order as cell_sum()), lane c < F then holds both sums of channel c and does the fp64 arithmetic.  The cells are final when the
// kernel starts (kernel boundary; under synchronised BatchNorm the all-reduce sits in between), so plain loads do.
// History (round 4): every lane summing every cell cost three dependent rounds of loads per kernel; a table finished ONCE by workgroup 0
// and published through a flag cost four round trips on the critical path of every workgroup (device-scope stores / loads: the XCDs' L2s
// are not coherent with each other); a last-arriver ticket at the end of the producing kernel needs a device-scope release fence, which
// writes the L2 back: ~40 us per kernel.
// COHERENT (the persistent small-batch launch, stgcn_train_mx.hip): the cells were completed by OTHER workgroups of this same launch
// (agent-scope atomics, then a counter) -- read them with agent-scope atomic loads (L1-bypassing `sc1` loads: atomics on both sides).
constexpr int BN_TABLE_ROWS = 7;        // mean, istd, gamma, beta, gamma istd, mean(dy), mean(dy xhat)
template <bool COHERENT = false>
__device__ __forceinline__ void bn_pair_to_lds(const double* cells, const float* prm, float* bnc, int L, int N, bool fwd, int b, int lane) {
    const int CS = cell_stride(L);
    const int base = (fwd ? cell_fwd(L) : cell_bwd(L)) + b * 2 * F;
    double v = 0.0;
    if (lane < 2 * F) {
        if constexpr (COHERENT) {
            double t[CELL_REPLICAS];
#pragma unroll
            for (int r = 0; r < CELL_REPLICAS; ++r) t[r] = __hip_atomic_load(&cells[r * CS + base + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int r = 0; r < CELL_REPLICAS; ++r) v += t[r];
        } else {
#pragma unroll
        for (int r = 0; r < CELL_REPLICAS; ++r) v += cells[r * CS + base + lane];
        }
    }
    const double s1 = v, s2 = __shfl(v, (lane + F) & 63, 64);
    if (lane < F) {
        const int c = lane;
        const double cnt = step_scratch(const_cast<double*>(cells), L)->bn_count;
        float* o = bnc + b * BN_TABLE_ROWS * F;
        if (fwd) {
            const double mean = s1 / cnt;
            double var = s2 / cnt - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            const double istd = 1.0 / sqrt(var + (double)BN_EPS);
            const int LS = layer_stride(N);
            const double g = prm[(b / 2) * LS + off_bn_g(N, b % 2) + c];
            o[0 * F + c] = (float)mean;
            o[1 * F + c] = (float)istd;
            o[2 * F + c] = (float)g;
            o[3 * F + c] = prm[(b / 2) * LS + off_bn_b(N, b % 2) + c];
            o[4 * F + c] = (float)(g * istd);
        } else {
            o[5 * F + c] = (float)(s1 / cnt);
            o[6 * F + c] = (float)(s2 / cnt);
        }
    }
}

}  // namespace rulgnn
