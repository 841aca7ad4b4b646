// Fused eval-mode ST_GCN forward for gfx950: ONE kernel from raw windows to the RUL prediction.
// Reference path replaced: ST_GCN_model.forward under model.eval() -- models/ST_GCN/Model.py:208-222
// (~90 ATen dispatches per call in the reference; here x is read once from HBM and 4 bytes per
// sample are written back).
//
// Two kernels share this entry point:
//   * stgcn_forward_mx.hip  (num_patch <= 15, 16-byte copyable tiles, 1..3 layers: the C-MAPSS shapes): channel mixing,
//     aggregation and projection on the f16 matrix cores with 2-way split operands, exact-path fallback inside the launch;
//   * the row-mapped exact-fp32 kernel below (every other num_patch <= 64, and the fallback's arithmetic).
#include "stgcn_eval_tile.hpp"
#include "stgcn_host.hpp"

namespace rulgnn {

struct FwdArgs {
    int64_t B;
    int64_t ntiles;
    int N, P, Ppad, L;
    int K;              // MPNN order
    uint32_t magicP;
    int vec4;
    int stage_floats;   // per-wave LDS staging floats
};

// NFIX / PFIX / LFIX: num_patch, patch_size, num_layers known at compile time (0 = from the arguments): the C-MAPSS shapes
// (14 sensors x 30 or 50 points, two layers) get immediate offsets, fully unrolled statistics passes and a flat layer loop.
template <int RW, int NFIX = 0, int PFIX = 0, int LFIX = 0>
__global__ __launch_bounds__(BLOCK) void stgcn_forward_eval_kernel(const float* __restrict__ gx,
                                                                   const float* __restrict__ prm,
                                                                   const float* __restrict__ bn,
                                                                   float* __restrict__ out, FwdArgs a) {
    // prm / bn are separate __restrict__ kernel arguments so that wave-uniform weight reads
    // become scalar loads (s_load_dwordx16 -> SGPR operands) instead of per-lane VMEM loads.
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int SPW = Row<RW>::SPW;
    const int N = NFIX ? NFIX : a.N, L = LFIX ? LFIX : a.L, P = PFIX ? PFIX : a.P;
    const int K = NFIX ? 1 : a.K;                 // (the compile-time shapes are the reference's wirings: order 1)
    EvalWeightsLds<RW> w;
    w.bind(smem, L, K);
    float* stage_all = smem + EvalWeightsLds<RW>::floats(L, K);
    eval_weights_fill<RW>(w, prm, bn, N, L, threadIdx.x, BLOCK, K);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int srow = lane / RW, t = lane % RW;
    float* stage = stage_all + wave * a.stage_floats;
    const int64_t sampleNP = (int64_t)N * P;

    for (int64_t tile = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave; tile < a.ntiles;
         tile += (int64_t)gridDim.x * WAVES_PER_BLOCK) {
        const int64_t s0 = tile * SPW;
        const int ns = (int)((a.B - s0) < SPW ? (a.B - s0) : SPW);
        __builtin_amdgcn_wave_barrier();
        stage_tile(gx + s0 * sampleNP, stage, ns * (int)sampleNP, P, a.Ppad, a.magicP, a.vec4, lane);
        __builtin_amdgcn_wave_barrier();
        const float pred = eval_tile_valu<RW>(stage, ns, N, P, a.Ppad, L, w, prm, lane, K);
        if (t == 0 && srow < ns) out[s0 + srow] = pred;
    }
}

template <int RW, int NFIX, int PFIX, int LFIX>
static int launch_forward_fix(const TileGeom& g, const rulgnn_stgcn_shape* s, const float* x, const float* prm,
                          const float* bn, float* out, hipStream_t stream) {
    FwdArgs a;
    a.B = s->batch; a.ntiles = g.ntiles; a.N = s->num_patch; a.P = s->patch_size; a.Ppad = g.Ppad; a.L = s->num_layers;
    a.K = s->mpnn_k;
    a.magicP = g.magicP; a.vec4 = g.vec4 && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    a.stage_floats = g.stage_floats;
    const size_t lds = sizeof(float) * ((size_t)EvalWeightsLds<RW>::floats(a.L, a.K) + (size_t)WAVES_PER_BLOCK * g.stage_floats);
    if (lds > 160 * 1024) return RULGNN_EUNSUPPORTED;
    if (lds > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&stgcn_forward_eval_kernel<RW, NFIX, PFIX, LFIX>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return RULGNN_EHIP;
    }
    const int grid = persistent_grid(stgcn_forward_eval_kernel<RW, NFIX, PFIX, LFIX>, g.ntiles, lds, 4);
    (void)hipGetLastError();   // drop any stale error of the caller's earlier HIP calls
    hipLaunchKernelGGL((stgcn_forward_eval_kernel<RW, NFIX, PFIX, LFIX>), dim3(grid), dim3(BLOCK), lds, stream, x, prm, bn, out, a);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

template <int RW>
static int launch_forward(const TileGeom& g, const rulgnn_stgcn_shape* s, const float* x, const float* prm,
                          const float* bn, float* out, hipStream_t stream) {
    if constexpr (RW == 16) {
        if (s->mpnn_k == 1 && s->num_patch == 14 && s->num_layers == 2 && s->patch_size == 30) return launch_forward_fix<RW, 14, 30, 2>(g, s, x, prm, bn, out, stream);
        if (s->mpnn_k == 1 && s->num_patch == 14 && s->num_layers == 2 && s->patch_size == 50) return launch_forward_fix<RW, 14, 50, 2>(g, s, x, prm, bn, out, stream);
    }
    return launch_forward_fix<RW, 0, 0, 0>(g, s, x, prm, bn, out, stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// The launch behind the wide matrix-core kernel (stgcn_forward_mx.hip: stgcn_forward_mxw_kernel): scans the predictions and
// recomputes every non-finite one with the exact routine -- a sample whose arithmetic left the f16 range, or whose statistics are
// NaN (constant patch, Model.py:41-52: the exact routine reproduces the reference's NaN placement).  A workgroup that finds
// nothing (every dataset the reference wires is scaled to [0, 1]) has read 4 bytes per sample and leaves without touching a weight.
template <int RW>
__global__ __launch_bounds__(BLOCK) void stgcn_forward_fixup_kernel(const float* __restrict__ gx, const float* __restrict__ prm,
                                                                    const float* __restrict__ bn, float* out, FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int SPW = Row<RW>::SPW;
    const int N = a.N, L = a.L, P = a.P;
    const int K = a.K;                            // 1 -- passed like the exact kernel passes it: the two must stay the SAME code, bit for bit
    EvalWeightsLds<RW> w;
    w.bind(smem, L, K);
    float* stage = smem + EvalWeightsLds<RW>::floats(L, K) + (threadIdx.x >> 6) * a.stage_floats;
    const int lane = threadIdx.x & 63;
    const int srow = lane / RW, t = lane % RW;
    const int64_t sampleNP = (int64_t)N * P;
    bool filled = false;
    for (int64_t base = (int64_t)blockIdx.x * BLOCK; base < a.B; base += (int64_t)gridDim.x * BLOCK) {
        const int64_t mine = base + threadIdx.x;
        const bool bad = mine < a.B && !(__builtin_fabsf(out[mine]) <= 3.0e38f);
        if (!__syncthreads_or(bad)) continue;
        if (!filled) {
            eval_weights_fill<RW>(w, prm, bn, N, L, threadIdx.x, BLOCK, K);
            __syncthreads();
            filled = true;
        }
        uint64_t todo = __builtin_amdgcn_ballot_w64(bad);
        while (todo) {
            const int b = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int64_t smp = base + (threadIdx.x & ~63) + b;
            const int64_t s0 = smp / SPW * SPW;
            const int ns = (int)((a.B - s0) < SPW ? (a.B - s0) : SPW);
            __builtin_amdgcn_wave_barrier();
            stage_tile(gx + s0 * sampleNP, stage, ns * (int)sampleNP, P, a.Ppad, a.magicP, a.vec4, lane);
            __builtin_amdgcn_wave_barrier();
            const float pred = eval_tile_valu<RW>(stage, ns, N, P, a.Ppad, L, w, prm, lane, K);
            if (t == 0 && s0 + srow == smp) out[smp] = pred;
        }
    }
}

template <int RW>
static int launch_fixup(const TileGeom& g, const rulgnn_stgcn_shape* s, const float* x, const float* prm, const float* bn, float* out,
                        hipStream_t stream) {
    FwdArgs a;
    a.B = s->batch; a.ntiles = g.ntiles; a.N = s->num_patch; a.P = s->patch_size; a.Ppad = g.Ppad; a.L = s->num_layers;
    a.K = 1;                                      // (the matrix-core kernels this scan follows are order 1 only)
    a.magicP = g.magicP; a.vec4 = g.vec4 && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    a.stage_floats = g.stage_floats;
    const size_t lds = sizeof(float) * ((size_t)EvalWeightsLds<RW>::floats(a.L) + (size_t)WAVES_PER_BLOCK * g.stage_floats);
    if (lds > 160 * 1024) return RULGNN_EUNSUPPORTED;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(&stgcn_forward_fixup_kernel<RW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
        return RULGNN_EHIP;
    int64_t grid = (s->batch + BLOCK - 1) / BLOCK;
    if (grid > 1024) grid = 1024;
    (void)hipGetLastError();
    hipLaunchKernelGGL((stgcn_forward_fixup_kernel<RW>), dim3((unsigned)grid), dim3(BLOCK), lds, stream, x, prm, bn, out, a);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

int stgcn_forward_fixup(const rulgnn_stgcn_shape* s, const float* x, const float* prm, const float* bn, float* out, hipStream_t stream) {
    TileGeom g;
    const int rc = tile_geometry(s, &g);
    if (rc != RULGNN_OK) return rc;
    if (s->batch == 0) return RULGNN_OK;
    switch (g.RW) {
        case 16: return launch_fixup<16>(g, s, x, prm, bn, out, stream);
        case 32: return launch_fixup<32>(g, s, x, prm, bn, out, stream);
        default: return launch_fixup<64>(g, s, x, prm, bn, out, stream);
    }
}

int stgcn_forward_eval(const rulgnn_stgcn_shape* s, const float* x, const float* prm, const float* bn, float* out,
                       hipStream_t stream, int path) {
    TileGeom g;
    const int rc = tile_geometry(s, &g);
    if (rc != RULGNN_OK) return rc;
    if (s->batch == 0) return RULGNN_OK;
    if (path != STGCN_EVAL_EXACT) {
        int mrc = stgcn_forward_eval_mx(s, x, prm, bn, out, stream);
        if (mrc == RULGNN_EUNSUPPORTED) {                                         // 16 <= num_patch <= 47: the wide kernel + its scan
            mrc = stgcn_forward_eval_mxw(s, x, prm, bn, out, stream);
            if (mrc == RULGNN_OK) return stgcn_forward_fixup(s, x, prm, bn, out, stream);
        }
        if (mrc != RULGNN_EUNSUPPORTED || path == STGCN_EVAL_MX) return mrc;      // launched (or failed for a real reason)
    }
    switch (g.RW) {
        case 16: return launch_forward<16>(g, s, x, prm, bn, out, stream);
        case 32: return launch_forward<32>(g, s, x, prm, bn, out, stream);
        default: return launch_forward<64>(g, s, x, prm, bn, out, stream);
    }
}

}  // namespace rulgnn
