// Fused eval-mode ST_GCN forward for gfx950: ONE kernel from raw windows to the RUL prediction.
// Reference path replaced: ST_GCN_model.forward under model.eval() -- models/ST_GCN/Model.py:208-222
// (~90 ATen dispatches per call in the reference; here x is read once from HBM and 4 bytes per
// sample are written back).
#include "stgcn_device.hpp"
#ifndef EVAL_LDS_CONV
#define EVAL_LDS_CONV false   // measured: 93-94 us vs 87-88 us with scalar-operand weights (batch 65536): the eval kernel has no spill problem
#endif
#include "stgcn_host.hpp"

namespace rulgnn {

struct FwdArgs {
    int64_t B;
    int64_t ntiles;
    int N, P, Ppad, L;
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
    constexpr int WS = wstride<RW>();
    constexpr int SPW = Row<RW>::SPW;
    const int N = NFIX ? NFIX : a.N, L = LFIX ? LFIX : a.L, P = PFIX ? PFIX : a.P, LS = layer_stride(N);
    float* wlds = smem;                          // [L+1][RW][WS] theta rows per layer, then fc1 rows
    float* bnf = wlds + (L + 1) * RW * WS;       // [L][2][2][F]   folded BatchNorm scale / shift
    float* vecs = bnf + L * 4 * F;               // [L+2][RW]      theta bias per layer, fc1 bias, fc2 weight
    float* convw = vecs + (L + 2) * RW;          // [L][2][F][2F]  conv weights as stored ([co][ci][tap])
    float* stage_all = convw + (EVAL_LDS_CONV ? L * 2 * F * F * 2 : 0);

    // ---- block prologue: weights that vary per lane go to LDS, zero padded to the row width ----
    for (int i = threadIdx.x; i < (L + 1) * RW * RW; i += BLOCK) {
        const int m = i / (RW * RW), j = (i / RW) % RW, k = i % RW;
        const float* src = m < L ? prm + m * LS + off_theta_w(N) : prm + off_fc1_w(N, L);
        wlds[(m * RW + j) * WS + k] = (j < N && k < N) ? src[j * N + k] : 0.f;
    }
    for (int i = threadIdx.x; i < (L + 2) * RW; i += BLOCK) {
        const int m = i / RW, j = i % RW;
        const float* src = m < L ? prm + m * LS + off_theta_b(N) : (m == L ? prm + off_fc1_b(N, L) : prm + off_fc2_w(N, L));
        vecs[i] = j < N ? src[j] : 0.f;
    }
    if constexpr (EVAL_LDS_CONV) {
        for (int i = threadIdx.x; i < L * 2 * F * F * 2; i += BLOCK) {
            const int l = i / (2 * F * F * 2), r = i % (2 * F * F * 2);
            convw[i] = prm[l * LS + off_conv_w(N, r / (F * F * 2)) + r % (F * F * 2)];
        }
    }
    for (int i = threadIdx.x; i < L * 2 * F; i += BLOCK) {
        const int l = i / (2 * F), blk = (i / F) % 2, c = i % F;
        const float mean = bn[((l * 2 + blk) * 2 + 0) * F + c];
        const float var = bn[((l * 2 + blk) * 2 + 1) * F + c];
        const float g = prm[l * LS + off_bn_g(N, blk) + c], b = prm[l * LS + off_bn_b(N, blk) + c];
        const float sc = g / sqrtf(var + BN_EPS);
        bnf[((l * 2 + blk) * 2 + 0) * F + c] = sc;
        bnf[((l * 2 + blk) * 2 + 1) * F + c] = b - mean * sc;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int srow = lane / RW, t = lane % RW;
    float* stage = stage_all + wave * a.stage_floats;
    const float fc2_b = prm[off_fc2_b(N, L)];
    const int64_t sampleNP = (int64_t)N * P;

    for (int64_t tile = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave; tile < a.ntiles;
         tile += (int64_t)gridDim.x * WAVES_PER_BLOCK) {
        const int64_t s0 = tile * SPW;
        const int ns = (int)((a.B - s0) < SPW ? (a.B - s0) : SPW);
        __builtin_amdgcn_wave_barrier();
        stage_tile(gx + s0 * sampleNP, stage, ns * (int)sampleNP, P, a.Ppad, a.magicP, a.vec4, lane);
        __builtin_amdgcn_wave_barrier();

        const bool valid = (srow < ns) && (t < N);
        float X[F];
#pragma unroll
        for (int c = 0; c < F; ++c) X[c] = 0.f;
        if (valid) patch_statistics(stage + (srow * N + t) * a.Ppad, P, X);

        constexpr int NA = RW == 16 ? F : NPAIR;      // RW 16: lane-distributed adjacency rows (MFMA path)
        float A[NA];
        if constexpr (RW == 16) {
            pearson_rows_mfma(X, srow < ns, N, stage, lane, A);
        } else {
            pearson_adjacency<RW>(X, valid, N, A);
        }

        for (int l = 0; l < L; ++l) {
            const float* lp = prm + l * LS;
            const float* bl = bnf + l * 4 * F;
            float AX[F], H[F], z[F], o0[F];
            if constexpr (RW == 16) {
                adj_aggregate_mfma(A, X, AX);
            } else {
                adj_aggregate(A, X, AX);
            }
            const float tb = vecs[l * RW + t];
#pragma unroll
            for (int c = 0; c < F; ++c) H[c] = tb;
            Row<RW>::project10(H, AX, wlds + (l * RW + t) * WS, N);     // theta(A.X), Model.py:87
#pragma unroll
            for (int c = 0; c < F; ++c) H[c] = leaky(H[c]);
            if constexpr (EVAL_LDS_CONV) causal_conv_lds<RW, 1>(H, convw + (l * 2 + 0) * F * F * 2, t, z);
            else causal_conv<RW, 1>(H, lp + off_conv_w(N, 0), t, z);             // conv_block1, Model.py:134-146
#pragma unroll
            for (int c = 0; c < F; ++c) o0[c] = relu(relu(fmaf(z[c], bl[c], bl[F + c])) + H[c]);
            if constexpr (EVAL_LDS_CONV) causal_conv_lds<RW, 2>(o0, convw + (l * 2 + 1) * F * F * 2, t, z);
            else causal_conv<RW, 2>(o0, lp + off_conv_w(N, 1), t, z);            // conv_block2 (dilation 2), Model.py:148-160
#pragma unroll
            for (int c = 0; c < F; ++c) {
                const float o1 = relu(relu(fmaf(z[c], bl[2 * F + c], bl[3 * F + c])) + o0[c]);
                X[c] = valid ? o1 + X[c] : 0.f;                          // Dropout is identity in eval; out += res
            }
        }
        // AdaptiveMaxPool1d over the ten channels (NaN-propagating like torch), Model.py:218-219
        float pooled = X[0];
#pragma unroll
        for (int c = 1; c < F; ++c) pooled = (X[c] > pooled || X[c] != X[c]) ? X[c] : pooled;
        pooled = valid ? pooled : 0.f;
        float y1 = vecs[L * RW + t];
        Row<RW>::project1(y1, pooled, wlds + (L * RW + t) * WS, N);     // fc1, Model.py:220
        y1 = relu(y1);
        const float pred = Row<RW>::allsum(y1 * vecs[(L + 1) * RW + t]) + fc2_b;   // fc2, Model.py:221
        if (t == 0 && srow < ns) out[s0 + srow] = pred;
    }
}

template <int RW, int NFIX, int PFIX, int LFIX>
static int launch_forward_fix(const TileGeom& g, const rulgnn_stgcn_shape* s, const float* x, const float* prm,
                          const float* bn, float* out, hipStream_t stream) {
    FwdArgs a;
    a.B = s->batch; a.ntiles = g.ntiles; a.N = s->num_patch; a.P = s->patch_size; a.Ppad = g.Ppad; a.L = s->num_layers;
    a.magicP = g.magicP; a.vec4 = g.vec4 && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    a.stage_floats = g.stage_floats;
    constexpr int WS = wstride<RW>();
    const size_t lds = sizeof(float) * ((size_t)(a.L + 1) * RW * WS + (size_t)a.L * 4 * F + (size_t)(a.L + 2) * RW +
                                        (EVAL_LDS_CONV ? (size_t)a.L * 2 * F * F * 2 : 0) +
                                        (size_t)WAVES_PER_BLOCK * g.stage_floats);
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
        if (s->num_patch == 14 && s->num_layers == 2 && s->patch_size == 30) return launch_forward_fix<RW, 14, 30, 2>(g, s, x, prm, bn, out, stream);
        if (s->num_patch == 14 && s->num_layers == 2 && s->patch_size == 50) return launch_forward_fix<RW, 14, 50, 2>(g, s, x, prm, bn, out, stream);
    }
    return launch_forward_fix<RW, 0, 0, 0>(g, s, x, prm, bn, out, stream);
}

int stgcn_forward_eval(const rulgnn_stgcn_shape* s, const float* x, const float* prm, const float* bn, float* out,
                       hipStream_t stream) {
    TileGeom g;
    const int rc = tile_geometry(s, &g);
    if (rc != RULGNN_OK) return rc;
    if (s->batch == 0) return RULGNN_OK;
    switch (g.RW) {
        case 16: return launch_forward<16>(g, s, x, prm, bn, out, stream);
        case 32: return launch_forward<32>(g, s, x, prm, bn, out, stream);
        default: return launch_forward<64>(g, s, x, prm, bn, out, stream);
    }
}

}  // namespace rulgnn
