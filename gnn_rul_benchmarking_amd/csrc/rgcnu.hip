// RGCNU on gfx950 (a GCNLayer user).
// Reference path replaced: RGCNU_model.forward -- models/RGCNU/Model.py:96-119 (adj_construction :80-93, SCL :25-43 with GCNLayer :7-22,
// TDL :46-53, FusionModule :56-78) -- and RGCNU.update, algorithms/algorithms.py:284-296 (MSE of the first head, backward, Adam).
//
// Everything per sample is small (N <= 32 sensor nodes, L <= 64 time steps, hidden widths <= 64), so the step is a handful of
// one-workgroup-per-sample kernels with their operands in LDS plus the persistent LSTM kernels of bilstm.hip in their one-direction
// form:
//   rg_adj        x -> A1, A2 = tanh(alpha (x W^T + b)); S = alpha (A1 A2^T - A2 A1^T); A = relu(tanh S); A_hat = D^-1/2 (A + I) D^-1/2
//   rg_scl        per (sample b, step l): graph convolutions with the adjacency of sample (b L + l) % batch -- the reference tiles the
//                 adjacency batch L times along the batch axis while the node signals are sample-major (Model.py:104-106); two GCN
//                 layers (1 -> H -> H), dropout, 1x1 convolution -> the LSTM's input sequence
//   lstm          nn.LSTM(N -> E) over the L steps of every sample (bilstm.hip, ndir = 1)
//   rg_fusion     1x1 convolution of x + LSTM output, 'same' convolution (k taps), the two linear heads, squared error
// and their mirror images backwards.  Parameter gradients: every workgroup owns fixed gradient entries in registers over its
// grid-stride loop over samples and writes one partial row; rows are summed in a fixed order (rows_sum) -- reproducible bit for bit.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sgemm_mfma.hpp"
#include "stgcn_device.hpp"
#include "stgcn_host.hpp"

namespace rulgnn {

namespace {

constexpr int RB = 256;                 // threads per workgroup
constexpr int RG_MAXN = 32, RG_MAXL = 64, RG_MAXH = 64, RG_MAXE = 64, RG_MAXK = 7;
constexpr int RG_OWN = 16;              // gradient entries a thread owns per tensor (RB * RG_OWN >= the largest tensor)

struct RgGeom {
    int64_t B, G;                       // samples, graphs = B * L
    int N, L, H, E, K, pad;
    float alpha;
    // flat parameter offsets (floats), the reference's named_parameters() order
    int o_t1w, o_t1b, o_t2w, o_t2b, o_g1w, o_g1b, o_g2w, o_g2b, o_cw, o_cb, o_wih, o_whh, o_bih, o_bhh, o_c1w, o_c1b, o_c2w, o_c2b,
        o_f1w, o_f1b, o_f2w, o_f2b, pcount;
    int nA, nS, nF;                     // gradient entries of the adjacency / SCL / fusion groups
    int blocks;                         // workgroups of the per-sample kernels (= their partial rows)
    int gblocks;                        // workgroups of the per-graph kernels (= partial rows of rg_scl_bwd)
    // workspace offsets (floats)
    int64_t w_A1, w_A2, w_T, w_dinv, w_Ahat, w_ax1, w_ah1, w_z2, w_sp, w_hseq, w_M, w_M2, w_dpred, w_sq, w_dM, w_dsp, w_dAg,
        w_partA, w_partS, w_partF, w_lstm, total;
};

int rg_geometry(const rulgnn_rgcnu_shape* s, RgGeom* g) {
    if (!s) return RULGNN_EINVAL;
    if (s->batch < 0 || s->num_nodes < 1 || s->time_length < 1 || s->hidden_dim < 1 || s->encoder_hidden_dim < 1 || s->kernel_size < 1)
        return RULGNN_EINVAL;
    if (s->num_nodes > RG_MAXN || s->time_length > RG_MAXL || s->hidden_dim > RG_MAXH || s->encoder_hidden_dim > RG_MAXE ||
        s->kernel_size > RG_MAXK || (s->kernel_size & 1) == 0)
        return RULGNN_EUNSUPPORTED;          // padding='same' with an even kernel pads asymmetrically: not wired by the reference
    const int N = s->num_nodes, L = s->time_length, H = s->hidden_dim, E = s->encoder_hidden_dim, K = s->kernel_size;
    if (E * E * K > RB * RG_OWN || H * H > RB * RG_OWN || E * L > RB * RG_OWN || N * L > RB * RG_OWN || E * N > RB * RG_OWN)
        return RULGNN_EUNSUPPORTED;
    g->B = s->batch; g->G = s->batch * L;
    g->N = N; g->L = L; g->H = H; g->E = E; g->K = K; g->pad = (K - 1) / 2; g->alpha = s->alpha;
    int o = 0;
    auto tk = [&](int n) { const int r = o; o += n; return r; };
    g->o_t1w = tk(N * L); g->o_t1b = tk(N); g->o_t2w = tk(N * L); g->o_t2b = tk(N);
    g->nA = o;
    g->o_g1w = tk(H); g->o_g1b = tk(H); g->o_g2w = tk(H * H); g->o_g2b = tk(H); g->o_cw = tk(H); g->o_cb = tk(1);
    g->nS = o - g->nA;
    g->o_wih = tk(4 * E * N); g->o_whh = tk(4 * E * E); g->o_bih = tk(4 * E); g->o_bhh = tk(4 * E);
    g->o_c1w = tk(E * N); g->o_c1b = tk(E); g->o_c2w = tk(E * E * K); g->o_c2b = tk(E); g->o_f1w = tk(E * L); g->o_f1b = tk(1);
    g->nF = o - g->o_c1w;
    g->o_f2w = tk(E * L); g->o_f2b = tk(1);
    g->pcount = o;
    g->blocks = (int)(g->B < 512 ? (g->B > 0 ? g->B : 1) : 512);
    g->gblocks = (int)(g->G < 2048 ? (g->G > 0 ? g->G : 1) : 2048);
    int64_t w = 0;
    auto wk = [&](int64_t n) { const int64_t r = w; w += (n + 63) & ~(int64_t)63; return r; };
    const int64_t B = g->B, G = g->G;
    g->w_A1 = wk(B * N * N); g->w_A2 = wk(B * N * N); g->w_T = wk(B * N * N); g->w_dinv = wk(B * N); g->w_Ahat = wk(B * N * N);
    g->w_ax1 = wk(G * N); g->w_ah1 = wk(G * N * H); g->w_z2 = wk(G * N * H);
    g->w_sp = wk(G * N); g->w_hseq = wk(G * E); g->w_M = wk(B * E * L); g->w_M2 = wk(B * E * L);
    g->w_dpred = wk(B); g->w_sq = wk(B); g->w_dM = wk(G * E); g->w_dsp = wk(G * N); g->w_dAg = wk(G * N * N);
    g->w_partA = wk((int64_t)g->blocks * g->nA); g->w_partS = wk((int64_t)g->gblocks * g->nS); g->w_partF = wk((int64_t)g->blocks * g->nF);
    rulgnn_bilstm_shape ls{L, (int32_t)(B > 0 ? B : 1), N, E};
    const size_t lb = bilstm_workspace_bytes(&ls);
    if (lb == 0) return RULGNN_EUNSUPPORTED;
    g->w_lstm = wk((int64_t)(lb / sizeof(float)) + 64);
    g->total = w;
    return RULGNN_OK;
}

__device__ __forceinline__ float rg_tanh(float v) { return tanhf(v); }

// ---- adjacency ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RB) void rg_adj_kernel(RgGeom g, const float* __restrict__ x, const float* __restrict__ prm, float* __restrict__ ws) {
    __shared__ float xs[RG_MAXN * RG_MAXL], a1[RG_MAXN * RG_MAXN], a2[RG_MAXN * RG_MAXN], tt[RG_MAXN * RG_MAXN], dv[RG_MAXN];
    const int N = g.N, L = g.L, tid = threadIdx.x;
    for (int64_t b = blockIdx.x; b < g.B; b += gridDim.x) {
        for (int i = tid; i < N * L; i += RB) xs[i] = x[b * N * L + i];
        __syncthreads();
        for (int i = tid; i < N * N; i += RB) {
            const int n = i / N, m = i % N;
            float u1 = prm[g.o_t1b + m], u2 = prm[g.o_t2b + m];
            for (int l = 0; l < L; ++l) {
                const float xv = xs[n * L + l];
                u1 = fmaf(xv, prm[g.o_t1w + m * L + l], u1);
                u2 = fmaf(xv, prm[g.o_t2w + m * L + l], u2);
            }
            a1[i] = rg_tanh(g.alpha * u1);
            a2[i] = rg_tanh(g.alpha * u2);
        }
        __syncthreads();
        for (int i = tid; i < N * N; i += RB) {
            const int r = i / N, c = i % N;
            float sacc = 0.f;
            for (int m = 0; m < N; ++m) sacc += a1[r * N + m] * a2[c * N + m] - a2[r * N + m] * a1[c * N + m];
            tt[i] = rg_tanh(g.alpha * sacc);
        }
        __syncthreads();
        if (tid < N) {
            float r = 1.0f;                                    // the self loop
            for (int c = 0; c < N; ++c) r += fmaxf(tt[tid * N + c], 0.f);
            dv[tid] = 1.0f / sqrtf(r);
        }
        __syncthreads();
        for (int i = tid; i < N * N; i += RB) {
            const int r = i / N, c = i % N;
            const float at = fmaxf(tt[i], 0.f) + (r == c ? 1.f : 0.f);
            ws[g.w_A1 + b * N * N + i] = a1[i];
            ws[g.w_A2 + b * N * N + i] = a2[i];
            ws[g.w_T + b * N * N + i] = tt[i];
            ws[g.w_Ahat + b * N * N + i] = dv[r] * at * dv[c];
        }
        if (tid < N) ws[g.w_dinv + b * N + tid] = dv[tid];
        __syncthreads();
    }
}

// dropout keep-scale of element (global sample, step, node, channel); counter-based hash shared with the oracle
__device__ __forceinline__ float rg_keep(uint32_t key, uint32_t thr, float scale, int64_t sample, int l, int n, int h, const RgGeom& g) {
    if (thr == 0u) return 1.f;
    const uint32_t ctr = (uint32_t)(((sample * g.L + l) * g.N + n) * g.H + h);
    return lowbias32(ctr ^ key) >= thr ? scale : 0.f;
}

// ---- spatial correlation layer --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RB) void rg_scl_kernel(RgGeom g, const float* __restrict__ x, const float* __restrict__ prm, float* __restrict__ ws,
                                                    uint32_t key, uint32_t thr, float scale, int64_t sample_offset) {
    // LDS sized by the actual widths (the static worst case, 45 KB, allowed three workgroups per CU: 124 us at batch 256)
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int N = g.N, L = g.L, H = g.H, tid = threadIdx.x;
    float* ah = sm;                       // [N][N]
    float* xs = ah + N * N;               // [N]
    float* ax = xs + N;                   // [N]
    float* h1 = ax + N;                   // [N][H]
    float* a1 = h1 + N * H;               // [N][H]
    float* h2 = a1 + N * H;               // [N][H]
    float* w2t = h2 + N * H;              // [H][H + 1]: W2[h][k] -> [k][h], padded rows
    for (int i = tid; i < H * H; i += RB) w2t[(i % H) * (H + 1) + i / H] = prm[g.o_g2w + i];
    __syncthreads();
    for (int64_t gi = blockIdx.x; gi < g.G; gi += gridDim.x) {
        const int64_t b = gi / L;
        const int l = (int)(gi % L);
        const int64_t a = gi % g.B;                             // the adjacency this graph convolves with (Model.py:104-106)
        for (int i = tid; i < N * N; i += RB) ah[i] = ws[g.w_Ahat + a * N * N + i];
        if (tid < N) xs[tid] = x[(b * N + tid) * L + l];
        __syncthreads();
        if (tid < N) {
            float v = 0.f;
            for (int j = 0; j < N; ++j) v = fmaf(ah[tid * N + j], xs[j], v);
            ax[tid] = v;
            ws[g.w_ax1 + gi * N + tid] = v;
        }
        __syncthreads();
        for (int i = tid; i < N * H; i += RB) {
            const int n = i / H, h = i % H;
            h1[i] = fmaxf(fmaf(ax[n], prm[g.o_g1w + h], prm[g.o_g1b + h]), 0.f);
        }
        __syncthreads();
        for (int i = tid; i < N * H; i += RB) {
            const int n = i / H, h = i % H;
            float v = 0.f;
            for (int j = 0; j < N; ++j) v = fmaf(ah[n * N + j], h1[j * H + h], v);
            a1[i] = v;
            ws[g.w_ah1 + gi * N * H + i] = v;
        }
        __syncthreads();
        for (int i = tid; i < N * H; i += RB) {
            const int n = i / H, h = i % H;
            float v = prm[g.o_g2b + h];
            for (int k = 0; k < H; ++k) v = fmaf(a1[n * H + k], w2t[k * (H + 1) + h], v);
            ws[g.w_z2 + gi * N * H + i] = v;
            h2[i] = fmaxf(v, 0.f) * rg_keep(key, thr, scale, sample_offset + b, l, n, h, g);
        }
        __syncthreads();
        if (tid < N) {
            float v = prm[g.o_cb];
            for (int h = 0; h < H; ++h) v = fmaf(h2[tid * H + h], prm[g.o_cw + h], v);
            ws[g.w_sp + (b * L + l) * N + tid] = v;             // [sample][step][node]: the LSTM's batch-first input
        }
        __syncthreads();
    }
}

// ---- fusion module, heads, squared error ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(RB) void rg_fusion_kernel(RgGeom g, const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ prm,
                                                       float* __restrict__ ws, float* __restrict__ pred, float* __restrict__ stdv, float inv_gb) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int N = g.N, L = g.L, E = g.E, K = g.K, pad = g.pad, LP = L + K - 1, tid = threadIdx.x;
    float* Mp = sm;                       // [E][LP] zero-padded
    float* xs = Mp + E * LP;              // [N][L]
    float* w2 = xs + N * L;               // [E][E][K]
    float* red = w2 + E * E * K;          // [2][RB]
    for (int i = tid; i < E * E * K; i += RB) w2[i] = prm[g.o_c2w + i];
    for (int64_t b = blockIdx.x; b < g.B; b += gridDim.x) {
        for (int i = tid; i < E * LP; i += RB) Mp[i] = 0.f;
        for (int i = tid; i < N * L; i += RB) xs[i] = x[b * N * L + i];
        __syncthreads();
        for (int i = tid; i < E * L; i += RB) {
            const int e = i / L, l = i % L;
            float v = prm[g.o_c1b + e];
            for (int n = 0; n < N; ++n) v = fmaf(prm[g.o_c1w + e * N + n], xs[n * L + l], v);
            v += ws[g.w_hseq + (b * L + l) * E + e];
            Mp[e * LP + pad + l] = v;
            ws[g.w_M + b * E * L + i] = v;
        }
        __syncthreads();
        float p1 = 0.f, p2 = 0.f;
        for (int i = tid; i < E * L; i += RB) {
            const int o = i / L, l = i % L;
            float v = prm[g.o_c2b + o];
            for (int e = 0; e < E; ++e)
                for (int j = 0; j < K; ++j) v = fmaf(w2[(o * E + e) * K + j], Mp[e * LP + l + j], v);
            ws[g.w_M2 + b * E * L + i] = v;
            p1 = fmaf(v, prm[g.o_f1w + i], p1);
            p2 = fmaf(v, prm[g.o_f2w + i], p2);
        }
        red[tid] = p1;
        red[RB + tid] = p2;
        __syncthreads();
        for (int m = RB / 2; m > 0; m >>= 1) {
            if (tid < m) { red[tid] += red[tid + m]; red[RB + tid] += red[RB + tid + m]; }
            __syncthreads();
        }
        if (tid == 0) {
            const float pr = red[0] + prm[g.o_f1b];
            pred[b] = pr;
            if (stdv) stdv[b] = red[RB] + prm[g.o_f2b];
            if (y) {
                const float d = pr - y[b];
                ws[g.w_sq + b] = d * d * inv_gb;
                ws[g.w_dpred + b] = 2.f * d * inv_gb;
            }
        }
        __syncthreads();
    }
}

// thread-owned gradient entries: entry e of a tensor of n values belongs to thread e % RB, slot e / RB
#define RG_FOR_OWNED(n, e, s) _Pragma("unroll") for (int s = 0, e = threadIdx.x; s < RG_OWN; ++s, e += RB) if (e < (n))

// ---- fusion backward ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RB) void rg_fusion_bwd_kernel(RgGeom g, const float* __restrict__ x, const float* __restrict__ dpred_in,
                                                           const float* __restrict__ prm, float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int N = g.N, L = g.L, E = g.E, K = g.K, pad = g.pad, LP = L + K - 1, tid = threadIdx.x;
    float* Mp = sm;                       // [E][LP]  forward M, zero padded
    float* d2 = Mp + E * LP;              // [E][LP]  d M2, zero padded by pad' = K - 1 - pad on the left (transposed convolution)
    float* xs = d2 + E * LP;              // [N][L]
    float* w2 = xs + N * L;               // [E][E][K]
    float* dMs = w2 + E * E * K;          // [E][L]
    for (int i = tid; i < E * E * K; i += RB) w2[i] = prm[g.o_c2w + i];
    float gc1w[RG_OWN], gc1b[RG_OWN], gc2w[RG_OWN], gc2b[RG_OWN], gf1w[RG_OWN], gf1b = 0.f;
#pragma unroll
    for (int s = 0; s < RG_OWN; ++s) gc1w[s] = gc1b[s] = gc2w[s] = gc2b[s] = gf1w[s] = 0.f;
    const float* dpred = dpred_in ? dpred_in : ws + g.w_dpred;
    for (int64_t b = blockIdx.x; b < g.B; b += gridDim.x) {
        const float dp = dpred[b];
        for (int i = tid; i < E * LP; i += RB) { Mp[i] = 0.f; d2[i] = 0.f; }
        for (int i = tid; i < N * L; i += RB) xs[i] = x[b * N * L + i];
        __syncthreads();
        for (int i = tid; i < E * L; i += RB) {
            const int e = i / L, l = i % L;
            Mp[e * LP + pad + l] = ws[g.w_M + b * E * L + i];
            d2[e * LP + (K - 1 - pad) + l] = dp * prm[g.o_f1w + i];
        }
        RG_FOR_OWNED(E * L, e, s) gf1w[s] = fmaf(dp, ws[g.w_M2 + b * E * L + e], gf1w[s]);
        if (tid == 0) gf1b += dp;
        __syncthreads();
        // d M[e][l] = sum_{o, j} W2[o][e][j] dM2[o][l - j + pad]
        for (int i = tid; i < E * L; i += RB) {
            const int e = i / L, l = i % L;
            float v = 0.f;
            for (int o = 0; o < E; ++o)
                for (int j = 0; j < K; ++j) v = fmaf(w2[(o * E + e) * K + j], d2[o * LP + (K - 1 - pad) + l - j + pad], v);
            dMs[i] = v;
            ws[g.w_dM + (b * L + l) * E + e] = v;               // [sample][step][E]: d (LSTM output)
        }
        // d W2[o][e][j] = sum_l dM2[o][l] M[e][l + j - pad]
        RG_FOR_OWNED(E * E * K, e_, s) {
            const int o = e_ / (E * K), e = (e_ / K) % E, j = e_ % K;
            float v = 0.f;
            for (int l = 0; l < L; ++l) v = fmaf(d2[o * LP + (K - 1 - pad) + l], Mp[e * LP + l + j], v);
            gc2w[s] += v;
        }
        RG_FOR_OWNED(E, o, s) {
            float v = 0.f;
            for (int l = 0; l < L; ++l) v += d2[o * LP + (K - 1 - pad) + l];
            gc2b[s] += v;
        }
        __syncthreads();
        RG_FOR_OWNED(E * N, e_, s) {
            const int e = e_ / N, n = e_ % N;
            float v = 0.f;
            for (int l = 0; l < L; ++l) v = fmaf(dMs[e * L + l], xs[n * L + l], v);
            gc1w[s] += v;
        }
        RG_FOR_OWNED(E, e, s) {
            float v = 0.f;
            for (int l = 0; l < L; ++l) v += dMs[e * L + l];
            gc1b[s] += v;
        }
        __syncthreads();
    }
    float* row = ws + g.w_partF + (int64_t)blockIdx.x * g.nF;
    const int base = g.o_c1w;
    RG_FOR_OWNED(E * N, e, s) row[g.o_c1w - base + e] = gc1w[s];
    RG_FOR_OWNED(E, e, s) row[g.o_c1b - base + e] = gc1b[s];
    RG_FOR_OWNED(E * E * K, e, s) row[g.o_c2w - base + e] = gc2w[s];
    RG_FOR_OWNED(E, e, s) row[g.o_c2b - base + e] = gc2b[s];
    RG_FOR_OWNED(E * L, e, s) row[g.o_f1w - base + e] = gf1w[s];
    if (tid == 0) row[g.o_f1b - base] = gf1b;
}

// ---- spatial correlation layer backward -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(RB) void rg_scl_bwd_kernel(RgGeom g, const float* __restrict__ x, const float* __restrict__ prm, float* __restrict__ ws,
                                                        uint32_t key, uint32_t thr, float scale, int64_t sample_offset) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int N = g.N, L = g.L, H = g.H, tid = threadIdx.x;
    float* ah = sm;                       // [N][N]
    float* xs = ah + N * N;               // [N] x 4: xs, ax, dsp, dax
    float* ax = xs + N;
    float* dsp = ax + N;
    float* dax = dsp + N;
    float* h1 = dax + N;                  // [N][H] x 5
    float* a1 = h1 + N * H;
    float* dz2 = a1 + N * H;
    float* da1 = dz2 + N * H;
    float* dz1 = da1 + N * H;
    float* w2 = dz1 + N * H;              // [H][H + 1]: W2[h][k], padded rows
    for (int i = tid; i < H * H; i += RB) w2[(i / H) * (H + 1) + i % H] = prm[g.o_g2w + i];
    float gg2w[RG_OWN], gg1w = 0.f, gg1b = 0.f, gg2b = 0.f, gcw = 0.f, gcb = 0.f;
#pragma unroll
    for (int s = 0; s < RG_OWN; ++s) gg2w[s] = 0.f;
    __syncthreads();
    // graphs over the workgroups in a fixed stride (grid = rg_graph_blocks(), a function of the shape alone): every partial row
    // sums the same graphs on every run
    for (int64_t gi = blockIdx.x; gi < g.G; gi += gridDim.x) {
        {
            const int64_t b = gi / L, a = gi % g.B;
            const int l = (int)(gi % L);
            for (int i = tid; i < N * N; i += RB) ah[i] = ws[g.w_Ahat + a * N * N + i];
            if (tid < N) {
                xs[tid] = x[(b * N + tid) * L + l];
                ax[tid] = ws[g.w_ax1 + gi * N + tid];
                dsp[tid] = ws[g.w_dsp + gi * N + tid];
            }
            __syncthreads();
            for (int i = tid; i < N * H; i += RB) {
                const int n = i / H, h = i % H;
                h1[i] = fmaxf(fmaf(ax[n], prm[g.o_g1w + h], prm[g.o_g1b + h]), 0.f);
                a1[i] = ws[g.w_ah1 + gi * N * H + i];
                const float z = ws[g.w_z2 + gi * N * H + i];
                const float kp = rg_keep(key, thr, scale, sample_offset + b, l, n, h, g);
                dz2[i] = z > 0.f ? dsp[n] * prm[g.o_cw + h] * kp : 0.f;
                dz1[i] = fmaxf(z, 0.f) * kp;                     // the dropped-out hidden features (dz1 is free until gcn1's backward)
            }
            __syncthreads();
            // conv1d (1x1) and gcn2 parameter gradients
            if (tid < H) {
                float vw = 0.f, vb = 0.f;
                for (int n = 0; n < N; ++n) {
                    vw = fmaf(dsp[n], dz1[n * H + tid], vw);
                    vb += dz2[n * H + tid];
                }
                gcw += vw;
                gg2b += vb;
            }
            if (tid == 0) {
                float v = 0.f;
                for (int n = 0; n < N; ++n) v += dsp[n];
                gcb += v;
            }
            RG_FOR_OWNED(H * H, e, s) {
                const int h = e / H, k = e % H;
                float v = 0.f;
                for (int n = 0; n < N; ++n) v = fmaf(dz2[n * H + h], a1[n * H + k], v);
                gg2w[s] += v;
            }
            // d (A_hat h1)[n][k] = sum_h dz2[n][h] W2[h][k]
            for (int i = tid; i < N * H; i += RB) {
                const int n = i / H, k = i % H;
                float v = 0.f;
                for (int h = 0; h < H; ++h) v = fmaf(dz2[n * H + h], w2[h * (H + 1) + k], v);
                da1[i] = v;
            }
            __syncthreads();
            // d h1[j][h] = sum_i A_hat[i][j] da1[i][h];  gcn1
            for (int i = tid; i < N * H; i += RB) {
                const int j = i / H, h = i % H;
                float v = 0.f;
                for (int r = 0; r < N; ++r) v = fmaf(ah[r * N + j], da1[r * H + h], v);
                dz1[i] = h1[i] > 0.f ? v : 0.f;
            }
            __syncthreads();
            if (tid < H) {
                float vw = 0.f, vb = 0.f;
                for (int n = 0; n < N; ++n) {
                    vw = fmaf(dz1[n * H + tid], ax[n], vw);
                    vb += dz1[n * H + tid];
                }
                gg1w += vw;
                gg1b += vb;
            }
            if (tid < N) {
                float v = 0.f;
                for (int h = 0; h < H; ++h) v = fmaf(dz1[tid * H + h], prm[g.o_g1w + h], v);
                dax[tid] = v;
            }
            __syncthreads();
            // d A_hat of this graph: da1 h1^T + dax x^T
            for (int i = tid; i < N * N; i += RB) {
                const int r = i / N, c = i % N;
                float v = dax[r] * xs[c];
                for (int h = 0; h < H; ++h) v = fmaf(da1[r * H + h], h1[c * H + h], v);
                ws[g.w_dAg + gi * N * N + i] = v;
            }
            __syncthreads();
        }
    }
    float* row = ws + g.w_partS + (int64_t)blockIdx.x * g.nS;
    const int base = g.o_g1w;
    if (tid < H) {
        row[g.o_g1w - base + tid] = gg1w;
        row[g.o_g1b - base + tid] = gg1b;
        row[g.o_g2b - base + tid] = gg2b;
        row[g.o_cw - base + tid] = gcw;
    }
    RG_FOR_OWNED(H * H, e, s) row[g.o_g2w - base + e] = gg2w[s];
    if (tid == 0) row[g.o_cb - base] = gcb;
}

// ---- spatial correlation layer on the fp32 matrix cores (N <= 16 nodes, H = 32: every wiring of the reference but N-CMAPSS's 20 nodes) ----
// The kernels above give a graph to a WORKGROUP: eight barriers per graph and two LDS reads per multiply-add (89 / 203 us at batch 256,
// half of the step).  Here a graph belongs to a WAVEFRONT -- four graphs in flight per workgroup, no workgroup barrier in the loop, LDS
// operations of one wavefront execute in order -- and every product of the layer is v_mfma_f32_16x16x4f32 with the node axis padded to
// 16: lane (kq, li) = (lane / 16, lane % 16) feeds A[m = li][k = 4 s + kq] and B[k = 4 s + kq][n = li] and receives
// C[m = 4 kq + r][n = li].  Operands come from per-wavefront LDS tiles (pitch 33) or, where they are rank-one (h1 = relu(ax w1 + b1)),
// are formed in registers; W2 sits in 16 registers per lane in the operand form of the kernel's one product with it.  Pad rows / columns
// of A_hat are zero, so whatever the pad rows of the other operand hold never reaches a result.
constexpr int RG_MXW = 4;                // wavefronts (graphs in flight) per workgroup
constexpr int RG_TP = 33;                // tile pitch
// (NT: 16-row tiles of the node axis -- 1 for the 14-node wirings, 2 for N-CMAPSS's 20 nodes; LDS per wavefront: A_hat [16 NT][16 NT + 1] |
//  three [16 NT][33] tiles | four vectors of 16 NT)
constexpr int RG_MXH = 32;
__host__ __device__ constexpr int rg_mx_wave_floats(int NT) { return 16 * NT * (16 * NT + 1) + 3 * 16 * NT * RG_TP + 4 * 16 * NT; }
__host__ __device__ constexpr int rg_mx_red_floats() { return RG_MXW * (RG_MXH * RG_MXH + 4 * 4 * RG_MXH + 32); }
inline size_t rg_scl_mx_lds(int NT, bool bwd) {
    const int a = RG_MXW * rg_mx_wave_floats(NT), b = bwd ? rg_mx_red_floats() : 0;
    return sizeof(float) * (size_t)(a > b ? a : b);
}

__device__ __forceinline__ f32x4t rg_mfma(float a, float b, f32x4t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// sum over the 16 lanes of a row (li), every lane of the row receives it
__device__ __forceinline__ float rg_row16_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}

template <int NT>
__global__ __launch_bounds__(64 * RG_MXW, NT == 1 ? 4 : 2) void rg_scl_mx_kernel(RgGeom g, const float* __restrict__ x, const float* __restrict__ prm,
                                                                                   float* __restrict__ ws, uint32_t key, uint32_t thr, float scale,
                                                                                   int64_t sample_offset) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int H = RG_MXH, NP = 16 * NT, AP = NP + 1, WF = rg_mx_wave_floats(NT);
    const int N = g.N, L = g.L, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, kq = lane >> 4;
    float* ah = sm + wave * WF;                      // [NP][AP], zero outside [N][N]
    float* a1t = ah + NP * AP;                       // [NP][33]
    float* xs = a1t + 3 * NP * RG_TP;                // [NP]
    float* axs = xs + NP;                            // [NP]
    float w1[2], b1[2], b2[2], cw[2], w2t[2][8];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int c = 16 * t + li;
        w1[t] = prm[g.o_g1w + c]; b1[t] = prm[g.o_g1b + c]; b2[t] = prm[g.o_g2b + c]; cw[t] = prm[g.o_cw + c];
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) w2t[t][s4] = prm[g.o_g2w + c * H + 4 * s4 + kq];      // B(k, n = h) = W2[h][k]
    }
    const float cb = prm[g.o_cb];
    for (int i = lane; i < NP * AP; i += 64) ah[i] = 0.f;
    __builtin_amdgcn_wave_barrier();
    const int64_t nw = (int64_t)gridDim.x * RG_MXW;
    for (int64_t gi = (int64_t)blockIdx.x * RG_MXW + wave; gi < g.G; gi += nw) {
        const int64_t b = gi / L, a = gi % g.B;                 // the adjacency this graph convolves with (Model.py:104-106)
        const int l = (int)(gi % L);
        for (int i = lane; i < N * N; i += 64) ah[(i / N) * AP + i % N] = ws[g.w_Ahat + a * N * N + i];
        if (lane < NP) xs[lane] = lane < N ? x[(b * N + lane) * L + l] : 0.f;
        __builtin_amdgcn_wave_barrier();
        if (lane < NP) {
            float v = 0.f;
            for (int j = 0; j < N; ++j) v = fmaf(ah[lane * AP + j], xs[j], v);
            axs[lane] = v;
            if (lane < N) ws[g.w_ax1 + gi * N + lane] = v;
        }
        __builtin_amdgcn_wave_barrier();
        // a1 = A_hat h1, h1[k][h] = relu(ax[k] w1[h] + b1[h]) formed in the B operand
        {
            f32x4t acc[NT][2];
#pragma unroll
            for (int i = 0; i < NT; ++i) acc[i][0] = acc[i][1] = (f32x4t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s4 = 0; s4 < 4 * NT; ++s4) {
                const float axk = axs[4 * s4 + kq];
                const float h0 = fmaxf(fmaf(axk, w1[0], b1[0]), 0.f), h1v = fmaxf(fmaf(axk, w1[1], b1[1]), 0.f);
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    const float av = ah[(16 * i + li) * AP + 4 * s4 + kq];
                    acc[i][0] = rg_mfma(av, h0, acc[i][0]);
                    acc[i][1] = rg_mfma(av, h1v, acc[i][1]);
                }
            }
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = 16 * i + 4 * kq + r, h = 16 * t + li;
                        a1t[n * RG_TP + h] = acc[i][t][r];
                        if (n < N) ws[g.w_ah1 + gi * N * H + n * H + h] = acc[i][t][r];
                    }
        }
        __builtin_amdgcn_wave_barrier();
        // z2 = a1 W2^T + b2; dropout; the 1x1 convolution over the hidden axis
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            f32x4t acc[2] = {(f32x4t){0.f, 0.f, 0.f, 0.f}, (f32x4t){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int s4 = 0; s4 < 8; ++s4) {
                const float av = a1t[(16 * i + li) * RG_TP + 4 * s4 + kq];
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t] = rg_mfma(av, w2t[t][s4], acc[t]);
            }
            float p[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = 16 * i + 4 * kq + r, h = 16 * t + li;
                    const float z = acc[t][r] + b2[t];
                    if (n < N) {
                        ws[g.w_z2 + gi * N * H + n * H + h] = z;
                        p[r] = fmaf(fmaxf(z, 0.f) * rg_keep(key, thr, scale, sample_offset + b, l, n, h, g), cw[t], p[r]);
                    }
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = rg_row16_sum(p[r]) + cb;
                const int n = 16 * i + 4 * kq + r;
                if (li == 0 && n < N) ws[g.w_sp + (b * L + l) * N + n] = v;      // [sample][step][node]: the LSTM's batch-first input
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int NT>
__global__ __launch_bounds__(64 * RG_MXW, NT == 1 ? 3 : 2) void rg_scl_bwd_mx_kernel(RgGeom g, const float* __restrict__ x, const float* __restrict__ prm,
                                                                                       float* __restrict__ ws, uint32_t key, uint32_t thr, float scale,
                                                                                       int64_t sample_offset) {
    // after the loop the wavefronts' regions are reused for the fixed-order sum of their partial gradients:
    // g2w [RG_MXW][H H] | column sums [RG_MXW][4 kinds][4 kq][H] | cb [RG_MXW][32]
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int H = RG_MXH, NP = 16 * NT, AP = NP + 1, WF = rg_mx_wave_floats(NT);
    const int N = g.N, L = g.L, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, kq = lane >> 4;
    float* ah = sm + wave * WF;                      // [NP][AP], zero outside [N][N]
    float* dz2t = ah + NP * AP;                      // [NP][33] x 3: dz2, a1, da1
    float* a1t = dz2t + NP * RG_TP;
    float* da1t = a1t + NP * RG_TP;
    float* xs = da1t + NP * RG_TP;                   // [NP] x 4: xs, ax, dsp, dax
    float* axs = xs + NP;
    float* dsp = axs + NP;
    float* dax = dsp + NP;
    float w1[2], b1[2], cw[2], w2b[2][8], w1k[8], b1k[8];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int c = 16 * t + li;
        w1[t] = prm[g.o_g1w + c]; b1[t] = prm[g.o_g1b + c]; cw[t] = prm[g.o_cw + c];
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) w2b[t][s4] = prm[g.o_g2w + (4 * s4 + kq) * H + c];      // B(k = h, n) = W2[h][n]
    }
#pragma unroll
    for (int s4 = 0; s4 < 8; ++s4) { w1k[s4] = prm[g.o_g1w + 4 * s4 + kq]; b1k[s4] = prm[g.o_g1b + 4 * s4 + kq]; }
    f32x4t g2w[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) g2w[i][j] = (f32x4t){0.f, 0.f, 0.f, 0.f};
    float gg1w[2] = {0.f, 0.f}, gg1b[2] = {0.f, 0.f}, gg2b[2] = {0.f, 0.f}, gcw[2] = {0.f, 0.f}, gcb = 0.f;
    for (int i = lane; i < NP * AP; i += 64) ah[i] = 0.f;
    __builtin_amdgcn_wave_barrier();
    // graphs over the wavefronts in a fixed stride (the grid is a function of the shape alone): every partial row sums the same graphs
    // in the same order on every run
    const int64_t nw = (int64_t)gridDim.x * RG_MXW;
    for (int64_t gi = (int64_t)blockIdx.x * RG_MXW + wave; gi < g.G; gi += nw) {
        const int64_t b = gi / L, a = gi % g.B;
        const int l = (int)(gi % L);
        for (int i = lane; i < N * N; i += 64) ah[(i / N) * AP + i % N] = ws[g.w_Ahat + a * N * N + i];
        if (lane < NP) {
            const bool in = lane < N;
            xs[lane] = in ? x[(b * N + lane) * L + l] : 0.f;
            axs[lane] = in ? ws[g.w_ax1 + gi * N + lane] : 0.f;
            const float d = in ? ws[g.w_dsp + gi * N + lane] : 0.f;
            dsp[lane] = d;
            gcb += d;
        }
        __builtin_amdgcn_wave_barrier();
        // element-wise part in the result layout of the products: rows 16 i + 4 kq + r, columns 16 t + li
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = 16 * i + 4 * kq + r, h = 16 * t + li;
                    float dz = 0.f, av = 0.f;
                    if (n < N) {
                        const float z = ws[g.w_z2 + gi * N * H + n * H + h];
                        av = ws[g.w_ah1 + gi * N * H + n * H + h];
                        const float kp = rg_keep(key, thr, scale, sample_offset + b, l, n, h, g), d = dsp[n];
                        dz = z > 0.f ? d * cw[t] * kp : 0.f;
                        gcw[t] = fmaf(d, fmaxf(z, 0.f) * kp, gcw[t]);          // conv1d (1x1) weight: the dropped-out hidden features
                        gg2b[t] += dz;
                    }
                    dz2t[n * RG_TP + h] = dz;
                    a1t[n * RG_TP + h] = av;
                }
        __builtin_amdgcn_wave_barrier();
        // gcn2 weight gradient: g2w[h][k] += sum_n dz2[n][h] a1[n][k] (accumulators live across the graphs)
#pragma unroll
        for (int s4 = 0; s4 < 4 * NT; ++s4) {
            const int n = 4 * s4 + kq;
            const float d0 = dz2t[n * RG_TP + li], d1 = dz2t[n * RG_TP + 16 + li];
            const float e0 = a1t[n * RG_TP + li], e1 = a1t[n * RG_TP + 16 + li];
            g2w[0][0] = rg_mfma(d0, e0, g2w[0][0]);
            g2w[0][1] = rg_mfma(d0, e1, g2w[0][1]);
            g2w[1][0] = rg_mfma(d1, e0, g2w[1][0]);
            g2w[1][1] = rg_mfma(d1, e1, g2w[1][1]);
        }
        // d (A_hat h1) = dz2 W2
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            f32x4t acc[2] = {(f32x4t){0.f, 0.f, 0.f, 0.f}, (f32x4t){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int s4 = 0; s4 < 8; ++s4) {
                const float av = dz2t[(16 * i + li) * RG_TP + 4 * s4 + kq];
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t] = rg_mfma(av, w2b[t][s4], acc[t]);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) da1t[(16 * i + 4 * kq + r) * RG_TP + 16 * t + li] = acc[t][r];
        }
        __builtin_amdgcn_wave_barrier();
        // d h1 = A_hat^T da1 through the ReLU of gcn1; its parameter gradients; d ax
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            f32x4t acc[2] = {(f32x4t){0.f, 0.f, 0.f, 0.f}, (f32x4t){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int s4 = 0; s4 < 4 * NT; ++s4) {
                const int i = 4 * s4 + kq;
                const float av = ah[i * AP + 16 * jt + li];
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t] = rg_mfma(av, da1t[i * RG_TP + 16 * t + li], acc[t]);
            }
            float pd[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float axj = axs[16 * jt + 4 * kq + r];
                    const float dz1 = fmaf(axj, w1[t], b1[t]) > 0.f ? acc[t][r] : 0.f;      // pad rows: A_hat's pad columns made acc zero
                    gg1w[t] = fmaf(dz1, axj, gg1w[t]);
                    gg1b[t] += dz1;
                    pd[r] = fmaf(dz1, w1[t], pd[r]);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = rg_row16_sum(pd[r]);
                if (li == 0) dax[16 * jt + 4 * kq + r] = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // d A_hat of this graph: da1 h1^T + dax x^T
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            const float axc = axs[16 * jt + li], xc = xs[16 * jt + li];
            float hb[8];
#pragma unroll
            for (int s4 = 0; s4 < 8; ++s4) hb[s4] = fmaxf(fmaf(axc, w1k[s4], b1k[s4]), 0.f);
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                f32x4t acc = (f32x4t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s4 = 0; s4 < 8; ++s4) acc = rg_mfma(da1t[(16 * i + li) * RG_TP + 4 * s4 + kq], hb[s4], acc);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = 16 * i + 4 * kq + r, cc = 16 * jt + li;
                    if (rr < N && cc < N) ws[g.w_dAg + gi * N * N + rr * N + cc] = fmaf(dax[rr], xc, acc[r]);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // ---- one partial row per workgroup: the four wavefronts' partial sums in a fixed order ----
    __syncthreads();
    float* r2w = sm;                                 // [RG_MXW][H H]
    float* rcol = r2w + RG_MXW * H * H;              // [RG_MXW][4][4][H]: g1w, g1b, g2b, cw by (kq, column)
    float* rcb = rcol + RG_MXW * 4 * 4 * H;          // [RG_MXW][32]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) r2w[wave * H * H + (16 * i + 4 * kq + r) * H + 16 * j + li] = g2w[i][j][r];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float* c = rcol + wave * 4 * 4 * H + kq * H + 16 * t + li;
        c[0 * 4 * H] = gg1w[t];
        c[1 * 4 * H] = gg1b[t];
        c[2 * 4 * H] = gg2b[t];
        c[3 * 4 * H] = gcw[t];
    }
    if (lane < 32) rcb[wave * 32 + lane] = lane < NP ? gcb : 0.f;
    __syncthreads();
    float* row = ws + g.w_partS + (int64_t)blockIdx.x * g.nS;
    const int base = g.o_g1w, tid = threadIdx.x;
    for (int e = tid; e < H * H; e += 64 * RG_MXW) {
        float v = 0.f;
        for (int w = 0; w < RG_MXW; ++w) v += r2w[w * H * H + e];
        row[g.o_g2w - base + e] = v;
    }
    if (tid < 4 * H) {
        const int kind = tid / H, c = tid % H;
        float v = 0.f;
        for (int w = 0; w < RG_MXW; ++w)
            for (int q = 0; q < 4; ++q) v += rcol[w * 4 * 4 * H + kind * 4 * H + q * H + c];
        const int off = kind == 0 ? g.o_g1w : kind == 1 ? g.o_g1b : kind == 2 ? g.o_g2b : g.o_cw;
        row[off - base + c] = v;
    }
    if (tid == 0) {
        float v = 0.f;
        for (int i = 0; i < RG_MXW * 32; ++i) v += rcb[i];
        row[g.o_cb - base] = v;
    }
}

// ---- fusion module on the fp32 matrix cores (E = 32, k = 3, N <= 16, L <= 64: the reference's wirings) --------------------------------
// The k-tap convolution of a sample is the product [E x E k] x [E k x L] over the zero-padded tile (96 x 50 at the reference's widths),
// its transposed form the same with the weights regrouped, its weight gradient [E x L] x [L x E k]; the 1x1 convolution over the nodes
// and its weight gradient are products over N and L.  One workgroup per sample as above, but a wavefront owns a 16-column tile of the
// time axis (forward, d M) or three 16 x 16 tiles of d W2 (accumulated in registers over the workgroup's samples); the weights sit in
// operand form in registers for the whole kernel (48 per lane), the tiles in LDS with odd pitches.  rg_fusion 51 -> / rg_fusion_bwd 56 ->
// see DESIGN 3i.
constexpr int RF_E = 32, RF_K = 3, RF_PAD = 1, RF_P = 67, RF_XP = 65, RF_KE = RF_E * RF_K;      // 96 = (e, j) pairs

__host__ __device__ __forceinline__ bool rg_fusion_mx_ok(const RgGeom& g) { return g.E == RF_E && g.K == RF_K && g.N <= 32 && g.L <= 64; }

__global__ __launch_bounds__(RB) void rg_fusion_mx_kernel(RgGeom g, const float* __restrict__ x, const float* __restrict__ y,
                                                          const float* __restrict__ prm, float* __restrict__ ws, float* __restrict__ pred,
                                                          float* __restrict__ stdv, float inv_gb) {
    __shared__ __attribute__((aligned(16))) float Mp[RF_E * RF_P];      // [E][67]: M at column pad + l, zero elsewhere
    __shared__ __attribute__((aligned(16))) float xs[32 * RF_XP];       // [32][65]: x, zero rows beyond N
    __shared__ float red[2 * RB];
    constexpr int E = RF_E, P = RF_P, XP = RF_XP;
    const int N = g.N, L = g.L, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    float c1a[2][8], w2a[2][24], c1b[2][4], c2b[2][4];
    int boff[24];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) c1a[i][s4] = 4 * s4 + kq < N ? prm[g.o_c1w + (16 * i + li) * N + 4 * s4 + kq] : 0.f;
#pragma unroll
        for (int s4 = 0; s4 < 24; ++s4) w2a[i][s4] = prm[g.o_c2w + (16 * i + li) * RF_KE + 4 * s4 + kq];
#pragma unroll
        for (int r = 0; r < 4; ++r) { c1b[i][r] = prm[g.o_c1b + 16 * i + 4 * kq + r]; c2b[i][r] = prm[g.o_c2b + 16 * i + 4 * kq + r]; }
    }
#pragma unroll
    for (int s4 = 0; s4 < 24; ++s4) boff[s4] = ((4 * s4 + kq) / RF_K) * P + (4 * s4 + kq) % RF_K;
    for (int i = tid; i < 32 * XP; i += RB) xs[i] = 0.f;
    for (int i = tid; i < E * P; i += RB) Mp[i] = 0.f;
    const int l = 16 * wave + li;                 // this lane's column of the time axis (operand B and the results)
    const int nsteps = N <= 16 ? 4 : 8;           // reduction steps of the 1x1 convolution over the nodes
    for (int64_t b = blockIdx.x; b < g.B; b += gridDim.x) {
        __syncthreads();
        for (int i = tid; i < N * L; i += RB) xs[(i / L) * XP + i % L] = x[b * N * L + i];
        __syncthreads();
        {   // M = conv1x1_N(x) + LSTM output
            f32x4t acc[2] = {(f32x4t){0.f, 0.f, 0.f, 0.f}, (f32x4t){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int s4 = 0; s4 < 8; ++s4) {
                if (s4 < nsteps) {
                    const float bv = xs[(4 * s4 + kq) * XP + l];
                    acc[0] = rg_mfma(c1a[0][s4], bv, acc[0]);
                    acc[1] = rg_mfma(c1a[1][s4], bv, acc[1]);
                }
            }
            if (l < L) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float4 hv = *reinterpret_cast<const float4*>(ws + g.w_hseq + (b * L + l) * E + 16 * i + 4 * kq);
                    const float h4[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int e = 16 * i + 4 * kq + r;
                        const float v = acc[i][r] + c1b[i][r] + h4[r];
                        Mp[e * P + RF_PAD + l] = v;
                        ws[g.w_M + b * E * L + e * L + l] = v;
                    }
                }
            }
        }
        __syncthreads();
        float p1 = 0.f, p2 = 0.f;
        {   // M2 = conv_k(M) + bias; the two heads
            f32x4t acc[2] = {(f32x4t){0.f, 0.f, 0.f, 0.f}, (f32x4t){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int s4 = 0; s4 < 24; ++s4) {
                const float bv = Mp[boff[s4] + l];
                acc[0] = rg_mfma(w2a[0][s4], bv, acc[0]);
                acc[1] = rg_mfma(w2a[1][s4], bv, acc[1]);
            }
            if (l < L) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int idx = (16 * i + 4 * kq + r) * L + l;
                        const float v = acc[i][r] + c2b[i][r];
                        ws[g.w_M2 + b * E * L + idx] = v;
                        p1 = fmaf(v, prm[g.o_f1w + idx], p1);
                        p2 = fmaf(v, prm[g.o_f2w + idx], p2);
                    }
            }
        }
        red[tid] = p1;
        red[RB + tid] = p2;
        __syncthreads();
        for (int m = RB / 2; m > 0; m >>= 1) {
            if (tid < m) { red[tid] += red[tid + m]; red[RB + tid] += red[RB + tid + m]; }
            __syncthreads();
        }
        if (tid == 0) {
            const float pr = red[0] + prm[g.o_f1b];
            pred[b] = pr;
            if (stdv) stdv[b] = red[RB] + prm[g.o_f2b];
            if (y) {
                const float d = pr - y[b];
                ws[g.w_sq + b] = d * d * inv_gb;
                ws[g.w_dpred + b] = 2.f * d * inv_gb;
            }
        }
    }
}

__global__ __launch_bounds__(RB) void rg_fusion_bwd_mx_kernel(RgGeom g, const float* __restrict__ x, const float* __restrict__ dpred_in,
                                                              const float* __restrict__ prm, float* __restrict__ ws) {
    __shared__ __attribute__((aligned(16))) float Mp[RF_E * RF_P];      // forward M at column pad + l
    __shared__ __attribute__((aligned(16))) float d2[RF_E * RF_P];      // d M2 at column (k - 1 - pad) + l
    __shared__ __attribute__((aligned(16))) float dMs[RF_E * RF_XP];    // d M at column l
    __shared__ __attribute__((aligned(16))) float xs[32 * RF_XP];
    constexpr int E = RF_E, P = RF_P, XP = RF_XP, K = RF_K, D2O = RF_K - 1 - RF_PAD;
    const int N = g.N, L = g.L, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    float w2d[2][24];                  // A(m = e, k = (o, j)) = W2[o][e][j]
    int doff[24];                      // B(k = (o, j), n = l) = d2[o][(k - 1) + l - j]
#pragma unroll
    for (int s4 = 0; s4 < 24; ++s4) {
        const int o = (4 * s4 + kq) / K, j = (4 * s4 + kq) % K;
        doff[s4] = o * P + (K - 1) - j;
#pragma unroll
        for (int i = 0; i < 2; ++i) w2d[i][s4] = prm[g.o_c2w + (o * E + 16 * i + li) * K + j];
    }
    // d W2: wavefront w owns the row tile i = w & 1 and the column tiles 3 (w >> 1) .. + 2 of the [E][E k] gradient
    const int wi = wave & 1, wt0 = 3 * (wave >> 1);
    int eoff[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) eoff[t] = ((16 * (wt0 + t) + li) / K) * P + (16 * (wt0 + t) + li) % K;
    f32x4t gw2[3] = {(f32x4t){0.f, 0.f, 0.f, 0.f}, (f32x4t){0.f, 0.f, 0.f, 0.f}, (f32x4t){0.f, 0.f, 0.f, 0.f}};
    f32x4t gw1[2] = {(f32x4t){0.f, 0.f, 0.f, 0.f}, (f32x4t){0.f, 0.f, 0.f, 0.f}};      // d W1[e][n]: wavefronts 0, 1 own the row tile i = wave; column tiles of the nodes
    const int ntn = N <= 16 ? 1 : 2;
    float gc1b[RG_OWN], gc2b[RG_OWN], gf1w[RG_OWN], gf1b = 0.f;
#pragma unroll
    for (int s = 0; s < RG_OWN; ++s) gc1b[s] = gc2b[s] = gf1w[s] = 0.f;
    for (int i = tid; i < E * P; i += RB) { Mp[i] = 0.f; d2[i] = 0.f; }
    for (int i = tid; i < E * XP; i += RB) dMs[i] = 0.f;
    for (int i = tid; i < 32 * XP; i += RB) xs[i] = 0.f;
    const float* dpred = dpred_in ? dpred_in : ws + g.w_dpred;
    const int l = 16 * wave + li;
    const int ksteps = (L + 3) / 4;
    for (int64_t b = blockIdx.x; b < g.B; b += gridDim.x) {
        const float dp = dpred[b];
        __syncthreads();
        for (int i = tid; i < N * L; i += RB) xs[(i / L) * XP + i % L] = x[b * N * L + i];
        for (int i = tid; i < E * L; i += RB) {
            const int e = i / L, ll = i % L;
            Mp[e * P + RF_PAD + ll] = ws[g.w_M + b * E * L + i];
            d2[e * P + D2O + ll] = dp * prm[g.o_f1w + i];
        }
        RG_FOR_OWNED(E * L, e, s) gf1w[s] = fmaf(dp, ws[g.w_M2 + b * E * L + e], gf1w[s]);
        if (tid == 0) gf1b += dp;
        __syncthreads();
        {   // d M[e][l] = sum_{o, j} W2[o][e][j] dM2[o][l - j + pad]
            f32x4t acc[2] = {(f32x4t){0.f, 0.f, 0.f, 0.f}, (f32x4t){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int s4 = 0; s4 < 24; ++s4) {
                const float bv = d2[doff[s4] + l];
                acc[0] = rg_mfma(w2d[0][s4], bv, acc[0]);
                acc[1] = rg_mfma(w2d[1][s4], bv, acc[1]);
            }
            if (l < L) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) dMs[(16 * i + 4 * kq + r) * XP + l] = acc[i][r];
                    *reinterpret_cast<float4*>(ws + g.w_dM + (b * L + l) * E + 16 * i + 4 * kq) =
                        make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);          // [sample][step][E]: d (LSTM output)
                }
            }
        }
        // d W2[o][(e, j)] += sum_l dM2[o][l] M[e][l + j - pad]
        for (int s4 = 0; s4 < ksteps; ++s4) {
            const int k = 4 * s4 + kq;
            const float av = d2[(16 * wi + li) * P + D2O + k];
#pragma unroll
            for (int t = 0; t < 3; ++t) gw2[t] = rg_mfma(av, Mp[eoff[t] + k], gw2[t]);
        }
        RG_FOR_OWNED(E, o, s) {
            float v = 0.f;
            for (int ll = 0; ll < L; ++ll) v += d2[o * P + D2O + ll];
            gc2b[s] += v;
        }
        __syncthreads();
        // d W1[e][n] += sum_l dM[e][l] x[n][l]
        if (wave < 2)
            for (int s4 = 0; s4 < ksteps; ++s4) {
                const int k = 4 * s4 + kq;
                const float av = dMs[(16 * wave + li) * XP + k];
                gw1[0] = rg_mfma(av, xs[li * XP + k], gw1[0]);
                if (ntn > 1) gw1[1] = rg_mfma(av, xs[(16 + li) * XP + k], gw1[1]);
            }
        RG_FOR_OWNED(E, e, s) {
            float v = 0.f;
            for (int ll = 0; ll < L; ++ll) v += dMs[e * XP + ll];
            gc1b[s] += v;
        }
    }
    float* row = ws + g.w_partF + (int64_t)blockIdx.x * g.nF;
    const int base = g.o_c1w;
    if (wave < 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (li < N) row[g.o_c1w - base + (16 * wave + 4 * kq + r) * N + li] = gw1[0][r];
            if (16 + li < N) row[g.o_c1w - base + (16 * wave + 4 * kq + r) * N + 16 + li] = gw1[1][r];
        }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) row[g.o_c2w - base + (16 * wi + 4 * kq + r) * RF_KE + 16 * (wt0 + t) + li] = gw2[t][r];
    RG_FOR_OWNED(E, e, s) row[g.o_c1b - base + e] = gc1b[s];
    RG_FOR_OWNED(E, e, s) row[g.o_c2b - base + e] = gc2b[s];
    RG_FOR_OWNED(E * L, e, s) row[g.o_f1w - base + e] = gf1w[s];
    if (tid == 0) row[g.o_f1b - base] = gf1b;
}

// ---- adjacency backward ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RB) void rg_adj_bwd_kernel(RgGeom g, const float* __restrict__ x, const float* __restrict__ prm, float* __restrict__ ws) {
    __shared__ float xs[RG_MAXN * RG_MAXL], a1[RG_MAXN * RG_MAXN], a2[RG_MAXN * RG_MAXN], tt[RG_MAXN * RG_MAXN], dh[RG_MAXN * RG_MAXN],
        ds[RG_MAXN * RG_MAXN], du1[RG_MAXN * RG_MAXN], du2[RG_MAXN * RG_MAXN], dv[RG_MAXN], dd[RG_MAXN];
    const int N = g.N, L = g.L, tid = threadIdx.x;
    float gw1[RG_OWN], gw2[RG_OWN], gb1 = 0.f, gb2 = 0.f;
#pragma unroll
    for (int s = 0; s < RG_OWN; ++s) gw1[s] = gw2[s] = 0.f;
    for (int64_t b = blockIdx.x; b < g.B; b += gridDim.x) {
        for (int i = tid; i < N * L; i += RB) xs[i] = x[b * N * L + i];
        for (int i = tid; i < N * N; i += RB) {
            a1[i] = ws[g.w_A1 + b * N * N + i];
            a2[i] = ws[g.w_A2 + b * N * N + i];
            tt[i] = ws[g.w_T + b * N * N + i];
            float v = 0.f;                                       // the graphs that used this sample's adjacency: b, b + B, b + 2B, ...
            for (int j = 0; j < L; ++j) v += ws[g.w_dAg + (b + (int64_t)j * g.B) * N * N + i];
            dh[i] = v;
        }
        if (tid < N) dv[tid] = ws[g.w_dinv + b * N + tid];
        __syncthreads();
        // A_hat_ij = d_i At_ij d_j, d = rowsum(At)^-1/2
        if (tid < N) {
            float v = 0.f;
            for (int c = 0; c < N; ++c) {
                const float at_rc = fmaxf(tt[tid * N + c], 0.f) + (tid == c ? 1.f : 0.f);
                const float at_cr = fmaxf(tt[c * N + tid], 0.f) + (tid == c ? 1.f : 0.f);
                v = fmaf(dh[tid * N + c] * at_rc, dv[c], v);
                v = fmaf(dh[c * N + tid] * at_cr, dv[c], v);
            }
            dd[tid] = v * (-0.5f) * dv[tid] * dv[tid] * dv[tid];
        }
        __syncthreads();
        for (int i = tid; i < N * N; i += RB) {
            const int r = i / N, c = i % N;
            const float dA = dv[r] * dh[i] * dv[c] + dd[r];
            const float t = tt[i];
            ds[i] = t > 0.f ? dA * (1.f - t * t) * g.alpha : 0.f;
        }
        __syncthreads();
        // dA1 = dS A2 - dS^T A2;  dA2 = dS^T A1 - dS A1;  through tanh(alpha u)
        for (int i = tid; i < N * N; i += RB) {
            const int n = i / N, m = i % N;
            float v1 = 0.f, v2 = 0.f;
            for (int c = 0; c < N; ++c) {
                const float sd = ds[n * N + c] - ds[c * N + n];
                v1 = fmaf(sd, a2[c * N + m], v1);
                v2 = fmaf(-sd, a1[c * N + m], v2);
            }
            du1[i] = v1 * (1.f - a1[i] * a1[i]) * g.alpha;
            du2[i] = v2 * (1.f - a2[i] * a2[i]) * g.alpha;
        }
        __syncthreads();
        RG_FOR_OWNED(N * L, e, s) {
            const int m = e / L, l = e % L;
            float v1 = 0.f, v2 = 0.f;
            for (int n = 0; n < N; ++n) {
                v1 = fmaf(du1[n * N + m], xs[n * L + l], v1);
                v2 = fmaf(du2[n * N + m], xs[n * L + l], v2);
            }
            gw1[s] += v1;
            gw2[s] += v2;
        }
        if (tid < N) {
            float v1 = 0.f, v2 = 0.f;
            for (int n = 0; n < N; ++n) { v1 += du1[n * N + tid]; v2 += du2[n * N + tid]; }
            gb1 += v1;
            gb2 += v2;
        }
        __syncthreads();
    }
    float* row = ws + g.w_partA + (int64_t)blockIdx.x * g.nA;
    RG_FOR_OWNED(N * L, e, s) { row[g.o_t1w + e] = gw1[s]; row[g.o_t2w + e] = gw2[s]; }
    if (tid < N) { row[g.o_t1b + tid] = gb1; row[g.o_t2b + tid] = gb2; }
}

__global__ void rg_zero_kernel(float* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.f;
}

size_t rg_fusion_lds(const RgGeom& g, bool bwd) {
    const int LP = g.L + g.K - 1;
    size_t fl = (size_t)g.E * LP + (size_t)g.N * g.L + (size_t)g.E * g.E * g.K;
    fl += bwd ? (size_t)g.E * LP + (size_t)g.E * g.L : (size_t)2 * RB;
    return fl * sizeof(float);
}

}  // namespace

int64_t rgcnu_param_count(const rulgnn_rgcnu_shape* s) {
    RgGeom g;
    return rg_geometry(s, &g) == RULGNN_OK ? g.pcount : -1;
}

size_t rgcnu_workspace_bytes(const rulgnn_rgcnu_shape* s) {
    RgGeom g;
    return rg_geometry(s, &g) == RULGNN_OK ? (size_t)g.total * sizeof(float) : 0;
}

#define RG_RC(call)                        \
    do {                                   \
        const int rc_ = (call);            \
        if (rc_ != RULGNN_OK) return rc_;  \
    } while (0)

// mode bit 0: forward, bit 1: backward (after a forward with the same args / workspace)
int rgcnu_run(const rulgnn_rgcnu_shape* s, const rulgnn_rgcnu_args* a, int mode, hipStream_t st) {
    RgGeom g;
    RG_RC(rg_geometry(s, &g));
    if (a->workspace_bytes < (size_t)g.total * sizeof(float)) return RULGNN_EWORKSPACE;
    if (g.B == 0) return RULGNN_OK;
    float* ws = static_cast<float*>(a->workspace);
    const float* prm = a->params;
    const float p = a->training ? a->dropout_p : 0.f;
    uint32_t thr = 0;
    if (p > 0.f) {
        const uint64_t ti = (uint64_t)((double)p * 4294967296.0 + 0.5);
        thr = ti > 4294967295ull ? 4294967295u : (uint32_t)ti;
    }
    const float scale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    const uint32_t key = dropout_layer_key(a->seed, a->step, 0);
    const float inv_gb = 1.0f / (float)(a->global_batch > 0 ? a->global_batch : g.B);
    rulgnn_bilstm_shape ls{g.L, (int32_t)g.B, g.N, g.E};
    rulgnn_bilstm_args la{};
    la.x = ws + g.w_sp;
    la.w_ih[0] = prm + g.o_wih; la.w_hh[0] = prm + g.o_whh; la.b_ih[0] = prm + g.o_bih; la.b_hh[0] = prm + g.o_bhh;
    la.w_ih[1] = la.w_ih[0]; la.w_hh[1] = la.w_hh[0]; la.b_ih[1] = la.b_ih[0]; la.b_hh[1] = la.b_hh[0];      // unused (ndir = 1)
    la.out = ws + g.w_hseq;
    la.workspace = ws + g.w_lstm;
    la.workspace_bytes = bilstm_workspace_bytes(&ls);
    // the matrix-core SCL kernels give a graph to a wavefront: half the workgroups (and partial rows), twice the graphs in flight
    const bool scl_mx = g.N <= 32 && g.H == RG_MXH;
    const int scl_nt = g.N <= 16 ? 1 : 2;
    const int blocks = g.blocks, gblocks = scl_mx ? (g.gblocks + 1) / 2 : g.gblocks;
    const size_t lds_scl = sizeof(float) * ((size_t)g.N * g.N + 2 * g.N + 3 * (size_t)g.N * g.H + (size_t)g.H * (g.H + 1));
    const size_t lds_sclb = sizeof(float) * ((size_t)g.N * g.N + 4 * g.N + 5 * (size_t)g.N * g.H + (size_t)g.H * (g.H + 1));
    (void)hipGetLastError();
    if (mode & 1) {
        hipLaunchKernelGGL(rg_adj_kernel, dim3(blocks), dim3(RB), 0, st, g, a->x, prm, ws);
        if (scl_mx) {
            const size_t lm = rg_scl_mx_lds(scl_nt, false);
            auto go = [&](auto kernel) {
                static bool raised = false;                          // once per instantiation and process
                if (lm > 48 * 1024 && !raised) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lm);
                    raised = true;
                }
                hipLaunchKernelGGL(kernel, dim3((unsigned)gblocks), dim3(64 * RG_MXW), lm, st, g, a->x, prm, ws, key, thr, scale, a->sample_offset);
            };
            if (scl_nt == 1) go(rg_scl_mx_kernel<1>); else go(rg_scl_mx_kernel<2>);
        } else
        hipLaunchKernelGGL(rg_scl_kernel, dim3((unsigned)gblocks), dim3(RB), lds_scl, st, g, a->x, prm, ws, key, thr, scale, a->sample_offset);
        if (hipGetLastError() != hipSuccess) return RULGNN_EHIP;
        RG_RC(bilstm_forward(&ls, &la, st, 1));
        const size_t lds = rg_fusion_lds(g, false);
        if (lds > 160 * 1024) return RULGNN_EUNSUPPORTED;
        if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(rg_fusion_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)lds) != hipSuccess)
            return RULGNN_EHIP;
        if (rg_fusion_mx_ok(g))
            hipLaunchKernelGGL(rg_fusion_mx_kernel, dim3(blocks), dim3(RB), 0, st, g, a->x, a->y, prm, ws, a->pred, a->std_pred, inv_gb);
        else
        hipLaunchKernelGGL(rg_fusion_kernel, dim3(blocks), dim3(RB), lds, st, g, a->x, a->y, prm, ws, a->pred, a->std_pred, inv_gb);
        if (a->y && a->loss) (void)block_sum((const float*)(ws + g.w_sq), g.B, a->loss, st);
        if (hipGetLastError() != hipSuccess) return RULGNN_EHIP;
    }
    if (mode & 2) {
        if (!a->grads) return RULGNN_EINVAL;
        const size_t lds = rg_fusion_lds(g, true);
        if (lds > 160 * 1024) return RULGNN_EUNSUPPORTED;
        if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(rg_fusion_bwd_kernel),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return RULGNN_EHIP;
        if (rg_fusion_mx_ok(g))
            hipLaunchKernelGGL(rg_fusion_bwd_mx_kernel, dim3(blocks), dim3(RB), 0, st, g, a->x, a->dpred, prm, ws);
        else
        hipLaunchKernelGGL(rg_fusion_bwd_kernel, dim3(blocks), dim3(RB), lds, st, g, a->x, a->dpred, prm, ws);
        if (hipGetLastError() != hipSuccess) return RULGNN_EHIP;
        la.dout = ws + g.w_dM;
        la.dx = ws + g.w_dsp;
        la.dw_ih[0] = a->grads + g.o_wih; la.dw_hh[0] = a->grads + g.o_whh; la.db_ih[0] = a->grads + g.o_bih; la.db_hh[0] = a->grads + g.o_bhh;
        la.dw_ih[1] = la.dw_ih[0]; la.dw_hh[1] = la.dw_hh[0]; la.db_ih[1] = la.db_ih[0]; la.db_hh[1] = la.db_hh[0];
        RG_RC(bilstm_backward(&ls, &la, st, 1));
        if (lds_sclb > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(rg_scl_bwd_kernel),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sclb) != hipSuccess)
            return RULGNN_EHIP;
        if (scl_mx) {
            const size_t lm = rg_scl_mx_lds(scl_nt, true);
            auto go = [&](auto kernel) {
                static bool raised = false;                          // once per instantiation and process
                if (lm > 48 * 1024 && !raised) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lm);
                    raised = true;
                }
                hipLaunchKernelGGL(kernel, dim3((unsigned)gblocks), dim3(64 * RG_MXW), lm, st, g, a->x, prm, ws, key, thr, scale, a->sample_offset);
            };
            if (scl_nt == 1) go(rg_scl_bwd_mx_kernel<1>); else go(rg_scl_bwd_mx_kernel<2>);
        } else
        hipLaunchKernelGGL(rg_scl_bwd_kernel, dim3((unsigned)gblocks), dim3(RB), lds_sclb, st, g, a->x, prm, ws, key, thr, scale, a->sample_offset);
        hipLaunchKernelGGL(rg_adj_bwd_kernel, dim3(blocks), dim3(RB), 0, st, g, a->x, prm, ws);
        if (hipGetLastError() != hipSuccess) return RULGNN_EHIP;
        {   // the three groups' partial rows in one launch
            const float* const part[3] = {ws + g.w_partA, ws + g.w_partS, ws + g.w_partF};
            float* const out[3] = {a->grads, a->grads + g.o_g1w, a->grads + g.o_c1w};
            const int rows[3] = {blocks, gblocks, blocks}, n[3] = {g.nA, g.nS, g.nF};
            const int64_t ld[3] = {g.nA, g.nS, g.nF};
            RG_RC(rows_sum_three(part, out, rows, ld, n, st));
        }
        const int nz = g.E * g.L + 1;                                    // the `std` head is not in the loss (algorithms.py:287-290)
        hipLaunchKernelGGL(rg_zero_kernel, dim3((nz + 255) / 256), dim3(256), 0, st, a->grads + g.o_f2w, nz);
        if (hipGetLastError() != hipSuccess) return RULGNN_EHIP;
    }
    return RULGNN_OK;
}

}  // namespace rulgnn
