// STAGNN on gfx950 (the last of the GCNLayer users).
// Reference path replaced: STAGNN_model.forward -- models/STAGNN/Model.py:184-230 (GCNLayer :8-22, GraphAttentionLayer :26-61, GAT
// :63-74, TemporalConvNet :85-159, MultiHeadTemporalEncoder :163-180) -- and STAGNN.update, algorithms/algorithms.py:314-323.
//
//   x [bs, N, L] -> adj = (covariance of the sensor rows > threshold) -> gcn1 (L -> h, leaky 0.01) -> gat1 (mean of `heads`
//   attention layers: softmax over ALL neighbours, then masked by adj) -> gcn2 -> gat2 -> [N, h] read as N channels of length T = h
//   -> tcn1 (causal k = 2 convolutions, dilation 1 and 2, each + BatchNorm1d + ReLU, residuals; N -> h -> h channels) -> temporal
//   encoder 1 (softmax over the length of sigmoid(Linear over the channels), mean over the heads, x scaled) -> tcn2 (h -> out -> out)
//   -> temporal encoder 2 -> Linear(out * h -> 1).
//
// A sample is a few [<= 64, <= 64] matrices: one workgroup per sample and stage, everything of the sample in LDS.  The four train-mode
// BatchNorms need the statistics of the whole batch before they can be applied, so the model is cut at them: graph | conv1 | BN,
// residual, conv2 | BN, residual, encoder (| the same three for tcn2, the last one with the head) forward, and the mirror image
// backward (each BatchNorm backward needs two more batch-wide sums).  Per-workgroup fp64 partial sums are combined by every consumer
// in a fixed order; weight gradients accumulate in per-workgroup rows (each slot owned by one thread) that one kernel adds in a fixed
// order: the step is deterministic.  Eval mode normalises with the running statistics (no cut needed, same kernels).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sgemm_mfma.hpp"
#include "stgcn_host.hpp"

namespace rulgnn {

namespace {

constexpr int TB = 256;
constexpr int TG_MAXN = 32, TG_MAXL = 128, TG_MAXH = 64, TG_MAXOUT = 16, TG_MAXHEADS = 4, TG_CMAX = 64;
constexpr int TG_WREG = 2 * TG_MAXH * TG_MAXH / TB;      // convolution-weight gradients a thread owns (<= 64 x 64 x 2 over TB threads)
constexpr float TG_BN_EPS = 1e-5f, TG_GCN_SLOPE = 0.01f, TG_GAT_SLOPE = 0.1f, TG_BN_MOMENTUM = 0.1f;

struct TgGeom {
    int64_t B;
    int N, L, h, out, heads, T, nblk;
    float thr;
    int Ci[2], Co[2];
    // flat parameter offsets (reference named_parameters() order, live parameters only)
    int o_gcn_w[2], o_gcn_b[2], o_gat_w[2][TG_MAXHEADS], o_gat_b[2][TG_MAXHEADS], o_gat_a[2][TG_MAXHEADS], o_gat_ab[2][TG_MAXHEADS];
    int o_ds_w[2], o_ds_b[2], o_c1_w[2], o_bn_g[4], o_bn_b[4], o_c2_w[2], o_enc_w[2][TG_MAXHEADS], o_enc_b[2][TG_MAXHEADS], o_fc_w, o_fc_b, pcount;
    int bn_off[4], bn_total;                    // running_mean at bn_off[k], running_var at bn_off[k] + channels
    // workspace (float offsets; per-sample arrays are [B][stride])
    int64_t w_adj, w_ahat, w_ax1, w_pre1, w_wh[2], w_gpre[2], w_att[2], w_g1, w_ax2, w_pre2, w_G;
    int64_t w_z1[2], w_o0[2], w_z2[2], w_o1[2], w_es[2], w_ew[2], w_e[2];
    int64_t w_dy2[2], w_dres[2], w_dy1[2], w_dxin[2], w_dpred, w_sq, w_bnpart, w_dbnpart, w_gpart, total;
};

int tg_geometry(const rulgnn_stagnn_shape* s, TgGeom* g) {
    if (!s) return RULGNN_EINVAL;
    if (s->batch < 0 || s->num_nodes < 1 || s->time_length < 2 || s->hidden_dim < 1 || s->output_dim < 1 || s->num_heads < 1) return RULGNN_EINVAL;
    if (s->num_nodes > TG_MAXN || s->time_length > TG_MAXL || s->hidden_dim > TG_MAXH || s->output_dim > TG_MAXOUT || s->num_heads > TG_MAXHEADS)
        return RULGNN_EUNSUPPORTED;
    // the residual branches: downsample0 exists when the channel counts differ (Model.py:104); downsample1 never (both widths equal)
    if (s->num_nodes == s->hidden_dim || s->hidden_dim == s->output_dim) return RULGNN_EUNSUPPORTED;
    if (s->hidden_dim < 3) return RULGNN_EUNSUPPORTED;          // the dilation-2 convolution runs along a length of hidden_dim
    g->B = s->batch; g->N = s->num_nodes; g->L = s->time_length; g->h = s->hidden_dim; g->out = s->output_dim; g->heads = s->num_heads;
    g->T = g->h; g->thr = s->threshold;
    g->Ci[0] = g->N; g->Co[0] = g->h; g->Ci[1] = g->h; g->Co[1] = g->out;
    g->nblk = (int)(g->B < 256 ? (g->B > 0 ? g->B : 1) : 256);
    const int N = g->N, L = g->L, h = g->h, T = g->T, Hd = g->heads;
    int o = 0;
    for (int l = 0; l < 2; ++l) {
        g->o_gcn_w[l] = o; o += h * (l == 0 ? L : h);
        g->o_gcn_b[l] = o; o += h;
        for (int i = 0; i < Hd; ++i) {
            g->o_gat_w[l][i] = o; o += h * h;
            g->o_gat_b[l][i] = o; o += h;
            g->o_gat_a[l][i] = o; o += 2 * h;
            g->o_gat_ab[l][i] = o; o += 1;
        }
    }
    int bo = 0;
    for (int l = 0; l < 2; ++l) {
        const int Ci = g->Ci[l], Co = g->Co[l];
        g->o_ds_w[l] = o; o += Co * Ci;
        g->o_ds_b[l] = o; o += Co;
        g->o_c1_w[l] = o; o += Co * Ci * 2;
        g->o_bn_g[2 * l] = o; o += Co;
        g->o_bn_b[2 * l] = o; o += Co;
        g->o_c2_w[l] = o; o += Co * Co * 2;
        g->o_bn_g[2 * l + 1] = o; o += Co;
        g->o_bn_b[2 * l + 1] = o; o += Co;
        for (int i = 0; i < Hd; ++i) {
            g->o_enc_w[l][i] = o; o += Co;
            g->o_enc_b[l][i] = o; o += 1;
        }
        g->bn_off[2 * l] = bo; bo += 2 * Co;
        g->bn_off[2 * l + 1] = bo; bo += 2 * Co;
    }
    g->o_fc_w = o; o += g->out * h;
    g->o_fc_b = o; o += 1;
    g->pcount = o;
    g->bn_total = bo;
    int64_t w = 0;
    const int64_t B = g->B;
    auto take = [&w](int64_t nfl) { const int64_t at = w; w += (nfl + 63) & ~(int64_t)63; return at; };
    g->w_adj = take(B * N * N); g->w_ahat = take(B * N * N);
    g->w_ax1 = take(B * N * L); g->w_pre1 = take(B * N * h);
    for (int l = 0; l < 2; ++l) { g->w_wh[l] = take(B * Hd * N * h); g->w_gpre[l] = take(B * Hd * N * N); g->w_att[l] = take(B * Hd * N * N); }
    g->w_g1 = take(B * N * h); g->w_ax2 = take(B * N * h); g->w_pre2 = take(B * N * h); g->w_G = take(B * N * h);
    for (int l = 0; l < 2; ++l) {
        const int64_t ct = (int64_t)g->Co[l] * T;
        g->w_z1[l] = take(B * ct); g->w_o0[l] = take(B * ct); g->w_z2[l] = take(B * ct); g->w_o1[l] = take(B * ct);
        g->w_es[l] = take(B * Hd * T); g->w_ew[l] = take(B * Hd * T); g->w_e[l] = take(B * ct);
        g->w_dy2[l] = take(B * ct); g->w_dres[l] = take(B * ct); g->w_dy1[l] = take(B * ct); g->w_dxin[l] = take(B * g->Ci[l] * T);
    }
    g->w_dpred = take(B); g->w_sq = take(B);
    g->w_bnpart = take((int64_t)2 * 4 * g->nblk * 2 * TG_CMAX);          // doubles: two floats each
    g->w_dbnpart = take((int64_t)2 * 4 * g->nblk * 2 * TG_CMAX);
    g->w_gpart = take((int64_t)g->nblk * g->pcount);
    g->total = w;
    return RULGNN_OK;
}

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : slope * v; }

// ---- per-workgroup fp64 partial sums -> batch totals, by every workgroup in the same fixed order ----------------------------------
// part[b][0..CMAX) and part[b][CMAX..2 CMAX) hold two sums per channel of workgroup b.  The TB threads split the workgroups four ways
// (independent loads in flight instead of one dependent chain per channel), the four slices are added in a fixed order.
__device__ void reduce_partials(int Co, const double* __restrict__ part, int nblk, double* tot0, double* tot1) {
    __shared__ double slice[4][2 * TG_CMAX];
    const int c = threadIdx.x & (TG_CMAX - 1), q = threadIdx.x / TG_CMAX;
    __syncthreads();
    if (c < Co) {
        // eight partial rows (16 loads) requested per pass (it was two: one L2 round trip per pass, 32 passes at 256 workgroups -- a fifth of
        // the step; sixteen rows per pass measured no better); the running sums are added in a fixed order
        constexpr int U = 8;
        double a0[U], a1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) a0[u] = a1[u] = 0.0;
        int b = q;
        for (; b + 4 * (U - 1) < nblk; b += 4 * U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                a0[u] += part[(int64_t)(b + 4 * u) * 2 * TG_CMAX + c];
                a1[u] += part[(int64_t)(b + 4 * u) * 2 * TG_CMAX + TG_CMAX + c];
            }
        }
        for (; b < nblk; b += 4) { a0[0] += part[(int64_t)b * 2 * TG_CMAX + c]; a1[0] += part[(int64_t)b * 2 * TG_CMAX + TG_CMAX + c]; }
#pragma unroll
        for (int w = U / 2; w > 0; w >>= 1)
#pragma unroll
            for (int u = 0; u < w; ++u) { a0[u] += a0[u + w]; a1[u] += a1[u + w]; }
        const double s0 = a0[0], t0 = 0.0, s1 = a1[0], t1 = 0.0;
        slice[q][c] = s0 + t0;
        slice[q][TG_CMAX + c] = s1 + t1;
    }
    __syncthreads();
    if (threadIdx.x < Co) {
        const int k = threadIdx.x;
        *tot0 = (slice[0][k] + slice[1][k]) + (slice[2][k] + slice[3][k]);
        *tot1 = (slice[0][TG_CMAX + k] + slice[1][TG_CMAX + k]) + (slice[2][TG_CMAX + k] + slice[3][TG_CMAX + k]);
    }
}

// BatchNorm constants of one layer (batch statistics when training, the running ones otherwise)
__device__ void bn_consts(int Co, const double* __restrict__ part, int nblk, const float* __restrict__ running, int training, double cnt,
                          float* mu, float* istd) {
    double s = 0.0, q = 0.0;
    if (training) reduce_partials(Co, part, nblk, &s, &q);
    if ((int)threadIdx.x < Co) {
        const int c = threadIdx.x;
        double mean, var;
        if (training) {
            mean = s / cnt;
            var = q / cnt - mean * mean;
            if (var < 0.0) var = 0.0;
        } else {
            mean = running[c];
            var = running[Co + c];
        }
        mu[c] = (float)mean;
        istd[c] = (float)(1.0 / sqrt(var + (double)TG_BN_EPS));
    }
}

// mean over the batch of dy and of dy * xhat (the two sums of the BatchNorm backward)
__device__ void bn_bwd_means(int Co, const double* __restrict__ part, int nblk, double cnt, float* m1, float* m2) {
    double s = 0.0, q = 0.0;
    reduce_partials(Co, part, nblk, &s, &q);
    if ((int)threadIdx.x < Co) {
        m1[threadIdx.x] = (float)(s / cnt);
        m2[threadIdx.x] = (float)(q / cnt);
    }
}

// LDS rows get an odd stride wherever the lanes of a wavefront walk DIFFERENT rows at the same column (weight-gradient loops, the
// attention scores): a row stride of 64 floats puts all of them in one bank.
__device__ __forceinline__ int odd(int v) { return v | 1; }

// coalesced copy of a [rows][cols] matrix from global memory into LDS rows of stride ld (weights are staged once per use: read from
// global memory inside the inner loops, every product waited for its own L2 round trip)
__device__ __forceinline__ void stage(float* dst, const float* __restrict__ src, int rows, int cols, int ld) {
    for (int i = threadIdx.x; i < rows * cols; i += TB) dst[(i / cols) * ld + i % cols] = src[i];
}

// group-strided variant of stage(): the TB threads of one head group copy a [rows][cols] matrix into LDS rows of stride ld
__device__ __forceinline__ void stage_group(float* dst, const float* __restrict__ src, int rows, int cols, int ld, int gt) {
    for (int i = gt; i < rows * cols; i += TB) dst[(i / cols) * ld + i % cols] = src[i];
}

// floats of one head group's private LDS region (forward / backward) and the whole dynamic LDS of the two graph kernels
__host__ __device__ inline int tg_fwd_group_floats(int N, int h) {
    const int hp = h | 1, wb = h * hp > N * h ? h * hp : N * h;
    return N * hp + N * N + 2 * N + wb;
}
__host__ __device__ inline int tg_bwd_group_floats(int N, int h) {
    const int hp = h | 1;
    return N * hp + N * h + 2 * N * N + 2 * N + h * hp;
}

// ---- graph part, forward ---------------------------------------------------------------------------------------------------------
// The heads of an attention layer are independent until their mean: G of them run side by side, one group of TB threads each
// (blockDim = TB * G; with one head after the other the kernel was a chain of ~40 barrier-separated phases on one sample).
// LDS: X[N*LP] | adj[N*N] | ah[N*N] | AX[N*Lh] | H0[N*h] | H1[N*h] | Wg[h*odd(Lh)] | G x { Wh[N*hp] | att[N*N] | f1[N] | f2[N] | Wb }
__global__ __launch_bounds__(1024) void tg_graph_fwd_kernel(TgGeom g, int G, const float* __restrict__ x, const float* __restrict__ prm,
                                                            float* __restrict__ ws) {
    extern __shared__ float lds[];
    const int N = g.N, L = g.L, h = g.h, Hd = g.heads, tid = threadIdx.x, NT = blockDim.x;
    const int grp = tid / TB, gt = tid % TB;
    const int Lh = L > h ? L : h, LP = odd(L), hp = odd(h);
    float* X = lds;
    float* adj = X + N * LP;
    float* ah = adj + N * N;
    float* AX = ah + N * N;
    float* H0 = AX + N * Lh;
    float* H1 = H0 + N * h;
    float* Wg = H1 + N * h;
    float* groups = Wg + h * odd(Lh);
    const int gsz = tg_fwd_group_floats(N, h);
    float* Wh = groups + grp * gsz;
    float* att = Wh + N * hp;
    float* f1 = att + N * N;
    float* f2 = f1 + N;
    float* Wb = f2 + N;                        // the head's weights, then its output [N][h]
    const int wb_off = N * hp + N * N + 2 * N;
    const float ih = 1.0f / (float)Hd;
    for (int64_t b = blockIdx.x; b < g.B; b += gridDim.x) {
        __syncthreads();
        for (int i = tid; i < N * L; i += NT) X[(i / L) * LP + i % L] = x[b * N * L + i];
        __syncthreads();
        if (tid < N) {
            float s = 0.f;
#pragma unroll 8
            for (int l = 0; l < L; ++l) s += X[tid * LP + l];
            f1[tid] = s / (float)L;            // (group 0's f1 / f2 double as scratch here)
        }
        __syncthreads();
        for (int e = tid; e < N * N; e += NT) {
            const int i = e / N, j = e % N;
            float c = 0.f;
#pragma unroll 8
            for (int l = 0; l < L; ++l) c = fmaf(X[i * LP + l] - groups[N * hp + N * N + i], X[j * LP + l] - groups[N * hp + N * N + j], c);
            adj[e] = c / (float)(L - 1) > g.thr ? 1.f : 0.f;
        }
        __syncthreads();
        if (tid < N) {
            float d = 1.f;
#pragma unroll 8
            for (int j = 0; j < N; ++j) d += adj[tid * N + j];
            f2[tid] = 1.0f / sqrtf(d);
        }
        __syncthreads();
        for (int e = tid; e < N * N; e += NT) {
            const int i = e / N, j = e % N;
            const float* dinv = groups + N * hp + N * N + N;          // group 0's f2
            const float v = dinv[i] * (adj[e] + (i == j ? 1.f : 0.f)) * dinv[j];
            ah[e] = v;
            ws[g.w_adj + b * N * N + e] = adj[e];
            ws[g.w_ahat + b * N * N + e] = v;
        }
        __syncthreads();
        for (int layer = 0; layer < 2; ++layer) {
            const int K = layer == 0 ? L : h, KP = odd(K);          // input width of this GCN layer
            const float* in = layer == 0 ? X : H1;
            const int inld = layer == 0 ? LP : h;
            for (int i = tid; i < h * K; i += NT) Wg[(i / K) * KP + i % K] = prm[g.o_gcn_w[layer] + i];
            for (int e = tid; e < N * K; e += NT) {                   // AX = A_hat in
                const int i = e / K, k = e % K;
                float a = 0.f;
#pragma unroll 8
                for (int j = 0; j < N; ++j) a = fmaf(ah[i * N + j], in[j * inld + k], a);
                AX[e] = a;
                ws[(layer == 0 ? g.w_ax1 : g.w_ax2) + b * N * K + e] = a;
            }
            __syncthreads();
            for (int e = tid; e < N * h; e += NT) {
                const int i = e / h, o = e % h;
                float a = prm[g.o_gcn_b[layer] + o];
#pragma unroll 8
                for (int k = 0; k < K; ++k) a = fmaf(AX[i * K + k], Wg[o * KP + k], a);
                ws[(layer == 0 ? g.w_pre1 : g.w_pre2) + b * N * h + e] = a;
                H0[e] = lrelu(a, TG_GCN_SLOPE);
                H1[e] = 0.f;
            }
            __syncthreads();
            for (int hd0 = 0; hd0 < Hd; hd0 += G) {
                const int hd = hd0 + grp;
                const bool act = hd < Hd;
                const int hdc = act ? hd : 0;
                const float* av = prm + g.o_gat_a[layer][hdc];
                const int64_t at_wh = g.w_wh[layer] + (b * Hd + hdc) * N * h, at_nn = (b * Hd + hdc) * N * N;
                if (act) stage_group(Wb, prm + g.o_gat_w[layer][hdc], h, h, hp, gt);
                __syncthreads();
                if (act)
                    for (int e = gt; e < N * h; e += TB) {
                        const int i = e / h, o = e % h;
                        float a = prm[g.o_gat_b[layer][hdc] + o];
#pragma unroll 8
                        for (int k = 0; k < h; ++k) a = fmaf(H0[i * h + k], Wb[o * hp + k], a);
                        Wh[i * hp + o] = a;
                        ws[at_wh + e] = a;
                    }
                __syncthreads();
                if (act && gt < 2 * N) {
                    const int i = gt % N, half = gt / N;
                    float a = 0.f;
#pragma unroll 8
                    for (int o = 0; o < h; ++o) a = fmaf(av[half * h + o], Wh[i * hp + o], a);
                    (half ? f2 : f1)[i] = a;
                }
                __syncthreads();
                if (act && gt < N) {
                    const float ab = prm[g.o_gat_ab[layer][hdc]];
                    const int i = gt;
                    float m = -INFINITY;
#pragma unroll 8
                    for (int j = 0; j < N; ++j) {
                        const float pre = f1[i] + f2[j] + ab;
                        ws[g.w_gpre[layer] + at_nn + i * N + j] = pre;
                        m = fmaxf(m, lrelu(pre, TG_GAT_SLOPE));
                    }
                    float s = 0.f;
#pragma unroll 8
                    for (int j = 0; j < N; ++j) {
                        const float ev = expf(lrelu(f1[i] + f2[j] + ab, TG_GAT_SLOPE) - m);
                        att[i * N + j] = ev;
                        s += ev;
                    }
                    const float inv = 1.0f / s;
#pragma unroll 8
                    for (int j = 0; j < N; ++j) {
                        const float a = att[i * N + j] * inv;
                        att[i * N + j] = a;
                        ws[g.w_att[layer] + at_nn + i * N + j] = a;
                    }
                }
                __syncthreads();
                if (act)
                    for (int e = gt; e < N * h; e += TB) {             // the head's output over its (no longer needed) weights
                        const int i = e / h, o = e % h;
                        float a = 0.f;
#pragma unroll 8
                        for (int j = 0; j < N; ++j) a = fmaf(att[i * N + j] * adj[i * N + j], Wh[j * hp + o], a);
                        Wb[e] = a * ih;
                    }
                __syncthreads();
                for (int e = tid; e < N * h; e += NT) {                // mean over the heads, in head order
                    float a = H1[e];
                    for (int q = 0; q < G && hd0 + q < Hd; ++q) a += groups[q * gsz + wb_off + e];
                    H1[e] = a;
                }
                __syncthreads();
            }
            for (int e = tid; e < N * h; e += NT) ws[(layer == 0 ? g.w_g1 : g.w_G) + b * N * h + e] = H1[e];
            __syncthreads();
        }
    }
}

// ---- TCN stage 1: z1 = causal conv (dilation 1) of the input; BatchNorm partial sums -----------------------------------------------
// LDS: xin[Ci*TP] | z[Co*TP] | W[Co*Ci*2]
__global__ __launch_bounds__(TB) void tg_conv1_fwd_kernel(TgGeom g, int l, const float* __restrict__ xin_g, const float* __restrict__ prm,
                                                          float* __restrict__ ws) {
    extern __shared__ float lds[];
    const int Ci = g.Ci[l], Co = g.Co[l], T = g.T, TP = odd(T), tid = threadIdx.x;
    float* xin = lds;
    float* z = xin + Ci * TP;
    float* W = z + Co * TP;
    stage(W, prm + g.o_c1_w[l], 1, Co * Ci * 2, 0);
    double* part = reinterpret_cast<double*>(ws + g.w_bnpart) + ((int64_t)(2 * l) * g.nblk + blockIdx.x) * 2 * TG_CMAX;
    double s = 0.0, q = 0.0;                                       // thread c < Co: sums of channel c over this workgroup's samples
    for (int64_t b = blockIdx.x; b < g.B; b += gridDim.x) {
        __syncthreads();
        for (int i = tid; i < Ci * T; i += TB) xin[(i / T) * TP + i % T] = xin_g[b * Ci * T + i];
        __syncthreads();
        for (int e = tid; e < Co * T; e += TB) {
            const int c = e / T, t = e % T;
            float a = 0.f;
#pragma unroll 8
            for (int ci = 0; ci < Ci; ++ci) {
                a = fmaf(W[(c * Ci + ci) * 2 + 1], xin[ci * TP + t], a);
                if (t >= 1) a = fmaf(W[(c * Ci + ci) * 2], xin[ci * TP + t - 1], a);
            }
            z[c * TP + t] = a;
            ws[g.w_z1[l] + b * Co * T + e] = a;
        }
        __syncthreads();
        if (tid < Co)
#pragma unroll 8
            for (int t = 0; t < T; ++t) { const double v = z[tid * TP + t]; s += v; q += v * v; }
    }
    if (tid < Co) { part[tid] = s; part[TG_CMAX + tid] = q; }
}

// ---- TCN stage 2: BN1, ReLU, + residual (1x1 convolution of the input), ReLU -> out0; z2 = causal conv (dilation 2); BN partials --
// LDS: xin[Ci*TP] | o0[Co*TP] | z[Co*TP] | mu[Co] | istd[Co] | Wd[Co*Ci] | W2[Co*Co*2]
__global__ __launch_bounds__(TB) void tg_mid_fwd_kernel(TgGeom g, int l, const float* __restrict__ xin_g, const float* __restrict__ prm,
                                                        const float* __restrict__ bnstate, float* __restrict__ ws, int training) {
    extern __shared__ float lds[];
    const int Ci = g.Ci[l], Co = g.Co[l], T = g.T, TP = odd(T), tid = threadIdx.x;
    float* xin = lds;
    float* o0 = xin + Ci * TP;
    float* z = o0 + Co * TP;
    float* mu = z + Co * TP;
    float* istd = mu + Co;
    float* Wd = istd + Co;
    float* W2 = Wd + Co * Ci;
    const double cnt = (double)g.B * T;
    bn_consts(Co, reinterpret_cast<const double*>(ws + g.w_bnpart) + (int64_t)(2 * l) * g.nblk * 2 * TG_CMAX, g.nblk, bnstate + g.bn_off[2 * l], training,
              cnt, mu, istd);
    stage(Wd, prm + g.o_ds_w[l], 1, Co * Ci, 0);
    stage(W2, prm + g.o_c2_w[l], 1, Co * Co * 2, 0);
    const float* gam = prm + g.o_bn_g[2 * l];
    const float* bet = prm + g.o_bn_b[2 * l];
    double* part = reinterpret_cast<double*>(ws + g.w_bnpart) + ((int64_t)(2 * l + 1) * g.nblk + blockIdx.x) * 2 * TG_CMAX;
    double s = 0.0, q = 0.0;
    for (int64_t b = blockIdx.x; b < g.B; b += gridDim.x) {
        __syncthreads();
        for (int i = tid; i < Ci * T; i += TB) xin[(i / T) * TP + i % T] = xin_g[b * Ci * T + i];
        __syncthreads();
        for (int e = tid; e < Co * T; e += TB) {
            const int c = e / T, t = e % T;
            const float y1 = fmaf((ws[g.w_z1[l] + b * Co * T + e] - mu[c]) * istd[c], gam[c], bet[c]);
            float r = prm[g.o_ds_b[l] + c];
#pragma unroll 8
            for (int ci = 0; ci < Ci; ++ci) r = fmaf(Wd[c * Ci + ci], xin[ci * TP + t], r);
            const float v = fmaxf(fmaxf(y1, 0.f) + r, 0.f);
            o0[c * TP + t] = v;
            ws[g.w_o0[l] + b * Co * T + e] = v;
        }
        __syncthreads();
        for (int e = tid; e < Co * T; e += TB) {
            const int c = e / T, t = e % T;
            float a = 0.f;
#pragma unroll 8
            for (int ci = 0; ci < Co; ++ci) {
                a = fmaf(W2[(c * Co + ci) * 2 + 1], o0[ci * TP + t], a);
                if (t >= 2) a = fmaf(W2[(c * Co + ci) * 2], o0[ci * TP + t - 2], a);
            }
            z[c * TP + t] = a;
            ws[g.w_z2[l] + b * Co * T + e] = a;
        }
        __syncthreads();
        if (training && tid < Co)
#pragma unroll 8
            for (int t = 0; t < T; ++t) { const double v = z[tid * TP + t]; s += v; q += v * v; }
    }
    if (training && tid < Co) { part[tid] = s; part[TG_CMAX + tid] = q; }
}

// ---- TCN stage 2 on the fp32 matrix cores (Co = T = 64, Ci <= 16: tcn1 of the reference's 64-wide wirings) -------------------------
// The stage above is scalar multiply-adds with two LDS reads each (107 us at batch 256).  Here the 1x1 down-sampling convolution is the
// product [Co x Ci] x [Ci x T] and the dilated two-tap convolution [Co x 2 Co] x [2 Co x T] over the tile padded by two zero columns
// (tap 1 reads column t + 2, tap 0 column t), both as v_mfma_f32_16x16x4f32: wavefront w owns the 16 columns t = 16 w .. 16 w + 15 and
// all four row tiles; lane (kq, li) = (lane / 16, lane % 16) feeds A[m = li][k = 4 s + kq], B[k = 4 s + kq][n = li] and receives
// C[m = 4 kq + r][n = li].  W2 sits in LDS as [Co][2 Co] at pitch 129.
// LDS: xin[32][65] | o0p[64][67] | z[64][65] | mu[64] | istd[64] | W2s[64][129]
constexpr int TM_C = 64, TM_P = 65, TM_PP = 67, TM_WP = 129;
constexpr size_t TM_FWD_LDS = sizeof(float) * (32 * TM_P + TM_C * TM_PP + TM_C * TM_P + 2 * TM_C + TM_C * TM_WP);
constexpr size_t TM_BWD_LDS = sizeof(float) * (32 * TM_P + 2 * TM_C * TM_PP + TM_C * TM_P + 6 * TM_C + TM_C * TM_WP);

__host__ __device__ inline bool tg_mid_mx_ok(const TgGeom& g, int l) { return g.Co[l] == TM_C && g.T == TM_C && g.Ci[l] <= 32; }

__device__ __forceinline__ f32x4t tg_mfma(float a, float b, f32x4t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

__global__ __launch_bounds__(TB) void tg_mid_fwd_mx_kernel(TgGeom g, int l, const float* __restrict__ xin_g, const float* __restrict__ prm,
                                                           const float* __restrict__ bnstate, float* __restrict__ ws, int training) {
    extern __shared__ float lds[];
    constexpr int C = TM_C, T = TM_C, P = TM_P, PP = TM_PP, WP = TM_WP;
    const int Ci = g.Ci[l], tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    float* xin = lds;                     // [32][P], rows beyond Ci zero
    float* o0p = xin + 32 * P;            // [C][PP]: out0 at column 2 + t behind two zero columns
    float* z = o0p + C * PP;              // [C][P]
    float* mu = z + C * P;
    float* istd = mu + C;
    float* W2s = istd + C;                // [C][WP]: W2[c][(ci, k)]
    const double cnt = (double)g.B * T;
    bn_consts(C, reinterpret_cast<const double*>(ws + g.w_bnpart) + (int64_t)(2 * l) * g.nblk * 2 * TG_CMAX, g.nblk, bnstate + g.bn_off[2 * l], training,
              cnt, mu, istd);
    stage(W2s, prm + g.o_c2_w[l], C, 2 * C, WP);
    for (int i = tid; i < 32 * P; i += TB) xin[i] = 0.f;
    for (int i = tid; i < C * PP; i += TB) o0p[i] = 0.f;
    float wd[4][8];                        // A(m = c, k = ci) of the 1x1 convolution
    const int dsteps = Ci <= 16 ? 4 : 8;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) wd[i][s4] = 4 * s4 + kq < Ci ? prm[g.o_ds_w[l] + (16 * i + li) * Ci + 4 * s4 + kq] : 0.f;
    const float* gam = prm + g.o_bn_g[2 * l];
    const float* bet = prm + g.o_bn_b[2 * l];
    const float* dsb = prm + g.o_ds_b[l];
    double* part = reinterpret_cast<double*>(ws + g.w_bnpart) + ((int64_t)(2 * l + 1) * g.nblk + blockIdx.x) * 2 * TG_CMAX;
    double s = 0.0, q = 0.0;
    const int t = 16 * wave + li;
    for (int64_t b = blockIdx.x; b < g.B; b += gridDim.x) {
        __syncthreads();
        for (int i = tid; i < Ci * T; i += TB) xin[(i / T) * P + i % T] = xin_g[b * Ci * T + i];
        __syncthreads();
        {   // out0 = relu(relu(bn1(z1)) + conv1x1(x) + bias)
            f32x4t acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = (f32x4t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s4 = 0; s4 < 8; ++s4) {
                if (s4 < dsteps) {
                    const float bv = xin[(4 * s4 + kq) * P + t];
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = tg_mfma(wd[i][s4], bv, acc[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 16 * i + 4 * kq + r, e = c * T + t;
                    const float y1 = fmaf((ws[g.w_z1[l] + b * C * T + e] - mu[c]) * istd[c], gam[c], bet[c]);
                    const float v = fmaxf(fmaxf(y1, 0.f) + (acc[i][r] + dsb[c]), 0.f);
                    o0p[c * PP + 2 + t] = v;
                    ws[g.w_o0[l] + b * C * T + e] = v;
                }
        }
        __syncthreads();
        {   // z2 = conv2_dilation2(out0)
            f32x4t acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = (f32x4t){0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
            for (int s4 = 0; s4 < 32; ++s4) {
                const int kk = 4 * s4 + kq;
                const float bv = o0p[(kk >> 1) * PP + t + 2 * (kk & 1)];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = tg_mfma(W2s[(16 * i + li) * WP + kk], bv, acc[i]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 16 * i + 4 * kq + r;
                    z[c * P + t] = acc[i][r];
                    ws[g.w_z2[l] + b * C * T + c * T + t] = acc[i][r];
                }
        }
        __syncthreads();
        if (training && tid < C)
#pragma unroll 8
            for (int tt = 0; tt < T; ++tt) { const double v = z[tid * P + tt]; s += v; q += v * v; }
    }
    if (training && tid < C) { part[tid] = s; part[TG_CMAX + tid] = q; }
}

// ---- TCN stage 3: BN2, ReLU, + out0, ReLU -> out1; temporal encoder -> e; for the second TCN also the head and the loss terms -------
// LDS: o1[Co*TP] | u[heads*T] | m[T] | mu[Co] | istd[Co] | red[TB] | ew[heads*Co]
__global__ __launch_bounds__(TB) void tg_end_fwd_kernel(TgGeom g, int l, const float* __restrict__ prm, const float* __restrict__ bnstate,
                                                        float* __restrict__ ws, int training, const float* __restrict__ y, float* __restrict__ pred,
                                                        float inv_gb) {
    extern __shared__ float lds[];
    const int Co = g.Co[l], T = g.T, TP = odd(T), Hd = g.heads, tid = threadIdx.x;
    float* o1 = lds;
    float* u = o1 + Co * TP;
    float* m = u + Hd * T;
    float* mu = m + T;
    float* istd = mu + Co;
    float* red = istd + Co;
    float* ew = red + TB;
    const double cnt = (double)g.B * T;
    bn_consts(Co, reinterpret_cast<const double*>(ws + g.w_bnpart) + (int64_t)(2 * l + 1) * g.nblk * 2 * TG_CMAX, g.nblk, bnstate + g.bn_off[2 * l + 1],
              training, cnt, mu, istd);
    for (int e = tid; e < Hd * Co; e += TB) ew[e] = prm[g.o_enc_w[l][e / Co] + e % Co];
    const float* gam = prm + g.o_bn_g[2 * l + 1];
    const float* bet = prm + g.o_bn_b[2 * l + 1];
    for (int64_t b = blockIdx.x; b < g.B; b += gridDim.x) {
        __syncthreads();
        for (int e = tid; e < Co * T; e += TB) {
            const int c = e / T, t = e % T;
            const float y2 = fmaf((ws[g.w_z2[l] + b * Co * T + e] - mu[c]) * istd[c], gam[c], bet[c]);
            const float v = fmaxf(fmaxf(y2, 0.f) + ws[g.w_o0[l] + b * Co * T + e], 0.f);
            o1[c * TP + t] = v;
            ws[g.w_o1[l] + b * Co * T + e] = v;
        }
        __syncthreads();
        for (int e = tid; e < Hd * T; e += TB) {
            const int hd = e / T, t = e % T;
            float a = prm[g.o_enc_b[l][hd]];
#pragma unroll 8
            for (int c = 0; c < Co; ++c) a = fmaf(ew[hd * Co + c], o1[c * TP + t], a);
            const float sg = 1.0f / (1.0f + expf(-a));
            u[e] = sg;
            ws[g.w_es[l] + (b * Hd + hd) * T + t] = sg;
        }
        __syncthreads();
        {   // softmax over the T steps of a head: one wavefront per head, a lane per step (T <= 64) -- three threads walked it alone before
            const int hd = tid >> 6, lane = tid & 63;
            if (hd < Hd) {
                const float v = lane < T ? u[hd * T + lane] : -INFINITY;
                float mx = v;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
                const float ex = lane < T ? expf(v - mx) : 0.f;
                float sum = ex;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
                if (lane < T) {
                    const float w = ex * (1.0f / sum);
                    u[hd * T + lane] = w;
                    ws[g.w_ew[l] + (b * Hd + hd) * T + lane] = w;
                }
            }
        }
        __syncthreads();
        for (int t = tid; t < T; t += TB) {
            float a = 0.f;
            for (int hd = 0; hd < Hd; ++hd) a += u[hd * T + t];
            m[t] = a / (float)Hd;
        }
        __syncthreads();
        float acc = 0.f;
        for (int e = tid; e < Co * T; e += TB) {
            const float v = o1[(e / T) * TP + e % T] * m[e % T];
            ws[g.w_e[l] + b * Co * T + e] = v;
            if (l == 1) acc = fmaf(v, prm[g.o_fc_w + e], acc);
        }
        if (l == 1) {
            red[tid] = acc;
            __syncthreads();
            for (int k = TB / 2; k > 0; k >>= 1) {
                if (tid < k) red[tid] += red[tid + k];
                __syncthreads();
            }
            if (tid == 0) {
                const float pr = red[0] + prm[g.o_fc_b];
                pred[b] = pr;
                if (y) {
                    const float d = pr - y[b];
                    ws[g.w_sq + b] = d * d * inv_gb;
                    ws[g.w_dpred + b] = 2.f * d * inv_gb;
                }
            }
        }
    }
}

// running statistics after a train-mode forward (BatchNorm1d: momentum 0.1, unbiased variance); one workgroup of TB threads per layer
__global__ __launch_bounds__(TB) void tg_running_kernel(TgGeom g, const float* __restrict__ ws, float* __restrict__ bnstate) {
    const int k = blockIdx.x, Co = g.Co[k / 2];
    const double cnt = (double)g.B * g.T;
    double s = 0.0, q = 0.0;
    reduce_partials(Co, reinterpret_cast<const double*>(ws + g.w_bnpart) + (int64_t)k * g.nblk * 2 * TG_CMAX, g.nblk, &s, &q);
    if ((int)threadIdx.x < Co) {
        const int c = threadIdx.x;
        const double mean = s / cnt;
        double var = q / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const double unb = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
        float* rm = bnstate + g.bn_off[k];
        rm[c] = (float)((1.0 - TG_BN_MOMENTUM) * rm[c] + TG_BN_MOMENTUM * mean);
        rm[Co + c] = (float)((1.0 - TG_BN_MOMENTUM) * rm[Co + c] + TG_BN_MOMENTUM * unb);
    }
}

// ---- backward of stage 3: (head,) encoder, ReLU, residual split, ReLU of the BN2 branch; BN2-backward partial sums ----------------
// LDS: o1[Co*TP] | de[Co*TP] | s[heads*T] | w[heads*T] | dm[T] | du[heads*T] | mu[Co] | istd[Co] | dot[heads] | ew[heads*Co]
__global__ __launch_bounds__(TB) void tg_end_bwd_kernel(TgGeom g, int l, const float* __restrict__ prm, const float* __restrict__ bnstate,
                                                        float* __restrict__ ws, const float* __restrict__ dpred, const float* __restrict__ de_g) {
    extern __shared__ float lds[];
    const int Co = g.Co[l], T = g.T, TP = odd(T), Hd = g.heads, tid = threadIdx.x;
    float* o1 = lds;
    float* de = o1 + Co * TP;
    float* sg = de + Co * TP;
    float* wv = sg + Hd * T;
    float* dm = wv + Hd * T;
    float* du = dm + T;
    float* mu = du + Hd * T;
    float* istd = mu + Co;
    float* dot = istd + Co;
    float* ew = dot + Hd;
    const double cnt = (double)g.B * T;
    bn_consts(Co, reinterpret_cast<const double*>(ws + g.w_bnpart) + (int64_t)(2 * l + 1) * g.nblk * 2 * TG_CMAX, g.nblk, bnstate + g.bn_off[2 * l + 1], 1,
              cnt, mu, istd);
    for (int e = tid; e < Hd * Co; e += TB) ew[e] = prm[g.o_enc_w[l][e / Co] + e % Co];
    const float* gam = prm + g.o_bn_g[2 * l + 1];
    const float* bet = prm + g.o_bn_b[2 * l + 1];
    float* gp = ws + g.w_gpart + (int64_t)blockIdx.x * g.pcount;
    double* part = reinterpret_cast<double*>(ws + g.w_dbnpart) + ((int64_t)(2 * l + 1) * g.nblk + blockIdx.x) * 2 * TG_CMAX;
    double s1 = 0.0, s2 = 0.0;
    for (int64_t b = blockIdx.x; b < g.B; b += gridDim.x) {
        __syncthreads();
        for (int e = tid; e < Co * T; e += TB) {
            const int at = (e / T) * TP + e % T;
            o1[at] = ws[g.w_o1[l] + b * Co * T + e];
            if (l == 1) {
                const float dp = dpred[b];
                de[at] = dp * prm[g.o_fc_w + e];
                gp[g.o_fc_w + e] += dp * ws[g.w_e[l] + b * Co * T + e];
            } else {
                de[at] = de_g[b * Co * T + e];
            }
        }
        if (l == 1 && tid == 0) gp[g.o_fc_b] += dpred[b];
        for (int e = tid; e < Hd * T; e += TB) {
            sg[e] = ws[g.w_es[l] + b * Hd * T + e];
            wv[e] = ws[g.w_ew[l] + b * Hd * T + e];
        }
        __syncthreads();
        for (int t = tid; t < T; t += TB) {
            float a = 0.f;
#pragma unroll 8
            for (int c = 0; c < Co; ++c) a = fmaf(de[c * TP + t], o1[c * TP + t], a);
            dm[t] = a / (float)Hd;                               // d w_i[t], the same for every head
        }
        __syncthreads();
        if (tid < Hd) {
            float a = 0.f;
#pragma unroll 8
            for (int t = 0; t < T; ++t) a = fmaf(dm[t], wv[tid * T + t], a);
            dot[tid] = a;
        }
        __syncthreads();
        for (int e = tid; e < Hd * T; e += TB) {
            const int hd = e / T, t = e % T;
            const float s = sg[e];
            du[e] = wv[e] * (dm[t] - dot[hd]) * s * (1.f - s);
        }
        __syncthreads();
        // encoder parameter gradients: thread (hd, c) owns linears.hd.weight[c]; thread hd owns its bias
        for (int e = tid; e < Hd * Co; e += TB) {
            const int hd = e / Co, c = e % Co;
            float a = 0.f;
#pragma unroll 8
            for (int t = 0; t < T; ++t) a = fmaf(du[hd * T + t], o1[c * TP + t], a);
            gp[g.o_enc_w[l][hd] + c] += a;
        }
        if (tid < Hd) {
            float a = 0.f;
#pragma unroll 8
            for (int t = 0; t < T; ++t) a += du[tid * T + t];
            gp[g.o_enc_b[l][tid]] += a;
        }
        __syncthreads();
        for (int e = tid; e < Co * T; e += TB) {
            const int c = e / T, t = e % T, at = c * TP + t;
            float mt = 0.f, dx = 0.f;
            for (int hd = 0; hd < Hd; ++hd) {
                mt += wv[hd * T + t];
                dx = fmaf(du[hd * T + t], ew[hd * Co + c], dx);
            }
            dx = fmaf(de[at], mt / (float)Hd, dx);
            const float d = o1[at] > 0.f ? dx : 0.f;             // through the last ReLU
            const float xh = (ws[g.w_z2[l] + b * Co * T + e] - mu[c]) * istd[c];
            const float y2 = fmaf(xh, gam[c], bet[c]);
            const float dy = y2 > 0.f ? d : 0.f;
            ws[g.w_dres[l] + b * Co * T + e] = d;
            ws[g.w_dy2[l] + b * Co * T + e] = dy;
            de[at] = dy;                                         // (every thread rewrites only its own elements)
            o1[at] = dy * xh;
        }
        __syncthreads();
        if (tid < Co)
#pragma unroll 8
            for (int t = 0; t < T; ++t) { s1 += (double)de[tid * TP + t]; s2 += (double)o1[tid * TP + t]; }
    }
    if (tid < Co) { part[tid] = s1; part[TG_CMAX + tid] = s2; }
}

// ---- backward of stage 2: BN2 backward, conv2 backward, residual, ReLU(out0), 1x1 convolution backward, ReLU of the BN1 branch ------
// LDS: xin[Ci*TP] | o0[Co*TP] | dz[Co*TP] | d[Co*TP] | mu1,istd1,mu2,istd2,m1,m2 [6*Co] | Wd[Co*Ci] | W2[Co*Co*2]
__global__ __launch_bounds__(TB) void tg_mid_bwd_kernel(TgGeom g, int l, const float* __restrict__ xin_g, const float* __restrict__ prm,
                                                        const float* __restrict__ bnstate, float* __restrict__ ws) {
    extern __shared__ float lds[];
    const int Ci = g.Ci[l], Co = g.Co[l], T = g.T, TP = odd(T), tid = threadIdx.x;
    float* xin = lds;
    float* o0 = xin + Ci * TP;
    float* dz = o0 + Co * TP;
    float* d = dz + Co * TP;
    float* mu1 = d + Co * TP;
    float* istd1 = mu1 + Co;
    float* mu2 = istd1 + Co;
    float* istd2 = mu2 + Co;
    float* m1 = istd2 + Co;
    float* m2 = m1 + Co;
    float* Wd = m2 + Co;
    float* W2 = Wd + Co * Ci;
    const double cnt = (double)g.B * T;
    const double* bp = reinterpret_cast<const double*>(ws + g.w_bnpart);
    bn_consts(Co, bp + (int64_t)(2 * l) * g.nblk * 2 * TG_CMAX, g.nblk, bnstate + g.bn_off[2 * l], 1, cnt, mu1, istd1);
    bn_consts(Co, bp + (int64_t)(2 * l + 1) * g.nblk * 2 * TG_CMAX, g.nblk, bnstate + g.bn_off[2 * l + 1], 1, cnt, mu2, istd2);
    bn_bwd_means(Co, reinterpret_cast<const double*>(ws + g.w_dbnpart) + (int64_t)(2 * l + 1) * g.nblk * 2 * TG_CMAX, g.nblk, cnt, m1, m2);
    stage(Wd, prm + g.o_ds_w[l], 1, Co * Ci, 0);
    stage(W2, prm + g.o_c2_w[l], 1, Co * Co * 2, 0);
    __syncthreads();
    const float* gam1 = prm + g.o_bn_g[2 * l];
    const float* bet1 = prm + g.o_bn_b[2 * l];
    const float* gam2 = prm + g.o_bn_g[2 * l + 1];
    float* gp = ws + g.w_gpart + (int64_t)blockIdx.x * g.pcount;
    if (blockIdx.x == 0)                                         // BatchNorm affine gradients are the batch sums themselves
        for (int c = tid; c < Co; c += TB) {
            gp[g.o_bn_g[2 * l + 1] + c] += (float)((double)m2[c] * cnt);
            gp[g.o_bn_b[2 * l + 1] + c] += (float)((double)m1[c] * cnt);
        }
    double* part = reinterpret_cast<double*>(ws + g.w_dbnpart) + ((int64_t)(2 * l) * g.nblk + blockIdx.x) * 2 * TG_CMAX;
    double s1 = 0.0, s2 = 0.0;
    float gw2[TG_WREG], gwd[TG_WREG / 2];
#pragma unroll
    for (int q = 0; q < TG_WREG; ++q) gw2[q] = 0.f;
#pragma unroll
    for (int q = 0; q < TG_WREG / 2; ++q) gwd[q] = 0.f;
    for (int64_t b = blockIdx.x; b < g.B; b += gridDim.x) {
        __syncthreads();
        for (int i = tid; i < Ci * T; i += TB) xin[(i / T) * TP + i % T] = xin_g[b * Ci * T + i];
        for (int e = tid; e < Co * T; e += TB) {
            const int c = e / T, at = c * TP + e % T;
            o0[at] = ws[g.w_o0[l] + b * Co * T + e];
            const float xh = (ws[g.w_z2[l] + b * Co * T + e] - mu2[c]) * istd2[c];
            dz[at] = gam2[c] * istd2[c] * (ws[g.w_dy2[l] + b * Co * T + e] - m1[c] - xh * m2[c]);
        }
        __syncthreads();
        // conv2 weight gradient: thread owns (c, ci, k); summed in registers over the workgroup's samples (a read-modify-write of
        // the partial row per sample was a chain of dependent global round trips: 32 per thread)
#pragma unroll
        for (int q = 0; q < TG_WREG; ++q) {
            const int e = tid + q * TB;
            if (e < Co * Co * 2) {
                const int k = e & 1, ci = (e >> 1) % Co, c = (e >> 1) / Co;
                const int sh = k ? 0 : 2;
                float a = 0.f;
#pragma unroll 8
                for (int t = sh; t < T; ++t) a = fmaf(dz[c * TP + t], o0[ci * TP + t - sh], a);
                gw2[q] += a;
            }
        }
        // d out0 = residual path + conv2 backward; through ReLU(out0)
        for (int e = tid; e < Co * T; e += TB) {
            const int ci = e / T, t = e % T;
            float a = ws[g.w_dres[l] + b * Co * T + e];
#pragma unroll 8
            for (int c = 0; c < Co; ++c) {
                a = fmaf(W2[(c * Co + ci) * 2 + 1], dz[c * TP + t], a);
                if (t + 2 < T) a = fmaf(W2[(c * Co + ci) * 2], dz[c * TP + t + 2], a);
            }
            d[ci * TP + t] = o0[ci * TP + t] > 0.f ? a : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < TG_WREG / 2; ++q) {
            const int e = tid + q * TB;
            if (e < Co * Ci) {
                const int c = e / Ci, ci = e % Ci;
                float a = 0.f;
#pragma unroll 8
                for (int t = 0; t < T; ++t) a = fmaf(d[c * TP + t], xin[ci * TP + t], a);
                gwd[q] += a;
            }
        }
        if (tid < Co) {
            float a = 0.f;
#pragma unroll 8
            for (int t = 0; t < T; ++t) a += d[tid * TP + t];
            gp[g.o_ds_b[l] + tid] += a;
        }
        for (int e = tid; e < Ci * T; e += TB) {
            const int ci = e / T, t = e % T;
            float a = 0.f;
#pragma unroll 8
            for (int c = 0; c < Co; ++c) a = fmaf(Wd[c * Ci + ci], d[c * TP + t], a);
            ws[g.w_dxin[l] + b * Ci * T + e] = a;
        }
        __syncthreads();
        for (int e = tid; e < Co * T; e += TB) {
            const int c = e / T, at = c * TP + e % T;
            const float xh = (ws[g.w_z1[l] + b * Co * T + e] - mu1[c]) * istd1[c];
            const float y1 = fmaf(xh, gam1[c], bet1[c]);
            const float dy = y1 > 0.f ? d[at] : 0.f;
            ws[g.w_dy1[l] + b * Co * T + e] = dy;
            dz[at] = dy;
            o0[at] = dy * xh;
        }
        __syncthreads();
        if (tid < Co)
#pragma unroll 8
            for (int t = 0; t < T; ++t) { s1 += (double)dz[tid * TP + t]; s2 += (double)o0[tid * TP + t]; }
    }
    if (tid < Co) { part[tid] = s1; part[TG_CMAX + tid] = s2; }
#pragma unroll
    for (int q = 0; q < TG_WREG; ++q)
        if (tid + q * TB < Co * Co * 2) gp[g.o_c2_w[l] + tid + q * TB] = gw2[q];
#pragma unroll
    for (int q = 0; q < TG_WREG / 2; ++q)
        if (tid + q * TB < Co * Ci) gp[g.o_ds_w[l] + tid + q * TB] = gwd[q];
}

// ---- backward of TCN stage 2 on the fp32 matrix cores (same shapes as tg_mid_fwd_mx_kernel) ------------------------------------------
// d W2 = dz [Co x T] x out0-taps [T x 2 Co] (32 tiles of 16 x 16, eight per wavefront, accumulated in registers over the workgroup's
// samples), d out0 = W2^T-taps [Co x 2 Co] x dz-taps [2 Co x T], d Wd = d [Co x T] x x^T [T x Ci], d x = Wd^T [Ci x Co] x d [Co x T].
// LDS: xin[32][65] | o0p[64][67] | dzp[64][67] (two zero columns behind) | d[64][65] | six [64] constant vectors | W2T[64][129] = W2[c][ci][k] at [ci][(c, k)]
__global__ __launch_bounds__(TB) void tg_mid_bwd_mx_kernel(TgGeom g, int l, const float* __restrict__ xin_g, const float* __restrict__ prm,
                                                           const float* __restrict__ bnstate, float* __restrict__ ws) {
    extern __shared__ float lds[];
    constexpr int C = TM_C, T = TM_C, P = TM_P, PP = TM_PP, WP = TM_WP;
    const int Ci = g.Ci[l], tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    float* xin = lds;
    float* o0p = xin + 32 * P;
    float* dzp = o0p + C * PP;
    float* d = dzp + C * PP;
    float* mu1 = d + C * P;
    float* istd1 = mu1 + C;
    float* mu2 = istd1 + C;
    float* istd2 = mu2 + C;
    float* m1 = istd2 + C;
    float* m2 = m1 + C;
    float* W2T = m2 + C;
    const double cnt = (double)g.B * T;
    const double* bp = reinterpret_cast<const double*>(ws + g.w_bnpart);
    bn_consts(C, bp + (int64_t)(2 * l) * g.nblk * 2 * TG_CMAX, g.nblk, bnstate + g.bn_off[2 * l], 1, cnt, mu1, istd1);
    bn_consts(C, bp + (int64_t)(2 * l + 1) * g.nblk * 2 * TG_CMAX, g.nblk, bnstate + g.bn_off[2 * l + 1], 1, cnt, mu2, istd2);
    bn_bwd_means(C, reinterpret_cast<const double*>(ws + g.w_dbnpart) + (int64_t)(2 * l + 1) * g.nblk * 2 * TG_CMAX, g.nblk, cnt, m1, m2);
    for (int i = tid; i < C * C * 2; i += TB) {
        const int k = i & 1, ci = (i >> 1) % C, c = (i >> 1) / C;
        W2T[ci * WP + 2 * c + k] = prm[g.o_c2_w[l] + i];
    }
    for (int i = tid; i < 32 * P; i += TB) xin[i] = 0.f;
    for (int i = tid; i < C * PP; i += TB) { o0p[i] = 0.f; dzp[i] = 0.f; }
    const int cit = Ci <= 16 ? 1 : 2;      // 16-row tiles of the input channels
    float wdT[2][16];                      // A(m = ci, k = c) = Wd[c][ci]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s4 = 0; s4 < 16; ++s4) wdT[j][s4] = 16 * j + li < Ci ? prm[g.o_ds_w[l] + (4 * s4 + kq) * Ci + 16 * j + li] : 0.f;
    __syncthreads();
    const float* gam1 = prm + g.o_bn_g[2 * l];
    const float* bet1 = prm + g.o_bn_b[2 * l];
    const float* gam2 = prm + g.o_bn_g[2 * l + 1];
    float* gp = ws + g.w_gpart + (int64_t)blockIdx.x * g.pcount;
    if (blockIdx.x == 0)                                         // BatchNorm affine gradients are the batch sums themselves
        for (int c = tid; c < C; c += TB) {
            gp[g.o_bn_g[2 * l + 1] + c] += (float)((double)m2[c] * cnt);
            gp[g.o_bn_b[2 * l + 1] + c] += (float)((double)m1[c] * cnt);
        }
    double* part = reinterpret_cast<double*>(ws + g.w_dbnpart) + ((int64_t)(2 * l) * g.nblk + blockIdx.x) * 2 * TG_CMAX;
    double s1 = 0.0, s2 = 0.0;
    f32x4t gw2[8], gwd[2] = {(f32x4t){0.f, 0.f, 0.f, 0.f}, (f32x4t){0.f, 0.f, 0.f, 0.f}};      // rows c = 16 wave + ..: columns (ci, k) = 16 j + li | ci = 16 j + li
#pragma unroll
    for (int j = 0; j < 8; ++j) gw2[j] = (f32x4t){0.f, 0.f, 0.f, 0.f};
    const int t = 16 * wave + li;
    for (int64_t b = blockIdx.x; b < g.B; b += gridDim.x) {
        __syncthreads();
        for (int i = tid; i < Ci * T; i += TB) xin[(i / T) * P + i % T] = xin_g[b * Ci * T + i];
        for (int e = tid; e < C * T; e += TB) {
            const int c = e / T, tt = e % T;
            o0p[c * PP + 2 + tt] = ws[g.w_o0[l] + b * C * T + e];
            const float xh = (ws[g.w_z2[l] + b * C * T + e] - mu2[c]) * istd2[c];
            dzp[c * PP + tt] = gam2[c] * istd2[c] * (ws[g.w_dy2[l] + b * C * T + e] - m1[c] - xh * m2[c]);
        }
        __syncthreads();
        // d W2[c][(ci, k)] += sum_t dz[c][t] out0[ci][t - 2 (1 - k)]
#pragma unroll 4
        for (int s4 = 0; s4 < 16; ++s4) {
            const int tt = 4 * s4 + kq;
            const float av = dzp[(16 * wave + li) * PP + tt];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int n = 16 * j + li;
                gw2[j] = tg_mfma(av, o0p[(n >> 1) * PP + tt + 2 * (n & 1)], gw2[j]);
            }
        }
        {   // d out0[ci][t] = residual path + sum_{c, k} W2[c][ci][k] dz[c][t + 2 (1 - k)]; through ReLU(out0)
            f32x4t acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = (f32x4t){0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
            for (int s4 = 0; s4 < 32; ++s4) {
                const int kk = 4 * s4 + kq;
                const float bv = dzp[(kk >> 1) * PP + t + 2 * (1 - (kk & 1))];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = tg_mfma(W2T[(16 * i + li) * WP + kk], bv, acc[i]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ci = 16 * i + 4 * kq + r;
                    const float a = acc[i][r] + ws[g.w_dres[l] + b * C * T + ci * T + t];
                    d[ci * P + t] = o0p[ci * PP + 2 + t] > 0.f ? a : 0.f;
                }
        }
        __syncthreads();
        // d Wd[c][ci] += sum_t d[c][t] x[ci][t]
#pragma unroll 4
        for (int s4 = 0; s4 < 16; ++s4) {
            const int tt = 4 * s4 + kq;
            const float av = d[(16 * wave + li) * P + tt];
            gwd[0] = tg_mfma(av, xin[li * P + tt], gwd[0]);
            if (cit > 1) gwd[1] = tg_mfma(av, xin[(16 + li) * P + tt], gwd[1]);
        }
        if (tid < C) {
            float a = 0.f;
#pragma unroll 8
            for (int tt = 0; tt < T; ++tt) a += d[tid * P + tt];
            gp[g.o_ds_b[l] + tid] += a;
        }
        {   // d x[ci][t] = sum_c Wd[c][ci] d[c][t]
            f32x4t acc[2] = {(f32x4t){0.f, 0.f, 0.f, 0.f}, (f32x4t){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int s4 = 0; s4 < 16; ++s4) {
                const float bv = d[(4 * s4 + kq) * P + t];
                acc[0] = tg_mfma(wdT[0][s4], bv, acc[0]);
                if (cit > 1) acc[1] = tg_mfma(wdT[1][s4], bv, acc[1]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (16 * j + 4 * kq + r < Ci) ws[g.w_dxin[l] + b * Ci * T + (16 * j + 4 * kq + r) * T + t] = acc[j][r];
        }
        __syncthreads();
        for (int e = tid; e < C * T; e += TB) {
            const int c = e / T, tt = e % T;
            const float xh = (ws[g.w_z1[l] + b * C * T + e] - mu1[c]) * istd1[c];
            const float y1 = fmaf(xh, gam1[c], bet1[c]);
            const float dy = y1 > 0.f ? d[c * P + tt] : 0.f;
            ws[g.w_dy1[l] + b * C * T + e] = dy;
            dzp[c * PP + tt] = dy;
            o0p[c * PP + 2 + tt] = dy * xh;
        }
        __syncthreads();
        if (tid < C)
#pragma unroll 8
            for (int tt = 0; tt < T; ++tt) { s1 += (double)dzp[tid * PP + tt]; s2 += (double)o0p[tid * PP + 2 + tt]; }
    }
    if (tid < C) { part[tid] = s1; part[TG_CMAX + tid] = s2; }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) gp[g.o_c2_w[l] + (16 * wave + 4 * kq + r) * 2 * C + 16 * j + li] = gw2[j][r];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (li < Ci) gp[g.o_ds_w[l] + (16 * wave + 4 * kq + r) * Ci + li] = gwd[0][r];
        if (16 + li < Ci) gp[g.o_ds_w[l] + (16 * wave + 4 * kq + r) * Ci + 16 + li] = gwd[1][r];
    }
}

// ---- backward of stage 1: BN1 backward, conv1 backward -> gradient of the stage's input ---------------------------------------------
// LDS: xin[Ci*TP] | dz[Co*TP] | mu,istd,m1,m2 [4*Co] | W1[Co*Ci*2]
__global__ __launch_bounds__(TB) void tg_conv1_bwd_kernel(TgGeom g, int l, const float* __restrict__ xin_g, const float* __restrict__ prm,
                                                          const float* __restrict__ bnstate, float* __restrict__ ws, float* __restrict__ dxin_out) {
    extern __shared__ float lds[];
    const int Ci = g.Ci[l], Co = g.Co[l], T = g.T, TP = odd(T), tid = threadIdx.x;
    float* xin = lds;
    float* dz = xin + Ci * TP;
    float* mu = dz + Co * TP;
    float* istd = mu + Co;
    float* m1 = istd + Co;
    float* m2 = m1 + Co;
    float* W1 = m2 + Co;
    const double cnt = (double)g.B * T;
    bn_consts(Co, reinterpret_cast<const double*>(ws + g.w_bnpart) + (int64_t)(2 * l) * g.nblk * 2 * TG_CMAX, g.nblk, bnstate + g.bn_off[2 * l], 1, cnt, mu,
              istd);
    bn_bwd_means(Co, reinterpret_cast<const double*>(ws + g.w_dbnpart) + (int64_t)(2 * l) * g.nblk * 2 * TG_CMAX, g.nblk, cnt, m1, m2);
    stage(W1, prm + g.o_c1_w[l], 1, Co * Ci * 2, 0);
    __syncthreads();
    const float* gam = prm + g.o_bn_g[2 * l];
    float* gp = ws + g.w_gpart + (int64_t)blockIdx.x * g.pcount;
    if (blockIdx.x == 0)
        for (int c = tid; c < Co; c += TB) {
            gp[g.o_bn_g[2 * l] + c] += (float)((double)m2[c] * cnt);
            gp[g.o_bn_b[2 * l] + c] += (float)((double)m1[c] * cnt);
        }
    float gw1[TG_WREG];
#pragma unroll
    for (int q = 0; q < TG_WREG; ++q) gw1[q] = 0.f;
    for (int64_t b = blockIdx.x; b < g.B; b += gridDim.x) {
        __syncthreads();
        for (int i = tid; i < Ci * T; i += TB) xin[(i / T) * TP + i % T] = xin_g[b * Ci * T + i];
        for (int e = tid; e < Co * T; e += TB) {
            const int c = e / T;
            const float xh = (ws[g.w_z1[l] + b * Co * T + e] - mu[c]) * istd[c];
            dz[c * TP + e % T] = gam[c] * istd[c] * (ws[g.w_dy1[l] + b * Co * T + e] - m1[c] - xh * m2[c]);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < TG_WREG; ++q) {
            const int e = tid + q * TB;
            if (e < Co * Ci * 2) {
                const int k = e & 1, ci = (e >> 1) % Ci, c = (e >> 1) / Ci;
                const int sh = k ? 0 : 1;
                float a = 0.f;
#pragma unroll 8
                for (int t = sh; t < T; ++t) a = fmaf(dz[c * TP + t], xin[ci * TP + t - sh], a);
                gw1[q] += a;
            }
        }
        for (int e = tid; e < Ci * T; e += TB) {
            const int ci = e / T, t = e % T;
            float a = ws[g.w_dxin[l] + b * Ci * T + e];
#pragma unroll 8
            for (int c = 0; c < Co; ++c) {
                a = fmaf(W1[(c * Ci + ci) * 2 + 1], dz[c * TP + t], a);
                if (t + 1 < T) a = fmaf(W1[(c * Ci + ci) * 2], dz[c * TP + t + 1], a);
            }
            dxin_out[b * Ci * T + e] = a;
        }
    }
#pragma unroll
    for (int q = 0; q < TG_WREG; ++q)
        if (tid + q * TB < Co * Ci * 2) gp[g.o_c1_w[l] + tid + q * TB] = gw1[q];
}


// ---- TCN stage 1 on the fp32 matrix cores (T = 64; any Ci, Co <= 64): z1 = conv1(x) as [Co x 2 Ci] x [2 Ci x T] over the tile with one zero
// column in front (tap 1 reads column t + 1, tap 0 column t).  LDS: xinp[CiP][67] | z[CoP][65] | W1s[CoP][WPF] = W1[c][(ci, k)], zero-padded
__host__ __device__ inline int tg_pad16(int v) { return (v + 15) / 16 * 16; }
__host__ __device__ inline int tg_c1_wpt(int Co) { return ((2 * Co + 3) / 4 * 4) | 1; }
__host__ __device__ inline bool tg_conv1_mx_ok(const TgGeom& g, int l) { return g.T == TM_C && g.Ci[l] <= 64 && g.Co[l] <= 64; }
__host__ __device__ inline int tg_c1_wpf(int Ci) { return ((2 * Ci + 3) / 4 * 4) | 1; }
inline size_t tg_conv1_fwd_mx_lds(const TgGeom& g, int l) {
    return sizeof(float) * ((size_t)tg_pad16(g.Ci[l]) * TM_PP + (size_t)tg_pad16(g.Co[l]) * TM_P + (size_t)tg_pad16(g.Co[l]) * tg_c1_wpf(g.Ci[l]));
}

__global__ __launch_bounds__(TB) void tg_conv1_fwd_mx_kernel(TgGeom g, int l, const float* __restrict__ xin_g, const float* __restrict__ prm,
                                                             float* __restrict__ ws) {
    extern __shared__ float lds[];
    constexpr int T = TM_C, P = TM_P, PP = TM_PP;
    const int Ci = g.Ci[l], Co = g.Co[l], CiP = tg_pad16(Ci), CoP = tg_pad16(Co), WPF = tg_c1_wpf(Ci), CT = CoP / 16, ksteps = (2 * Ci + 3) / 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    float* xinp = lds;
    float* z = xinp + CiP * PP;
    float* W1s = z + CoP * P;
    for (int i = tid; i < CiP * PP; i += TB) xinp[i] = 0.f;
    for (int i = tid; i < CoP * WPF; i += TB) {
        const int r = i / WPF, c = i - r * WPF;
        W1s[i] = r < Co && c < 2 * Ci ? prm[g.o_c1_w[l] + r * 2 * Ci + c] : 0.f;
    }
    double* part = reinterpret_cast<double*>(ws + g.w_bnpart) + ((int64_t)(2 * l) * g.nblk + blockIdx.x) * 2 * TG_CMAX;
    double s = 0.0, q = 0.0;                                       // thread c < Co: sums of channel c over this workgroup's samples
    const int t = 16 * wave + li;
    for (int64_t b = blockIdx.x; b < g.B; b += gridDim.x) {
        __syncthreads();
        for (int i = tid; i < Ci * T; i += TB) xinp[(i / T) * PP + 1 + i % T] = xin_g[b * Ci * T + i];
        __syncthreads();
        {
            f32x4t acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = (f32x4t){0.f, 0.f, 0.f, 0.f};
            for (int s4 = 0; s4 < ksteps; ++s4) {
                const int kk = 4 * s4 + kq;
                const float bv = xinp[(kk >> 1) * PP + t + (kk & 1)];          // rows beyond Ci meet zero weights (and exist: CiP >= Ci + 1 or 2 Ci % 4 == 0)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < CT) acc[i] = tg_mfma(W1s[(16 * i + li) * WPF + kk], bv, acc[i]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 16 * i + 4 * kq + r;
                    if (c < Co) {
                        z[c * P + t] = acc[i][r];
                        ws[g.w_z1[l] + b * Co * T + c * T + t] = acc[i][r];
                    }
                }
        }
        __syncthreads();
        if (tid < Co)
#pragma unroll 8
            for (int tt = 0; tt < T; ++tt) { const double v = z[tid * P + tt]; s += v; q += v * v; }
    }
    if (tid < Co) { part[tid] = s; part[TG_CMAX + tid] = q; }
}

// ---- backward of TCN stage 1 on the fp32 matrix cores (T = 64; any Ci, Co <= 64) ------------------------------------------------------
// d W1 = dz [Co x T] x x-taps [T x 2 Ci] (16 x 16 tiles over the wavefronts, accumulated in registers over the workgroup's samples),
// d x += W1^T-taps [Ci x 2 Co] x dz-taps [2 Co x T].  Operand rows are padded to 16 with zeros in LDS, so no product needs a guard.
// LDS: xinp[CiP][67] (x at column 1 + t) | dzp[CoP][67] (zero from column 64) | mu, istd, m1, m2 [4 Co] | W1T[CiP][WPT] = W1[c][ci][k] at [ci][(c, k)]
inline size_t tg_conv1_bwd_mx_lds(const TgGeom& g, int l) {
    return sizeof(float) * ((size_t)(tg_pad16(g.Ci[l]) + tg_pad16(g.Co[l])) * TM_PP + 4 * (size_t)g.Co[l] + (size_t)tg_pad16(g.Ci[l]) * tg_c1_wpt(g.Co[l]));
}

__global__ __launch_bounds__(TB) void tg_conv1_bwd_mx_kernel(TgGeom g, int l, const float* __restrict__ xin_g, const float* __restrict__ prm,
                                                             const float* __restrict__ bnstate, float* __restrict__ ws, float* __restrict__ dxin_out) {
    extern __shared__ float lds[];
    constexpr int T = TM_C, PP = TM_PP;
    const int Ci = g.Ci[l], Co = g.Co[l], CiP = tg_pad16(Ci), CoP = tg_pad16(Co), WPT = tg_c1_wpt(Co);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    float* xinp = lds;
    float* dzp = xinp + CiP * PP;
    float* mu = dzp + CoP * PP;
    float* istd = mu + Co;
    float* m1 = istd + Co;
    float* m2 = m1 + Co;
    float* W1T = m2 + Co;
    const double cnt = (double)g.B * T;
    bn_consts(Co, reinterpret_cast<const double*>(ws + g.w_bnpart) + (int64_t)(2 * l) * g.nblk * 2 * TG_CMAX, g.nblk, bnstate + g.bn_off[2 * l], 1, cnt, mu,
              istd);
    bn_bwd_means(Co, reinterpret_cast<const double*>(ws + g.w_dbnpart) + (int64_t)(2 * l) * g.nblk * 2 * TG_CMAX, g.nblk, cnt, m1, m2);
    for (int i = tid; i < CiP * WPT; i += TB) W1T[i] = 0.f;
    for (int i = tid; i < (CiP + CoP) * PP; i += TB) xinp[i] = 0.f;            // xinp and dzp are adjacent
    __syncthreads();
    for (int i = tid; i < Co * Ci * 2; i += TB) {
        const int k = i & 1, ci = (i >> 1) % Ci, c = (i >> 1) / Ci;
        W1T[ci * WPT + 2 * c + k] = prm[g.o_c1_w[l] + i];
    }
    const float* gam = prm + g.o_bn_g[2 * l];
    float* gp = ws + g.w_gpart + (int64_t)blockIdx.x * g.pcount;
    if (blockIdx.x == 0)
        for (int c = tid; c < Co; c += TB) {
            gp[g.o_bn_g[2 * l] + c] += (float)((double)m2[c] * cnt);
            gp[g.o_bn_b[2 * l] + c] += (float)((double)m1[c] * cnt);
        }
    // d W1 tiles: (row tile i of Co, column tile j of the 2 Ci (ci, k) pairs), tile q of this wavefront = wave + 4 q
    const int NT2 = (2 * Ci + 15) / 16, ntiles = (CoP / 16) * NT2, CIT = CiP / 16, ksteps = (2 * Co + 3) / 4;
    constexpr int MAXQ = 8;                 // 4 x 8 tiles at most (Co, Ci <= 64)
    f32x4t gw[MAXQ];
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) gw[q] = (f32x4t){0.f, 0.f, 0.f, 0.f};
    const int t = 16 * wave + li;
    for (int64_t b = blockIdx.x; b < g.B; b += gridDim.x) {
        __syncthreads();
        for (int i = tid; i < Ci * T; i += TB) xinp[(i / T) * PP + 1 + i % T] = xin_g[b * Ci * T + i];
        for (int e = tid; e < Co * T; e += TB) {
            const int c = e / T;
            const float xh = (ws[g.w_z1[l] + b * Co * T + e] - mu[c]) * istd[c];
            dzp[c * PP + e % T] = gam[c] * istd[c] * (ws[g.w_dy1[l] + b * Co * T + e] - m1[c] - xh * m2[c]);
        }
        __syncthreads();
        // d W1[c][(ci, k)] += sum_t dz[c][t] x[ci][t - (1 - k)]
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            const int idx = wave + 4 * q;
            if (idx < ntiles) {
                const int i = idx / NT2, n = 16 * (idx - i * NT2) + li;
                const float* ar = dzp + (16 * i + li) * PP + kq;
                const float* br = xinp + (n >> 1) * PP + (n & 1) + kq;
                f32x4t acc = gw[q];
#pragma unroll 4
                for (int s4 = 0; s4 < T / 4; ++s4) acc = tg_mfma(ar[4 * s4], br[4 * s4], acc);
                gw[q] = acc;
            }
        }
        {   // d x[ci][t] += sum_{c, k} W1[c][ci][k] dz[c][t + (1 - k)]
            f32x4t acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = (f32x4t){0.f, 0.f, 0.f, 0.f};
            for (int s4 = 0; s4 < ksteps; ++s4) {
                const int kk = 4 * s4 + kq;
                const float bv = dzp[(kk >> 1) * PP + t + 1 - (kk & 1)];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < CIT) acc[i] = tg_mfma(W1T[(16 * i + li) * WPT + kk], bv, acc[i]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ci = 16 * i + 4 * kq + r;
                    if (ci < Ci) {
                        const int64_t at = b * Ci * T + ci * T + t;
                        dxin_out[at] = acc[i][r] + ws[g.w_dxin[l] + at];
                    }
                }
        }
    }
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        const int idx = wave + 4 * q;
        if (idx < ntiles) {
            const int i = idx / NT2, n = 16 * (idx - i * NT2) + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * i + 4 * kq + r;
                if (c < Co && n < 2 * Ci) gp[g.o_c1_w[l] + c * 2 * Ci + n] = gw[q][r];
            }
        }
    }
}

// ---- graph part, backward --------------------------------------------------------------------------------------------------------
// LDS: adj[N*N] | ah[N*N] | dH[N*h] | dN[N*h] | H[N*h] | dhp[N*h] | AX[N*Lh] | Wg[h*hp] |
//      G x { Wh[N*hp] | dWh[N*h] | att[N*N] | dpre[N*N] | f1[N] | f2[N] | Wb[h*hp] }
__global__ __launch_bounds__(1024) void tg_graph_bwd_kernel(TgGeom g, int G, const float* __restrict__ prm, float* __restrict__ ws) {
    extern __shared__ float lds[];
    const int N = g.N, L = g.L, h = g.h, Hd = g.heads, tid = threadIdx.x, NT = blockDim.x;
    const int grp = tid / TB, gt = tid % TB;
    const int Lh = L > h ? L : h, hp = odd(h);
    float* adj = lds;
    float* ah = adj + N * N;
    float* dH = ah + N * N;          // gradient arriving at the GAT output / the GCN output
    float* dN = dH + N * h;          // gradient w.r.t. the GAT input (summed over the heads)
    float* H = dN + N * h;           // GAT input = leaky(pre) of the GCN below
    float* dhp = H + N * h;
    float* AX = dhp + N * h;
    float* Wg = AX + N * Lh;
    float* groups = Wg + h * hp;
    const int gsz = tg_bwd_group_floats(N, h);
    float* Wh = groups + grp * gsz;   // the head's Wh, later its share of dN ([N][h])
    float* dWh = Wh + N * hp;
    float* att = dWh + N * h;
    float* dpre = att + N * N;
    float* f1 = dpre + N * N;
    float* f2 = f1 + N;
    float* Wb = f2 + N;
    float* gp = ws + g.w_gpart + (int64_t)blockIdx.x * g.pcount;
    const float ih = 1.0f / (float)Hd;
    for (int64_t b = blockIdx.x; b < g.B; b += gridDim.x) {
        __syncthreads();
        for (int e = tid; e < N * N; e += NT) { adj[e] = ws[g.w_adj + b * N * N + e]; ah[e] = ws[g.w_ahat + b * N * N + e]; }
        for (int e = tid; e < N * h; e += NT) dH[e] = ws[g.w_dxin[0] + b * N * h + e];          // d G from tcn1's first stage (written in place)
        __syncthreads();
        for (int layer = 1; layer >= 0; --layer) {
            const int K = layer == 0 ? L : h;
            const float* pre_g = ws + (layer == 0 ? g.w_pre1 : g.w_pre2) + b * N * h;
            for (int e = tid; e < N * h; e += NT) { H[e] = lrelu(pre_g[e], TG_GCN_SLOPE); dN[e] = 0.f; dhp[e] = dH[e] * ih; }
            __syncthreads();
            for (int hd0 = 0; hd0 < Hd; hd0 += G) {
                const int hd = hd0 + grp;
                const bool act = hd < Hd;
                const int hdc = act ? hd : 0;
                const float* av = prm + g.o_gat_a[layer][hdc];
                const int64_t at_wh = g.w_wh[layer] + (b * Hd + hdc) * N * h, at_nn = (b * Hd + hdc) * N * N;
                if (act) {
                    for (int e = gt; e < N * h; e += TB) Wh[(e / h) * hp + e % h] = ws[at_wh + e];
                    for (int e = gt; e < N * N; e += TB) att[e] = ws[g.w_att[layer] + at_nn + e];
                    stage_group(Wb, prm + g.o_gat_w[layer][hdc], h, h, hp, gt);
                }
                __syncthreads();
                if (act)                                            // d att (masked)
                    for (int e = gt; e < N * N; e += TB) {
                        const int i = e / N, j = e % N;
                        float a = 0.f;
#pragma unroll 8
                        for (int o = 0; o < h; ++o) a = fmaf(dhp[i * h + o], Wh[j * hp + o], a);
                        dpre[e] = a * adj[e];
                    }
                __syncthreads();
                if (act && gt < N) {                                // softmax backward, one thread per row
                    const int i = gt;
                    float dot = 0.f;
#pragma unroll 8
                    for (int j = 0; j < N; ++j) dot = fmaf(dpre[i * N + j], att[i * N + j], dot);
                    float r = 0.f;
#pragma unroll 8
                    for (int j = 0; j < N; ++j) {
                        const float de = att[i * N + j] * (dpre[i * N + j] - dot);
                        const float v = ws[g.w_gpre[layer] + at_nn + i * N + j] > 0.f ? de : TG_GAT_SLOPE * de;
                        dpre[i * N + j] = v;
                        r += v;
                    }
                    f1[i] = r;                                      // d f1[i] = row sum
                }
                __syncthreads();
                if (act && gt < N) {
                    float cs = 0.f;
#pragma unroll 8
                    for (int i = 0; i < N; ++i) cs += dpre[i * N + gt];
                    f2[gt] = cs;                                    // d f2[j] = column sum
                }
                __syncthreads();
                if (act) {
                    if (gt == 0) {
                        float a = 0.f;
#pragma unroll 8
                        for (int i = 0; i < N; ++i) a += f1[i];
                        gp[g.o_gat_ab[layer][hdc]] += a;
                    }
                    if (gt < 2 * h) {
                        const int half = gt / h, o = gt % h;
                        const float* df = half ? f2 : f1;
                        float a = 0.f;
#pragma unroll 8
                        for (int i = 0; i < N; ++i) a = fmaf(df[i], Wh[i * hp + o], a);
                        gp[g.o_gat_a[layer][hdc] + gt] += a;
                    }
                    for (int e = gt; e < N * h; e += TB) {
                        const int j = e / h, o = e % h;
                        float a = fmaf(f1[j], av[o], f2[j] * av[h + o]);
#pragma unroll 8
                        for (int i = 0; i < N; ++i) a = fmaf(att[i * N + j] * adj[i * N + j], dhp[i * h + o], a);
                        dWh[e] = a;
                    }
                }
                __syncthreads();
                if (act) {
                    for (int e = gt; e < h * h; e += TB) {
                        const int o = e / h, k = e % h;
                        float a = 0.f;
#pragma unroll 8
                        for (int i = 0; i < N; ++i) a = fmaf(dWh[i * h + o], H[i * h + k], a);
                        gp[g.o_gat_w[layer][hdc] + e] += a;
                    }
                    if (gt < h) {
                        float a = 0.f;
#pragma unroll 8
                        for (int i = 0; i < N; ++i) a += dWh[i * h + gt];
                        gp[g.o_gat_b[layer][hdc] + gt] += a;
                    }
                    for (int e = gt; e < N * h; e += TB) {             // this head's share of dN, over its Wh (no longer needed)
                        const int i = e / h, k = e % h;
                        float a = 0.f;
#pragma unroll 8
                        for (int o = 0; o < h; ++o) a = fmaf(dWh[i * h + o], Wb[o * hp + k], a);
                        Wh[e] = a;
                    }
                }
                __syncthreads();
                for (int e = tid; e < N * h; e += NT) {                // summed in head order
                    float a = dN[e];
                    for (int q = 0; q < G && hd0 + q < Hd; ++q) a += groups[q * gsz + e];
                    dN[e] = a;
                }
                __syncthreads();
            }
            // GCN backward: d pre = dN * leaky'(pre)
            for (int e = tid; e < N * h; e += NT) dN[e] = pre_g[e] > 0.f ? dN[e] : TG_GCN_SLOPE * dN[e];
            const float* ax_g = ws + (layer == 0 ? g.w_ax1 : g.w_ax2) + b * N * K;
            for (int e = tid; e < N * K; e += NT) AX[e] = ax_g[e];
            if (layer == 1)
                for (int i = tid; i < h * h; i += NT) Wg[(i / h) * hp + i % h] = prm[g.o_gcn_w[1] + i];
            __syncthreads();
            for (int e = tid; e < h * K; e += NT) {
                const int o = e / K, k = e % K;
                float a = 0.f;
#pragma unroll 8
                for (int i = 0; i < N; ++i) a = fmaf(dN[i * h + o], AX[i * K + k], a);
                gp[g.o_gcn_w[layer] + e] += a;
            }
            if (tid < h) {
                float a = 0.f;
#pragma unroll 8
                for (int i = 0; i < N; ++i) a += dN[i * h + tid];
                gp[g.o_gcn_b[layer] + tid] += a;
            }
            __syncthreads();
            if (layer == 1) {
                // d AX = d pre W ; d (gat1 output) = A_hat^T d AX
                for (int e = tid; e < N * h; e += NT) {
                    const int i = e / h, k = e % h;
                    float a = 0.f;
#pragma unroll 8
                    for (int o = 0; o < h; ++o) a = fmaf(dN[i * h + o], Wg[o * hp + k], a);
                    AX[e] = a;
                }
                __syncthreads();
                for (int e = tid; e < N * h; e += NT) {
                    const int j = e / h, k = e % h;
                    float a = 0.f;
#pragma unroll 8
                    for (int i = 0; i < N; ++i) a = fmaf(ah[i * N + j], AX[i * h + k], a);
                    dH[e] = a;
                }
                __syncthreads();
            }
        }
    }
}

inline size_t tg_lds_graph_fwd(const TgGeom& g, int G) {
    const int Lh = g.L > g.h ? g.L : g.h;
    return sizeof(float) * ((size_t)g.N * (g.L | 1) + 2 * g.N * g.N + (size_t)g.N * Lh + 2 * (size_t)g.N * g.h + (size_t)g.h * (Lh | 1) +
                            (size_t)G * tg_fwd_group_floats(g.N, g.h));
}
inline size_t tg_lds_graph_bwd(const TgGeom& g, int G) {
    const int Lh = g.L > g.h ? g.L : g.h;
    return sizeof(float) * (2 * (size_t)g.N * g.N + 4 * (size_t)g.N * g.h + (size_t)g.N * Lh + (size_t)g.h * (g.h | 1) +
                            (size_t)G * tg_bwd_group_floats(g.N, g.h));
}
// head groups run side by side: as many as fit in the CU's LDS (<= 4 x TB threads)
inline int tg_head_groups(const TgGeom& g, bool bwd) {
    int G = g.heads < 4 ? g.heads : 4;
    while (G > 1 && (bwd ? tg_lds_graph_bwd(g, G) : tg_lds_graph_fwd(g, G)) > 160 * 1024) --G;
    return G;
}

template <typename K>
inline int tg_allow_lds(K kernel, size_t lds) {
    if (lds > 160 * 1024) return RULGNN_EUNSUPPORTED;
    if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return RULGNN_EHIP;
    return RULGNN_OK;
}

}  // namespace

int64_t stagnn_param_count(const rulgnn_stagnn_shape* s) {
    TgGeom g;
    return tg_geometry(s, &g) == RULGNN_OK ? g.pcount : -1;
}

int64_t stagnn_bn_state_count(const rulgnn_stagnn_shape* s) {
    TgGeom g;
    return tg_geometry(s, &g) == RULGNN_OK ? g.bn_total : -1;
}

size_t stagnn_workspace_bytes(const rulgnn_stagnn_shape* s) {
    TgGeom g;
    return tg_geometry(s, &g) == RULGNN_OK ? (size_t)g.total * sizeof(float) : 0;
}

int64_t stagnn_tap_offset(const rulgnn_stagnn_shape* s, int which) {
    TgGeom g;
    if (tg_geometry(s, &g) != RULGNN_OK) return -1;
    switch (which) {
        case 0: return g.w_adj;
        case 1: return g.w_G;
        case 2: return g.w_o1[0];
        case 3: return g.w_e[0];
        case 4: return g.w_o1[1];
        case 5: return g.w_e[1];
        default: return -1;
    }
}

#define TG_RC(call)                        \
    do {                                   \
        const int rc_ = (call);            \
        if (rc_ != RULGNN_OK) return rc_;  \
    } while (0)
#define TG_LAUNCH_OK()                                           \
    do {                                                         \
        if (hipGetLastError() != hipSuccess) return RULGNN_EHIP; \
    } while (0)

// mode bit 0: forward, bit 1: backward (after a TRAINING forward with the same args / workspace)
int stagnn_run(const rulgnn_stagnn_shape* s, const rulgnn_stagnn_args* a, int mode, hipStream_t st) {
    TgGeom g;
    TG_RC(tg_geometry(s, &g));
    if (a->workspace_bytes < (size_t)g.total * sizeof(float)) return RULGNN_EWORKSPACE;
    if (g.B == 0) return RULGNN_OK;
    float* ws = static_cast<float*>(a->workspace);
    const float* prm = a->params;
    const int64_t gb = a->global_batch > 0 ? a->global_batch : g.B;
    const float inv_gb = 1.0f / (float)gb;
    const int training = a->training ? 1 : 0;
    const int T = g.T, TP = g.T | 1, Hd = g.heads;
    const dim3 grid((unsigned)g.nblk), blk(TB);
    const float* stage_in[2] = {ws + g.w_G, ws + g.w_e[0]};
    (void)hipGetLastError();
    if (mode & 1) {
        const int G = tg_head_groups(g, false);
        const size_t lg = tg_lds_graph_fwd(g, G);
        TG_RC(tg_allow_lds(tg_graph_fwd_kernel, lg));
        hipLaunchKernelGGL(tg_graph_fwd_kernel, grid, dim3(TB * G), lg, st, g, G, a->x, prm, ws);
        TG_LAUNCH_OK();
        for (int l = 0; l < 2; ++l) {
            const int Ci = g.Ci[l], Co = g.Co[l];
            const size_t l1 = sizeof(float) * ((size_t)Ci * TP + (size_t)Co * TP + 2 * (size_t)Co * Ci);
            const size_t l2 = sizeof(float) * ((size_t)Ci * TP + 2 * (size_t)Co * TP + 2 * Co + (size_t)Co * Ci + 2 * (size_t)Co * Co);
            const size_t l3 = sizeof(float) * ((size_t)Co * TP + (size_t)Hd * T + T + 2 * Co + TB + (size_t)Hd * Co);
            TG_RC(tg_allow_lds(tg_conv1_fwd_kernel, l1));
            TG_RC(tg_allow_lds(tg_mid_fwd_kernel, l2));
            TG_RC(tg_allow_lds(tg_end_fwd_kernel, l3));
            if (tg_conv1_mx_ok(g, l)) {
                const size_t l1m = tg_conv1_fwd_mx_lds(g, l);
                TG_RC(tg_allow_lds(tg_conv1_fwd_mx_kernel, l1m));
                hipLaunchKernelGGL(tg_conv1_fwd_mx_kernel, grid, blk, l1m, st, g, l, stage_in[l], prm, ws);
            } else
            hipLaunchKernelGGL(tg_conv1_fwd_kernel, grid, blk, l1, st, g, l, stage_in[l], prm, ws);
            if (tg_mid_mx_ok(g, l)) {
                TG_RC(tg_allow_lds(tg_mid_fwd_mx_kernel, TM_FWD_LDS));
                hipLaunchKernelGGL(tg_mid_fwd_mx_kernel, grid, blk, TM_FWD_LDS, st, g, l, stage_in[l], prm, (const float*)a->bn_state, ws, training);
            } else
            hipLaunchKernelGGL(tg_mid_fwd_kernel, grid, blk, l2, st, g, l, stage_in[l], prm, (const float*)a->bn_state, ws, training);
            hipLaunchKernelGGL(tg_end_fwd_kernel, grid, blk, l3, st, g, l, prm, (const float*)a->bn_state, ws, training, a->y, a->pred, inv_gb);
            TG_LAUNCH_OK();
        }
        if (training && a->update_running_stats) hipLaunchKernelGGL(tg_running_kernel, dim3(4), dim3(TB), 0, st, g, (const float*)ws, a->bn_state);
        if (a->y && a->loss) (void)block_sum((const float*)(ws + g.w_sq), g.B, a->loss, st);
        TG_LAUNCH_OK();
    }
    if (mode & 2) {
        if (!a->grads) return RULGNN_EINVAL;
        if (!training) return RULGNN_EINVAL;                      // the backward differentiates the batch statistics
        const float* dpred = a->dpred ? a->dpred : ws + g.w_dpred;
        if (hipMemsetAsync(ws + g.w_gpart, 0, (size_t)g.nblk * g.pcount * sizeof(float), st) != hipSuccess) return RULGNN_EHIP;
        for (int l = 1; l >= 0; --l) {
            const int Ci = g.Ci[l], Co = g.Co[l];
            const size_t l3 = sizeof(float) * (2 * (size_t)Co * TP + 3 * (size_t)Hd * T + T + 2 * Co + Hd + (size_t)Hd * Co);
            const size_t l2 = sizeof(float) * ((size_t)Ci * TP + 3 * (size_t)Co * TP + 6 * Co + (size_t)Co * Ci + 2 * (size_t)Co * Co);
            const size_t l1 = sizeof(float) * ((size_t)Ci * TP + (size_t)Co * TP + 4 * Co + 2 * (size_t)Co * Ci);
            TG_RC(tg_allow_lds(tg_mid_bwd_kernel, l2));
            TG_RC(tg_allow_lds(tg_end_bwd_kernel, l3));
            TG_RC(tg_allow_lds(tg_conv1_bwd_kernel, l1));
            // (the gradient w.r.t. a stage's input is written over w_dxin[l], which the next kernel down the chain reads)
            hipLaunchKernelGGL(tg_end_bwd_kernel, grid, blk, l3, st, g, l, prm, (const float*)a->bn_state, ws, dpred, (const float*)(ws + g.w_dxin[1]));
            if (tg_mid_mx_ok(g, l)) {
                TG_RC(tg_allow_lds(tg_mid_bwd_mx_kernel, TM_BWD_LDS));
                hipLaunchKernelGGL(tg_mid_bwd_mx_kernel, grid, blk, TM_BWD_LDS, st, g, l, stage_in[l], prm, (const float*)a->bn_state, ws);
            } else
            hipLaunchKernelGGL(tg_mid_bwd_kernel, grid, blk, l2, st, g, l, stage_in[l], prm, (const float*)a->bn_state, ws);
            if (tg_conv1_mx_ok(g, l)) {
                const size_t l1m = tg_conv1_bwd_mx_lds(g, l);
                TG_RC(tg_allow_lds(tg_conv1_bwd_mx_kernel, l1m));
                hipLaunchKernelGGL(tg_conv1_bwd_mx_kernel, grid, blk, l1m, st, g, l, stage_in[l], prm, (const float*)a->bn_state, ws, ws + g.w_dxin[l]);
            } else
            hipLaunchKernelGGL(tg_conv1_bwd_kernel, grid, blk, l1, st, g, l, stage_in[l], prm, (const float*)a->bn_state, ws, ws + g.w_dxin[l]);
            TG_LAUNCH_OK();
        }
        const int G = tg_head_groups(g, true);
        const size_t lg = tg_lds_graph_bwd(g, G);
        TG_RC(tg_allow_lds(tg_graph_bwd_kernel, lg));
        hipLaunchKernelGGL(tg_graph_bwd_kernel, grid, dim3(TB * G), lg, st, g, G, prm, ws);
        TG_LAUNCH_OK();
        TG_RC(rows_sum(ws + g.w_gpart, g.nblk, g.pcount, g.pcount, a->grads, st));
    }
    return RULGNN_OK;
}

}  // namespace rulgnn
