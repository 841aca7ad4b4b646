// HAGCN graph stack for gfx950: cosine adjacency -> 3 x [GIN layer -> SAGPool (scores, KL prior, top-k gather)] ->
// node means, forward and backward.
//
// Reference: models/HAGCN/Model.py (GINLayer :6-24, SAGPool :75-120, cosine_distance :122-127, HAGCN_model.forward
// :164-183).  The Bi-LSTM stack in front of it (:26-73, 95 % of the reference's time, strictly sequential over batch*nodes)
// is delegated to the vendor library by the Python side; this file is the graph part.
//
// One workgroup per graph (a patch of one sample: <= 20 sensor nodes, 60/64-wide features).  Everything a graph needs
// lives in LDS; the 64x64 weight matrices are staged through LDS per linear layer.  The forward writes a tape (inputs of
// every linear layer, scores, the selected node indices) to the workspace; the backward walks it in reverse, computes all
// data gradients per graph and leaves the per-row output gradients of every linear layer in the workspace, from which the
// weight gradients are reduced over all graphs by split-K MFMA GEMMs (deterministic).
#include "sgemm_mfma.hpp"
#include "stgcn_host.hpp"

namespace rulgnn {

namespace {

constexpr int HB = 256;
constexpr int MAXN = 20;            // nodes
constexpr int MAXF = 64;            // feature width
constexpr int TS = MAXF + 1;        // LDS tile stride
constexpr int AS = MAXN + 1;        // adjacency stride
constexpr int NLV = 3;
constexpr float LEAKY = 0.01f;
__constant__ int kTopK[NLV] = {10, 5, 1};
constexpr int hTopK[NLV] = {10, 5, 1};       // SAGPool sizes, Model.py:137-141
constexpr int TOPK_SLOTS = 16;               // ints per graph in the index output: 10 | 5 | 1

struct HgGeom {
    int64_t G;
    int N, F0, Hd, Hh;
    int nin[NLV];                   // nodes entering each level: N, 10, 5
    int fin[NLV];                   // feature width entering each level: F0, Hd, Hd
    // parameter offsets per level
    int o_eps[NLV], o_w0[NLV], o_b0[NLV], o_w2[NLV], o_b2[NLV], o_rw[NLV], o_rb[NLV], o_mw[NLV], o_mb[NLV], o_pw0[NLV], o_pb0[NLV],
        o_pw2[NLV], o_pb2[NLV];
    int nparam;
    // tape offsets (floats) per level, each [rows_l x width]
    int64_t rows[NLV];
    int64_t t_x[NLV], t_adj[NLV], t_u[NLV], t_h[NLV], t_g[NLV], t_axs[NLV], t_xop[NLV], t_pm[NLV], t_p[NLV], t_s[NLV];
    int64_t d_za[NLV], d_zb[NLV], d_zm[NLV], d_zp0[NLV], d_zp1[NLV], d_zr[NLV], d_eps[NLV];
    int64_t t_kl, t_topk, t_one, t_split, total_floats;
};

// The parameter-gradient products of the graph backward -- per level six (weight = dZ^T In, bias = column sums of dZ) pairs and the GIN
// eps sum: 39 products, each a launch pair of 5-18 us at its latency floor (66 launches, 0.55 ms of an 11-ms step, all in front of the
// LSTM stack's BPTT).  They feed nothing in the call: ten at a time they are jobs of one split-K launch + one reduction
// (sgemm_splitk_batch).  ws == nullptr: shapes only (scratch sizing).
constexpr int HG_PGRAD_JOBS = NLV * 13, HG_PGRAD_BATCH = 10;
static int hg_pgrad_jobs(const HgGeom& g, float* ws, float* gr, SplitKJob* jobs) {
    int n = 0;
    float* one = ws ? ws + g.t_one : nullptr;
    auto P = [&](int64_t off) -> const float* { return ws ? ws + off : nullptr; };
    auto G_ = [&](int off) -> float* { return gr ? gr + off : nullptr; };
    for (int l = 0; l < NLV; ++l) {
        const int R = (int)(g.rows[l] > 0 ? g.rows[l] : 1), Fin = g.fin[l], Hd = g.Hd, Hh = g.Hh;
        auto wgrad = [&](const float* dz, int O, const float* in, int K, float* dw, float* db) {
            jobs[n++] = SplitKJob{dz, 1, O, in, 1, K, dw, K, O, K, R};
            jobs[n++] = SplitKJob{one, 0, 0, dz, 1, O, db, O, 1, O, R};
        };
        wgrad(P(g.d_za[l]), Hd, P(g.t_u[l]), Fin, G_(g.o_w0[l]), G_(g.o_b0[l]));
        wgrad(P(g.d_zb[l]), Hd, P(g.t_h[l]), Hd, G_(g.o_w2[l]), G_(g.o_b2[l]));
        wgrad(P(g.d_zm[l]), Hd, P(g.t_axs[l]), Hd, G_(g.o_mw[l]), G_(g.o_mb[l]));
        wgrad(P(g.d_zp0[l]), Hh, P(g.t_g[l]), Hd, G_(g.o_pw0[l]), G_(g.o_pb0[l]));
        wgrad(P(g.d_zp1[l]), 1, P(g.t_pm[l]), Hh, G_(g.o_pw2[l]), G_(g.o_pb2[l]));
        wgrad(P(g.d_zr[l]), 1, P(g.t_axs[l]), Hd, G_(g.o_rw[l]), G_(g.o_rb[l]));
        jobs[n++] = SplitKJob{P(g.d_eps[l]), 0, 1, one, 0, 0, G_(g.o_eps[l]), 1, 1, 1, (int)(g.G > 0 ? g.G : 1)};      // sum over the graphs
    }
    return n;
}

__host__ int hg_geometry(const rulgnn_hagcn_shape* s, HgGeom* g) {
    if (!s) return RULGNN_EINVAL;
    if (s->graphs < 0 || s->num_node < 1 || s->enc_dim < 1 || s->hidden_dim < 2) return RULGNN_EINVAL;
    if (s->num_node > MAXN || s->num_node < hTopK[0] || s->enc_dim > MAXF || s->hidden_dim > MAXF || (s->hidden_dim & 1))
        return RULGNN_EUNSUPPORTED;
    if (s->graphs * (int64_t)s->num_node * MAXF > ((int64_t)1 << 30)) return RULGNN_EUNSUPPORTED;
    g->G = s->graphs;
    g->N = s->num_node;
    g->F0 = s->enc_dim;
    g->Hd = s->hidden_dim;
    g->Hh = g->Hd / 2;
    int o = 0;
    auto take = [&](int n) { const int r = o; o += n; return r; };
    for (int l = 0; l < NLV; ++l) {
        g->nin[l] = l == 0 ? g->N : hTopK[l - 1];
        g->fin[l] = l == 0 ? g->F0 : g->Hd;
        g->o_eps[l] = take(1);
        g->o_w0[l] = take(g->Hd * g->fin[l]); g->o_b0[l] = take(g->Hd);
        g->o_w2[l] = take(g->Hd * g->Hd); g->o_b2[l] = take(g->Hd);
        g->o_rw[l] = take(g->Hd); g->o_rb[l] = take(1);
        g->o_mw[l] = take(g->Hd * g->Hd); g->o_mb[l] = take(g->Hd);
        g->o_pw0[l] = take(g->Hh * g->Hd); g->o_pb0[l] = take(g->Hh);
        g->o_pw2[l] = take(g->Hh); g->o_pb2[l] = take(1);
    }
    g->nparam = o;
    int64_t t = 0;
    auto tk = [&](int64_t n) { const int64_t r = t; t += (n + 63) & ~(int64_t)63; return r; };
    for (int l = 0; l < NLV; ++l) {
        const int64_t R = g->G * g->nin[l];
        g->rows[l] = R;
        g->t_x[l] = tk(R * g->fin[l]); g->t_adj[l] = tk(R * g->nin[l]); g->t_u[l] = tk(R * g->fin[l]);
        g->t_h[l] = tk(R * g->Hd); g->t_g[l] = tk(R * g->Hd); g->t_axs[l] = tk(R * g->Hd); g->t_xop[l] = tk(R * g->Hd);
        g->t_pm[l] = tk(R * g->Hh); g->t_p[l] = tk(R); g->t_s[l] = tk(R);
        g->d_za[l] = tk(R * g->Hd); g->d_zb[l] = tk(R * g->Hd); g->d_zm[l] = tk(R * g->Hd); g->d_zp0[l] = tk(R * g->Hh);
        g->d_zp1[l] = tk(R); g->d_zr[l] = tk(R); g->d_eps[l] = tk(g->G);
    }
    g->t_kl = tk(g->G * NLV);
    g->t_topk = tk(g->G * TOPK_SLOTS);
    g->t_one = tk(64);
    size_t mx = 1;
    {   // scratch of the parameter-gradient products: launched ten jobs at a time (hg_pgrad_jobs / sgemm_splitk_batch)
        SplitKJob jobs[HG_PGRAD_JOBS];
        const int nj = hg_pgrad_jobs(*g, nullptr, nullptr, jobs);
        for (int j0 = 0; j0 < nj; j0 += HG_PGRAD_BATCH) {
            const size_t v = sgemm_splitk_batch_floats(jobs + j0, nj - j0 < HG_PGRAD_BATCH ? nj - j0 : HG_PGRAD_BATCH);
            if (v > mx) mx = v;
        }
    }
    g->t_split = tk((int64_t)mx);
    g->total_floats = t;
    return RULGNN_OK;
}

// ---- block-cooperative helpers (operands in LDS, tile stride TS, adjacency stride AS) ----
__device__ inline void stage_w(const float* __restrict__ W, int O, int K, float* Wst) {       // Wst[o][k]
    for (int e = threadIdx.x; e < O * K; e += HB) Wst[(e / K) * TS + (e % K)] = W[e];
}
// out[i][o] = act(sum_k in[i][k] Wst[o][k] + b[o]); optionally the pre-activation goes to `pre` (global, [n][O])
__device__ inline void lin_fwd(const float* in, int n, int K, const float* Wst, const float* __restrict__ b, int O, float* out, int act,
                               float* __restrict__ pre) {
    for (int e = threadIdx.x; e < n * O; e += HB) {
        const int i = e / O, o = e - i * O;
        float a = b[o];
        for (int k = 0; k < K; ++k) a = fmaf(in[i * TS + k], Wst[o * TS + k], a);
        if (pre) pre[e] = a;
        out[i * TS + o] = act == 1 ? fmaxf(a, 0.f) : (act == 2 ? (a > 0.f ? a : LEAKY * a) : a);
    }
}
// din[i][k] (+)= sum_o dout[i][o] Wst[o][k]
__device__ inline void lin_bwd(const float* dout, int n, int O, const float* Wst, int K, float* din, bool accumulate) {
    for (int e = threadIdx.x; e < n * K; e += HB) {
        const int i = e / K, k = e - i * K;
        float a = accumulate ? din[i * TS + k] : 0.f;
        for (int o = 0; o < O; ++o) a = fmaf(dout[i * TS + o], Wst[o * TS + k], a);
        din[i * TS + k] = a;
    }
}
// out[i][c] = sum_j adj[i][j] x[j][c]  (+ scale * x[i][c])
__device__ inline void agg(const float* adj, const float* x, int n, int F, float* out, float self_scale) {
    for (int e = threadIdx.x; e < n * F; e += HB) {
        const int i = e / F, c = e - i * F;
        float a = self_scale * x[i * TS + c];
        for (int j = 0; j < n; ++j) a = fmaf(adj[i * AS + j], x[j * TS + c], a);
        out[i * TS + c] = a;
    }
}
// out[j][c] (+)= sum_i adj[i][j] d[i][c]  (+ scale * d[j][c])
__device__ inline void agg_t(const float* adj, const float* d, int n, int F, float* out, float self_scale, bool accumulate) {
    for (int e = threadIdx.x; e < n * F; e += HB) {
        const int j = e / F, c = e - j * F;
        float a = (accumulate ? out[j * TS + c] : 0.f) + self_scale * d[j * TS + c];
        for (int i = 0; i < n; ++i) a = fmaf(adj[i * AS + j], d[i * TS + c], a);
        out[j * TS + c] = a;
    }
}
// dadj[i][j] += sum_c a[i][c] b[j][c]
__device__ inline void outer_acc(const float* a, const float* b, int n, int F, float* dadj) {
    for (int e = threadIdx.x; e < n * n; e += HB) {
        const int i = e / n, j = e - i * n;
        float s = dadj[i * AS + j];
        for (int c = 0; c < F; ++c) s = fmaf(a[i * TS + c], b[j * TS + c], s);
        dadj[i * AS + j] = s;
    }
}
__device__ inline void tile_store(const float* t, int n, int F, float* __restrict__ dst) {     // LDS tile -> [n][F] global
    for (int e = threadIdx.x; e < n * F; e += HB) dst[e] = t[(e / F) * TS + (e % F)];
}
__device__ inline void tile_load(const float* __restrict__ src, int n, int F, float* t) {
    for (int e = threadIdx.x; e < n * F; e += HB) t[(e / F) * TS + (e % F)] = src[e];
}

// ---------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(HB) void hg_forward_kernel(HgGeom g, const float* __restrict__ nodes, const float* __restrict__ prm,
                                                       float* __restrict__ ws, float* __restrict__ feats, int* __restrict__ topk_out,
                                                       const int* __restrict__ forced) {
    extern __shared__ float sm[];
    float* X = sm;                      // [MAXN][TS] level input
    float* U = X + MAXN * TS;
    float* H = U + MAXN * TS;
    float* Gt = H + MAXN * TS;
    float* AX = Gt + MAXN * TS;
    float* XO = AX + MAXN * TS;
    float* PM = XO + MAXN * TS;
    float* Wst = PM + MAXN * TS;        // [MAXF][TS]
    float* ADJ = Wst + MAXF * TS;       // [MAXN][AS]
    float* ADJ2 = ADJ + MAXN * AS;
    float* nrm = ADJ2 + MAXN * AS;      // [MAXN]
    float* Pv = nrm + MAXN;
    float* Sv = Pv + MAXN;
    __shared__ int sel[MAXN];
    const int tid = threadIdx.x, Hd = g.Hd, Hh = g.Hh;
    int* topk_ws = reinterpret_cast<int*>(ws + g.t_topk);

    for (int64_t gi = blockIdx.x; gi < g.G; gi += gridDim.x) {
        // ---- cosine adjacency (Model.py:122-127) ----
        tile_load(nodes + gi * g.N * g.F0, g.N, g.F0, X);
        __syncthreads();
        if (tid < g.N) {
            float s = 0.f;
            for (int c = 0; c < g.F0; ++c) s = fmaf(X[tid * TS + c], X[tid * TS + c], s);
            nrm[tid] = sqrtf(s);
        }
        __syncthreads();
        for (int e = tid; e < g.N * g.N; e += HB) {
            const int i = e / g.N, j = e - i * g.N;
            float s = 0.f;
            for (int c = 0; c < g.F0; ++c) s = fmaf(X[i * TS + c], X[j * TS + c], s);
            ADJ[i * AS + j] = s / (nrm[i] * nrm[j]);
        }
        __syncthreads();
        int slot = 0;
        for (int l = 0; l < NLV; ++l) {
            const int n = g.nin[l], Fin = g.fin[l], k = kTopK[l];
            const int64_t r0 = gi * n;
            tile_store(X, n, Fin, ws + g.t_x[l] + r0 * Fin);
            for (int e = tid; e < n * n; e += HB) ws[g.t_adj[l] + r0 * n + e] = ADJ[(e / n) * AS + (e % n)];
            // ---- GIN (Model.py:16-24): u = A x + (1 + eps) x; mlp ----
            agg(ADJ, X, n, Fin, U, 1.0f + prm[g.o_eps[l]]);
            stage_w(prm + g.o_w0[l], Hd, Fin, Wst);
            __syncthreads();
            tile_store(U, n, Fin, ws + g.t_u[l] + r0 * Fin);
            lin_fwd(U, n, Fin, Wst, prm + g.o_b0[l], Hd, H, 1, nullptr);
            __syncthreads();
            tile_store(H, n, Hd, ws + g.t_h[l] + r0 * Hd);
            stage_w(prm + g.o_w2[l], Hd, Hd, Wst);
            __syncthreads();
            lin_fwd(H, n, Hd, Wst, prm + g.o_b2[l], Hd, Gt, 0, nullptr);
            __syncthreads();
            tile_store(Gt, n, Hd, ws + g.t_g[l] + r0 * Hd);
            // ---- SAGPool (Model.py:89-118) ----
            agg(ADJ, Gt, n, Hd, AX, 0.f);
            stage_w(prm + g.o_mw[l], Hd, Hd, Wst);
            __syncthreads();
            tile_store(AX, n, Hd, ws + g.t_axs[l] + r0 * Hd);
            lin_fwd(AX, n, Hd, Wst, prm + g.o_mb[l], Hd, XO, 2, ws + g.t_xop[l] + r0 * Hd);
            __syncthreads();
            stage_w(prm + g.o_pw0[l], Hh, Hd, Wst);
            __syncthreads();
            lin_fwd(Gt, n, Hd, Wst, prm + g.o_pb0[l], Hh, PM, 1, nullptr);
            __syncthreads();
            tile_store(PM, n, Hh, ws + g.t_pm[l] + r0 * Hh);
            if (tid < n) {                              // logits of the prior (mlp) and of the score (rank)
                float a = prm[g.o_pb2[l]], b = prm[g.o_rb[l]];
                for (int c = 0; c < Hh; ++c) a = fmaf(PM[tid * TS + c], prm[g.o_pw2[l] + c], a);
                for (int c = 0; c < Hd; ++c) b = fmaf(AX[tid * TS + c], prm[g.o_rw[l] + c], b);
                Pv[tid] = a;
                Sv[tid] = b;
            }
            __syncthreads();
            float pn = 0.f, sn = 0.f;                   // softmax over the nodes (dim = 1), both
            if (tid < n) {
                float ma = -INFINITY, mb = -INFINITY;
                for (int j = 0; j < n; ++j) { ma = fmaxf(ma, Pv[j]); mb = fmaxf(mb, Sv[j]); }
                float sa = 0.f, sb = 0.f;
                for (int j = 0; j < n; ++j) { sa += expf(Pv[j] - ma); sb += expf(Sv[j] - mb); }
                pn = expf(Pv[tid] - ma) / sa;
                sn = expf(Sv[tid] - mb) / sb;
            }
            __syncthreads();
            if (tid < n) {
                Pv[tid] = pn;
                Sv[tid] = sn;
                ws[g.t_p[l] + r0 + tid] = pn;
                ws[g.t_s[l] + r0 + tid] = sn;
            }
            __syncthreads();
            if (tid == 0) {                             // KL(score || P) contribution of this graph (Model.py:103)
                float kl = 0.f;
                for (int j = 0; j < n; ++j)
                    if (Sv[j] > 0.f) kl += Sv[j] * (logf(Sv[j]) - logf(Pv[j]));
                ws[g.t_kl + gi * NLV + l] = kl;
            }
            if (tid < n) {                              // rank in a stable descending sort of the scores
                int rnk = 0;
                const float s = Sv[tid];
                for (int j = 0; j < n; ++j) rnk += (Sv[j] > s || (Sv[j] == s && j < tid)) ? 1 : 0;
                if (rnk < k) sel[rnk] = tid;
            }
            __syncthreads();
            if (tid < k) {
                if (forced) sel[tid] = forced[gi * TOPK_SLOTS + slot + tid];
                topk_ws[gi * TOPK_SLOTS + slot + tid] = sel[tid];
                if (topk_out) topk_out[gi * TOPK_SLOTS + slot + tid] = sel[tid];
            }
            __syncthreads();
            // ---- gather: x_next[a] = xo[sel[a]]; adj_next[b][a] = adj[sel[a]][sel[b]] (Model.py:110-116) ----
            for (int e = tid; e < k * Hd; e += HB) {
                const int a = e / Hd, c = e - a * Hd;
                X[a * TS + c] = XO[sel[a] * TS + c];
            }
            for (int e = tid; e < k * k; e += HB) {
                const int b = e / k, a = e - b * k;
                ADJ2[b * AS + a] = ADJ[sel[a] * AS + sel[b]];
            }
            __syncthreads();
            for (int c = tid; c < Hd; c += HB) {        // mean over the kept nodes (Model.py:177-179)
                float s = 0.f;
                for (int a = 0; a < k; ++a) s += X[a * TS + c];
                feats[gi * (NLV * Hd) + l * Hd + c] = s / (float)k;
            }
            for (int e = tid; e < k * k; e += HB) ADJ[(e / k) * AS + (e % k)] = ADJ2[(e / k) * AS + (e % k)];
            slot += k;
            __syncthreads();
        }
    }
}

constexpr size_t HG_FWD_LDS = sizeof(float) * (7 * MAXN * TS + MAXF * TS + 2 * MAXN * AS + 3 * MAXN);

// total KL = sum over graphs and levels / G  (F.kl_div(..., reduction='batchmean'), summed over the three layers)
__global__ __launch_bounds__(1024) void hg_kl_kernel(const float* __restrict__ v, int64_t n, float inv_g, float* __restrict__ out) {
    __shared__ float red[1024];
    float a = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) a += v[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int m = 512; m > 0; m >>= 1) {
        if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0] * inv_g;
}

// ---------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(HB) void hg_backward_kernel(HgGeom g, const float* __restrict__ prm, float* __restrict__ ws,
                                                        const float* __restrict__ dfeats, const float* __restrict__ dkl,
                                                        float* __restrict__ dnodes) {
    extern __shared__ float sm[];
    float* X = sm;                      // tape: level input
    float* Hh_ = X + MAXN * TS;         // tape: GIN hidden
    float* Gt = Hh_ + MAXN * TS;        // tape: GIN output
    float* dXO = Gt + MAXN * TS;        // d xo -> d xo_pre
    float* dAX = dXO + MAXN * TS;
    float* dG = dAX + MAXN * TS;
    float* dH = dG + MAXN * TS;
    float* dU = dH + MAXN * TS;
    float* dXn = dU + MAXN * TS;        // gradient w.r.t. the pooled output of the current level (from the level above)
    float* dPM = dXn + MAXN * TS;
    float* Wst = dPM + MAXN * TS;
    float* ADJ = Wst + MAXF * TS;
    float* dADJ = ADJ + MAXN * AS;
    float* dADJn = dADJ + MAXN * AS;    // gradient w.r.t. the pooled adjacency of the current level
    float* dsl = dADJn + MAXN * AS;     // [MAXN] d score logit
    float* dpl = dsl + MAXN;            // [MAXN] d prior logit
    float* red = dpl + MAXN;            // [HB]
    __shared__ int sel[MAXN];
    const int tid = threadIdx.x, Hd = g.Hd, Hh = g.Hh;
    const int* topk_ws = reinterpret_cast<const int*>(ws + g.t_topk);
    const float ckl = dkl[0] / (float)g.G;

    for (int64_t gi = blockIdx.x; gi < g.G; gi += gridDim.x) {
        for (int l = NLV - 1; l >= 0; --l) {
            const int n = g.nin[l], Fin = g.fin[l], k = kTopK[l];
            const int64_t r0 = gi * n;
            int slot = 0;
            for (int q = 0; q < l; ++q) slot += kTopK[q];
            if (tid < k) sel[tid] = topk_ws[gi * TOPK_SLOTS + slot + tid];
            tile_load(ws + g.t_x[l] + r0 * Fin, n, Fin, X);
            tile_load(ws + g.t_h[l] + r0 * Hd, n, Hd, Hh_);
            tile_load(ws + g.t_g[l] + r0 * Hd, n, Hd, Gt);
            for (int e = tid; e < n * n; e += HB) {
                ADJ[(e / n) * AS + (e % n)] = ws[g.t_adj[l] + r0 * n + e];
                dADJ[(e / n) * AS + (e % n)] = 0.f;
            }
            for (int e = tid; e < n * Hd; e += HB) dXO[(e / Hd) * TS + (e % Hd)] = 0.f;
            __syncthreads();
            // d pooled -> scatter onto the kept nodes (mean over k nodes + what the level above sent down)
            for (int e = tid; e < k * Hd; e += HB) {
                const int a = e / Hd, c = e - a * Hd;
                float v = dfeats[gi * (NLV * Hd) + l * Hd + c] / (float)k;
                if (l < NLV - 1) v += dXn[a * TS + c];
                dXO[sel[a] * TS + c] = v;
            }
            if (l < NLV - 1)                            // adj_next[b][a] = adj[sel[a]][sel[b]]
                for (int e = tid; e < k * k; e += HB) {
                    const int b = e / k, a = e - b * k;
                    dADJ[sel[a] * AS + sel[b]] = dADJn[b * AS + a];
                }
            // KL backward through both softmaxes
            if (tid < n) {
                const float s = ws[g.t_s[l] + r0 + tid], p = ws[g.t_p[l] + r0 + tid];
                dsl[tid] = s > 0.f ? ckl * (logf(s) - logf(p) + 1.0f) : 0.f;     // d kl / d score
                dpl[tid] = s > 0.f ? -ckl * s / p : 0.f;                          // d kl / d P
            }
            __syncthreads();
            float na = 0.f, nb = 0.f;
            if (tid < n) {
                float ds = 0.f, dp = 0.f;
                for (int j = 0; j < n; ++j) {
                    ds = fmaf(dsl[j], ws[g.t_s[l] + r0 + j], ds);
                    dp = fmaf(dpl[j], ws[g.t_p[l] + r0 + j], dp);
                }
                const float s = ws[g.t_s[l] + r0 + tid], p = ws[g.t_p[l] + r0 + tid];
                na = s * (dsl[tid] - ds);
                nb = p * (dpl[tid] - dp);
            }
            __syncthreads();
            if (tid < n) {
                dsl[tid] = na;
                dpl[tid] = nb;
                ws[g.d_zr[l] + r0 + tid] = na;
                ws[g.d_zp1[l] + r0 + tid] = nb;
            }
            __syncthreads();
            // d axs = ds_logit (x) w_rank ; d pm = dp_logit (x) w_2 * [pm > 0]
            for (int e = tid; e < n * Hd; e += HB) {
                const int i = e / Hd, c = e - i * Hd;
                dAX[i * TS + c] = dsl[i] * prm[g.o_rw[l] + c];
                const float xop = ws[g.t_xop[l] + r0 * Hd + e];
                const float dz = dXO[i * TS + c] * (xop > 0.f ? 1.f : LEAKY);
                dXO[i * TS + c] = dz;                   // d xo_pre
                ws[g.d_zm[l] + r0 * Hd + e] = dz;
            }
            for (int e = tid; e < n * Hh; e += HB) {
                const int i = e / Hh, c = e - i * Hh;
                const float dz = ws[g.t_pm[l] + r0 * Hh + e] > 0.f ? dpl[i] * prm[g.o_pw2[l] + c] : 0.f;
                dPM[i * TS + c] = dz;
                ws[g.d_zp0[l] + r0 * Hh + e] = dz;
            }
            stage_w(prm + g.o_pw0[l], Hh, Hd, Wst);
            __syncthreads();
            lin_bwd(dPM, n, Hh, Wst, Hd, dG, false);    // d g (prior branch)
            __syncthreads();
            stage_w(prm + g.o_mw[l], Hd, Hd, Wst);
            __syncthreads();
            lin_bwd(dXO, n, Hd, Wst, Hd, dAX, true);    // d axs += d xo_pre W_model
            __syncthreads();
            outer_acc(dAX, Gt, n, Hd, dADJ);            // axs = adj g
            agg_t(ADJ, dAX, n, Hd, dG, 0.f, true);
            __syncthreads();
            tile_store(dG, n, Hd, ws + g.d_zb[l] + r0 * Hd);
            stage_w(prm + g.o_w2[l], Hd, Hd, Wst);
            __syncthreads();
            lin_bwd(dG, n, Hd, Wst, Hd, dH, false);
            __syncthreads();
            for (int e = tid; e < n * Hd; e += HB) {
                const int i = e / Hd, c = e - i * Hd;
                const float dz = Hh_[i * TS + c] > 0.f ? dH[i * TS + c] : 0.f;
                dH[i * TS + c] = dz;
                ws[g.d_za[l] + r0 * Hd + e] = dz;
            }
            stage_w(prm + g.o_w0[l], Hd, Fin, Wst);
            __syncthreads();
            lin_bwd(dH, n, Hd, Wst, Fin, dU, false);
            __syncthreads();
            // u = adj x + (1 + eps) x
            float pe = 0.f;
            for (int e = tid; e < n * Fin; e += HB) pe = fmaf(dU[(e / Fin) * TS + (e % Fin)], X[(e / Fin) * TS + (e % Fin)], pe);
            red[tid] = pe;
            outer_acc(dU, X, n, Fin, dADJ);
            agg_t(ADJ, dU, n, Fin, dXn, 1.0f + prm[g.o_eps[l]], false);      // becomes d (pooled output of the level below) / d nodes
            __syncthreads();
            for (int m = HB / 2; m > 0; m >>= 1) {
                if (tid < m) red[tid] += red[tid + m];
                __syncthreads();
            }
            if (tid == 0) ws[g.d_eps[l] + gi] = red[0];
            for (int e = tid; e < n * n; e += HB) dADJn[(e / n) * AS + (e % n)] = dADJ[(e / n) * AS + (e % n)];
            __syncthreads();
        }
        // ---- cosine adjacency backward: d x_i = d x_i + [ S xhat - rowsum(S * A) xhat ]_i / |x_i|, S = dA + dA^T ----
        const int N = g.N, F0 = g.F0;
        float* nrm = dsl;
        if (tid < N) {
            float s = 0.f;
            for (int c = 0; c < F0; ++c) s = fmaf(X[tid * TS + c], X[tid * TS + c], s);
            nrm[tid] = sqrtf(s);
        }
        __syncthreads();
        for (int e = tid; e < N * N; e += HB) {
            const int i = e / N, j = e - i * N;
            dADJ[i * AS + j] = dADJn[i * AS + j] + dADJn[j * AS + i];
        }
        __syncthreads();
        if (tid < N) {
            float s = 0.f;
            for (int j = 0; j < N; ++j) s = fmaf(dADJ[tid * AS + j], ADJ[tid * AS + j], s);
            dpl[tid] = s;
        }
        __syncthreads();
        for (int e = tid; e < N * F0; e += HB) {
            const int i = e / F0, c = e - i * F0;
            float a = 0.f;
            for (int j = 0; j < N; ++j) a = fmaf(dADJ[i * AS + j], X[j * TS + c] / nrm[j], a);
            a -= dpl[i] * X[i * TS + c] / nrm[i];
            dnodes[gi * N * F0 + e] = dXn[i * TS + c] + a / nrm[i];
        }
        __syncthreads();
    }
}

constexpr size_t HG_BWD_LDS = sizeof(float) * (10 * MAXN * TS + MAXF * TS + 3 * MAXN * AS + 2 * MAXN + HB);

__global__ void hg_fill_one_kernel(float* p) { p[0] = 1.f; }

template <typename K>
int hg_grid(K kernel, int64_t items, size_t lds) {
    int dev = 0, cus = 256, per_cu = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, HB, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    int64_t want = (int64_t)cus * per_cu;
    if (want > items) want = items;
    return want < 1 ? 1 : (int)want;
}

}  // namespace

int64_t hagcn_graph_param_count(const rulgnn_hagcn_shape* s) {
    HgGeom g;
    return hg_geometry(s, &g) == RULGNN_OK ? g.nparam : -1;
}

size_t hagcn_workspace_bytes(const rulgnn_hagcn_shape* s) {
    HgGeom g;
    if (hg_geometry(s, &g) != RULGNN_OK) return 0;
    return (size_t)g.total_floats * sizeof(float);
}

#define HG_RC(call)                        \
    do {                                   \
        const int rc_ = (call);            \
        if (rc_ != RULGNN_OK) return rc_;  \
    } while (0)

int hagcn_graph_forward(const rulgnn_hagcn_shape* s, const rulgnn_hagcn_args* a, hipStream_t st) {
    HgGeom g;
    HG_RC(hg_geometry(s, &g));
    if (a->workspace_bytes < (size_t)g.total_floats * sizeof(float)) return RULGNN_EWORKSPACE;
    float* ws = static_cast<float*>(a->workspace);
    (void)hipGetLastError();
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(hg_forward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)HG_FWD_LDS) != hipSuccess)
        return RULGNN_EHIP;
    hipLaunchKernelGGL(hg_forward_kernel, dim3(hg_grid(hg_forward_kernel, g.G, HG_FWD_LDS)), dim3(HB), HG_FWD_LDS, st, g, a->nodes,
                       a->params, ws, a->feats, a->topk, a->forced_topk);
    hipLaunchKernelGGL(hg_kl_kernel, dim3(1), dim3(1024), 0, st, (const float*)(ws + g.t_kl), g.G * NLV, 1.0f / (float)g.G, a->kl);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

int hagcn_graph_backward(const rulgnn_hagcn_shape* s, const rulgnn_hagcn_args* a, hipStream_t st) {
    HgGeom g;
    HG_RC(hg_geometry(s, &g));
    if (a->workspace_bytes < (size_t)g.total_floats * sizeof(float)) return RULGNN_EWORKSPACE;
    float* ws = static_cast<float*>(a->workspace);
    (void)hipGetLastError();
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(hg_backward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)HG_BWD_LDS) != hipSuccess)
        return RULGNN_EHIP;
    hipLaunchKernelGGL(hg_backward_kernel, dim3(hg_grid(hg_backward_kernel, g.G, HG_BWD_LDS)), dim3(HB), HG_BWD_LDS, st, g, a->params, ws,
                       a->dfeats, a->dkl, a->dnodes);
    float* one = ws + g.t_one;
    float* split = ws + g.t_split;
    hipLaunchKernelGGL(hg_fill_one_kernel, dim3(1), dim3(1), 0, st, one);
    if (g.G > 0) {
        SplitKJob jobs[HG_PGRAD_JOBS];
        const int nj = hg_pgrad_jobs(g, ws, a->grads, jobs);
        const size_t split_floats = (size_t)(g.total_floats - g.t_split);
        for (int j0 = 0; j0 < nj; j0 += HG_PGRAD_BATCH)
            HG_RC(sgemm_splitk_batch(jobs + j0, nj - j0 < HG_PGRAD_BATCH ? nj - j0 : HG_PGRAD_BATCH, split, split_floats, st));
    } else if (hipMemsetAsync(a->grads, 0, sizeof(float) * (size_t)g.nparam, st) != hipSuccess) {      // (no graphs: every sum is empty)
        return RULGNN_EHIP;
    }
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

}  // namespace rulgnn
