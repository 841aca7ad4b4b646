// STGNN graph path for gfx950 (first ChebNet user): Gaussian-kernel top-k adjacency of every
// (sample, patch) graph and the Chebyshev terms T_k(A) X, then the ChebNet projection and its filter gradient as MFMA GEMMs.
//
// Reference: models/STGNN/Model.py (compute_adjacency_matrix :8-25, ChebNet :29-61, the reshapes of STGNN_model.forward
// :75-91).  The model input carries no gradient and the adjacency depends on the input alone, so the graph construction
// is forward-only and ChebNet's backward is one contraction: d filters[k] = T_k^T d out.
//
//   stgnn_terms_kernel   one workgroup per graph: X [N, f] to LDS, pairwise squared distances (torch.cdist's exact path
//                        for <= 25 rows: sqrt of the summed squares, squared again by the reference), exp, rank-based top-k
//                        mask (ties towards the smaller column), T_1 = A X, T_k = 2 A T_{k-1} - T_{k-2};
//                        writes terms [G*N, K*f] (row = graph*N + node, column = k*f + j) and, optionally, A.
//   projection           out [G*N, H] = terms x filters viewed as [K*f, H]            (sgemm_mfma.hpp)
//   filter gradient      d filters [K*f, H] = terms^T x d out, split-K, fixed reduction order (deterministic)
#include "sgemm_mfma.hpp"
#include "stgcn_host.hpp"

namespace rulgnn {

namespace {

constexpr int GB = 256;             // threads per graph workgroup
constexpr int MAXN = 32;            // nodes per graph
constexpr int MAXF = 128;           // features per node (patch_size)
constexpr int MAXK = 4;

struct GnGeom {
    int64_t B, G;
    int N, L, f, H, K, topk, T, KF;
};

__host__ int gn_geometry(const rulgnn_stgnn_shape* s, GnGeom* g) {
    if (!s) return RULGNN_EINVAL;
    if (s->batch < 0 || s->num_nodes < 1 || s->num_patch < 1 || s->patch_size < 1 || s->hidden_dim < 1 || s->K < 1 || s->top_k < 1)
        return RULGNN_EINVAL;
    if (s->top_k > s->num_nodes) return RULGNN_EINVAL;                  // torch.topk raises
    if (s->num_nodes > MAXN || s->patch_size > MAXF || s->K > MAXK || s->hidden_dim > 1024 || s->num_patch > 4096)
        return RULGNN_EUNSUPPORTED;
    if (s->batch * (int64_t)s->num_patch * s->num_nodes > ((int64_t)1 << 30)) return RULGNN_EUNSUPPORTED;
    g->B = s->batch;
    g->N = s->num_nodes;
    g->L = s->num_patch;
    g->f = s->patch_size;
    g->H = s->hidden_dim;
    g->K = s->K;
    g->topk = s->top_k;
    g->G = g->B * g->L;
    g->T = g->L * g->f;
    g->KF = g->K * g->f;
    return RULGNN_OK;
}

__global__ __launch_bounds__(GB) void stgnn_terms_kernel(GnGeom g, const float* __restrict__ x, float* __restrict__ terms,
                                                         float* __restrict__ adj_out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int N = g.N, f = g.f, fp = f + 1;                 // +1: rows of X are read by 32 threads at once
    float* X = sm;                                          // [N][fp]
    float* T1 = X + N * fp;                                 // [N][fp]   T_{k-1}
    float* T0 = T1 + N * fp;                                // [N][fp]   T_{k-2}
    float* S = T0 + N * fp;                                 // [N][N+1]  similarities, then the masked adjacency
    float* A = S + N * (N + 1);
    const int64_t gi = blockIdx.x;
    const int64_t b = gi / g.L;
    const int l = (int)(gi % g.L);
    const float* xb = x + b * (int64_t)N * g.T + (int64_t)l * f;       // node n, feature j at xb[n*T + j]  (Model.py:79-80)
    for (int e = threadIdx.x; e < N * f; e += GB) {
        const int n = e / f, j = e % f;
        const float v = xb[(int64_t)n * g.T + j];
        X[n * fp + j] = v;
        T0[n * fp + j] = v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < N * N; e += GB) {
        const int i = e / N, j = e % N;
        float s = 0.f;
        for (int k = 0; k < f; ++k) {
            const float d = X[i * fp + k] - X[j * fp + k];
            s = fmaf(d, d, s);
        }
        const float d = sqrtf(s);
        S[i * (N + 1) + j] = expf(-(d * d));
    }
    __syncthreads();
    for (int e = threadIdx.x; e < N * N; e += GB) {        // keep entry (i, j) iff fewer than top_k entries of row i beat it
        const int i = e / N, j = e % N;
        const float v = S[i * (N + 1) + j];
        int beat = 0;
        for (int m = 0; m < N; ++m) {
            const float u = S[i * (N + 1) + m];
            beat += (u > v || (u == v && m < j)) ? 1 : 0;
        }
        const float a = beat < g.topk ? v : 0.f;
        A[i * (N + 1) + j] = a;
        if (adj_out) adj_out[gi * N * N + e] = a;
    }
    __syncthreads();
    float* trow = terms + gi * (int64_t)N * g.KF;
    for (int e = threadIdx.x; e < N * f; e += GB) trow[(int64_t)(e / f) * g.KF + e % f] = X[(e / f) * fp + e % f];
    // T_1 = A X, T_k = 2 A T_{k-1} - T_{k-2}  (Model.py:49-59)
    for (int k = 1; k < g.K; ++k) {
        const float* prev = k == 1 ? X : T1;
        constexpr int PER = (MAXN * MAXF + GB - 1) / GB;
        float acc[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int e = threadIdx.x + q * GB;
            acc[q] = 0.f;
            if (e < N * f) {
                const int i = e / f, j = e % f;
                float s = 0.f;
                for (int m = 0; m < N; ++m) s = fmaf(A[i * (N + 1) + m], prev[m * fp + j], s);
                acc[q] = k == 1 ? s : 2.f * s - T0[i * fp + j];
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int e = threadIdx.x + q * GB;
            if (e < N * f) {
                const int i = e / f, j = e % f;
                if (k > 1) T0[i * fp + j] = T1[i * fp + j];
                T1[i * fp + j] = acc[q];
                trow[(int64_t)i * g.KF + k * f + j] = acc[q];
            }
        }
        __syncthreads();
    }
}

size_t terms_lds_bytes(const GnGeom& g) { return ((size_t)3 * g.N * (g.f + 1) + (size_t)2 * g.N * (g.N + 1)) * sizeof(float); }

}  // namespace

size_t stgnn_workspace_bytes(const rulgnn_stgnn_shape* s) {
    GnGeom g;
    if (gn_geometry(s, &g) != RULGNN_OK) return 0;
    return sgemm_splitk_partial_floats(g.KF, g.H) * sizeof(float);
}

int stgnn_terms(const rulgnn_stgnn_shape* s, const float* x, float* terms, float* adj, hipStream_t st) {
    GnGeom g;
    const int rc = gn_geometry(s, &g);
    if (rc != RULGNN_OK) return rc;
    if (g.G == 0) return RULGNN_OK;
    const size_t lds = terms_lds_bytes(g);
    if (lds > 64 * 1024) return RULGNN_EUNSUPPORTED;
    (void)hipGetLastError();
    hipLaunchKernelGGL(stgnn_terms_kernel, dim3((unsigned)g.G), dim3(GB), lds, st, g, x, terms, adj);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

int stgnn_cheb_forward(const rulgnn_stgnn_shape* s, const float* terms, const float* filters, float* out, hipStream_t st) {
    GnGeom g;
    const int rc = gn_geometry(s, &g);
    if (rc != RULGNN_OK) return rc;
    const int64_t M = g.G * g.N;
    if (M == 0) return RULGNN_OK;
    // out[m][h] = sum_q terms[m][q] * filters[q][h]
    return sgemm(terms, g.KF, 1, filters, 1, g.H, out, g.H, (int)M, g.H, g.KF, false, st);
}

int stgnn_cheb_backward(const rulgnn_stgnn_shape* s, const float* terms, const float* dout, float* dfilters, void* workspace,
                        size_t workspace_bytes, hipStream_t st) {
    GnGeom g;
    const int rc = gn_geometry(s, &g);
    if (rc != RULGNN_OK) return rc;
    if (!workspace || workspace_bytes < stgnn_workspace_bytes(s)) return RULGNN_EWORKSPACE;
    const int64_t M = g.G * g.N;
    if (M == 0) return hipMemsetAsync(dfilters, 0, sizeof(float) * g.KF * g.H, st) == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
    // dfilters[q][h] = sum_m terms[m][q] * dout[m][h]
    return sgemm_splitk(terms, 1, g.KF, dout, 1, g.H, dfilters, g.H, g.KF, g.H, (int)M, false, static_cast<float*>(workspace), st);
}

// ---------------------------------------------------------------------------------------------------
// whole model: forward / backward / both, flat parameters
//   chebnet.filters [K, f, H] | gru.weight_ih_l0 [3H, H] | gru.weight_hh_l0 [3H, H] | gru.bias_ih_l0 [3H] | gru.bias_hh_l0 [3H] |
//   fc.weight [N*L*H] | fc.bias [1]        (the reference's state_dict order, Model.py:69-72)
// ---------------------------------------------------------------------------------------------------
namespace {

struct GnOff {
    int filters, wih, whh, bih, bhh, fcw, fcb, total;
};
GnOff gn_offsets(const GnGeom& g) {
    GnOff o;
    int t = 0;
    o.filters = t; t += g.KF * g.H;
    o.wih = t; t += 3 * g.H * g.H;
    o.whh = t; t += 3 * g.H * g.H;
    o.bih = t; t += 3 * g.H;
    o.bhh = t; t += 3 * g.H;
    o.fcw = t; t += g.H * g.L * g.N;
    o.fcb = t; t += 1;
    o.total = t;
    return o;
}

// rows (b, l, n) <-> (b, n, l) of an [*, H] tensor (Model.py:93-95: view(bs, L, N, H).permute(0, 2, 1, 3))
__global__ __launch_bounds__(GB) void stgnn_permute_kernel(GnGeom g, const float* __restrict__ src, float* __restrict__ dst, int to_seq) {
    const int64_t total = g.G * g.N * g.H;
    for (int64_t e = (int64_t)blockIdx.x * GB + threadIdx.x; e < total; e += (int64_t)gridDim.x * GB) {
        const int h = (int)(e % g.H);
        const int64_t r = e / g.H;                           // source row
        int64_t b, l, n;
        if (to_seq) { n = r % g.N; l = (r / g.N) % g.L; b = r / ((int64_t)g.N * g.L); }
        else { l = r % g.L; n = (r / g.L) % g.N; b = r / ((int64_t)g.N * g.L); }
        const int64_t d = to_seq ? (b * g.N + n) * g.L + l : (b * g.L + l) * g.N + n;
        dst[d * g.H + h] = src[e];
    }
}

// pred[b] = flat[b] . fc_w + fc_b (Model.py:101-104); d loss / d pred and the squared error for the MSE
__global__ __launch_bounds__(GB) void stgnn_head_kernel(GnGeom g, const float* __restrict__ flat, const float* __restrict__ fcw,
                                                        const float* __restrict__ fcb, const float* __restrict__ y,
                                                        float* __restrict__ pred, float* __restrict__ dpred, float* __restrict__ sqerr,
                                                        float inv_gb) {
    __shared__ float red[GB];
    const int64_t b = blockIdx.x;
    const int Q = g.N * g.L * g.H;
    float a = 0.f;
    for (int q = threadIdx.x; q < Q; q += GB) a = fmaf(flat[b * Q + q], fcw[q], a);
    red[threadIdx.x] = a;
    __syncthreads();
    for (int m = GB / 2; m > 0; m >>= 1) {
        if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float p = red[0] + fcb[0];
        pred[b] = p;
        if (y) {
            const float d = p - y[b];
            dpred[b] = 2.f * d * inv_gb;
            sqerr[b] = d * d * inv_gb;
        }
    }
}

// d flat[b][q] = dpred[b] * fc_w[q]
__global__ __launch_bounds__(GB) void stgnn_dflat_kernel(GnGeom g, const float* __restrict__ dpred, const float* __restrict__ fcw,
                                                         float* __restrict__ dflat) {
    const int Q = g.N * g.L * g.H;
    const int64_t total = g.B * Q;
    for (int64_t e = (int64_t)blockIdx.x * GB + threadIdx.x; e < total; e += (int64_t)gridDim.x * GB) dflat[e] = dpred[e / Q] * fcw[e % Q];
}

__global__ void stgnn_fill_kernel(float* p, int n, float v) {
    if ((int)threadIdx.x < n) p[threadIdx.x] = v;
}

struct GnWs {
    size_t terms, cheb, seq, hs, dhs, dseq, dcheb, dpred, sqerr, one, gru, split, total;
    size_t gru_bytes;
};
void gn_ws(const GnGeom& g, GnWs* w) {
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t R = (size_t)g.G * g.N;
    size_t o = 0;
    w->terms = o; o = al(o + R * g.KF * sizeof(float));
    w->cheb = o; o = al(o + R * g.H * sizeof(float));
    w->seq = o; o = al(o + R * g.H * sizeof(float));
    w->hs = o; o = al(o + R * g.H * sizeof(float));
    w->dhs = o; o = al(o + R * g.H * sizeof(float));
    w->dseq = o; o = al(o + R * g.H * sizeof(float));
    w->dcheb = o; o = al(o + R * g.H * sizeof(float));
    w->dpred = o; o = al(o + (size_t)(g.B > 0 ? g.B : 1) * sizeof(float));
    w->sqerr = o; o = al(o + (size_t)(g.B > 0 ? g.B : 1) * sizeof(float));
    w->one = o; o = al(o + 64 * sizeof(float));
    rulgnn_gru_shape gs{(int64_t)(g.B * g.N), g.L, g.H, g.H};
    w->gru_bytes = gru_workspace_bytes(&gs);
    w->gru = o; o = al(o + w->gru_bytes);
    const int Q = g.N * g.L * g.H;
    size_t sp = sgemm_splitk_partial_floats(g.KF, g.H);
    const size_t s2 = sgemm_splitk_need_floats(1, Q, (int)g.B), s3 = sgemm_splitk_need_floats(1, 1, (int)g.B);
    if (s2 > sp) sp = s2;
    if (s3 > sp) sp = s3;
    w->split = o; o = al(o + sp * sizeof(float));
    w->total = o;
}

inline unsigned gn_blocks(int64_t n) {
    int64_t b = (n + GB - 1) / GB;
    return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

int64_t stgnn_param_count(const rulgnn_stgnn_shape* s) {
    GnGeom g;
    if (gn_geometry(s, &g) != RULGNN_OK) return -1;
    return gn_offsets(g).total;
}

size_t stgnn_step_workspace_bytes(const rulgnn_stgnn_shape* s) {
    GnGeom g;
    if (gn_geometry(s, &g) != RULGNN_OK) return 0;
    GnWs w;
    gn_ws(g, &w);
    return w.gru_bytes == 0 && g.B > 0 ? 0 : w.total;
}

#define GN_RC(x) do { const int rc_ = (x); if (rc_ != RULGNN_OK) return rc_; } while (0)

// mode bit 0: forward (pred; with y also d pred and the loss terms), bit 1: backward (gradients; d pred from args->dpred or
// from the forward of this call)
int stgnn_run(const rulgnn_stgnn_shape* s, const rulgnn_stmsgcn_args* a, int mode, hipStream_t st) {
    GnGeom g;
    GN_RC(gn_geometry(s, &g));
    GnWs w;
    gn_ws(g, &w);
    if (!a->workspace || a->workspace_bytes < w.total) return RULGNN_EWORKSPACE;
    const GnOff o = gn_offsets(g);
    if (g.B == 0) {
        if ((mode & 2) && hipMemsetAsync(a->grads, 0, sizeof(float) * o.total, st) != hipSuccess) return RULGNN_EHIP;
        if ((mode & 2) && a->loss && hipMemsetAsync(a->loss, 0, sizeof(float), st) != hipSuccess) return RULGNN_EHIP;
        return RULGNN_OK;
    }
    char* ws = static_cast<char*>(a->workspace);
    auto Fp = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    const float* prm = a->params;
    const int64_t R = g.G * g.N;
    const int Q = g.N * g.L * g.H;
    rulgnn_gru_shape gs{(int64_t)(g.B * g.N), g.L, g.H, g.H};
    rulgnn_gru_args ga{};
    ga.w_ih = prm + o.wih; ga.w_hh = prm + o.whh; ga.b_ih = prm + o.bih; ga.b_hh = prm + o.bhh;
    ga.workspace = ws + w.gru; ga.workspace_bytes = w.gru_bytes;
    float* seq = g.L == 1 ? Fp(w.cheb) : Fp(w.seq);          // one patch: the permutation is the identity
    float* dseq = Fp(w.dseq);
    float* dcheb = g.L == 1 ? dseq : Fp(w.dcheb);
    (void)hipGetLastError();
    if (mode & 1) {
        GN_RC(stgnn_terms(s, a->x, Fp(w.terms), nullptr, st));
        GN_RC(stgnn_cheb_forward(s, Fp(w.terms), prm + o.filters, Fp(w.cheb), st));
        if (g.L > 1) hipLaunchKernelGGL(stgnn_permute_kernel, dim3(gn_blocks(R * g.H)), dim3(GB), 0, st, g, (const float*)Fp(w.cheb), seq, 1);
        ga.x = seq; ga.out = Fp(w.hs);
        GN_RC(gru_forward(&gs, &ga, st));
        const float inv_gb = 1.0f / (float)(a->global_batch > 0 ? a->global_batch : g.B);
        hipLaunchKernelGGL(stgnn_head_kernel, dim3((unsigned)g.B), dim3(GB), 0, st, g, (const float*)Fp(w.hs), prm + o.fcw, prm + o.fcb, a->y,
                           a->pred, Fp(w.dpred), Fp(w.sqerr), inv_gb);
        if (hipGetLastError() != hipSuccess) return RULGNN_EHIP;
    }
    if (mode & 2) {
        const float* dpred = a->dpred ? a->dpred : Fp(w.dpred);
        float* gr = a->grads;
        float* split = Fp(w.split);
        hipLaunchKernelGGL(stgnn_fill_kernel, dim3(1), dim3(64), 0, st, Fp(w.one), 64, 1.0f);
        // fc: d w[q] = sum_b dpred[b] flat[b][q], d b = sum_b dpred[b], d flat = dpred (x) w
        GN_RC(sgemm_splitk(dpred, 0, 1, Fp(w.hs), 1, Q, gr + o.fcw, Q, 1, Q, (int)g.B, false, split, st));
        GN_RC(sgemm_splitk(dpred, 0, 1, Fp(w.one), 0, 0, gr + o.fcb, 1, 1, 1, (int)g.B, false, split, st));
        hipLaunchKernelGGL(stgnn_dflat_kernel, dim3(gn_blocks(g.B * Q)), dim3(GB), 0, st, g, dpred, prm + o.fcw, Fp(w.dhs));
        ga.x = seq; ga.dout = Fp(w.dhs); ga.dx = dseq;
        ga.dw_ih = gr + o.wih; ga.dw_hh = gr + o.whh; ga.db_ih = gr + o.bih; ga.db_hh = gr + o.bhh;
        GN_RC(gru_backward(&gs, &ga, st));
        if (g.L > 1) hipLaunchKernelGGL(stgnn_permute_kernel, dim3(gn_blocks(R * g.H)), dim3(GB), 0, st, g, (const float*)dseq, dcheb, 0);
        GN_RC(stgnn_cheb_backward(s, Fp(w.terms), dcheb, gr + o.filters, split, sgemm_splitk_partial_floats(g.KF, g.H) * sizeof(float), st));
        if (!a->dpred && a->loss)
            (void)block_sum((const float*)Fp(w.sqerr), (int64_t)g.B, a->loss, st);
        if (hipGetLastError() != hipSuccess) return RULGNN_EHIP;
    }
    return RULGNN_OK;
}

}  // namespace rulgnn
