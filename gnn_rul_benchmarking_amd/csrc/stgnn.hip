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

}  // namespace rulgnn
