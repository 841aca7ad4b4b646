// STNet on gfx950 (a ChebNet user).
// Reference path replaced: STNet_model.forward -- models/STNet/Model.py:77-169 (ChebNet :7-40) -- and STNet.update,
// algorithms/algorithms.py:454-463 (loss = MSE(pred, y) + the auto-encoder's reconstruction MSE, backward, Adam).
//
//   x [bs, T * P] -> per (sample, patch) graph: |STFT| (n_fft = hop = nperseg, periodic Hann window, reflect-centred: torch.stft's
//   defaults) = N frequency nodes x f frames -> node weight = 1x1 convolution of (mean, max over the frames) -> nodes above 0.7 are
//   fully connected (A = m m^T) -> three ChebNets (K = 3, no non-linearity in between) -> auto-encoder (4 + 4 Linear layers; its
//   reconstruction error is part of the loss) -> LSTM over the T patches -> Linear.
//
// The adjacency is the outer product of a 0/1 mask, so A x is a masked node sum broadcast back (no [N, N] matrix is ever built); the
// Chebyshev terms [T0 | T1 | T2] of a layer are written side by side so that the layer is ONE matrix-core GEMM
// [rows, 3 C_in] x [3 C_in, C_out] (filters [3, C_in, C_out] are exactly that matrix) and its backward one GEMM + one split-K GEMM.
// The auto-encoder and the head are GEMMs over [bs * T] rows; the LSTM runs on the persistent kernels of bilstm.hip (one direction).
// The threshold has no gradient: the 1x1 convolution's parameters stay untouched (grad is None in the reference).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sgemm_mfma.hpp"
#include "stgcn_host.hpp"

namespace rulgnn {

namespace {

constexpr int SB = 256;
constexpr int SN_MAX_CHEB = 4, SN_MAXSEG = 64, SN_MAXFR = 64, SN_MAXNODE = 33;
constexpr float SN_THRESHOLD = 0.7f;            // Model.py:106

struct SnGeom {
    int64_t B, G, R, BT;                        // samples, graphs = B * T, node rows = G * N, patch rows = B * T
    int T, P, nseg, N, f, ncheb, A, E;
    int C[SN_MAX_CHEB + 1];                     // channel widths: f, Cheb_layers...
    int D;                                      // N * C[last]: auto-encoder width
    // flat parameter offsets
    int o_cw, o_cb, o_f[SN_MAX_CHEB], o_enc_w[4], o_enc_b[4], o_dec_w[4], o_dec_b[4], o_wih, o_whh, o_bih, o_bhh, o_lw, o_lb, pcount;
    // workspace offsets (floats)
    int64_t w_fal[SN_MAX_CHEB];          // 16-byte aligned copies of the ChebNet filters (their flat-buffer offsets are odd: the big-tile GEMM needs aligned operands)
    int64_t w_mag, w_mask, w_terms[SN_MAX_CHEB], w_out[SN_MAX_CHEB], w_enc[4], w_dec[4], w_hseq, w_dpred, w_sq, w_rsq, w_dD1, w_dD2,
        w_dA1, w_dA2, w_dterms, w_dc1, w_dc2, w_dhs, w_dH, w_one, w_split, w_lstm, total;
    int rblocks;                                // workgroups of the reconstruction-error kernel (= partial sums)
};

int sn_geometry(const rulgnn_stnet_shape* s, SnGeom* g) {
    if (!s) return RULGNN_EINVAL;
    if (s->batch < 0 || s->num_patch < 1 || s->patch_size < 2 || s->nperseg < 2 || s->num_cheb < 1 || s->lstm_hidden_dim < 1 ||
        s->autoencoder_hidden_dim < 1)
        return RULGNN_EINVAL;
    if (s->num_cheb > SN_MAX_CHEB || s->nperseg > SN_MAXSEG || (s->nperseg & 1) || s->patch_size % s->nperseg != 0 ||
        s->patch_size <= s->nperseg / 2)
        return RULGNN_EUNSUPPORTED;              // reflect padding needs more than nperseg / 2 points; frames = 1 + P / nperseg
    g->B = s->batch; g->T = s->num_patch; g->P = s->patch_size; g->nseg = s->nperseg;
    g->N = s->nperseg / 2 + 1;
    g->f = 1 + s->patch_size / s->nperseg;
    if (g->f > SN_MAXFR || g->N > SN_MAXNODE) return RULGNN_EUNSUPPORTED;
    if (s->num_nodes != g->N || s->input_dim != g->f) return RULGNN_EINVAL;         // the reference's kwargs must describe the STFT's shape
    g->ncheb = s->num_cheb; g->A = s->autoencoder_hidden_dim; g->E = s->lstm_hidden_dim;
    g->C[0] = g->f;
    for (int i = 0; i < g->ncheb; ++i) {
        if (s->cheb_layers[i] < 1 || s->cheb_layers[i] > 4096) return RULGNN_EINVAL;
        g->C[i + 1] = s->cheb_layers[i];
    }
    if (g->E > 128 || g->A > 1024) return RULGNN_EUNSUPPORTED;
    g->G = g->B * g->T; g->R = g->G * g->N; g->BT = g->G;
    g->D = g->N * g->C[g->ncheb];
    if (g->R * 3 * 4096 > ((int64_t)1 << 40)) return RULGNN_EUNSUPPORTED;
    int o = 0;
    auto tk = [&](int n) { const int r = o; o += n; return r; };
    g->o_cw = tk(2); g->o_cb = tk(1);
    for (int i = 0; i < g->ncheb; ++i) g->o_f[i] = tk(3 * g->C[i] * g->C[i + 1]);
    const int A = g->A, D = g->D;
    const int ein[4] = {D, A, A, A}, eout[4] = {A, A, A, A}, din[4] = {A, A, A, A}, dout[4] = {A, A, A, D};
    for (int i = 0; i < 4; ++i) { g->o_enc_w[i] = tk(eout[i] * ein[i]); g->o_enc_b[i] = tk(eout[i]); }
    for (int i = 0; i < 4; ++i) { g->o_dec_w[i] = tk(dout[i] * din[i]); g->o_dec_b[i] = tk(dout[i]); }
    g->o_wih = tk(4 * g->E * A); g->o_whh = tk(4 * g->E * g->E); g->o_bih = tk(4 * g->E); g->o_bhh = tk(4 * g->E);
    g->o_lw = tk(g->E * g->T); g->o_lb = tk(1);
    g->pcount = o;
    int64_t w = 0;
    auto wk = [&](int64_t n) { const int64_t r = w; w += (n + 63) & ~(int64_t)63; return r; };
    const int64_t R = g->R, BT = g->BT;
    g->w_mag = wk(R * g->f); g->w_mask = wk(R);
    int cmax = 0;
    for (int i = 0; i < g->ncheb; ++i) {
        g->w_terms[i] = wk(R * 3 * g->C[i]);
        g->w_out[i] = wk(R * g->C[i + 1]);
        if (g->C[i] > cmax) cmax = g->C[i];
        if (g->C[i + 1] > cmax) cmax = g->C[i + 1];
    }
    for (int i = 0; i < 4; ++i) g->w_enc[i] = wk(BT * A);
    for (int i = 0; i < 3; ++i) g->w_dec[i] = wk(BT * A);
    g->w_dec[3] = wk(BT * D);
    g->w_hseq = wk(BT * g->E);
    g->w_dpred = wk(g->B); g->w_sq = wk(g->B);
    g->rblocks = 1024;
    g->w_rsq = wk(g->rblocks);
    g->w_dD1 = wk(BT * D); g->w_dD2 = wk(BT * D);
    g->w_dA1 = wk(BT * A); g->w_dA2 = wk(BT * A);
    g->w_dterms = wk(R * 3 * cmax); g->w_dc1 = wk(R * cmax); g->w_dc2 = wk(R * cmax);
    g->w_dhs = wk(BT * g->E); g->w_dH = wk(BT * A);
    g->w_one = wk(64);
    for (int i = 0; i < g->ncheb; ++i) g->w_fal[i] = wk((int64_t)3 * g->C[i] * g->C[i + 1]);
    int64_t sp = 1024;
    auto need = [&](int M, int Nn, int64_t K) {
        if (K > 0x7fffffff) return;
        const int64_t v = (int64_t)sgemm_splitk_need_floats(M, Nn, (int)K);
        if (v > sp) sp = v;
    };
    for (int i = 0; i < g->ncheb; ++i) need(3 * g->C[i], g->C[i + 1], R);
    for (int i = 0; i < 4; ++i) { need(eout[i], ein[i], BT); need(1, eout[i], BT); need(dout[i], din[i], BT); need(1, dout[i], BT); }
    need(1, g->E * g->T, g->B);
    for (int i = 0; i < 4; ++i) { need((int)BT, eout[i], ein[i]); need((int)BT, din[i], dout[i]); }      // activations over a long reduction: sn_gemm_rows
    g->w_split = wk(sp);
    rulgnn_bilstm_shape ls{g->T, (int32_t)(g->B > 0 ? g->B : 1), A, g->E};
    const size_t lb = bilstm_workspace_bytes(&ls);
    if (lb == 0) return RULGNN_EUNSUPPORTED;
    g->w_lstm = wk((int64_t)(lb / sizeof(float)) + 64);
    g->total = w;
    return RULGNN_OK;
}

// ---- STFT magnitude, node weights, mask: one workgroup per (sample, patch) --------------------------------------------------------
__global__ __launch_bounds__(SB) void sn_stft_kernel(SnGeom g, const float* __restrict__ x, const float* __restrict__ prm, float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int P = g.P, ns = g.nseg, half = ns / 2, N = g.N, f = g.f, tid = threadIdx.x;
    float* xp = sm;                       // [P + ns] reflect-padded patch
    float* cs = xp + P + ns;              // [ns] cos(2 pi j / ns)
    float* sn = cs + ns;                  // [ns] sin
    float* win = sn + ns;                 // [ns] periodic Hann
    float* mg = win + ns;                 // [N][f]
    for (int j = tid; j < ns; j += SB) {
        const float a = 6.283185307179586f * (float)j / (float)ns;
        cs[j] = cosf(a);
        sn[j] = sinf(a);
        win[j] = 0.5f - 0.5f * cosf(a);
    }
    for (int64_t gi = blockIdx.x; gi < g.G; gi += gridDim.x) {
        const float* px = x + gi * P;
        for (int i = tid; i < P + ns; i += SB) {
            int q = i - half;                                   // reflect without repeating the edge sample (torch 'reflect')
            if (q < 0) q = -q;
            if (q >= P) q = 2 * (P - 1) - q;
            xp[i] = px[q];
        }
        __syncthreads();
        for (int i = tid; i < N * f; i += SB) {
            const int k = i / f, t = i % f;
            float re = 0.f, im = 0.f;
            for (int m = 0; m < ns; ++m) {
                const float v = xp[t * ns + m] * win[m];
                const int j = (k * m) % ns;
                re = fmaf(v, cs[j], re);
                im = fmaf(-v, sn[j], im);
            }
            const float a = sqrtf(re * re + im * im);
            mg[i] = a;
            ws[g.w_mag + gi * N * f + i] = a;
        }
        __syncthreads();
        if (tid < N) {
            float s = 0.f, mx = -INFINITY;
            for (int t = 0; t < f; ++t) { s += mg[tid * f + t]; mx = fmaxf(mx, mg[tid * f + t]); }
            const float nw = prm[g.o_cw] * (s / (float)f) + prm[g.o_cw + 1] * mx + prm[g.o_cb];
            ws[g.w_mask + gi * N + tid] = nw > SN_THRESHOLD ? 1.f : 0.f;
        }
        __syncthreads();
    }
}

// ---- Chebyshev terms of A = m m^T: [T0 | T1 | T2] per node row ---------------------------------------------------------------------
__global__ __launch_bounds__(SB) void sn_terms_kernel(SnGeom g, int C, const float* __restrict__ xin, const float* __restrict__ mask,
                                                      float* __restrict__ terms) {
    const int N = g.N;
    for (int64_t gi = blockIdx.x; gi < g.G; gi += gridDim.x) {
        const float* m = mask + gi * N;
        float cnt = 0.f;
        for (int n = 0; n < N; ++n) cnt += m[n];
        for (int c = threadIdx.x; c < C; c += SB) {
            float s = 0.f;
            for (int n = 0; n < N; ++n) s = fmaf(m[n], xin[(gi * N + n) * C + c], s);
            for (int n = 0; n < N; ++n) {
                const float xv = xin[(gi * N + n) * C + c];
                float* tr = terms + (gi * N + n) * 3 * C;
                const float t1 = m[n] * s;
                tr[c] = xv;
                tr[C + c] = t1;
                tr[2 * C + c] = 2.f * m[n] * (cnt * s) - xv;       // 2 A T1 - T0 with A T1 = m (m . T1) = m cnt s
            }
        }
    }
}

// dx = dT0 - dT2 + A (dT1 + 2 A dT2)      (A symmetric)
__global__ __launch_bounds__(SB) void sn_terms_bwd_kernel(SnGeom g, int C, const float* __restrict__ dterms, const float* __restrict__ mask,
                                                          float* __restrict__ dx) {
    const int N = g.N;
    for (int64_t gi = blockIdx.x; gi < g.G; gi += gridDim.x) {
        const float* m = mask + gi * N;
        for (int c = threadIdx.x; c < C; c += SB) {
            float a = 0.f;
            for (int n = 0; n < N; ++n) a = fmaf(m[n], dterms[(gi * N + n) * 3 * C + 2 * C + c], a);
            float b = 0.f;
            for (int n = 0; n < N; ++n) b = fmaf(m[n], dterms[(gi * N + n) * 3 * C + C + c] + 2.f * m[n] * a, b);
            for (int n = 0; n < N; ++n) {
                const float* dr = dterms + (gi * N + n) * 3 * C;
                dx[(gi * N + n) * C + c] = dr[c] - dr[2 * C + c] + m[n] * b;
            }
        }
    }
}

__global__ void sn_bias_act_kernel(float* __restrict__ z, const float* __restrict__ bias, int64_t rows, int C, int relu) {
    const int64_t n = rows * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = z[i] + bias[i % C];
        z[i] = relu ? fmaxf(v, 0.f) : v;
    }
}

__global__ void sn_relu_bwd_kernel(float* __restrict__ d, const float* __restrict__ h, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        d[i] = h[i] > 0.f ? d[i] : 0.f;
}

// a *= w[0], b *= w[0]: the reconstruction term's gradients under a weight other than 1 (autograd path)
__global__ void sn_scale2_kernel(float* __restrict__ a, float* __restrict__ b, int64_t n, const float* __restrict__ w) {
    const float s = w[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        a[i] *= s;
        b[i] *= s;
    }
}
__global__ void sn_add_kernel(float* __restrict__ a, const float* __restrict__ b, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) a[i] += b[i];
}

// reconstruction error: partial sums of (Yo - Yp)^2 (one per workgroup, fixed assignment), d Yp = -2 s (Yo - Yp), d Yo = +2 s (Yo - Yp)
__global__ __launch_bounds__(SB) void sn_recon_kernel(const float* __restrict__ yo, const float* __restrict__ yp, int64_t n, float scale,
                                                      float* __restrict__ dyp, float* __restrict__ dyo, float* __restrict__ part) {
    __shared__ float red[SB];
    float a = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * SB + threadIdx.x; i < n; i += (int64_t)gridDim.x * SB) {
        const float d = yo[i] - yp[i];
        a = fmaf(d, d, a);
        if (dyp) { dyp[i] = -2.f * scale * d; dyo[i] = 2.f * scale * d; }
    }
    red[threadIdx.x] = a;
    __syncthreads();
    for (int m = SB / 2; m > 0; m >>= 1) {
        if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] * scale;
}

// head: pred = hseq_flat . w + b; squared error and d loss / d pred
__global__ __launch_bounds__(SB) void sn_head_kernel(SnGeom g, const float* __restrict__ hseq, const float* __restrict__ prm, const float* __restrict__ y,
                                                     float* __restrict__ pred, float* __restrict__ ws, float inv_gb) {
    __shared__ float red[SB];
    const int n = g.E * g.T;
    for (int64_t b = blockIdx.x; b < g.B; b += gridDim.x) {
        float a = 0.f;
        for (int i = threadIdx.x; i < n; i += SB) a = fmaf(hseq[b * n + i], prm[g.o_lw + i], a);
        red[threadIdx.x] = a;
        __syncthreads();
        for (int m = SB / 2; m > 0; m >>= 1) {
            if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const float pr = red[0] + prm[g.o_lb];
            pred[b] = pr;
            if (y) {
                const float d = pr - y[b];
                ws[g.w_sq + b] = d * d * inv_gb;
                ws[g.w_dpred + b] = 2.f * d * inv_gb;
            }
        }
        __syncthreads();
    }
}

// d hseq[b][i] = dpred[b] w[i]
__global__ void sn_head_bwd_kernel(SnGeom g, const float* __restrict__ dpred, const float* __restrict__ prm, float* __restrict__ dhs) {
    const int n = g.E * g.T;
    const int64_t tot = g.B * n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (int64_t)gridDim.x * blockDim.x)
        dhs[i] = dpred[i / n] * prm[g.o_lw + i % n];
}

__global__ void sn_loss_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ loss) {
    if (threadIdx.x == 0 && blockIdx.x == 0) loss[0] = a[0] + b[0];
}

__global__ void sn_fill_kernel(float* p, int n, float v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

inline unsigned sn_grid(int64_t n) {
    int64_t b = (n + SB - 1) / SB;
    return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

int64_t stnet_param_count(const rulgnn_stnet_shape* s) {
    SnGeom g;
    return sn_geometry(s, &g) == RULGNN_OK ? g.pcount : -1;
}

size_t stnet_workspace_bytes(const rulgnn_stnet_shape* s) {
    SnGeom g;
    return sn_geometry(s, &g) == RULGNN_OK ? (size_t)g.total * sizeof(float) : 0;
}

#define SN_RC(call)                        \
    do {                                   \
        const int rc_ = (call);            \
        if (rc_ != RULGNN_OK) return rc_;  \
    } while (0)
#define SN_LAUNCH_OK()                                        \
    do {                                                      \
        if (hipGetLastError() != hipSuccess) return RULGNN_EHIP; \
    } while (0)

// mode bit 0: forward, bit 1: backward (after a forward with the same args / workspace)
int stnet_run(const rulgnn_stnet_shape* s, const rulgnn_stnet_args* a, int mode, hipStream_t st) {
    SnGeom g;
    SN_RC(sn_geometry(s, &g));
    if (a->workspace_bytes < (size_t)g.total * sizeof(float)) return RULGNN_EWORKSPACE;
    if (g.B == 0) return RULGNN_OK;
    float* ws = static_cast<float*>(a->workspace);
    const float* prm = a->params;
    const int64_t gb = a->global_batch > 0 ? a->global_batch : g.B;
    const float inv_gb = 1.0f / (float)gb;
    const int A = g.A, D = g.D, E = g.E, nc = g.ncheb;
    const int R = (int)g.R, BT = (int)g.BT;
    const int ein[4] = {D, A, A, A}, eout[4] = {A, A, A, A}, din[4] = {A, A, A, A}, dout[4] = {A, A, A, D};
    const float rscale = 1.0f / ((float)gb * (float)g.T * (float)D);          // 1 / elements of the global batch's Y_o
    const float* mask = ws + g.w_mask;
    float* split = ws + g.w_split;
    rulgnn_bilstm_shape ls{g.T, (int32_t)g.B, A, E};
    rulgnn_bilstm_args la{};
    la.x = ws + g.w_enc[3];
    la.w_ih[0] = prm + g.o_wih; la.w_hh[0] = prm + g.o_whh; la.b_ih[0] = prm + g.o_bih; la.b_hh[0] = prm + g.o_bhh;
    la.w_ih[1] = la.w_ih[0]; la.w_hh[1] = la.w_hh[0]; la.b_ih[1] = la.b_ih[0]; la.b_hh[1] = la.b_hh[0];
    la.out = ws + g.w_hseq;
    la.workspace = ws + g.w_lstm;
    la.workspace_bytes = bilstm_workspace_bytes(&ls);
    const unsigned ggrid = (unsigned)(g.G < 4096 ? g.G : 4096);
    (void)hipGetLastError();
    // [BT x K] x [K x 50] with K = 900: 32 output tiles, each walking all of K alone (45 us); split over k it is a 6-us product and a
    // 6-us fixed-order sum (the scratch need of these shapes is registered in sn_geometry)
    auto gemm_rows = [&](const float* Am, int64_t sAm, int64_t sAk, const float* Bm, int64_t sBn, int64_t sBk, float* Cm, int64_t ldc, int M, int N,
                         int K) -> int {
        if (K >= 512 && (int64_t)((M + 63) / 64) * ((N + 63) / 64) < 128) return sgemm_splitk(Am, sAm, sAk, Bm, sBn, sBk, Cm, ldc, M, N, K, false, split, st);
        return sgemm(Am, sAm, sAk, Bm, sBn, sBk, Cm, ldc, M, N, K, false, st);
    };
    // The filters as GEMM operands: the flat parameter buffer puts them behind three scalars, 12 bytes off a 16-byte boundary, and the
    // large-tile GEMM (sgemm.hip: 16-byte vector loads) falls back to the 64 x 64 fp32 tiles for a misaligned operand -- 123 instead of
    // 98 us for the [18 000 x 900] x [900 x 200] product, 98 instead of 80 us for its transpose.  A misaligned filter is copied into the
    // workspace once per call (1 MB at the reference's widths, ~3 us a copy).
    const float* fop[SN_MAX_CHEB];
    for (int i = 0; i < nc; ++i) {
        fop[i] = prm + g.o_f[i];
        const bool used_big = i > 0 || (mode & 1);                  // layer 0's transpose product is never formed (the input carries no gradient)
        if ((reinterpret_cast<uintptr_t>(fop[i]) & 15) && used_big) {
            if (hipMemcpyAsync(ws + g.w_fal[i], fop[i], sizeof(float) * 3 * g.C[i] * g.C[i + 1], hipMemcpyDeviceToDevice, st) != hipSuccess)
                return RULGNN_EHIP;
            fop[i] = ws + g.w_fal[i];
        }
    }
    if (mode & 1) {
        const size_t lds = sizeof(float) * ((size_t)g.P + g.nseg + 3 * (size_t)g.nseg + (size_t)g.N * g.f);
        if (lds > 48 * 1024) return RULGNN_EUNSUPPORTED;
        hipLaunchKernelGGL(sn_fill_kernel, dim3(1), dim3(64), 0, st, ws + g.w_one + 3, 1, 0.0f);
        hipLaunchKernelGGL(sn_stft_kernel, dim3(ggrid), dim3(SB), lds, st, g, a->x, prm, ws);
        SN_LAUNCH_OK();
        const float* cur = ws + g.w_mag;
        for (int i = 0; i < nc; ++i) {
            hipLaunchKernelGGL(sn_terms_kernel, dim3(ggrid), dim3(SB), 0, st, g, g.C[i], cur, mask, ws + g.w_terms[i]);
            SN_LAUNCH_OK();
            SN_RC(sgemm(ws + g.w_terms[i], 3 * g.C[i], 1, fop[i], 1, g.C[i + 1], ws + g.w_out[i], g.C[i + 1], R, g.C[i + 1], 3 * g.C[i],
                        false, st));
            cur = ws + g.w_out[i];
        }
        const float* h = cur;                         // Y_o as [BT, D] rows
        for (int i = 0; i < 4; ++i) {
            SN_RC(gemm_rows(h, ein[i], 1, prm + g.o_enc_w[i], ein[i], 1, ws + g.w_enc[i], eout[i], BT, eout[i], ein[i]));
            hipLaunchKernelGGL(sn_bias_act_kernel, dim3(sn_grid((int64_t)BT * eout[i])), dim3(SB), 0, st, ws + g.w_enc[i], prm + g.o_enc_b[i],
                               (int64_t)BT, eout[i], i < 3 ? 1 : 0);
            h = ws + g.w_enc[i];
        }
        for (int i = 0; i < 4; ++i) {
            SN_RC(sgemm(h, din[i], 1, prm + g.o_dec_w[i], din[i], 1, ws + g.w_dec[i], dout[i], BT, dout[i], din[i], false, st));
            hipLaunchKernelGGL(sn_bias_act_kernel, dim3(sn_grid((int64_t)BT * dout[i])), dim3(SB), 0, st, ws + g.w_dec[i], prm + g.o_dec_b[i],
                               (int64_t)BT, dout[i], i < 3 ? 1 : 0);
            h = ws + g.w_dec[i];
        }
        SN_LAUNCH_OK();
        // reconstruction error (and, when a backward follows, its gradients w.r.t. both of its arguments)
        hipLaunchKernelGGL(sn_recon_kernel, dim3(g.rblocks), dim3(SB), 0, st, (const float*)cur, (const float*)(ws + g.w_dec[3]), (int64_t)BT * D,
                           rscale, ws + g.w_dD1, ws + g.w_dD2, ws + g.w_rsq);
        (void)block_sum((const float*)(ws + g.w_rsq), (int64_t)g.rblocks, ws + g.w_one + 1, st);
        SN_LAUNCH_OK();
        SN_RC(bilstm_forward(&ls, &la, st, 1));
        hipLaunchKernelGGL(sn_head_kernel, dim3((unsigned)(g.B < 1024 ? g.B : 1024)), dim3(SB), 0, st, g, (const float*)(ws + g.w_hseq), prm, a->y,
                           a->pred, ws, inv_gb);
        if (a->recon) hipLaunchKernelGGL(sn_loss_kernel, dim3(1), dim3(1), 0, st, (const float*)(ws + g.w_one + 1), (const float*)(ws + g.w_one + 3),
                                         a->recon);          // w_one + 3 holds 0
        if (a->y && a->loss) {
            (void)block_sum((const float*)(ws + g.w_sq), g.B, ws + g.w_one + 2, st);
            hipLaunchKernelGGL(sn_loss_kernel, dim3(1), dim3(1), 0, st, (const float*)(ws + g.w_one + 1), (const float*)(ws + g.w_one + 2), a->loss);
        }
        SN_LAUNCH_OK();
    }
    if (mode & 2) {
        if (!a->grads) return RULGNN_EINVAL;
        float* gr = a->grads;
        const float* dpred = a->dpred ? a->dpred : ws + g.w_dpred;
        float* one = ws + g.w_one;
        hipLaunchKernelGGL(sn_fill_kernel, dim3(1), dim3(64), 0, st, one, 1, 1.0f);
        hipLaunchKernelGGL(sn_fill_kernel, dim3(1), dim3(64), 0, st, gr + g.o_cw, 3, 0.0f);          // the 1x1 convolution has no gradient
        // head
        SN_RC(sgemm_splitk(dpred, 0, 1, ws + g.w_hseq, 1, E * g.T, gr + g.o_lw, E * g.T, 1, E * g.T, (int)g.B, false, split, st));
        SN_RC(sgemm_splitk(dpred, 0, 1, one, 0, 0, gr + g.o_lb, 1, 1, 1, (int)g.B, false, split, st));
        hipLaunchKernelGGL(sn_head_bwd_kernel, dim3(sn_grid(g.B * E * g.T)), dim3(SB), 0, st, g, dpred, prm, ws + g.w_dhs);
        SN_LAUNCH_OK();
        la.dout = ws + g.w_dhs;
        la.dx = ws + g.w_dH;
        la.dw_ih[0] = gr + g.o_wih; la.dw_hh[0] = gr + g.o_whh; la.db_ih[0] = gr + g.o_bih; la.db_hh[0] = gr + g.o_bhh;
        la.dw_ih[1] = la.dw_ih[0]; la.dw_hh[1] = la.dw_hh[0]; la.db_ih[1] = la.db_ih[0]; la.db_hh[1] = la.db_hh[0];
        SN_RC(bilstm_backward(&ls, &la, st, 1));
        // decoder, from d Yp = w_dD1 (the reconstruction scale of a data-parallel shard is the global one: recon is in the loss with weight 1)
        if (a->recon_weight)
            hipLaunchKernelGGL(sn_scale2_kernel, dim3(sn_grid((int64_t)BT * D)), dim3(SB), 0, st, ws + g.w_dD1, ws + g.w_dD2, (int64_t)BT * D,
                               a->recon_weight);
        float* d = ws + g.w_dD1;
        float* dA[2] = {ws + g.w_dA1, ws + g.w_dA2};
        for (int i = 3; i >= 0; --i) {
            const float* hin = i > 0 ? ws + g.w_dec[i - 1] : ws + g.w_enc[3];
            if (i < 3) hipLaunchKernelGGL(sn_relu_bwd_kernel, dim3(sn_grid((int64_t)BT * dout[i])), dim3(SB), 0, st, d, (const float*)(ws + g.w_dec[i]),
                                          (int64_t)BT * dout[i]);
            SN_RC(sgemm_splitk(d, 1, dout[i], hin, 1, din[i], gr + g.o_dec_w[i], din[i], dout[i], din[i], BT, false, split, st));
            SN_RC(sgemm_splitk(one, 0, 0, d, 1, dout[i], gr + g.o_dec_b[i], dout[i], 1, dout[i], BT, false, split, st));
            float* dn = dA[i & 1];
            SN_RC(gemm_rows(d, dout[i], 1, prm + g.o_dec_w[i], 1, din[i], dn, din[i], BT, din[i], dout[i]));
            d = dn;
        }
        // d H = decoder path + LSTM path
        hipLaunchKernelGGL(sn_add_kernel, dim3(sn_grid((int64_t)BT * A)), dim3(SB), 0, st, d, (const float*)(ws + g.w_dH), (int64_t)BT * A);
        SN_LAUNCH_OK();
        const float* yo = ws + g.w_out[nc - 1];
        for (int i = 3; i >= 0; --i) {
            const float* hin = i > 0 ? ws + g.w_enc[i - 1] : yo;
            if (i < 3) hipLaunchKernelGGL(sn_relu_bwd_kernel, dim3(sn_grid((int64_t)BT * eout[i])), dim3(SB), 0, st, d, (const float*)(ws + g.w_enc[i]),
                                          (int64_t)BT * eout[i]);
            SN_RC(sgemm_splitk(d, 1, eout[i], hin, 1, ein[i], gr + g.o_enc_w[i], ein[i], eout[i], ein[i], BT, false, split, st));
            SN_RC(sgemm_splitk(one, 0, 0, d, 1, eout[i], gr + g.o_enc_b[i], eout[i], 1, eout[i], BT, false, split, st));
            float* dn = i > 0 ? (d == dA[0] ? dA[1] : dA[0]) : ws + g.w_dD1;          // the last one is [BT, D]: d Y_o through the encoder
            SN_RC(sgemm(d, eout[i], 1, prm + g.o_enc_w[i], 1, ein[i], dn, ein[i], BT, ein[i], eout[i], false, st));
            d = dn;
        }
        // d Y_o = encoder path + the reconstruction term's own gradient
        hipLaunchKernelGGL(sn_add_kernel, dim3(sn_grid((int64_t)BT * D)), dim3(SB), 0, st, d, (const float*)(ws + g.w_dD2), (int64_t)BT * D);
        SN_LAUNCH_OK();
        const float* dcur = d;                        // [R, C_last]
        float* dc[2] = {ws + g.w_dc1, ws + g.w_dc2};
        for (int i = nc - 1; i >= 0; --i) {
            const int Ci = g.C[i], Co = g.C[i + 1];
            SN_RC(sgemm_splitk(ws + g.w_terms[i], 1, 3 * Ci, dcur, 1, Co, gr + g.o_f[i], Co, 3 * Ci, Co, R, false, split, st));
            if (i == 0) break;                        // the input carries no gradient
            SN_RC(sgemm(dcur, Co, 1, fop[i], Co, 1, ws + g.w_dterms, 3 * Ci, R, 3 * Ci, Co, false, st));
            float* dn = dc[i & 1];
            hipLaunchKernelGGL(sn_terms_bwd_kernel, dim3(ggrid), dim3(SB), 0, st, g, Ci, (const float*)(ws + g.w_dterms), mask, dn);
            SN_LAUNCH_OK();
            dcur = dn;
        }
    }
    return RULGNN_OK;
}

}  // namespace rulgnn
