// One-layer GRU over MANY SHORT sequences for gfx950 (STGNN: batch*nodes = 1400 .. 100 k sequences of 1-5 patches, hidden 64):
// nn.GRU(input_dim, hidden_dim, batch_first=True), h0 = 0, gate order (r, z, n); forward and backward.
//
// Reference call sites: models/STGNN/Model.py:71,97-98 (self.gru(chebnet_output_reshaped)); the arithmetic is torch's
// (aten _thnn_fused_gru_cell): r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r * gh_n), h' = (1 - z) n + z h with
// gi = W_ih x + b_ih, gh = W_hh h + b_hh.
//
// The regime is the opposite of HAGCN's LSTM (few sequences of thousands of steps, csrc/bilstm.hip): here every step is a
// fat GEMM over all sequences, so the step loop stays on the host side and each step is
//     gh_t = h_{t-1} W_hh^T (MFMA GEMM, skipped at t = 0 where h = 0)  ->  gate kernel (one lane per (sequence, unit)).
// The input projection of ALL steps is one GEMM in front.  Backward walks the steps in reverse: gate-backward kernel
// (recomputes the gates from the saved gi / gh), d h_{t-1} += d gh_t W_hh (GEMM); the weight gradients are four split-K
// GEMMs over all (sequence, step) rows at the end (fixed reduction order: deterministic).
#include "sgemm_mfma.hpp"
#include "stgcn_host.hpp"

namespace rulgnn {

namespace {

struct GruGeom {
    int64_t S, R;      // sequences, rows = S * L
    int L, I, H, H3;
};

__host__ int gru_geometry(const rulgnn_gru_shape* s, GruGeom* g) {
    if (!s) return RULGNN_EINVAL;
    if (s->num_seq < 0 || s->seq_len < 1 || s->input_dim < 1 || s->hidden_dim < 1) return RULGNN_EINVAL;
    if (s->hidden_dim > 1024 || s->input_dim > 4096 || s->seq_len > 4096) return RULGNN_EUNSUPPORTED;
    if (s->num_seq * (int64_t)s->seq_len > ((int64_t)1 << 30) / 4) return RULGNN_EUNSUPPORTED;
    if (s->num_seq * (int64_t)s->seq_len * 3 * s->hidden_dim > ((int64_t)1 << 31) - 1) return RULGNN_EUNSUPPORTED;   // GEMM indices are int
    g->S = s->num_seq;
    g->L = s->seq_len;
    g->I = s->input_dim;
    g->H = s->hidden_dim;
    g->H3 = 3 * g->H;
    g->R = g->S * g->L;
    return RULGNN_OK;
}

__device__ __forceinline__ float sigmoidf(float v) { return 1.f / (1.f + expf(-v)); }

// step t of the forward: gates from gi (no bias yet) and gh (no bias; absent at t = 0), new state into out[:, t]
__global__ __launch_bounds__(256) void gru_gate_kernel(GruGeom g, int t, const float* __restrict__ gi, const float* __restrict__ gh,
                                                       const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                                       float* __restrict__ out, float* __restrict__ hprev) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= g.S * g.H) return;
    const int64_t s = e / g.H;
    const int j = (int)(e % g.H);
    const int64_t row = s * g.L + t;
    const float* gir = gi + row * g.H3;
    const float hr = t > 0 ? gh[s * g.H3 + j] : 0.f, hz = t > 0 ? gh[s * g.H3 + g.H + j] : 0.f, hn = t > 0 ? gh[s * g.H3 + 2 * g.H + j] : 0.f;
    const float hp = t > 0 ? out[(row - 1) * g.H + j] : 0.f;
    const float r = sigmoidf(gir[j] + b_ih[j] + hr + b_hh[j]);
    const float z = sigmoidf(gir[g.H + j] + b_ih[g.H + j] + hz + b_hh[g.H + j]);
    const float n = tanhf(gir[2 * g.H + j] + b_ih[2 * g.H + j] + r * (hn + b_hh[2 * g.H + j]));
    out[row * g.H + j] = (1.f - z) * n + z * hp;
    hprev[row * g.H + j] = hp;
}

// step t of the backward: g = dout[:, t] + dh; writes dgi[:, t] (3H), dgh[:, t] (3H) and dh = g * z (the direct path to h_{t-1})
__global__ __launch_bounds__(256) void gru_gate_bwd_kernel(GruGeom g, int t, const float* __restrict__ gi, const float* __restrict__ ghall,
                                                           const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                                           const float* __restrict__ hprev, const float* __restrict__ dout,
                                                           float* __restrict__ dh, float* __restrict__ dgi, float* __restrict__ dgh,
                                                           int first) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= g.S * g.H) return;
    const int64_t s = e / g.H;
    const int j = (int)(e % g.H);
    const int64_t row = s * g.L + t;
    const float* gir = gi + row * g.H3;
    const float* ghr = ghall + row * g.H3;                  // zeros at t = 0
    const float hp = hprev[row * g.H + j];
    const float ghn = ghr[2 * g.H + j] + b_hh[2 * g.H + j];
    const float r = sigmoidf(gir[j] + b_ih[j] + ghr[j] + b_hh[j]);
    const float z = sigmoidf(gir[g.H + j] + b_ih[g.H + j] + ghr[g.H + j] + b_hh[g.H + j]);
    const float n = tanhf(gir[2 * g.H + j] + b_ih[2 * g.H + j] + r * ghn);
    const float gg = dout[row * g.H + j] + (first ? 0.f : dh[s * g.H + j]);
    const float dn = gg * (1.f - z);
    const float dz = gg * (hp - n);
    const float dpn = dn * (1.f - n * n);
    const float dpr = dpn * ghn * r * (1.f - r);
    const float dpz = dz * z * (1.f - z);
    float* a = dgi + row * g.H3;
    float* b = dgh + row * g.H3;
    a[j] = dpr; a[g.H + j] = dpz; a[2 * g.H + j] = dpn;
    b[j] = dpr; b[g.H + j] = dpz; b[2 * g.H + j] = dpn * r;
    dh[s * g.H + j] = gg * z;
}

// gather / scatter between the step-major scratch [S, 3H] and rows (s, t) of an [S*L, 3H] tensor
__global__ __launch_bounds__(256) void gru_rows_copy_kernel(GruGeom g, int t, const float* __restrict__ src, float* __restrict__ dst,
                                                            int to_rows) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= g.S * g.H3) return;
    const int64_t s = e / g.H3;
    const int q = (int)(e % g.H3);
    const int64_t row = s * g.L + t;
    if (to_rows) dst[row * g.H3 + q] = src[e]; else dst[e] = src[row * g.H3 + q];
}

__global__ void gru_fill_kernel(float* p, int64_t n, float v) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < n) p[e] = v;
}

struct GruWs {
    size_t gi, gh, hprev, ghstep, dgi, dgh, dh, dgstep, one, split, total;
};

void gru_ws(const GruGeom& g, GruWs* w) {
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t o = 0;
    const size_t R = (size_t)g.R, S = (size_t)g.S;
    w->gi = o; o = al(o + R * g.H3 * sizeof(float));           // tape: W_ih x (no bias)
    w->gh = o; o = al(o + R * g.H3 * sizeof(float));           // tape: W_hh h_{t-1} (no bias), zeros at t = 0
    w->hprev = o; o = al(o + R * g.H * sizeof(float));         // tape: h_{t-1}
    w->ghstep = o; o = al(o + S * g.H3 * sizeof(float));
    w->dgi = o; o = al(o + R * g.H3 * sizeof(float));
    w->dgh = o; o = al(o + R * g.H3 * sizeof(float));
    w->dh = o; o = al(o + S * g.H * sizeof(float));
    w->dgstep = o; o = al(o + S * g.H3 * sizeof(float));
    w->one = o; o = al(o + R * sizeof(float));
    const int mx = g.I > g.H ? g.I : g.H;
    w->split = o; o = al(o + sgemm_splitk_partial_floats(g.H3, mx) * sizeof(float));
    w->total = o;
}

inline unsigned blocks(int64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

size_t gru_workspace_bytes(const rulgnn_gru_shape* s) {
    GruGeom g;
    if (gru_geometry(s, &g) != RULGNN_OK) return 0;
    GruWs w;
    gru_ws(g, &w);
    return w.total;
}

#define GRU_RC(x) do { const int rc_ = (x); if (rc_ != RULGNN_OK) return rc_; } while (0)

int gru_forward(const rulgnn_gru_shape* s, const rulgnn_gru_args* a, hipStream_t st) {
    GruGeom g;
    GRU_RC(gru_geometry(s, &g));
    GruWs w;
    gru_ws(g, &w);
    if (!a->workspace || a->workspace_bytes < w.total) return RULGNN_EWORKSPACE;
    if (g.S == 0) return RULGNN_OK;
    char* ws = static_cast<char*>(a->workspace);
    auto Fp = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    // gi[row][q] = sum_i x[row][i] W_ih[q][i]
    GRU_RC(sgemm(a->x, g.I, 1, a->w_ih, g.I, 1, Fp(w.gi), g.H3, (int)g.R, g.H3, g.I, false, st));
    if (hipMemsetAsync(Fp(w.gh), 0, (size_t)g.R * g.H3 * sizeof(float), st) != hipSuccess) return RULGNN_EHIP;
    (void)hipGetLastError();
    for (int t = 0; t < g.L; ++t) {
        if (t > 0) {
            // gh[s][q] = sum_j h_{t-1}[s][j] W_hh[q][j]; h_{t-1} = out[s, t-1, :] (row stride L*H)
            GRU_RC(sgemm(a->out + (int64_t)(t - 1) * g.H, (int64_t)g.L * g.H, 1, a->w_hh, g.H, 1, Fp(w.ghstep), g.H3, (int)g.S, g.H3, g.H,
                         false, st));
            hipLaunchKernelGGL(gru_rows_copy_kernel, dim3(blocks(g.S * g.H3)), dim3(256), 0, st, g, t, (const float*)Fp(w.ghstep), Fp(w.gh), 1);
        }
        hipLaunchKernelGGL(gru_gate_kernel, dim3(blocks(g.S * g.H)), dim3(256), 0, st, g, t, (const float*)Fp(w.gi), (const float*)Fp(w.ghstep),
                           a->b_ih, a->b_hh, a->out, Fp(w.hprev));
    }
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

int gru_backward(const rulgnn_gru_shape* s, const rulgnn_gru_args* a, hipStream_t st) {
    GruGeom g;
    GRU_RC(gru_geometry(s, &g));
    GruWs w;
    gru_ws(g, &w);
    if (!a->workspace || a->workspace_bytes < w.total) return RULGNN_EWORKSPACE;
    if (g.S == 0) {
        if (hipMemsetAsync(a->dw_ih, 0, sizeof(float) * g.H3 * g.I, st) != hipSuccess || hipMemsetAsync(a->dw_hh, 0, sizeof(float) * g.H3 * g.H, st) != hipSuccess ||
            hipMemsetAsync(a->db_ih, 0, sizeof(float) * g.H3, st) != hipSuccess || hipMemsetAsync(a->db_hh, 0, sizeof(float) * g.H3, st) != hipSuccess)
            return RULGNN_EHIP;
        return RULGNN_OK;
    }
    char* ws = static_cast<char*>(a->workspace);
    auto Fp = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    (void)hipGetLastError();
    for (int t = g.L - 1; t >= 0; --t) {
        hipLaunchKernelGGL(gru_gate_bwd_kernel, dim3(blocks(g.S * g.H)), dim3(256), 0, st, g, t, (const float*)Fp(w.gi), (const float*)Fp(w.gh),
                           a->b_ih, a->b_hh, (const float*)Fp(w.hprev), a->dout, Fp(w.dh), Fp(w.dgi), Fp(w.dgh), t == g.L - 1 ? 1 : 0);
        if (t > 0) {
            // dh[s][j] += sum_q dgh_t[s][q] W_hh[q][j]
            hipLaunchKernelGGL(gru_rows_copy_kernel, dim3(blocks(g.S * g.H3)), dim3(256), 0, st, g, t, (const float*)Fp(w.dgh), Fp(w.dgstep), 0);
            GRU_RC(sgemm(Fp(w.dgstep), g.H3, 1, a->w_hh, 1, g.H, Fp(w.dh), g.H, (int)g.S, g.H, g.H3, true, st));
        }
    }
    hipLaunchKernelGGL(gru_fill_kernel, dim3(blocks(g.R)), dim3(256), 0, st, Fp(w.one), g.R, 1.0f);
    if (hipGetLastError() != hipSuccess) return RULGNN_EHIP;
    float* split = Fp(w.split);
    // dW_ih[q][i] = sum_row dgi[row][q] x[row][i];  dW_hh[q][j] = sum_row dgh[row][q] hprev[row][j];  biases: column sums
    GRU_RC(sgemm_splitk(Fp(w.dgi), 1, g.H3, a->x, 1, g.I, a->dw_ih, g.I, g.H3, g.I, (int)g.R, false, split, st));
    GRU_RC(sgemm_splitk(Fp(w.dgh), 1, g.H3, Fp(w.hprev), 1, g.H, a->dw_hh, g.H, g.H3, g.H, (int)g.R, false, split, st));
    GRU_RC(sgemm_splitk(Fp(w.dgi), 1, g.H3, Fp(w.one), 0, 1, a->db_ih, 1, g.H3, 1, (int)g.R, false, split, st));
    GRU_RC(sgemm_splitk(Fp(w.dgh), 1, g.H3, Fp(w.one), 0, 1, a->db_hh, 1, g.H3, 1, (int)g.R, false, split, st));
    if (a->dx)   // dx[row][i] = sum_q dgi[row][q] W_ih[q][i]
        GRU_RC(sgemm(Fp(w.dgi), g.H3, 1, a->w_ih, 1, g.I, a->dx, g.I, (int)g.R, g.I, g.H3, false, st));
    return RULGNN_OK;
}

}  // namespace rulgnn
