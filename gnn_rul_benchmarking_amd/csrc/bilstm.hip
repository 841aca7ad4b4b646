// Persistent bidirectional LSTM layer for gfx950 (the HAGCN encoder): out = h_forward + h_backward, forward and BPTT.
//
// Reference: Bi_LSTM_Standard, models/HAGCN/Model.py:26-73 (three nn.LSTM(bidirectional=True) whose two halves are
// summed), called with the reference's axis convention (Model.py:153-157): "batch" = num_patch (1..5 sequences), sequence
// length T = batch_size * num_node (1 400 .. 5 000 steps).  That is the worst case for a library RNN (a handful of kernel
// launches per time step); here the recurrence is ONE persistent workgroup per (direction, sequence):
//   * the input projection x W_ih^T of all time steps is one MFMA GEMM up front (sgemm_mfma.hpp);
//   * each of the 4H threads keeps its row of W_hh in registers for the whole sequence, reads h from LDS (broadcast),
//     and the H cell threads apply the gate non-linearities -- two workgroup barriers per time step, no launches;
//   * BPTT mirrors it with the transposed rows in registers; the per-step gate gradients are written out and the weight
//     gradients are split-K GEMMs over the whole sequence (dW_ih = dG^T x, dW_hh = dG^T h_prev), dx = dG W_ih.
// The 2 * num_patch recurrences of a layer run concurrently on different CUs; layers are sequentially dependent.
#include <utility>

#include "sgemm_mfma.hpp"
#include "stgcn_host.hpp"

namespace rulgnn {

namespace {

struct LstmGeom {
    int64_t T;                  // time steps (batch_size * num_node)
    int Bq, I, H, H4;           // sequences, input width, hidden width
    int64_t rows;               // Bq * T
    // workspace offsets (floats); every per-direction tensor is [dir][q][t][.]
    int64_t o_gi, o_gates, o_c, o_h, o_hprev, o_dgates, o_one, o_split, total;
};

__host__ int lstm_geometry(const rulgnn_bilstm_shape* s, LstmGeom* g) {
    if (!s) return RULGNN_EINVAL;
    if (s->seq_len < 1 || s->num_seq < 1 || s->input_dim < 1 || s->hidden_dim < 1) return RULGNN_EINVAL;
    if (s->hidden_dim > 128 || s->input_dim > 1024 || s->num_seq > 65535) return RULGNN_EUNSUPPORTED;
    if (s->seq_len * (int64_t)s->num_seq * 4 * s->hidden_dim > ((int64_t)1 << 30)) return RULGNN_EUNSUPPORTED;
    g->T = s->seq_len;
    g->Bq = s->num_seq;
    g->I = s->input_dim;
    g->H = s->hidden_dim;
    g->H4 = 4 * g->H;
    g->rows = g->T * g->Bq;
    int64_t o = 0;
    auto tk = [&](int64_t n) { const int64_t r = o; o += (n + 63) & ~(int64_t)63; return r; };
    g->o_gi = tk(2 * g->rows * g->H4);
    g->o_gates = tk(2 * g->rows * g->H4);
    g->o_c = tk(2 * g->rows * g->H);
    g->o_h = tk(2 * g->rows * g->H);
    g->o_hprev = tk(2 * g->rows * g->H);
    g->o_dgates = tk(2 * g->rows * g->H4);
    g->o_one = tk(64);
    int64_t mx = 1;
    for (const auto& mn : {std::pair<int, int>(g->H4, g->I), std::pair<int, int>(g->H4, g->H), std::pair<int, int>(1, g->H4)}) {
        const int64_t v = (int64_t)sgemm_splitk_need_floats(mn.first, mn.second, (int)g->rows);
        if (v > mx) mx = v;
    }
    g->o_split = tk(mx);
    g->total = o;
    return RULGNN_OK;
}

// The gate non-linearities sit on the critical path of every sequential step (0.27 us of 0.95 with libm's expf / tanhf and
// IEEE division): hardware exp2 and reciprocal instead (v_exp_f32, v_rcp_f32: ~1 ulp each, ~2e-7 absolute on the outputs).
__device__ inline float sigm(float v) { return __frcp_rn(1.0f + __expf(-v)); }
__device__ inline float tanh_gate(float v) {
    const float a = fabsf(v);
    const float t = 1.0f - 2.0f * __frcp_rn(1.0f + __expf(2.0f * a));        // exp overflow -> rcp(inf) = 0 -> 1
    return copysignf(t, v);
}

// ---------------------------------------------------------------------------------------------------
// forward recurrence: workgroup = (direction, sequence); thread j < 4H owns gate row j of W_hh
// ---------------------------------------------------------------------------------------------------
template <int HMAX>
__global__ __launch_bounds__(4 * HMAX) void lstm_forward_kernel(LstmGeom g, const float* __restrict__ gi, const float* __restrict__ w_hh0,
                                    const float* __restrict__ w_hh1, const float* __restrict__ b_ih0, const float* __restrict__ b_hh0,
                                    const float* __restrict__ b_ih1, const float* __restrict__ b_hh1, float* __restrict__ gates,
                                    float* __restrict__ cseq, float* __restrict__ hseq, float* __restrict__ hprev) {
    __shared__ __attribute__((aligned(16))) float hs[HMAX];
    __shared__ float gl[4 * HMAX];
    const int dir = blockIdx.x / g.Bq, q = blockIdx.x % g.Bq;
    const int j = threadIdx.x, H = g.H, H4 = g.H4;
    const bool live = j < H4;
    const float* w_hh = dir ? w_hh1 : w_hh0;
    float w[HMAX];
#pragma unroll
    for (int k = 0; k < HMAX; ++k) w[k] = (live && k < H) ? w_hh[j * H + k] : 0.f;
    const float bias = live ? (dir ? b_ih1[j] + b_hh1[j] : b_ih0[j] + b_hh0[j]) : 0.f;
    const bool is_tanh = j >= 2 * H && j < 3 * H;                      // gate order (i, f, g, o): g is the tanh row
    const float gate_in = is_tanh ? 1.0f : 0.5f;
    const int64_t base = ((int64_t)dir * g.Bq + q) * g.T;          // row of (dir, q, t = 0)
    if (j < HMAX) hs[j] = 0.f;
    float c = 0.f;
    __syncthreads();
    int64_t t = dir ? g.T - 1 : 0;
    const int64_t dt = dir ? -1 : 1;
    float nxt = live ? gi[(base + t) * H4 + j] : 0.f;
    for (int64_t s = 0; s < g.T; ++s, t += dt) {
        float acc = nxt + bias;
        if (live && s + 1 < g.T) nxt = gi[(base + t + dt) * H4 + j];       // next step's input projection, ahead of the matvec
#pragma unroll
        for (int k = 0; k < HMAX; k += 4) {
            const float4 hv = *reinterpret_cast<const float4*>(&hs[k]);
            acc = fmaf(w[k], hv.x, acc);
            acc = fmaf(w[k + 1], hv.y, acc);
            acc = fmaf(w[k + 2], hv.z, acc);
            acc = fmaf(w[k + 3], hv.w, acc);
        }
        // every gate row applies its own non-linearity before the barrier (all four wavefronts busy; sigmoid written as
        // 0.5 + 0.5 tanh(x / 2) so that the tanh row and the sigmoid rows run the same instructions): behind the barrier a
        // unit only combines its four gates and takes tanh(c)
        if (live) {
            const float tt = tanh_gate(acc * gate_in);
            gl[j] = is_tanh ? tt : fmaf(0.5f, tt, 0.5f);
        }
        if (j < H) hprev[(base + t) * H + j] = hs[j];
        __syncthreads();
        if (j < H) {
            const float ig = gl[j], fg = gl[H + j], gg = gl[2 * H + j], og = gl[3 * H + j];
            c = fmaf(fg, c, ig * gg);
            const float h = og * tanh_gate(c);
            float* gr = gates + (base + t) * H4;
            gr[j] = ig; gr[H + j] = fg; gr[2 * H + j] = gg; gr[3 * H + j] = og;
            cseq[(base + t) * H + j] = c;
            hseq[(base + t) * H + j] = h;
            hs[j] = h;
        }
        __syncthreads();
    }
}

// out[q][t][:] = h_fwd + h_bwd   (Model.py:59-61)
__global__ void lstm_sum_kernel(LstmGeom g, const float* __restrict__ hseq, float* __restrict__ out) {
    const int64_t n = g.rows * g.H;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
        out[e] = hseq[e] + hseq[n + e];
}

// ---------------------------------------------------------------------------------------------------
// BPTT: thread tid < 4H = (part p, unit k) keeps W_hh[p*H + :, k] in registers
// ---------------------------------------------------------------------------------------------------
template <int HMAX>
__global__ __launch_bounds__(4 * HMAX) void lstm_backward_kernel(LstmGeom g, const float* __restrict__ w_hh0, const float* __restrict__ w_hh1,
                                     const float* __restrict__ gates, const float* __restrict__ cseq, const float* __restrict__ dout,
                                     float* __restrict__ dgates) {
    __shared__ __attribute__((aligned(16))) float dgl[4 * HMAX];
    __shared__ float part[4][HMAX];
    const int dir = blockIdx.x / g.Bq, q = blockIdx.x % g.Bq;
    const int tid = threadIdx.x, H = g.H, H4 = g.H4;
    const bool live = tid < H4;
    const int p = live ? tid / H : 0, k = live ? tid - p * H : 0;
    const float* w_hh = dir ? w_hh1 : w_hh0;
    float wt[HMAX];
#pragma unroll
    for (int jj = 0; jj < HMAX; ++jj) wt[jj] = (live && jj < H) ? w_hh[(p * H + jj) * H + k] : 0.f;
    for (int e = tid; e < 4 * HMAX; e += blockDim.x) dgl[e] = 0.f;
    const int64_t base = ((int64_t)dir * g.Bq + q) * g.T;
    const int64_t obase = (int64_t)q * g.T;                          // dout is [q][t][H], shared by both directions
    float dh = 0.f, dc = 0.f;
    __syncthreads();
    // opposite to the forward order of this direction
    int64_t t = dir ? 0 : g.T - 1;
    const int64_t dt = dir ? 1 : -1;
    // The tape of a step (gates, cell states, incoming gradient) does not depend on the recurrence: it is fetched one step
    // ahead, so that a step does not start with a global-memory round trip.
    float n_ig = 0.f, n_fg = 0.f, n_gg = 0.f, n_og = 0.f, n_ct = 0.f, n_cp = 0.f, n_do = 0.f;
    auto fetch = [&](int64_t tt) {
        const float* gr = gates + (base + tt) * H4;
        n_ig = gr[tid]; n_fg = gr[H + tid]; n_gg = gr[2 * H + tid]; n_og = gr[3 * H + tid];
        n_ct = cseq[(base + tt) * H + tid];
        // cell state entering step tt: the forward visited tt - dt_fwd before tt, i.e. tt + dt in this loop's direction
        const bool first = dir ? (tt == g.T - 1) : (tt == 0);
        n_cp = first ? 0.f : cseq[(base + tt + dt) * H + tid];
        n_do = dout[(obase + tt) * H + tid];
    };
    // ... and everything of a step that does not involve dh / dc is folded into six coefficients while the tape is in
    // registers, off the recurrence's critical path (they are prepared behind the matvec of the step before):
    //   dct = dht * k_c + dc;  (di, df, dg) = dct * (k_i, k_f, k_g);  do = dht * k_o;  dc' = dct * k_fg
    float k_c = 0.f, k_i = 0.f, k_f = 0.f, k_g = 0.f, k_o = 0.f, k_fg = 0.f, k_do = 0.f;
    auto prepare = [&]() {
        const float tc = tanh_gate(n_ct);
        k_c = n_og * (1.0f - tc * tc);
        k_i = n_gg * n_ig * (1.0f - n_ig);
        k_f = n_cp * n_fg * (1.0f - n_fg);
        k_g = n_ig * (1.0f - n_gg * n_gg);
        k_o = tc * n_og * (1.0f - n_og);
        k_fg = n_fg;
        k_do = n_do;
    };
    if (tid < H && g.T > 0) { fetch(t); prepare(); }
    for (int64_t s = 0; s < g.T; ++s, t += dt) {
        if (tid < H) {
            const float dht = k_do + dh;
            const float dct = fmaf(dht, k_c, dc);
            const float di = dct * k_i, df = dct * k_f, dg = dct * k_g, dov = dht * k_o;
            dc = dct * k_fg;
            if (s + 1 < g.T) fetch(t + dt);
            dgl[tid] = di; dgl[H + tid] = df; dgl[2 * H + tid] = dg; dgl[3 * H + tid] = dov;
            float* dr = dgates + (base + t) * H4;
            dr[tid] = di; dr[H + tid] = df; dr[2 * H + tid] = dg; dr[3 * H + tid] = dov;
        }
        __syncthreads();
        if (live) {                                   // d h_prev[k] = sum over the four gate blocks of W_hh[block]^T d gate
            float a = 0.f;
            const float* dsrc = dgl + p * H;
            if ((H & 3) == 0) {                                                   // 16-byte LDS reads (p * H floats is a multiple of 16 B)
#pragma unroll
                for (int jj = 0; jj < HMAX; jj += 4) {                            // wt is zero beyond H; dsrc stays inside dgl
                    const float4 dv = *reinterpret_cast<const float4*>(dsrc + jj);
                    a = fmaf(dv.x, wt[jj], a);
                    a = fmaf(dv.y, wt[jj + 1], a);
                    a = fmaf(dv.z, wt[jj + 2], a);
                    a = fmaf(dv.w, wt[jj + 3], a);
                }
            } else {
#pragma unroll
                for (int jj = 0; jj < HMAX; ++jj) a = fmaf(dsrc[jj], wt[jj], a);
            }
            part[p][k] = a;
        }
        if (tid < H && s + 1 < g.T) prepare();        // the next step's coefficients (its tape was requested at the top of this step)
        __syncthreads();
        if (tid < H) dh = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
    }
}

__global__ void lstm_fill_one_kernel(float* p) { p[0] = 1.f; }
// db_ih = db_hh: copy
__global__ void lstm_copy_kernel(const float* __restrict__ a, float* __restrict__ b, int n) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) b[e] = a[e];
}

}  // namespace

size_t bilstm_workspace_bytes(const rulgnn_bilstm_shape* s) {
    LstmGeom g;
    if (lstm_geometry(s, &g) != RULGNN_OK) return 0;
    return (size_t)g.total * sizeof(float);
}

#define LS_RC(call)                        \
    do {                                   \
        const int rc_ = (call);            \
        if (rc_ != RULGNN_OK) return rc_;  \
    } while (0)

// ndir = 2: the bidirectional layer of the ABI; ndir = 1: the forward direction alone (nn.LSTM(bidirectional=False), e.g. RGCNU's
// TDL, models/RGCNU/Model.py:46-53) -- same kernels, half the workgroups, out = h of direction 0
int bilstm_forward(const rulgnn_bilstm_shape* s, const rulgnn_bilstm_args* a, hipStream_t st, int ndir) {
    LstmGeom g;
    LS_RC(lstm_geometry(s, &g));
    if (a->workspace_bytes < (size_t)g.total * sizeof(float)) return RULGNN_EWORKSPACE;
    float* ws = static_cast<float*>(a->workspace);
    (void)hipGetLastError();
    const int R = (int)g.rows;
    // input projections of both directions: gi[dir] = x W_ih[dir]^T   ([rows, I] x [I, 4H])
    if (ndir != 1 && ndir != 2) return RULGNN_EINVAL;
    for (int d = 0; d < ndir; ++d)
        LS_RC(sgemm(a->x, g.I, 1, a->w_ih[d], g.I, 1, ws + g.o_gi + (int64_t)d * g.rows * g.H4, g.H4, R, g.H4, g.I, false, st));
    const int threads = (g.H4 + 63) & ~63;
    const int d1 = ndir == 2 ? 1 : 0;                 // the one-direction launch never selects direction 1
    if (g.H <= 64)
        hipLaunchKernelGGL(lstm_forward_kernel<64>, dim3(ndir * g.Bq), dim3(threads), 0, st, g, (const float*)(ws + g.o_gi), a->w_hh[0],
                           a->w_hh[d1], a->b_ih[0], a->b_hh[0], a->b_ih[d1], a->b_hh[d1], ws + g.o_gates, ws + g.o_c, ws + g.o_h,
                           ws + g.o_hprev);
    else
        hipLaunchKernelGGL(lstm_forward_kernel<128>, dim3(ndir * g.Bq), dim3(threads), 0, st, g, (const float*)(ws + g.o_gi), a->w_hh[0],
                           a->w_hh[d1], a->b_ih[0], a->b_hh[0], a->b_ih[d1], a->b_hh[d1], ws + g.o_gates, ws + g.o_c, ws + g.o_h,
                           ws + g.o_hprev);
    if (ndir == 2) hipLaunchKernelGGL(lstm_sum_kernel, dim3(1024), dim3(256), 0, st, g, (const float*)(ws + g.o_h), a->out);
    else hipLaunchKernelGGL(lstm_copy_kernel, dim3((unsigned)((g.rows * g.H + 255) / 256)), dim3(256), 0, st, (const float*)(ws + g.o_h), a->out, (int)(g.rows * g.H));
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

int bilstm_backward(const rulgnn_bilstm_shape* s, const rulgnn_bilstm_args* a, hipStream_t st, int ndir) {
    LstmGeom g;
    LS_RC(lstm_geometry(s, &g));
    if (a->workspace_bytes < (size_t)g.total * sizeof(float)) return RULGNN_EWORKSPACE;
    float* ws = static_cast<float*>(a->workspace);
    (void)hipGetLastError();
    const int R = (int)g.rows, H = g.H, H4 = g.H4, I = g.I;
    const int threads = (H4 + 63) & ~63;
    if (ndir != 1 && ndir != 2) return RULGNN_EINVAL;
    const int d1 = ndir == 2 ? 1 : 0;
    if (H <= 64)
        hipLaunchKernelGGL(lstm_backward_kernel<64>, dim3(ndir * g.Bq), dim3(threads), 0, st, g, a->w_hh[0], a->w_hh[d1],
                           (const float*)(ws + g.o_gates), (const float*)(ws + g.o_c), a->dout, ws + g.o_dgates);
    else
        hipLaunchKernelGGL(lstm_backward_kernel<128>, dim3(ndir * g.Bq), dim3(threads), 0, st, g, a->w_hh[0], a->w_hh[d1],
                           (const float*)(ws + g.o_gates), (const float*)(ws + g.o_c), a->dout, ws + g.o_dgates);
    float* one = ws + g.o_one;
    float* split = ws + g.o_split;
    hipLaunchKernelGGL(lstm_fill_one_kernel, dim3(1), dim3(1), 0, st, one);
    for (int d = 0; d < ndir; ++d) {
        const float* dg = ws + g.o_dgates + (int64_t)d * g.rows * H4;
        // dW_ih = dG^T x ; dW_hh = dG^T h_prev ; db = column sums ; dx (+)= dG W_ih
        LS_RC(sgemm_splitk(dg, 1, H4, a->x, 1, I, a->dw_ih[d], I, H4, I, R, false, split, st));
        LS_RC(sgemm_splitk(dg, 1, H4, ws + g.o_hprev + (int64_t)d * g.rows * H, 1, H, a->dw_hh[d], H, H4, H, R, false, split, st));
        LS_RC(sgemm_splitk(one, 0, 0, dg, 1, H4, a->db_ih[d], H4, 1, H4, R, false, split, st));
        hipLaunchKernelGGL(lstm_copy_kernel, dim3((H4 + 255) / 256), dim3(256), 0, st, (const float*)a->db_ih[d], a->db_hh[d], H4);
        if (a->dx) LS_RC(sgemm(dg, H4, 1, a->w_ih[d], 1, I, a->dx, I, R, I, H4, d == 1, st));
    }
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

}  // namespace rulgnn
