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

#include "async_mem.hpp"
#include "aux_stream.hpp"
#include "sgemm_mfma.hpp"
#include "stgcn_host.hpp"

namespace rulgnn {

namespace {

struct LstmGeom {
    int64_t T;                  // time steps (batch_size * num_node)
    int Bq, I, H, H4;           // sequences, input width, hidden width
    int64_t rows;               // Bq * T
    // workspace offsets (floats); every per-direction tensor is [dir][q][t][.]
    int64_t o_gi, o_gates, o_c, o_tc, o_h, o_hprev, o_dgates, o_one, o_split, o_split_floats, total;
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
    g->o_tc = tk(2 * g->rows * g->H);
    g->o_h = tk(2 * g->rows * g->H);
    g->o_hprev = tk(2 * g->rows * g->H);
    g->o_dgates = tk(2 * g->rows * g->H4);
    g->o_one = tk(64);
    int64_t mx = 1;
    for (const auto& mn : {std::pair<int, int>(g->H4, g->I), std::pair<int, int>(g->H4, g->H), std::pair<int, int>(1, g->H4)}) {
        const int64_t v = (int64_t)sgemm_splitk_need_floats(mn.first, mn.second, (int)g->rows);
        if (v > mx) mx = v;
    }
    {   // ... or the layer's eight parameter-gradient products at once (bilstm_backward: sgemm_splitk_batch)
        const int R = (int)(g->rows > 0 ? g->rows : 1);
        SplitKJob dims[8];
        for (int d = 0; d < 2; ++d) {
            dims[4 * d + 0] = SplitKJob{nullptr, 0, 0, nullptr, 0, 0, nullptr, 0, g->H4, g->I, R};
            dims[4 * d + 1] = SplitKJob{nullptr, 0, 0, nullptr, 0, 0, nullptr, 0, g->H4, g->H, R};
            dims[4 * d + 2] = dims[4 * d + 3] = SplitKJob{nullptr, 0, 0, nullptr, 0, 0, nullptr, 0, 1, g->H4, R};
        }
        const int64_t v = (int64_t)sgemm_splitk_batch_floats(dims, 8);
        if (v > mx) mx = v;
    }
    g->o_split_floats = mx;
    g->o_split = tk(mx);
    g->total = o;
    return RULGNN_OK;
}

// The gate non-linearities sit on the critical path of every sequential step: hardware exp2 and reciprocal (v_exp_f32, v_rcp_f32: ~1 ulp
// each, ~2e-7 absolute on the outputs), and the argument scaling folded into the weights.  With e = 2^(k x):
//   sigmoid(x) = 1 / (1 + e), k = -log2(e);     tanh(x) = 1 - 2 / (1 + e), k = 2 log2(e)   (e = inf -> rcp = 0 -> 1, e = 0 -> -1)
// i.e. gate = fma(rcp(1 + exp2(k x)), m, b) with (m, b) = (1, 0) / (-2, 1): one instruction sequence for all four gate rows.
constexpr float LOG2E = 1.4426950408889634f;
__device__ __forceinline__ float rcp1p_exp2(float v) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v)); }
__device__ __forceinline__ float tanh_fast(float v) { return fmaf(-2.0f, rcp1p_exp2(v * (2.0f * LOG2E)), 1.0f); }

// One sequential step used to cost 1.6 us (H = 64) although its arithmetic is ~0.2 us: __syncthreads() waits for vmcnt(0), i.e. for the
// step's global stores (and the prefetched input row) to complete -- two HBM round trips per step.  Here the step barrier only waits
// for the LDS (lgkmcnt) and global traffic is asynchronous and counted by hand (async_mem.hpp): tape stores drain behind the
// recurrence, and the rows a step reads are requested LSTM_AHEAD steps earlier by LDS-DMA into a ring each wavefront reads back itself.
constexpr int LSTM_AHEAD = 8;       // even (the LDS vectors of the recurrence are double-buffered by step parity)

// acc += w * (lane N of this lane's row of 16 in hc): the matvec's broadcast.  Reading the whole vector in every lane (LDS broadcast
// reads) is bound by the LDS return path -- 4H lanes x H floats at 128 B/clk: 0.24 us per step at H = 64, 0.98 us at H = 128, what the
// recurrence measured -- so a lane reads H/16 floats (lane l: element 16c + l % 16 of chunk c) and the DPP row broadcast of gfx90a+
// feeds the FMA: 16x less LDS traffic, 4 issue cycles per FMA.
template <int N>
__device__ __forceinline__ void fma_row_bcast(float& acc, float hc, float w) {
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(hc), "v"(w), "n"(N));
}
template <int HMAX>
__device__ __forceinline__ void chunks_load(float (&hc)[HMAX / 16], const float* vec, int lane16) {
#pragma unroll
    for (int c = 0; c < HMAX / 16; ++c) hc[c] = vec[16 * c + lane16];
}
// (KLIVE <= HMAX: the hidden width when it is known at compile time -- the products beyond it meet zero weights and are not issued)
// (Measured and rejected at H = 120, two wavefronts per SIMD: a part of the products as plain FMAs on values every lane reads from LDS
// itself -- 2.4 VALU cycles instead of 4, the LDS return path being idle -- is SLOWER: HAGCN step 11.25 ms all-DPP, 11.5 / 11.8 /
// 12.2 ms with 24 / 40 / 56 products moved.)
template <int HMAX, int KLIVE>
__device__ __forceinline__ float matvec_row_bcast(const float (&w)[HMAX], const float (&hc)[HMAX / 16], float a0) {
    float a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int c = 0; c < (KLIVE + 15) / 16; ++c) {
        if (16 * c + 0 < KLIVE) fma_row_bcast<0>(a0, hc[c], w[16 * c + 0]);
        if (16 * c + 1 < KLIVE) fma_row_bcast<1>(a1, hc[c], w[16 * c + 1]);
        if (16 * c + 2 < KLIVE) fma_row_bcast<2>(a2, hc[c], w[16 * c + 2]);
        if (16 * c + 3 < KLIVE) fma_row_bcast<3>(a3, hc[c], w[16 * c + 3]);
        if (16 * c + 4 < KLIVE) fma_row_bcast<4>(a0, hc[c], w[16 * c + 4]);
        if (16 * c + 5 < KLIVE) fma_row_bcast<5>(a1, hc[c], w[16 * c + 5]);
        if (16 * c + 6 < KLIVE) fma_row_bcast<6>(a2, hc[c], w[16 * c + 6]);
        if (16 * c + 7 < KLIVE) fma_row_bcast<7>(a3, hc[c], w[16 * c + 7]);
        if (16 * c + 8 < KLIVE) fma_row_bcast<8>(a0, hc[c], w[16 * c + 8]);
        if (16 * c + 9 < KLIVE) fma_row_bcast<9>(a1, hc[c], w[16 * c + 9]);
        if (16 * c + 10 < KLIVE) fma_row_bcast<10>(a2, hc[c], w[16 * c + 10]);
        if (16 * c + 11 < KLIVE) fma_row_bcast<11>(a3, hc[c], w[16 * c + 11]);
        if (16 * c + 12 < KLIVE) fma_row_bcast<12>(a0, hc[c], w[16 * c + 12]);
        if (16 * c + 13 < KLIVE) fma_row_bcast<13>(a1, hc[c], w[16 * c + 13]);
        if (16 * c + 14 < KLIVE) fma_row_bcast<14>(a2, hc[c], w[16 * c + 14]);
        if (16 * c + 15 < KLIVE) fma_row_bcast<15>(a3, hc[c], w[16 * c + 15]);
    }
    return (a0 + a1) + (a2 + a3);
}
// sum over the four rows of 16 lanes of a wavefront, in every lane (v_permlane32_swap / v_permlane16_swap of gfx950)
__device__ __forceinline__ float rows_sum4(float a) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(a), false, false);
    const float b = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    const auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(b), __float_as_uint(b), false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

template <int CTRL>
__device__ __forceinline__ float quad_perm(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

// ---------------------------------------------------------------------------------------------------
// forward recurrence: workgroup = (direction, sequence); thread 4u + p owns gate row p*H + u of W_hh (gate order i, f, g, o), so the
// four gates of a unit sit in one quad of lanes: they are combined by DPP quad permutes (no LDS, no barrier), every lane of the quad
// carries the unit's cell state, and the new h goes to the other half of a double-buffered LDS vector -- ONE workgroup barrier per step.
// A wavefront is alone on its SIMD at H = 64 and issues one instruction per ~4 cycles whatever its type, so the step is written for
// instruction count: FULL (H == HMAX, HAGCN's layers) has no lane masks, every lane of a quad stores (its gate and one of c, h,
// h entering, tanh c -- the last saves the BPTT an exp and a reciprocal per step), ring slots and LDS halves are immediates.
// ---------------------------------------------------------------------------------------------------
template <int HMAX, bool FULL, int KLIVE = HMAX>
__global__ __launch_bounds__(4 * HMAX) void lstm_forward_kernel(LstmGeom g, const float* __restrict__ gi, const float* __restrict__ w_hh0,
                                    const float* __restrict__ w_hh1, const float* __restrict__ b_ih0, const float* __restrict__ b_hh0,
                                    const float* __restrict__ b_ih1, const float* __restrict__ b_hh1, float* __restrict__ gates,
                                    float* __restrict__ cseq, float* __restrict__ hseq, float* __restrict__ hprev, float* __restrict__ tseq) {
    __shared__ float hs[2][4 * HMAX];                                  // element 4u + p: every lane of a unit's quad writes its copy of h
    __shared__ float ring[LSTM_AHEAD][4 * HMAX];
    const int dir = blockIdx.x / g.Bq, q = blockIdx.x % g.Bq;
    const int tid = threadIdx.x, u = tid >> 2, p = tid & 3, H = g.H, H4 = g.H4, T = (int)g.T;
    // FULL: no lane masks.  Lanes beyond the hidden width (H < HMAX) repeat unit H - 1: same values to the same addresses -- benign --
    // and their slots of the h vector meet zero weights.
    const int uc = u < H ? u : H - 1;
    const bool live = FULL || u < H;
    const int row = p * H + uc;
    const float* w_hh = dir ? w_hh1 : w_hh0;
    const float scale = p == 2 ? 2.0f * LOG2E : -LOG2E;                // gate order (i, f, g, o): g is the tanh row
    const float gm = p == 2 ? -2.0f : 1.0f, gb = p == 2 ? 1.0f : 0.0f;
    float w[HMAX];
#pragma unroll
    for (int k = 0; k < HMAX; ++k) w[k] = k < H ? w_hh[(int64_t)row * H + k] * scale : 0.f;
    float bias_s = (dir ? b_ih1[row] + b_hh1[row] : b_ih0[row] + b_hh0[row]) * scale;
    const int64_t base = ((int64_t)dir * g.Bq + q) * g.T;          // row of (dir, q, t = 0)
    for (int e = tid; e < 8 * HMAX; e += blockDim.x) (&hs[0][0])[e] = 0.f;
    float c = 0.f, hlast = 0.f;
    const int64_t t0 = dir ? g.T - 1 : 0;
    const int64_t dt = dir ? -1 : 1;
    const unsigned ring_wave = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)&ring[0][tid & ~63]);   // LDS byte address
    // running pointers (a step moves them by one row in this direction's order): the row to request, the tape rows to store
    const float* gsrc = gi + (base + t0) * H4 + row;
    const int64_t gstep = dt * H4, xstep = dt * H;
    int requested = 0;                                                 // steps whose input projection has been requested
    auto request = [&](int slot) {                                     // clamped at the end: the instruction count must not vary
        dma_dword(gsrc, ring_wave + (unsigned)slot * (4 * HMAX * 4));
        if (++requested < T) gsrc += gstep;
    };
    float* gdst = gates + (base + t0) * H4 + row;
    float* xdst = (p == 0 ? cseq : (p == 1 ? hseq : (p == 2 ? hprev : tseq))) + (base + t0) * H + uc;
    // the weight rows above are the compiler's own loads: a use in front of the loop makes it wait for them HERE, not (conservatively,
    // every step) at their first use inside
#pragma unroll
    for (int k = 0; k < HMAX; ++k) asm volatile("" : "+v"(w[k]));
    asm volatile("" : "+v"(bias_s));
#pragma unroll
    for (int i = 0; i < LSTM_AHEAD; ++i) request(i);
    wait_vm<0>();
    __syncthreads();
    for (int s0 = 0; s0 < T; s0 += LSTM_AHEAD) {
#pragma unroll
        for (int i = 0; i < LSTM_AHEAD; ++i) {
            if (s0 + i >= T) break;
            const float* hv = hs[i & 1];                               // step s reads half s & 1 (LSTM_AHEAD is even)
            // issued after this row's request: the two stores of that step and (request, store, store) of the LSTM_AHEAD - 1 steps since
            wait_vm<2 + 3 * (LSTM_AHEAD - 1)>();
            float hc[HMAX / 16];
#pragma unroll
            for (int cc = 0; cc < (KLIVE + 15) / 16; ++cc) hc[cc] = hv[64 * cc + 4 * (tid & 15)];
            const float pre = matvec_row_bcast<HMAX, KLIVE>(w, hc, fmaf(ring[i][tid], scale, bias_s));
            const float gate = fmaf(rcp1p_exp2(pre), gm, gb);
            const float ig = quad_perm<0x00>(gate), fg = quad_perm<0x55>(gate), gg = quad_perm<0xAA>(gate), og = quad_perm<0xFF>(gate);
            c = fmaf(fg, c, ig * gg);
            const float tc = tanh_fast(c);
            float h = og * tc;
            if (!FULL && !live) h = 0.f;                               // a dead unit's slot is read (times a zero weight)
            hs[(i & 1) ^ 1][tid] = h;
            request(i);                                                // the slot's row was read in front of the matvec
            const float xv = p == 0 ? c : (p == 1 ? h : (p == 2 ? hlast : tc));
            if (FULL || live) {
                store_async(gdst, gate);
                store_async(xdst, xv);
            }
            gdst += gstep;
            xdst += xstep;
            hlast = h;
            lds_barrier();
        }
    }
}

// out[q][t][:] = h_fwd + h_bwd   (Model.py:59-61)
__global__ void lstm_sum_kernel(LstmGeom g, const float* __restrict__ hseq, float* __restrict__ out) {
    const int64_t n = g.rows * g.H;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
        out[e] = hseq[e] + hseq[n + e];
}

// ---------------------------------------------------------------------------------------------------
// BPTT: a wavefront owns 16 units; lane 16p + kl keeps W_hh[p*H + :, k] (gate block p, unit k = 16 wave + kl) in registers, so a row of
// 16 lanes shares its gate block: the block's gate gradients are broadcast along the row by DPP (matvec_row_bcast), the four partial
// sums of d h_prev[k] sit in the four rows (rows_sum4), every lane carries its unit's (dh, dc) and computes ITS gate's gradient: one
// LDS vector of 4H gate gradients, double-buffered, one workgroup barrier per step.  The tape of a step (gates, tanh c, cell state
// entering, incoming gradient) does not depend on the recurrence: it is requested LSTM_AHEAD steps ahead (lane 16p + kl: gate p and
// one of tanh c_t, c entering, d out; read back from the wavefront's own ring segment) and folded into four coefficients one step
// ahead, interleaved with the matvec:   dct = dht * k_c + dc;  d gate_p = (p < 3 ? dct : dht) * k_p;  dc' = dct * k_fg
//   k_c = o (1 - tc^2);   k_p = A (E (m - E) + b) with (A, E, m, b) = (g, i, 1, 0), (c entering, f, 1, 0), (i, g, 0, 1), (tc, o, 1, 0):
// the same three instructions in every row, A and E read through per-lane LDS offsets.
// ---------------------------------------------------------------------------------------------------
template <int HMAX, bool FULL, int KLIVE = HMAX>
__global__ __launch_bounds__(4 * HMAX) void lstm_backward_kernel(LstmGeom g, const float* __restrict__ w_hh0, const float* __restrict__ w_hh1,
                                     const float* __restrict__ gates, const float* __restrict__ cseq, const float* __restrict__ tseq,
                                     const float* __restrict__ dout, float* __restrict__ dgates) {
    constexpr int ROW = 4 * HMAX;                                              // floats per ring row
    __shared__ float dgl[2][ROW];                                              // [gate block][HMAX]
    __shared__ float ring[LSTM_AHEAD][2][ROW];                                 // [slot][gates | tanh c, c entering, d out, -][lane of the workgroup]
    const int dir = blockIdx.x / g.Bq, q = blockIdx.x % g.Bq;
    const int tid = threadIdx.x, wave0 = tid & ~63, kl = tid & 15, p = (tid >> 4) & 3, k = (wave0 >> 2) + kl, H = g.H, H4 = g.H4, T = (int)g.T;
    const bool live = FULL || k < H;
    const int kk = k < H ? k : H - 1;                                   // (lanes beyond the hidden width repeat unit H - 1, see the forward)
    const float* w_hh = dir ? w_hh1 : w_hh0;
    float wt[HMAX];
#pragma unroll
    for (int jj = 0; jj < HMAX; ++jj) wt[jj] = jj < H ? w_hh[((int64_t)p * H + jj) * H + kk] : 0.f;
    for (int e = tid; e < 2 * ROW; e += blockDim.x) (&dgl[0][0])[e] = 0.f;
    const int64_t base = ((int64_t)dir * g.Bq + q) * g.T;
    const int64_t obase = (int64_t)q * g.T;                          // dout is [q][t][H], shared by both directions
    // opposite to the forward order of this direction
    const int64_t t0 = dir ? 0 : g.T - 1;
    const int64_t dt = dir ? 1 : -1;
    // running pointers, one row per step in this loop's order.  The cell state entering step t is the one the forward left at t + dt
    // (this loop's direction) -- the forward's first step (this loop's last) has none: its address is clamped and the value replaced
    // by 0 when it is folded.  The second request of a step carries the row offset as the instruction's immediate (it moves the LDS
    // AND the global address), hence the pointer ROW floats back.
    const float* gptr = gates + (base + t0) * H4 + p * H + kk;
    const float* xptr = (p == 2 ? dout + (obase + t0) * H : (p == 1 ? cseq + (base + t0 + (T > 1 ? dt : 0)) * H : tseq + (base + t0) * H)) + kk - ROW;
    const int64_t gstep = dt * H4, xstep = dt * H;
    const unsigned ring_wave = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)&ring[0][0][wave0]);   // LDS byte address
    int requested = 0;
    auto request = [&](int slot) {                                     // clamped at the end: the instruction count must not vary
        asm volatile("s_mov_b32 m0, %2\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dword %0, off\n\t"
                     "global_load_lds_dword %1, off offset:%3"
                     :
                     : "v"(gptr), "v"(xptr), "s"(ring_wave + (unsigned)slot * (2 * ROW * 4)), "n"(ROW * 4)
                     : "memory");
        ++requested;
        if (requested < T) gptr += gstep;
        if (requested + (p == 1 ? 1 : 0) < T) xptr += xstep;
    };
    // per-lane ring offsets (floats from the slot's start)
    const int o_g = wave0 + kl;                                        // gates of this unit: + 0, 16, 32, 48 = i, f, g, o
    const int o_x = ROW + wave0 + kl;                                  // + 0, 16, 32 = tanh c_t, c entering, d out
    const int o_A = p == 0 ? o_g + 32 : (p == 1 ? o_x + 16 : (p == 2 ? o_g : o_x));
    const int o_E = p == 0 ? o_g : (p == 1 ? o_g + 16 : (p == 2 ? o_g + 32 : o_g + 48));
    const float km = p == 2 ? 0.0f : 1.0f, kb = p == 2 ? 1.0f : 0.0f;
    const int no_entering = p == 1 ? T - 1 : -1;                       // the loop step whose "c entering" does not exist (lanes of row 1)
    float k_c = 0.f, k_p = 0.f, k_fg = 0.f, k_do = 0.f;
    auto prepare = [&](int slot, int step) {                           // the coefficients of loop step `step` from ring slot `slot`
        const float* r = &ring[slot][0][0];
        const float og = r[o_g + 48], tc = r[o_x], fg = r[o_g + 16], d = r[o_x + 32], E = r[o_E];
        const float A = step == no_entering ? 0.f : r[o_A];
        k_c = fmaf(-og * tc, tc, og);
        k_p = A * fmaf(E, km - E, kb);
        k_fg = fg;
        k_do = d;
    };
    float* ddst = dgates + (base + t0) * H4 + p * H + kk;
#pragma unroll
    for (int jj = 0; jj < HMAX; ++jj) asm volatile("" : "+v"(wt[jj]));        // the compiler waits for its weight loads here (see the forward)
#pragma unroll
    for (int i = 0; i < LSTM_AHEAD; ++i) request(i);
    wait_vm<0>();
    prepare(0, 0);
    float dh = 0.f, dc = 0.f;
    __syncthreads();
    for (int s0 = 0; s0 < T; s0 += LSTM_AHEAD) {
#pragma unroll
        for (int i = 0; i < LSTM_AHEAD; ++i) {
            const int s = s0 + i;
            if (s >= T) break;
            float* dcur = dgl[i & 1];                                  // LSTM_AHEAD is even
            const float dht = k_do + dh;
            const float dct = fmaf(dht, k_c, dc);
            const float dval = (p == 3 ? dht : dct) * k_p;
            dc = dct * k_fg;
            if (FULL || live) {
                dcur[p * HMAX + k] = dval;
                store_async(ddst, dval);
            }
            ddst += gstep;
            lds_barrier();
            // d h_prev[k] = sum over the four gate blocks of W_hh[block]^T d gate: this row's block, then the four rows.  The next
            // step's tape is folded meanwhile; issued after its requests (LSTM_AHEAD - 1 steps ago): (store, two requests) of the
            // LSTM_AHEAD - 2 steps in between and this step's store
            float dchunk[HMAX / 16];
            chunks_load<HMAX>(dchunk, dcur + p * HMAX, kl);           // wt is zero beyond H; the LDS vector is zero there
            wait_vm<3 * (LSTM_AHEAD - 2) + 1>();
            prepare((i + 1) % LSTM_AHEAD, s + 1);
            const float part = matvec_row_bcast<HMAX, KLIVE>(wt, dchunk, 0.f);
            request(i);                                                // the slot held this step's tape, folded a step ago
            dh = rows_sum4(part);
        }
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
    auto fwd = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(ndir * g.Bq), dim3(threads), 0, st, g, (const float*)(ws + g.o_gi), a->w_hh[0], a->w_hh[d1], a->b_ih[0],
                           a->b_hh[0], a->b_ih[d1], a->b_hh[d1], ws + g.o_gates, ws + g.o_c, ws + g.o_h, ws + g.o_hprev, ws + g.o_tc);
    };
    // (the mask-free form serves every width: lanes beyond it repeat unit H - 1; HAGCN's widths also skip the products beyond them)
    if (g.H == 60) fwd(lstm_forward_kernel<64, true, 60>);
    else if (g.H <= 64) fwd(lstm_forward_kernel<64, true>);
    else if (g.H == 120) fwd(lstm_forward_kernel<128, true, 120>);
    else fwd(lstm_forward_kernel<128, true>);
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
    auto bwd = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(ndir * g.Bq), dim3(threads), 0, st, g, a->w_hh[0], a->w_hh[d1], (const float*)(ws + g.o_gates),
                           (const float*)(ws + g.o_c), (const float*)(ws + g.o_tc), a->dout, ws + g.o_dgates);
    };
    if (H == 60) bwd(lstm_backward_kernel<64, true, 60>);
    else if (H <= 64) bwd(lstm_backward_kernel<64, true>);
    else if (H == 120) bwd(lstm_backward_kernel<128, true, 120>);
    else bwd(lstm_backward_kernel<128, true>);
    float* one = ws + g.o_one;
    float* split = ws + g.o_split;
    hipLaunchKernelGGL(lstm_fill_one_kernel, dim3(1), dim3(1), 0, st, one);
    // The data gradient first: the layer below waits for nothing else.  dx (+)= dG W_ih
    if (a->dx)
        for (int d = 0; d < ndir; ++d)
            LS_RC(sgemm(ws + g.o_dgates + (int64_t)d * g.rows * H4, H4, 1, a->w_ih[d], 1, I, a->dx, I, R, I, H4, d == 1, st));
    // The parameter gradients feed nothing in this call: on the caller's second stream (args->aux_stream) they run beside what the caller
    // enqueues on `st` next -- in a stack of layers the BPTT of the layer below, a persistent recurrence on 2 * num_seq of the CUs.
    // NOT joined here: the caller joins before it reads them (include/rulgnn.h).
    hipStream_t wst = st;
    if (a->aux_stream) {
        wst = static_cast<hipStream_t>(a->aux_stream);
        hipEvent_t ev = aux_pooled_event();
        if (!ev || hipEventRecord(ev, st) != hipSuccess || hipStreamWaitEvent(wst, ev, 0) != hipSuccess) return RULGNN_EHIP;
    }
    // dW_ih = dG^T x ; dW_hh = dG^T h_prev ; db_ih = db_hh = column sums of dG -- of both directions: ONE split-K launch + ONE reduction
    // (they were seven launches per direction, 5-8 us each; behind the first layer's BPTT nothing is left to hide them under)
    SplitKJob jobs[8];
    int nj = 0;
    for (int d = 0; d < ndir; ++d) {
        const float* dg = ws + g.o_dgates + (int64_t)d * g.rows * H4;
        jobs[nj++] = SplitKJob{dg, 1, H4, a->x, 1, I, a->dw_ih[d], I, H4, I, R};
        jobs[nj++] = SplitKJob{dg, 1, H4, ws + g.o_hprev + (int64_t)d * g.rows * H, 1, H, a->dw_hh[d], H, H4, H, R};
        jobs[nj++] = SplitKJob{one, 0, 0, dg, 1, H4, a->db_ih[d], H4, 1, H4, R};
        jobs[nj++] = SplitKJob{one, 0, 0, dg, 1, H4, a->db_hh[d], H4, 1, H4, R};
    }
    if (R > 0) LS_RC(sgemm_splitk_batch(jobs, nj, split, (size_t)g.o_split_floats, wst));
    else
        for (int j = 0; j < nj; ++j)
            if (hipMemsetAsync(jobs[j].C, 0, sizeof(float) * (size_t)jobs[j].M * jobs[j].N, wst) != hipSuccess) return RULGNN_EHIP;
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

}  // namespace rulgnn
