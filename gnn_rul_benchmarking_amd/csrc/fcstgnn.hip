// FC_STGNN path for gfx950: per-(sample, patch, sensor) 1D-CNN encoder -> Linear + BatchNorm -> positional encoding
// (+ dropout) -> two windowed fully-connected space-time graph blocks (dot-product graph with softmax and decay mask,
// BatchNorm, MPNN, BatchNorm, mean over the window) -> 4-layer MLP; forward and backward.
//
// Reference: models/FC_STGNN/Model.py (FC_STGNN_RUL :5-84), Model_Base.py (Feature_extractor_1DCNN_RUL :12-41,
// Dot_Graph_Construction_weights :44-67, MPNN_mk_v2 :72-107, PositionalEncoding :111-134, Conv_GraphST :137-148,
// Mask_Matrix :150-170, GraphConvpoolMPNN_block_v6 :175-225) and algorithms/algorithms.py:51-76 (MSE + Adam).
//
// Decomposition.  Rows m = (sample, patch t, sensor) carry the encoder; the unfold of
// Conv_GraphST is index arithmetic (window w of block b covers patches w*stride + {0,1}; graph node q = tau*N + sensor),
// so the mapping Linear of the graph construction runs once per row, not per window copy.  Dense projections are
// [rows, .] GEMMs on the matrix cores (sgemm_mfma.hpp), one workgroup per window graph (<= 40 nodes) does the
// adjacency / softmax / aggregation in LDS, the rest is elementwise.  Seven BatchNorms cut the step into phases; batch
// statistics go through fp64 reduction cells in stream order.  BatchNorms on the unfolded windows are computed on the
// rows with multiplicity weights (how many windows contain a patch).
#include "adam_device.hpp"
#include "async_mem.hpp"
#include "aux_stream.hpp"
#include "sgemm_mfma.hpp"
#include "stgcn_host.hpp"

namespace rulgnn {

namespace {

constexpr int FB = 256;
constexpr int FC_GRAPH_FWD_THREADS = 128, FC_GRAPH_BWD_THREADS = 256;     // workgroups of the per-graph kernels (measured: 46 -> 42 us, 74 -> 70 us)
constexpr int MAXC = 64;            // BatchNorm channels
constexpr int MAXQ = 40;            // graph nodes = 2 * sensors
constexpr int MAXD = 64;            // graph feature width 2 * hidden_dim
constexpr int NBN = 7;
constexpr float BN_EPS = 1e-5f;
constexpr float LEAKY = 0.01f;
constexpr float DECAY = 0.7f;       // Model.py:13
constexpr float PE_P = 0.1f;        // Model.py:25

struct FcGeom {
    int64_t B, M;                   // samples; rows = B * NP * N
    int N, NP, PS, TL, H1, CO, K, L1, L2, CL, D2, HD, Q, FIN;
    int W[2], S[2], foff[2];
    int64_t G[2];
    int o_w1, o_ga, o_ba, o_w2, o_gb, o_bb, o_W3, o_b3, o_gc, o_bc;
    int o_map[2], o_bmap[2], o_gd[2], o_bd[2], o_th[2], o_thb[2], o_ge[2], o_be[2];
    int o_f1w, o_f1b, o_f2w, o_f2b, o_f3w, o_f3b, o_f4w, o_f4b, nparam;
    int bn_ch[NBN], bn_off[NBN], bn_g[NBN], bn_b[NBN], bn_total;
    double cnt[NBN];
};

__host__ int fc_geometry(const rulgnn_fcstgnn_shape* s, FcGeom* g) {
    if (!s) return RULGNN_EINVAL;
    if (s->batch < 0 || s->patch_size < 1 || s->num_patch < 2 || s->encoder_hidden_dim < 1 || s->encoder_out_dim < 1 ||
        s->encoder_conv_kernel < 1 || s->hidden_dim < 1 || s->num_node < 1)
        return RULGNN_EINVAL;
    g->K = s->encoder_conv_kernel;
    g->PS = s->patch_size;
    g->L1 = g->PS + 2 * (g->K / 2) - g->K + 1;
    g->L2 = g->L1 + 2 - g->K + 1;
    if (g->L1 < 1 || g->L2 < 1 || s->encoder_time_out != g->L2) return RULGNN_EINVAL;
    g->NP = s->num_patch;
    g->W[0] = g->NP - 1;
    g->W[1] = (g->NP - 2) / 2 + 1;
    if (s->num_windows != g->W[0] + g->W[1]) return RULGNN_EINVAL;
    if (s->num_node * 2 > MAXQ || s->hidden_dim * 2 > MAXD || s->encoder_hidden_dim > 16 || s->encoder_out_dim > MAXC ||
        g->K > 4 || g->PS > 64 || g->NP > 128)
        return RULGNN_EUNSUPPORTED;
    g->B = s->batch;
    g->N = s->num_node;
    g->TL = g->NP * g->PS;
    g->H1 = s->encoder_hidden_dim;
    g->CO = s->encoder_out_dim;
    g->CL = g->CO * g->L2;
    g->HD = s->hidden_dim;
    g->D2 = 2 * g->HD;
    g->Q = 2 * g->N;
    g->S[0] = 1;
    g->S[1] = 2;
    g->M = g->B * g->NP * g->N;
    if (g->M * (int64_t)(g->CL > g->D2 ? g->CL : g->D2) > ((int64_t)1 << 31) - 1) return RULGNN_EUNSUPPORTED;
    g->G[0] = g->B * g->W[0];
    g->G[1] = g->B * g->W[1];
    g->foff[0] = 0;
    g->foff[1] = g->W[0] * g->N * g->HD;
    g->FIN = g->HD * (g->W[0] + g->W[1]) * g->N;
    int o = 0;
    auto take = [&](int n) { const int r = o; o += n; return r; };
    g->o_w1 = take(g->H1 * g->K); g->o_ga = take(g->H1); g->o_ba = take(g->H1);
    g->o_w2 = take(g->CO * g->H1 * g->K); g->o_gb = take(g->CO); g->o_bb = take(g->CO);
    g->o_W3 = take(g->D2 * g->CL); g->o_b3 = take(g->D2); g->o_gc = take(g->D2); g->o_bc = take(g->D2);
    for (int b = 0; b < 2; ++b) {
        g->o_map[b] = take(g->D2 * g->D2); g->o_bmap[b] = take(g->D2); g->o_gd[b] = take(g->D2); g->o_bd[b] = take(g->D2);
        g->o_th[b] = take(g->HD * g->D2); g->o_thb[b] = take(g->HD); g->o_ge[b] = take(g->HD); g->o_be[b] = take(g->HD);
    }
    g->o_f1w = take(g->D2 * g->FIN); g->o_f1b = take(g->D2);
    g->o_f2w = take(g->D2 * g->D2); g->o_f2b = take(g->D2);
    g->o_f3w = take(g->HD * g->D2); g->o_f3b = take(g->HD);
    g->o_f4w = take(g->HD); g->o_f4b = take(1);
    g->nparam = o;
    const int ch[NBN] = {g->H1, g->CO, g->D2, g->D2, g->HD, g->D2, g->HD};
    const int og[NBN] = {g->o_ga, g->o_gb, g->o_gc, g->o_gd[0], g->o_ge[0], g->o_gd[1], g->o_ge[1]};
    const int ob[NBN] = {g->o_ba, g->o_bb, g->o_bc, g->o_bd[0], g->o_be[0], g->o_bd[1], g->o_be[1]};
    int bo = 0;
    for (int i = 0; i < NBN; ++i) {
        g->bn_ch[i] = ch[i]; g->bn_g[i] = og[i]; g->bn_b[i] = ob[i];
        g->bn_off[i] = bo;
        bo += 2 * ch[i];
    }
    g->bn_total = bo;
    g->cnt[0] = (double)g->M * g->L1;
    g->cnt[1] = (double)g->M * g->L2;
    g->cnt[2] = (double)g->M;
    for (int b = 0; b < 2; ++b) g->cnt[3 + 2 * b] = g->cnt[4 + 2 * b] = (double)g->G[b] * g->Q;
    return RULGNN_OK;
}

struct Cells {
    double fwd[NBN][MAXC][2];       // sum, sum of squares
    double bwd[NBN][MAXC][2];       // sum dy, sum dy * xhat
};

// Every workgroup adds its partial sums with one atomic per channel; thousands of workgroups hitting the same addresses
// serialise (~10 ns each: 40-50 us per streaming kernel at 4096 workgroups), so the cells exist CELL_REP times, workgroup b
// adds into replica b % CELL_REP and the readers sum the replicas in a fixed order.
constexpr int CELL_REP = 16;
__device__ inline double cell_fwd(const Cells* cells, int id, int c, int j) {
    double v = 0.0;
#pragma unroll
    for (int r = 0; r < CELL_REP; ++r) v += cells[r].fwd[id][c][j];
    return v;
}
__device__ inline double cell_bwd(const Cells* cells, int id, int c, int j) {
    double v = 0.0;
#pragma unroll
    for (int r = 0; r < CELL_REP; ++r) v += cells[r].bwd[id][c][j];
    return v;
}

struct BnCoef {
    float mean, inv, sc, sh;
};
__device__ inline BnCoef fbn(const FcGeom& g, const Cells* cells, const float* prm, const float* running, int training, int id, int c) {
    BnCoef r;
    float var;
    if (training) {
        const double m = cell_fwd(cells, id, c, 0) / g.cnt[id];
        double v = cell_fwd(cells, id, c, 1) / g.cnt[id] - m * m;
        if (v < 0.0) v = 0.0;
        r.mean = (float)m;
        var = (float)v;
    } else {
        r.mean = running[g.bn_off[id] + c];
        var = running[g.bn_off[id] + g.bn_ch[id] + c];
    }
    r.inv = 1.0f / sqrtf(var + BN_EPS);
    r.sc = prm[g.bn_g[id] + c] * r.inv;
    r.sh = prm[g.bn_b[id] + c] - r.mean * r.sc;
    return r;
}

// per-workgroup (sum, sumsq) accumulators in LDS doubles, flushed to the cells with one atomic per channel.  Two measures
// against same-address LDS atomics (they serialise: the streaming kernels spent 50-90 us in them at FD004 / batch 256):
// a thread keeps a private running pair while consecutive elements fall into the same channel, and the LDS slots exist
// BS_REP times, picked by lane bits, so lanes of one wavefront that do share a channel rarely share an address.
constexpr int BS_REP = 8;
constexpr int BS_DOUBLES = BS_REP * 2 * MAXC;
struct BlockStats {
    double* s;                      // [BS_REP][C][2] in LDS
    int C, cur;
    float pa, pb;
    __device__ void init(double* lds, int C_) {
        s = lds;
        C = C_;
        cur = -1;
        pa = pb = 0.f;
        for (int i = threadIdx.x; i < BS_REP * 2 * C; i += FB) s[i] = 0.0;
        __syncthreads();
    }
    __device__ void spill() {
        if (cur >= 0) {
            const int rep = (threadIdx.x ^ (threadIdx.x >> 3)) & (BS_REP - 1);
            atomicAdd(&s[(rep * C + cur) * 2], (double)pa);
            atomicAdd(&s[(rep * C + cur) * 2 + 1], (double)pb);
        }
    }
    __device__ void add(int c, float a, float b) {
        if (c != cur) {
            spill();
            cur = c;
            pa = pb = 0.f;
        }
        pa += a;
        pb += b;
    }
    __device__ void flush(double (*dst)[2], int C_) {
        spill();
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * C_; i += FB) {
            double v = 0.0;
#pragma unroll
            for (int r = 0; r < BS_REP; ++r) v += s[r * 2 * C_ + i];
            if (v != 0.0) atomicAdd(&dst[i >> 1][i & 1], v);
        }
    }
};

__device__ inline float leaky(float v) { return v > 0.f ? v : LEAKY * v; }
// Element loop of the streaming kernels with 32-BIT index arithmetic whenever the tensors are small enough (always, at the reference's
// wirings): an element index is decomposed by two to five divisions, and a 64-bit division is ~100 instructions against ~25 -- half of
// these kernels' time.  `body` is a generic lambda over the index type.
template <typename Body>
__device__ __forceinline__ void fc_walk(int64_t total, Body&& body) {
    if (total <= (int64_t)1 << 29) {
        const unsigned tot = (unsigned)total, step = gridDim.x * FB;
        for (unsigned e = blockIdx.x * FB + threadIdx.x; e < tot; e += step) body(e);
    } else {
        for (int64_t e = (int64_t)blockIdx.x * FB + threadIdx.x; e < total; e += (int64_t)gridDim.x * FB) body(e);
    }
}
// The two window blocks (stride 1 and 2) are independent between the positional encoding and the MLP: their kernels take per-block pointers
// in pairs and blockIdx.y picks the block, so one launch serves both (a launch less per pair on a chain of launches at their latency floor,
// and the smaller block's workgroups fill the slots the larger one leaves).
struct Ptr2 { float* p[2]; };
struct CPtr2 { const float* p[2]; };
// multiplicity of patch t in the unfolded windows of block b (how many window graphs contain it)
__device__ inline float mult(const FcGeom& g, int b, int t) {
    if (b == 0) return (g.W[0] > 1 && t > 0 && t < g.NP - 1) ? 2.f : 1.f;
    return t < 2 * g.W[1] ? 1.f : 0.f;
}
__device__ inline float pos_enc(int t, int d, int D2) {     // PositionalEncoding table entry (Model_Base.py:117-125), base 100
    const float div = expf((float)(d & ~1) * -(4.605170185988092f / (float)D2));
    const float ang = (float)t * div;
    return (d & 1) ? cosf(ang) : sinf(ang);
}

// ---------------------------------------------------------------------------------------------------
// encoder
// ---------------------------------------------------------------------------------------------------
// z1[m][c][p] = sum_k w1[c][k] * v[m][p + k - K/2]   (Conv1d(1 -> H1, K, padding K/2), Model_Base.py:17-18)
// (FD004: the reference's C-MAPSS wiring -- kernel 2, 3 output steps, 8 channels, 14 nodes x 25 patches of 2 points -- as compile-time
// constants: the element index is decomposed by five divisions, which are most of this kernel's instructions)
template <bool FD004>
__global__ __launch_bounds__(FB) void fc_conv1_kernel(FcGeom g, const float* __restrict__ x, const float* __restrict__ prm,
                                                     float* __restrict__ z1, Cells* cells, int training) {
    __shared__ double sl[BS_DOUBLES];
    BlockStats st;
    st.init(sl, g.H1);
    const int K = FD004 ? 2 : g.K, L1 = FD004 ? 3 : g.L1, H1 = FD004 ? 8 : g.H1, N = FD004 ? 14 : g.N, NP = FD004 ? 25 : g.NP,
              PS = FD004 ? 2 : g.PS, TL = FD004 ? 50 : g.TL;
    const int64_t total = g.M * H1 * L1;
    const int pad = K / 2;
    for (int64_t e = (int64_t)blockIdx.x * FB + threadIdx.x; e < total; e += (int64_t)gridDim.x * FB) {
        const int p = (int)(e % L1), c = (int)((e / L1) % H1);
        const int64_t m = e / ((int64_t)L1 * H1);
        const int node = (int)(m % N), t = (int)((m / N) % NP);
        const int64_t b = m / ((int64_t)N * NP);
        const float* v = x + (b * N + node) * TL + t * PS;
        float a = 0.f;
        for (int k = 0; k < K; ++k) {
            const int j = p + k - pad;
            if (j >= 0 && j < PS) a = fmaf(prm[g.o_w1 + c * K + k], v[j], a);
        }
        z1[e] = a;
        if (training) st.add(c, a, a * a);
    }
    if (training) st.flush(cells[blockIdx.x % CELL_REP].fwd[0], g.H1);
}

// Row-group mapping of the encoder kernels: a thread keeps ONE position of the per-row tile ((channel, time) = tid % tile) and walks
// rows; FB / tile rows are in flight per workgroup.  The element-per-thread form paid three 64-bit divisions per output, read every
// weight from global memory inside the innermost loop and changed its BatchNorm channel with every element (one LDS fp64 atomic pair
// per output): 33 us for the 2.1 M outputs of the second convolution at FD004 / batch 256.
struct RowGroup {
    int tile, rows_per, sub, pos;
    bool on;
    __device__ RowGroup(int tile_) : tile(tile_) {
        rows_per = FB / tile;                      // 0 when the tile is wider than the workgroup: see wide()
        sub = rows_per ? (int)threadIdx.x / tile : 0;
        pos = rows_per ? (int)threadIdx.x % tile : (int)threadIdx.x;
        on = rows_per ? sub < rows_per : true;
    }
    __device__ bool wide() const { return rows_per == 0; }
};
constexpr int FC_W2_MAX = 64 * 16 * 4;      // CO x H1 x K at the limits of fc_geometry

// z2[m][co][p] = sum_ci sum_k w2[co][ci][k] * relu(bn_a(z1))[m][ci][p + k - 1]   (padding 1, Model_Base.py:27-28)
template <int SK, int SL1, int SL2, int SH1, int SCO>
__global__ __launch_bounds__(FB) void fc_conv2_kernel(FcGeom g, const float* __restrict__ prm, const float* __restrict__ running,
                                                     const float* __restrict__ z1, float* __restrict__ z2, Cells* cells, int training) {
    __shared__ double sl[BS_DOUBLES];
    __shared__ BnCoef ca[16];
    __shared__ float wl[FC_W2_MAX];
    if (threadIdx.x < g.H1) ca[threadIdx.x] = fbn(g, cells, prm, running, training, 0, threadIdx.x);
    for (int e = threadIdx.x; e < g.CO * g.H1 * g.K; e += FB) wl[e] = prm[g.o_w2 + e];
    BlockStats st;
    st.init(sl, g.CO);                                                 // (ends in a barrier)
    const int H1 = SH1 ? SH1 : g.H1, K = SK ? SK : g.K, L1 = SL1 ? SL1 : g.L1, L2 = SL2 ? SL2 : g.L2, CL = (SCO ? SCO : g.CO) * L2;
    auto one = [&](int64_t m, int cl) {
        const int co = cl / L2, p = cl - co * L2;
        const float* zr = z1 + m * H1 * L1;
        const float* wr = wl + co * H1 * K;
        float a = 0.f;
        for (int ci = 0; ci < H1; ++ci)
            for (int k = 0; k < K; ++k) {
                const int q = p + k - 1;
                if (q >= 0 && q < L1) a = fmaf(wr[ci * K + k], fmaxf(fmaf(zr[ci * L1 + q], ca[ci].sc, ca[ci].sh), 0.f), a);
            }
        z2[m * CL + cl] = a;
        if (training) st.add(co, a, a * a);
    };
    const int T1 = H1 * L1;
    const RowGroup rg(T1 > CL ? T1 : CL);
    if (rg.wide()) {
        for (int64_t m = blockIdx.x; m < g.M; m += gridDim.x)
            for (int cl = threadIdx.x; cl < CL; cl += FB) one(m, cl);
    } else {
        // the row's activated input a1 = relu(bn_a(z1)) goes through LDS (one coalesced load per thread; read from global inside the
        // (channel, tap) loop the 16 loads of an output were served one after the other: 33 us of latency), double-buffered: one
        // barrier per group of rows
        __shared__ float tile[2][FB];
        const int co = rg.pos < CL ? rg.pos / L2 : 0, p = rg.pos - co * L2;
        const float* wr = wl + co * H1 * K;
        const int64_t stride = (int64_t)gridDim.x * rg.rows_per;
        int buf = 0;
        const int cis = rg.pos < T1 ? rg.pos / L1 : 0;
        auto fetch = [&](int64_t m0) {                                  // this thread's element of the row group that starts at m0
            const int64_t m = m0 + rg.sub;
            return (rg.on && m < g.M && rg.pos < T1) ? z1[m * T1 + rg.pos] : 0.f;
        };
        float nxt = fetch((int64_t)blockIdx.x * rg.rows_per);
        for (int64_t m0 = (int64_t)blockIdx.x * rg.rows_per; m0 < g.M; m0 += stride, buf ^= 1) {
            const int64_t m = m0 + rg.sub;
            const bool row_on = rg.on && m < g.M;
            const float cur = nxt;
            nxt = fetch(m0 + stride);                                  // the next group's element is in flight under this one's products
            if (row_on && rg.pos < T1) tile[buf][rg.sub * rg.tile + rg.pos] = fmaxf(fmaf(cur, ca[cis].sc, ca[cis].sh), 0.f);
            lds_barrier();                                             // LDS only: __syncthreads() would also wait for the stores of the group before
            if (row_on && rg.pos < CL) {
                const float* ar = &tile[buf][rg.sub * rg.tile];
                float a = 0.f;
                for (int ci = 0; ci < H1; ++ci)
                    for (int k = 0; k < K; ++k) {
                        const int q = p + k - 1;
                        if (q >= 0 && q < L1) a = fmaf(wr[ci * K + k], ar[ci * L1 + q], a);
                    }
                z2[m * CL + rg.pos] = a;
                if (training) st.add(co, a, a * a);
            }
        }
    }
    if (training) st.flush(cells[blockIdx.x % CELL_REP].fwd[1], g.CO);
}

// a2 = relu(bn_b(z2)), flattened [m][CO * L2]
__global__ __launch_bounds__(FB) void fc_act2_kernel(FcGeom g, const float* __restrict__ prm, const float* __restrict__ running,
                                                    const Cells* cells, int training, const float* __restrict__ z2, float* __restrict__ a2) {
    __shared__ BnCoef cb[MAXC];
    if (threadIdx.x < g.CO) cb[threadIdx.x] = fbn(g, cells, prm, running, training, 1, threadIdx.x);
    __syncthreads();
    const int64_t total = g.M * g.CL;
    for (int64_t e = (int64_t)blockIdx.x * FB + threadIdx.x; e < total; e += (int64_t)gridDim.x * FB) {
        const int co = (int)((e / g.L2) % g.CO);
        a2[e] = fmaxf(fmaf(z2[e], cb[co].sc, cb[co].sh), 0.f);
    }
}

// a2 = relu(bn_b(z2)), z3 = a2 W3^T + b3 and the BatchNorm-c statistics of z3 in ONE launch (row-group mapping, see fc_conv2_kernel):
// the activation kernel, the [M x CL] x [CL x D2] projection and the bias / statistics kernel were three launches of 8 + 12 + 6 us
// around a product of 24 MACs per output.  W3 sits in LDS with an odd row stride (lanes differ in the output: no bank conflicts).
__global__ __launch_bounds__(FB) void fc_proj3_kernel(FcGeom g, const float* __restrict__ prm, const float* __restrict__ running, Cells* cells,
                                                     int training, const float* __restrict__ z2, float* __restrict__ a2,
                                                     float* __restrict__ z3) {
    extern __shared__ float proj_lds[];                                // W3s[D2][CL + 1] | tile[2][FB]
    __shared__ double sl[BS_DOUBLES];
    __shared__ BnCoef cb[MAXC];
    const int CL = g.CL, D2 = g.D2, L2 = g.L2, WS = CL + 1;
    float* w3 = proj_lds;
    float* tile = proj_lds + D2 * WS;
    if (threadIdx.x < g.CO) cb[threadIdx.x] = fbn(g, cells, prm, running, training, 1, threadIdx.x);
    for (int e = threadIdx.x; e < D2 * CL; e += FB) w3[(e / CL) * WS + e % CL] = prm[g.o_W3 + e];
    BlockStats st;
    st.init(sl, D2);                                                   // (ends in a barrier)
    const RowGroup rg(CL > D2 ? CL : D2);
    const int co = rg.pos < CL ? rg.pos / L2 : 0;
    const float bias = rg.pos < D2 ? prm[g.o_b3 + rg.pos] : 0.f;
    const float* wr = w3 + (rg.pos < D2 ? rg.pos : 0) * WS;
    const int64_t stride = (int64_t)gridDim.x * rg.rows_per;
    auto fetch = [&](int64_t m0) {
        const int64_t m = m0 + rg.sub;
        return (rg.on && m < g.M && rg.pos < CL) ? z2[m * CL + rg.pos] : 0.f;
    };
    float nxt = fetch((int64_t)blockIdx.x * rg.rows_per);
    int buf = 0;
    for (int64_t m0 = (int64_t)blockIdx.x * rg.rows_per; m0 < g.M; m0 += stride, buf ^= 1) {
        const int64_t m = m0 + rg.sub;
        const bool row_on = rg.on && m < g.M;
        const float cur = nxt;
        nxt = fetch(m0 + stride);
        if (row_on && rg.pos < CL) {
            const float a = fmaxf(fmaf(cur, cb[co].sc, cb[co].sh), 0.f);
            a2[m * CL + rg.pos] = a;
            tile[buf * FB + rg.sub * rg.tile + rg.pos] = a;
        }
        lds_barrier();
        if (row_on && rg.pos < D2) {
            const float* ar = tile + buf * FB + rg.sub * rg.tile;
            float v = bias;
            for (int c = 0; c < CL; ++c) v = fmaf(ar[c], wr[c], v);
            z3[m * D2 + rg.pos] = v;
            if (training) st.add(rg.pos, v, v * v);
        }
    }
    if (training) st.flush(cells[blockIdx.x % CELL_REP].fwd[2], D2);
}

// z[r][c] += bias[c] in place; per-column (sum, sumsq) into cells->fwd[id]
__global__ __launch_bounds__(FB) void fc_bias_stats_kernel(float* __restrict__ z, const float* __restrict__ bias, int64_t rows, int C,
                                                          Cells* cells, int id, int training) {
    __shared__ double sl[BS_DOUBLES];
    BlockStats st;
    st.init(sl, C);
    const int64_t total = rows * C;
    for (int64_t e = (int64_t)blockIdx.x * FB + threadIdx.x; e < total; e += (int64_t)gridDim.x * FB) {
        const int c = (int)(e % C);
        const float v = z[e] + bias[c];
        z[e] = v;
        if (training) st.add(c, v, v * v);
    }
    if (training) st.flush(cells[blockIdx.x % CELL_REP].fwd[id], C);
}

// F = dropout(bn_c(z3) + pe[t]); weighted column statistics for the two window BatchNorms
__global__ __launch_bounds__(FB) void fc_pe_kernel(FcGeom g, const float* __restrict__ prm, const float* __restrict__ running,
                                                  Cells* cells, int training, const float* __restrict__ z3, float* __restrict__ F,
                                                  uint32_t drop_thr, float drop_scale, uint32_t drop_key, const uint32_t* key_dev,
                                                  int64_t row_offset) {
    __shared__ double sl[2 * BS_DOUBLES];
    __shared__ BnCoef cc[MAXC];
    if (threadIdx.x < g.D2) cc[threadIdx.x] = fbn(g, cells, prm, running, training, 2, threadIdx.x);
    BlockStats s0, s1;
    s0.init(sl, g.D2);
    s1.init(sl + BS_DOUBLES, g.D2);
    const uint32_t key = key_dev ? *key_dev : drop_key;
    const int64_t total = g.M * g.D2;
    fc_walk(total, [&](auto e) {
        const int d = (int)(e % g.D2);
        const auto m = e / g.D2;
        const int t = (int)((m / g.N) % g.NP);
        float y = fmaf(z3[e], cc[d].sc, cc[d].sh) + pos_enc(t, d, g.D2);
        if (training && drop_thr) {
            const uint32_t ctr = (uint32_t)(((int64_t)m + row_offset) * g.D2 + d);
            y = lowbias32(ctr ^ key) >= drop_thr ? y * drop_scale : 0.f;
        }
        F[e] = y;
        if (training) {
            const float c0 = mult(g, 0, t), c1 = mult(g, 1, t);
            s0.add(d, c0 * y, c0 * y * y);
            if (c1 != 0.f) s1.add(d, y, y * y);
        }
    });
    if (training) {
        s0.flush(cells[blockIdx.x % CELL_REP].fwd[3], g.D2);
        s1.flush(cells[blockIdx.x % CELL_REP].fwd[5], g.D2);
    }
}

// ---------------------------------------------------------------------------------------------------
// window graphs (one workgroup per graph)
// ---------------------------------------------------------------------------------------------------
__device__ inline int64_t graph_row(const FcGeom& g, int blk, int64_t gi, int q) {      // F row of node q of graph gi
    const int64_t b = gi / g.W[blk];
    const int w = (int)(gi - b * g.W[blk]);
    const int tau = q >= g.N ? 1 : 0;
    return (b * g.NP + w * g.S[blk] + tau) * g.N + (q - tau * g.N);
}

__global__ __launch_bounds__(FB) void fc_graph_kernel(FcGeom g, int blk, const float* __restrict__ prm, const float* __restrict__ running,
                                                     const Cells* cells, int training, const float* __restrict__ F,
                                                     const float* __restrict__ Mm, float* __restrict__ P, float* __restrict__ AX) {
    // dynamic LDS sized for THIS wiring's Q x D2 (sized for the limits, 40 x 64, the arrays took 28 KB and held the CU to 5 workgroups)
    extern __shared__ float fc_graph_lds[];
    __shared__ BnCoef cd[MAXD];
    const int Q = g.Q, D2 = g.D2, N = g.N, tid = threadIdx.x, FT = blockDim.x;
    const int DP = D2 + 1, QP = Q + 1;
    float* mm = fc_graph_lds;               // [Q][DP]
    float* xb = mm + Q * DP;                // [Q][DP]
    float* A = xb + Q * DP;                 // [Q][QP]
    if (tid < D2) cd[tid] = fbn(g, cells, prm, running, training, 3 + 2 * blk, tid);
    __syncthreads();
    for (int64_t gi = blockIdx.x; gi < g.G[blk]; gi += gridDim.x) {
        for (int e = tid; e < Q * D2; e += FT) {
            const int q = e / D2, d = e - q * D2;
            const int64_t r = graph_row(g, blk, gi, q);
            mm[q * DP + d] = Mm[r * D2 + d] + prm[g.o_bmap[blk] + d];
            xb[q * DP + d] = fmaf(F[r * D2 + d], cd[d].sc, cd[d].sh);
        }
        __syncthreads();
        for (int e = tid; e < Q * Q; e += FT) {
            const int i = e / Q, j = e - i * Q;
            float s = 0.f;
#pragma unroll 8
            for (int d = 0; d < D2; ++d) s = fmaf(mm[i * DP + d], mm[j * DP + d], s);
            if (i == j) s -= 1e8f;
            A[i * QP + j] = leaky(s);
        }
        __syncthreads();
        {
            // softmax over the row, then + I and the decay mask: four lanes per row (a quad: DPP reductions), seven columns each --
            // one lane per row left 100 of the 128 threads idle behind 3 x Q dependent LDS round trips with a libm exp and a division
            // per element, and was 60 % of this kernel
            const int part = tid & 3;
            for (int row = tid >> 2; row < ((Q + FT / 4 - 1) / (FT / 4)) * (FT / 4); row += FT / 4) {       // uniform trip count
            const bool on = row < Q;
            const float* ar = A + (on ? row : 0) * QP;
            float mx = -INFINITY;
            for (int j = part; j < Q; j += 4) mx = fmaxf(mx, ar[j]);
            mx = fmaxf(mx, __shfl_xor(mx, 1));
            mx = fmaxf(mx, __shfl_xor(mx, 2));
            float ev[(MAXQ + 3) / 4];
            float sum = 0.f;
#pragma unroll
            for (int c = 0; c < (MAXQ + 3) / 4; ++c) {
                const int j = part + 4 * c;
                ev[c] = j < Q ? __expf(ar[j] - mx) : 0.f;
                sum += ev[c];
            }
            sum += __shfl_xor(sum, 1);
            sum += __shfl_xor(sum, 2);
            const float inv = 1.0f / sum;
            if (on) {
                float* pr = P + (gi * Q + row) * Q;
                float* aw = A + row * QP;
#pragma unroll
                for (int c = 0; c < (MAXQ + 3) / 4; ++c) {
                    const int j = part + 4 * c;
                    if (j < Q) {
                        const float pv = ev[c] * inv;
                        pr[j] = pv;
                        aw[j] = (pv + (row == j ? 1.f : 0.f)) * (((row < N) == (j < N)) ? 1.f : DECAY);
                    }
                }
            }
            }
        }
        __syncthreads();
        for (int e = tid; e < Q * D2; e += FT) {
            const int i = e / D2, d = e - i * D2;
            float a = 0.f;
#pragma unroll 8
            for (int j = 0; j < Q; ++j) a = fmaf(A[i * QP + j], xb[j * DP + d], a);
            AX[(gi * Q + i) * D2 + d] = a;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// window graphs on the fp32 matrix cores: ONE WAVEFRONT per graph, no LDS staging, no workgroup barriers
// ---------------------------------------------------------------------------------------------------
// The workgroup-per-graph kernels above spend their time in LDS reads (two per FMA in the Q x Q x D products) and barriers.  A graph
// (Q = 2 x sensors <= 32 nodes, D = 2 x hidden = 16 or 32 features) fits the 32x32x2 fp32 MFMA tile, and its rows are consecutive in
// F / Mm (row of node q = first row + q): a wavefront keeps the whole graph in registers.
//   v_mfma_f32_32x32x2_f32: a-operand lane l = A[l & 31][l >> 5], b-operand lane l = B[l >> 5][l & 31], and of the result lane l holds
//   column l & 31, register r row krow(r, l >> 5) = 8 (r >> 2) + 4 (l >> 5) + (r & 3).
// A product sums over k in ANY order as long as both operands agree, so k runs in the order the registers already hold it: a result
// (register -> row, lane -> column) is fed back as the a-operand of the next product with k = krow(step, half) -- no transposition.
typedef float fc_f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ fc_f32x16 fc_mfma(float a, float b, const fc_f32x16& c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__host__ __device__ constexpr int fc_krow(int r, int h) { return 8 * (r >> 2) + 4 * h + (r & 3); }
__device__ __forceinline__ float fc_swap32(float v) {                 // the value of the lane 32 away
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return (threadIdx.x & 32) ? __uint_as_float(r[0]) : __uint_as_float(r[1]);
}
constexpr int FC_MX_WAVES = 4;

// A whole product of the window-graph kernels: acc += sum over the NS k slots (slot, half) of a[slot] b[slot].
//   BF = false: NS x v_mfma_f32_32x32x2_f32 (one k pair per instruction, exact fp32 operands);
//   BF = true (compute_dtype = bf16, BASELINE.json "FC_STGNN ... bf16"): the operands rounded to bf16 (v_cvt_pk_bf16_f32, nearest-even)
//   and eight slots per v_mfma_f32_32x32x16_bf16 -- a-operand lane l = A[l & 31][8 (l >> 5) + j], b-operand lane l = B[8 (l >> 5) + j][l & 31]:
//   slot j of half h is k = 8 h + j, the same (slot, half) pairing of the two operands as the fp32 form, so the register forms of the
//   kernels carry over unchanged; fp32 accumulation, fp32 softmax / BatchNorm / statistics around it.  NS = 8: one instruction instead of
//   eight (32 instead of 512 matrix-pipe cycles), NS = 16: two; NS = 4 pads four zero slots.
typedef __bf16 fc_bf16x8 __attribute__((ext_vector_type(8)));
// (the compiler's own conversion, not inline asm: the result feeds a matrix instruction directly and the hazard recogniser must see the
// VALU write -- with `asm("v_cvt_pk_bf16_f32")` the backward kernel read the operand registers too early: NaN gradients)
__device__ __forceinline__ unsigned fc_pk_bf16(float a, float b) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const f2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2));
}
template <bool BF, int NS>
__device__ __forceinline__ fc_f32x16 fc_prod(const float (&a)[NS], const float (&b)[NS], fc_f32x16 acc) {
    if constexpr (!BF) {
#pragma unroll
        for (int s = 0; s < NS; ++s) acc = fc_mfma(a[s], b[s], acc);
        return acc;
    } else {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int s0 = 0; s0 < NS; s0 += 8) {
            u32x4 pa, pb;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int s = s0 + 2 * q;
                pa[q] = fc_pk_bf16(s < NS ? a[s < NS ? s : 0] : 0.f, s + 1 < NS ? a[s + 1 < NS ? s + 1 : 0] : 0.f);
                pb[q] = fc_pk_bf16(s < NS ? b[s < NS ? s : 0] : 0.f, s + 1 < NS ? b[s + 1 < NS ? s + 1 : 0] : 0.f);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(fc_bf16x8, pa), __builtin_bit_cast(fc_bf16x8, pb), acc, 0, 0, 0);
        }
        return acc;
    }
}

// forward: S = M' M'^T, P = softmax(leaky(S - 1e8 I)) by rows, AX = ((P + I) o decay mask) X'   (Model_Base.py:44-78)
template <int D2T>
__global__ __launch_bounds__(64 * FC_MX_WAVES) void fc_graph_mx_kernel(FcGeom g, int blk, const float* __restrict__ prm, const float* __restrict__ running,
                                                                       const Cells* cells, int training, const float* __restrict__ F,
                                                                       const float* __restrict__ Mm, float* __restrict__ P, float* __restrict__ AX) {
    constexpr int HK = D2T / 2;                                        // k of a half-wave in the products over the features
    __shared__ BnCoef cd[D2T];
    __shared__ float bm[D2T];
    if (threadIdx.x < D2T) {
        cd[threadIdx.x] = fbn(g, cells, prm, running, training, 3 + 2 * blk, threadIdx.x);
        bm[threadIdx.x] = prm[g.o_bmap[blk] + threadIdx.x];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5, Q = g.Q, N = g.N;
    const float sc = c < D2T ? cd[c < D2T ? c : 0].sc : 0.f, sh = c < D2T ? cd[c < D2T ? c : 0].sh : 0.f;
    const int cq = c < Q ? c : Q - 1;                                  // lanes beyond the graph repeat its last node (never stored)
    for (int64_t gi = (int64_t)blockIdx.x * FC_MX_WAVES + (threadIdx.x >> 6); gi < g.G[blk]; gi += (int64_t)gridDim.x * FC_MX_WAVES) {
        const int64_t b = gi / g.W[blk];
        const int64_t row0 = (b * g.NP + (gi - b * g.W[blk]) * g.S[blk]) * g.N;       // rows row0 .. row0 + Q - 1 of F / Mm
        // S: both operands are node c's half row of the mapped features (k = HK h + step)
        float mh[HK];
        {
            const float4* src = reinterpret_cast<const float4*>(Mm + (row0 + cq) * D2T + HK * h);
#pragma unroll
            for (int v = 0; v < HK / 4; ++v) {
                const float4 t = src[v];
                mh[4 * v] = t.x + bm[HK * h + 4 * v];         mh[4 * v + 1] = t.y + bm[HK * h + 4 * v + 1];
                mh[4 * v + 2] = t.z + bm[HK * h + 4 * v + 2]; mh[4 * v + 3] = t.w + bm[HK * h + 4 * v + 3];
            }
        }
        // the normalised features by node for the second product: lane = feature, k = node krow(step, half)
        float xk[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int node = fc_krow(s, h);
            const float fv = F[(row0 + (node < Q ? node : Q - 1)) * D2T + (c < D2T ? c : 0)];      // (unconditional: the sixteen loads go out together)
            xk[s] = (c < D2T && node < Q) ? fmaf(fv, sc, sh) : 0.f;
        }
        fc_f32x16 S = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < HK; ++s) S = fc_mfma(mh[s], mh[s], S);
        // S is symmetric: the softmax over row c is the softmax over column c, which this lane and its partner 32 lanes away hold
        float t[16], mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = fc_krow(r, h);
            const float v = leaky(i == c ? S[r] - 1e8f : S[r]);
            t[r] = i < Q ? v : -INFINITY;
            mx = fmaxf(mx, t[r]);
        }
        mx = fmaxf(mx, fc_swap32(mx));
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            t[r] = __expf(t[r] - mx);
            sum += t[r];
        }
        sum += fc_swap32(sum);
        const float inv = 1.0f / sum;
        // t[r] = P[c][krow(r, h)]: four consecutive columns per register group
        if (c < Q) {
            float* pr = P + (gi * Q + c) * Q + 4 * h;
#pragma unroll
            for (int m = 0; m < 4; ++m)
                if (8 * m + 4 * h < Q)
                    *reinterpret_cast<float4*>(pr + 8 * m) = make_float4(t[4 * m] * inv, t[4 * m + 1] * inv, t[4 * m + 2] * inv, t[4 * m + 3] * inv);
        }
        // AX^T = X'^T Adj^T: a-operand the features by node (lane = feature), b-operand this lane's adjacency row -> lane = node,
        // register = feature krow(r, h): float4 stores of the node's row
        fc_f32x16 A = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int j = fc_krow(s, h);
            const float adj = (t[s] * inv + (j == c ? 1.f : 0.f)) * (((c < N) == (j < N)) ? 1.f : DECAY);
            A = fc_mfma(xk[s], j < Q ? adj : 0.f, A);
        }
        if (c < Q) {
            float* ar = AX + (gi * Q + c) * D2T + 4 * h;
#pragma unroll
            for (int m = 0; m < D2T / 8; ++m) *reinterpret_cast<float4*>(ar + 8 * m) = make_float4(A[4 * m], A[4 * m + 1], A[4 * m + 2], A[4 * m + 3]);
        }
    }
}

// features = mean over the two window steps of leaky(bn_e(z5))
__global__ __launch_bounds__(FB) void fc_pool_kernel(FcGeom g, const float* __restrict__ prm, const float* __restrict__ running,
                                                    const Cells* cells, int training, CPtr2 z5p, float* __restrict__ feat) {
    const int blk = blockIdx.y;
    const float* __restrict__ z5 = z5p.p[blk];
    __shared__ BnCoef ce[MAXC];
    if (threadIdx.x < g.HD) ce[threadIdx.x] = fbn(g, cells, prm, running, training, 4 + 2 * blk, threadIdx.x);
    __syncthreads();
    const int HD = g.HD, N = g.N, W = g.W[blk];
    const int64_t total = g.G[blk] * N * HD;
    fc_walk(total, [&](auto e) {
        const int h = (int)(e % HD), node = (int)((e / HD) % N);
        const auto gi = e / (HD * N), b = gi / W;
        const int w = (int)(gi - b * W);
        const float y0 = leaky(fmaf(z5[(gi * g.Q + node) * HD + h], ce[h].sc, ce[h].sh));
        const float y1 = leaky(fmaf(z5[(gi * g.Q + N + node) * HD + h], ce[h].sc, ce[h].sh));
        feat[b * g.FIN + g.foff[blk] + (w * N + node) * HD + h] = (y0 + y1) / 2.0f;
    });
}

// z = relu(z + bias) in place, [rows][C]
__global__ void fc_bias_relu_kernel(float* __restrict__ z, const float* __restrict__ bias, int64_t rows, int C) {
    const int64_t total = rows * C;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x)
        z[e] = fmaxf(z[e] + bias[e % C], 0.f);
}

// head: pred = h3 . w4 + b4; MSE pieces; dh3 = dpred * w4 * [h3 > 0]
__global__ void fc_head_kernel(FcGeom g, const float* __restrict__ prm, const float* __restrict__ h3, const float* __restrict__ y,
                               const float* __restrict__ dpred_in, float* __restrict__ pred, float* __restrict__ dpred,
                               float* __restrict__ sqerr, float* __restrict__ dh3, float inv_gb, int backward_only) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= g.B) return;
    float dp;
    if (!backward_only) {
        float a = prm[g.o_f4b];
        for (int j = 0; j < g.HD; ++j) a = fmaf(h3[b * g.HD + j], prm[g.o_f4w + j], a);
        pred[b] = a;
        if (!y) return;
        const float d = a - y[b];
        dp = 2.0f * d * inv_gb;
        sqerr[b] = d * d * inv_gb;
    } else {
        dp = dpred_in[b];
    }
    dpred[b] = dp;
    for (int j = 0; j < g.HD; ++j) dh3[b * g.HD + j] = h3[b * g.HD + j] > 0.f ? dp * prm[g.o_f4w + j] : 0.f;
}

// Every parameter gradient of the MLP (Model.py:30-40: fc1 [2h x FIN], fc2 [2h x 2h], fc3 [h x 2h], fc4 [1 x h] and their biases) and the
// loss sum in ONE launch, for the batches the reference protocol trains at (batch <= FC_MLPW_MAXB).  As split-K GEMMs + column sums these
// were nine latency-bound launches (sgemm_tiny / sgemm_mfma + reduce_slices / cols_sum_small / block_sum) at the head of the backward's side
// stream -- 50 us of host enqueue time during which the main stream's queue ran empty (the step is host-launch bound, tools/trace_family_step.sh).
// Workgroups [0, nfw): 64 columns of d fc1.weight each (thread = (column, row quarter): rows j = q, q + 4, ...; the batch walked in
// order, d h1 staged in LDS in chunks of 64 samples); the last workgroup: the small outputs, one per thread, each a dot product over the batch
// in order.  Fixed summation order: run-to-run reproducible.
constexpr int FC_MLPW_MAXB = 2048;
struct FcMlpW {
    const float *dpred, *dh3, *dh2, *dh1, *h3, *h2, *h1, *feat, *sqerr;
    float *f1w, *f1b, *f2w, *f2b, *f3w, *f3b, *f4w, *f4b, *loss;
    int B, D2, HD, FIN, nfw;
};
__global__ __launch_bounds__(256) void fc_mlp_wgrad_kernel(FcMlpW k) {
    __shared__ float dl[32 * (4 * 64 + 2 * 32 + 2)];                 // d fc1.weight part: [64 samples][D2 <= 64]; the small outputs: eight chunk arrays
    const int t = threadIdx.x, B = k.B, D2 = k.D2, HD = k.HD, FIN = k.FIN;
    if ((int)blockIdx.x < k.nfw) {
        const int c = blockIdx.x * 64 + (t & 63), q = t >> 6;
        float acc[16];                                               // rows q, q + 4, ..., < D2 <= 64
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int b0 = 0; b0 < B; b0 += 64) {
            const int nb = B - b0 < 64 ? B - b0 : 64;
            __syncthreads();
            for (int e = t; e < nb * D2; e += 256) dl[e] = k.dh1[(int64_t)b0 * D2 + e];
            __syncthreads();
            if (c < FIN) {
                for (int b = 0; b < nb; ++b) {
                    const float f = k.feat[(int64_t)(b0 + b) * FIN + c];
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (q + 4 * r < D2) acc[r] = fmaf(dl[b * D2 + q + 4 * r], f, acc[r]);
                }
            }
        }
        if (c < FIN) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (q + 4 * r < D2) k.f1w[(int64_t)(q + 4 * r) * FIN + c] = acc[r];
        }
        return;
    }
    // the small outputs: [f2w D2 x D2 | f3w HD x D2 | f4w HD | f1b D2 | f2b D2 | f3b HD | f4b 1 | loss 1], one per thread of the workgroups behind
    // the first nfw; the batch is staged through LDS in chunks of 32 samples (coalesced loads), every output a dot product over it in order
    const int n2 = D2 * D2, n3 = HD * D2;
    const int total = n2 + n3 + HD + D2 + D2 + HD + 2;
    const int o = ((int)blockIdx.x - k.nfw) * 256 + t;
    // which dot product this thread owns: a[b * sa + ia] * c[b * sc + ic] over the chunk arrays (c == nullptr: plain sum)
    float* const c_dh1 = dl, * const c_dh2 = c_dh1 + 32 * D2, * const c_h1 = c_dh2 + 32 * D2, * const c_h2 = c_h1 + 32 * D2;
    float* const c_dh3 = c_h2 + 32 * D2, * const c_h3 = c_dh3 + 32 * HD, * const c_dp = c_h3 + 32 * HD, * const c_sq = c_dp + 32;
    const float *pa = nullptr, *pc = nullptr;
    int sa = 0, sc = 0;
    float* dst = nullptr;
    {
        int i = o;
        if (i >= 0 && i < total) {
            if (i < n2) { pa = c_dh2 + i / D2; sa = D2; pc = c_h1 + i % D2; sc = D2; dst = k.f2w + i; }
            else if ((i -= n2) < n3) { pa = c_dh3 + i / D2; sa = HD; pc = c_h2 + i % D2; sc = D2; dst = k.f3w + i; }
            else if ((i -= n3) < HD) { pa = c_dp; sa = 1; pc = c_h3 + i; sc = HD; dst = k.f4w + i; }
            else if ((i -= HD) < D2) { pa = c_dh1 + i; sa = D2; dst = k.f1b + i; }
            else if ((i -= D2) < D2) { pa = c_dh2 + i; sa = D2; dst = k.f2b + i; }
            else if ((i -= D2) < HD) { pa = c_dh3 + i; sa = HD; dst = k.f3b + i; }
            else if ((i -= HD) == 0) { pa = c_dp; sa = 1; dst = k.f4b; }
            else if (k.loss) { pa = c_sq; sa = 1; dst = k.loss; }
        }
    }
    float v = 0.f;
    for (int b0 = 0; b0 < B; b0 += 32) {
        const int nb = B - b0 < 32 ? B - b0 : 32;
        __syncthreads();
        for (int e = t; e < nb * D2; e += 256) {
            c_dh1[e] = k.dh1[(int64_t)b0 * D2 + e];
            c_dh2[e] = k.dh2[(int64_t)b0 * D2 + e];
            c_h1[e] = k.h1[(int64_t)b0 * D2 + e];
            c_h2[e] = k.h2[(int64_t)b0 * D2 + e];
        }
        for (int e = t; e < nb * HD; e += 256) {
            c_dh3[e] = k.dh3[(int64_t)b0 * HD + e];
            c_h3[e] = k.h3[(int64_t)b0 * HD + e];
        }
        if (t < nb) {
            c_dp[t] = k.dpred[b0 + t];
            c_sq[t] = k.sqerr[b0 + t];
        }
        __syncthreads();
        if (dst) {
            if (pc) for (int b = 0; b < nb; ++b) v = fmaf(pa[b * sa], pc[b * sc], v);
            else for (int b = 0; b < nb; ++b) v += pa[b * sa];
        }
    }
    if (dst) *dst = v;
}

// The MLP behind the first (split-K) projection in ONE launch: bias + ReLU of fc1, fc2, fc3, the head, and -- when the loss is formed
// here (y) or its gradient comes in (dpred_in) -- the data gradients back to d h1.  Eight launches of the chain (bias-ReLU x 3, two
// [batch x 16 x 16] products, head; backward: head, two products, two masks) at their 5-7 us latency floor each; a thread owns a sample
// row, the 16..64-wide weights sit in LDS (broadcast reads).  mode bit 0: forward (h1 holds fc1's product without the bias), bit 1: backward.
template <int D2T>
__global__ __launch_bounds__(64) void fc_mlp_tail_kernel(FcGeom g, const float* __restrict__ prm, float* __restrict__ h1, float* __restrict__ h2,
                                                         float* __restrict__ h3, const float* __restrict__ y, const float* __restrict__ dpred_in,
                                                         float* __restrict__ pred, float* __restrict__ dpred, float* __restrict__ sqerr,
                                                         float* __restrict__ dh3, float* __restrict__ dh2, float* __restrict__ dh1, float inv_gb,
                                                         int mode) {
    constexpr int HDT = D2T / 2;
    __shared__ float w2[D2T * D2T], w3[HDT * D2T], w4[HDT], b1[D2T], b2[D2T], b3[HDT];
    for (int e = threadIdx.x; e < D2T * D2T; e += 64) w2[e] = prm[g.o_f2w + e];
    for (int e = threadIdx.x; e < HDT * D2T; e += 64) w3[e] = prm[g.o_f3w + e];
    for (int e = threadIdx.x; e < D2T; e += 64) { b1[e] = prm[g.o_f1b + e]; b2[e] = prm[g.o_f2b + e]; }
    for (int e = threadIdx.x; e < HDT; e += 64) { b3[e] = prm[g.o_f3b + e]; w4[e] = prm[g.o_f4w + e]; }
    __syncthreads();
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= g.B) return;
    float v1[D2T], v2[D2T], v3[HDT];
    float* r1 = h1 + b * D2T;
    float* r2 = h2 + b * D2T;
    float* r3 = h3 + b * HDT;
    float out = 0.f;
    if (mode & 1) {
#pragma unroll
        for (int i = 0; i < D2T; ++i) { v1[i] = fmaxf(r1[i] + b1[i], 0.f); r1[i] = v1[i]; }
#pragma unroll
        for (int o = 0; o < D2T; ++o) {
            float a = b2[o];
#pragma unroll
            for (int i = 0; i < D2T; ++i) a = fmaf(v1[i], w2[o * D2T + i], a);
            v2[o] = fmaxf(a, 0.f);
            r2[o] = v2[o];
        }
#pragma unroll
        for (int o = 0; o < HDT; ++o) {
            float a = b3[o];
#pragma unroll
            for (int i = 0; i < D2T; ++i) a = fmaf(v2[i], w3[o * D2T + i], a);
            v3[o] = fmaxf(a, 0.f);
            r3[o] = v3[o];
        }
        out = prm[g.o_f4b];
#pragma unroll
        for (int j = 0; j < HDT; ++j) out = fmaf(v3[j], w4[j], out);
        pred[b] = out;
        if (!y) return;
    } else {
#pragma unroll
        for (int i = 0; i < D2T; ++i) { v1[i] = r1[i]; v2[i] = r2[i]; }
#pragma unroll
        for (int i = 0; i < HDT; ++i) v3[i] = r3[i];
    }
    float dp;
    if (mode & 1) {
        const float d = out - y[b];
        dp = 2.0f * d * inv_gb;
        sqerr[b] = d * d * inv_gb;
    } else {
        dp = dpred_in[b];
    }
    dpred[b] = dp;
    float d3[HDT], d2[D2T];
#pragma unroll
    for (int j = 0; j < HDT; ++j) { d3[j] = v3[j] > 0.f ? dp * w4[j] : 0.f; dh3[b * HDT + j] = d3[j]; }
#pragma unroll
    for (int j = 0; j < D2T; ++j) {
        float a = 0.f;
#pragma unroll
        for (int o = 0; o < HDT; ++o) a = fmaf(d3[o], w3[o * D2T + j], a);
        d2[j] = v2[j] > 0.f ? a : 0.f;
        dh2[b * D2T + j] = d2[j];
    }
#pragma unroll
    for (int j = 0; j < D2T; ++j) {
        float a = 0.f;
#pragma unroll
        for (int o = 0; o < D2T; ++o) a = fmaf(d2[o], w2[o * D2T + j], a);
        dh1[b * D2T + j] = v1[j] > 0.f ? a : 0.f;
    }
}

// dz *= [h > 0]
__global__ void fc_relu_mask_kernel(float* __restrict__ dz, const float* __restrict__ h, int64_t n) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
        if (!(h[e] > 0.f)) dz[e] = 0.f;
}

// ---------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------
// d(bn_e output) from d features; BatchNorm-e backward sums
__global__ __launch_bounds__(FB) void fc_pool_bwd_kernel(FcGeom g, const float* __restrict__ prm, Cells* cells, CPtr2 z5p,
                                                        const float* __restrict__ dfeat, Ptr2 dy5p) {
    const int blk = blockIdx.y;
    const float* __restrict__ z5 = z5p.p[blk];
    float* __restrict__ dy5 = dy5p.p[blk];
    __shared__ double sl[BS_DOUBLES];
    __shared__ BnCoef ce[MAXC];
    const int id = 4 + 2 * blk;
    if (threadIdx.x < g.HD) ce[threadIdx.x] = fbn(g, cells, prm, nullptr, 1, id, threadIdx.x);
    BlockStats st;
    st.init(sl, g.HD);
    const int HD = g.HD, N = g.N, Q = g.Q, W = g.W[blk];
    const int64_t total = g.G[blk] * Q * HD;
    fc_walk(total, [&](auto e) {
        const int h = (int)(e % HD), q = (int)((e / HD) % Q);
        const auto gi = e / (HD * Q), b = gi / W;
        const int w = (int)(gi - b * W), node = q >= N ? q - N : q;
        const float zz = z5[e];
        const float yv = fmaf(zz, ce[h].sc, ce[h].sh);
        const float dy = dfeat[b * g.FIN + g.foff[blk] + (w * N + node) * HD + h] * 0.5f * (yv > 0.f ? 1.f : LEAKY);
        dy5[e] = dy;
        st.add(h, dy, dy * (zz - ce[h].mean) * ce[h].inv);
    });
    st.flush(cells[blockIdx.x % CELL_REP].bwd[id], HD);
}

// BatchNorm backward, row-major [rows][C]: dz = sc * (dy - sum_dy/m - xhat * sum_dyxhat/m), in place
// (blockIdx.y > 0: the second window block's tensor -- id + 2, its own rows)
__global__ __launch_bounds__(FB) void fc_bn_rows_bwd_kernel(FcGeom g, int id0, const float* __restrict__ prm, const Cells* cells, CPtr2 zp, Ptr2 dyp,
                                                           int64_t rows0, int64_t rows1) {
    const int id = id0 + 2 * blockIdx.y;
    const float* __restrict__ z = zp.p[blockIdx.y];
    float* __restrict__ dy = dyp.p[blockIdx.y];
    const int64_t rows = blockIdx.y ? rows1 : rows0;
    __shared__ BnCoef cf[MAXC];
    __shared__ float s1[MAXC], s2[MAXC];
    const int C = g.bn_ch[id];
    if (threadIdx.x < C) {
        cf[threadIdx.x] = fbn(g, cells, prm, nullptr, 1, id, threadIdx.x);
        s1[threadIdx.x] = (float)(cell_bwd(cells, id, threadIdx.x, 0) / g.cnt[id]);
        s2[threadIdx.x] = (float)(cell_bwd(cells, id, threadIdx.x, 1) / g.cnt[id]);
    }
    __syncthreads();
    const int64_t total = rows * C;
    fc_walk(total, [&](auto e) {
        const int c = (int)(e % C);
        const float xh = (z[e] - cf[c].mean) * cf[c].inv;
        dy[e] = cf[c].sc * (dy[e] - s1[c] - xh * s2[c]);
    });
}

// ... and as a transform in the LOAD of a consumer (round 4: the two channel-major passes over [M, C, L] were 16 + 17 us launches on the
// backward chain for 2 flops per element; the convolution gradient kernels that read their result apply it to each element they load)
struct BnBwdLds {
    BnCoef cf[MAXC];
    float s1[MAXC], s2[MAXC];
    __device__ __forceinline__ void init(const FcGeom& g, const Cells* cells, const float* prm, int id) {
        if ((int)threadIdx.x < g.bn_ch[id]) {
            cf[threadIdx.x] = fbn(g, cells, prm, nullptr, 1, id, threadIdx.x);
            s1[threadIdx.x] = (float)(cell_bwd(cells, id, threadIdx.x, 0) / g.cnt[id]);
            s2[threadIdx.x] = (float)(cell_bwd(cells, id, threadIdx.x, 1) / g.cnt[id]);
        }
    }
    __device__ __forceinline__ float apply(float dy, float z, int c) const {
        const float xh = (z - cf[c].mean) * cf[c].inv;
        return cf[c].sc * (dy - s1[c] - xh * s2[c]);
    }
};

// the same for channel-major rows [m][C][L]
__global__ __launch_bounds__(FB) void fc_bn_chan_bwd_kernel(FcGeom g, int id, int L, const float* __restrict__ prm, const Cells* cells,
                                                           const float* __restrict__ z, float* __restrict__ dy, int64_t total) {
    __shared__ BnCoef cf[MAXC];
    __shared__ float s1[MAXC], s2[MAXC];
    const int C = g.bn_ch[id];
    if (threadIdx.x < C) {
        cf[threadIdx.x] = fbn(g, cells, prm, nullptr, 1, id, threadIdx.x);
        s1[threadIdx.x] = (float)(cell_bwd(cells, id, threadIdx.x, 0) / g.cnt[id]);
        s2[threadIdx.x] = (float)(cell_bwd(cells, id, threadIdx.x, 1) / g.cnt[id]);
    }
    __syncthreads();
    for (int64_t e = (int64_t)blockIdx.x * FB + threadIdx.x; e < total; e += (int64_t)gridDim.x * FB) {
        const int c = (int)((e / L) % C);
        const float xh = (z[e] - cf[c].mean) * cf[c].inv;
        dy[e] = cf[c].sc * (dy[e] - s1[c] - xh * s2[c]);
    }
}

// graph backward: d adjacency, softmax / leaky backward, d mapping, d normalised features.  Every graph writes its own
// [Q, D2] contribution blocks (cX over its dAX block, which it has consumed; cM); fc_graph_gather_kernel adds the <= 2 graphs
// that contain a row -- no atomics: the first version scatter-added 5.5 M fp32 atomics per step onto the rows (slow, and the
// summation order of overlapping windows was not reproducible).
__global__ __launch_bounds__(FB) void fc_graph_bwd_kernel(FcGeom g, int blk, const float* __restrict__ prm, const Cells* cells,
                                                         const float* __restrict__ F, const float* __restrict__ Mm,
                                                         const float* __restrict__ P, float* dAX /* in: d AX, out: cX */,
                                                         float* __restrict__ cM) {
    // dynamic LDS sized for this wiring's Q x D2 (51 KB at the limits: three workgroups per CU)
    extern __shared__ float fc_graph_lds[];
    __shared__ BnCoef cd[MAXD];
    __shared__ int64_t rows[MAXQ];
    const int Q = g.Q, D2 = g.D2, N = g.N, tid = threadIdx.x, FT = blockDim.x;
    const int DP = D2 + 1, QP = Q + 1;
    float* mm = fc_graph_lds;               // [Q][DP]
    float* xb = mm + Q * DP;
    float* da = xb + Q * DP;
    float* Pm = da + Q * DP;                // [Q][QP]
    float* T = Pm + Q * QP;
    float* Sm = T + Q * QP;                 // pre-activation M M^T - 1e8 I (for the leaky slope)
    if (tid < D2) cd[tid] = fbn(g, cells, prm, nullptr, 1, 3 + 2 * blk, tid);
    __syncthreads();
    for (int64_t gi = blockIdx.x; gi < g.G[blk]; gi += gridDim.x) {
        if (tid < Q) rows[tid] = graph_row(g, blk, gi, tid);
        __syncthreads();
        for (int e = tid; e < Q * D2; e += FT) {
            const int q = e / D2, d = e - q * D2;
            const int64_t r = rows[q];
            mm[q * DP + d] = Mm[r * D2 + d] + prm[g.o_bmap[blk] + d];
            xb[q * DP + d] = fmaf(F[r * D2 + d], cd[d].sc, cd[d].sh);
            da[q * DP + d] = dAX[(gi * Q + q) * D2 + d];
        }
        for (int e = tid; e < Q * Q; e += FT) Pm[(e / Q) * QP + e % Q] = P[gi * Q * Q + e];
        __syncthreads();
        // d Adj -> d P (masked) ; d Xbn = Adj^T dAX
        for (int e = tid; e < Q * Q; e += FT) {
            const int i = e / Q, j = e - i * Q;
            float s = 0.f, sm = 0.f;
#pragma unroll 8
            for (int d = 0; d < D2; ++d) {
                s = fmaf(da[i * DP + d], xb[j * DP + d], s);
                sm = fmaf(mm[i * DP + d], mm[j * DP + d], sm);
            }
            T[i * QP + j] = s * (((i < N) == (j < N)) ? 1.f : DECAY);
            Sm[i * QP + j] = i == j ? sm - 1e8f : sm;
        }
        for (int e = tid; e < Q * D2; e += FT) {
            const int j = e / D2, d = e - j * D2;
            float s = 0.f;
#pragma unroll 8
            for (int i = 0; i < Q; ++i)
                s = fmaf((Pm[i * QP + j] + (i == j ? 1.f : 0.f)) * (((i < N) == (j < N)) ? 1.f : DECAY), da[i * DP + d], s);
            dAX[(gi * Q + j) * D2 + d] = s;               // (this graph's dAX block is in LDS since the barrier above)
        }
        __syncthreads();
        if (tid < Q) {                                   // softmax backward per row, then the leaky slope of the pre-activation
            float dot = 0.f;
#pragma unroll 8
            for (int j = 0; j < Q; ++j) dot = fmaf(T[tid * QP + j], Pm[tid * QP + j], dot);
#pragma unroll 8
            for (int j = 0; j < Q; ++j) T[tid * QP + j] = Pm[tid * QP + j] * (T[tid * QP + j] - dot) * (Sm[tid * QP + j] > 0.f ? 1.f : LEAKY);
        }
        __syncthreads();
        for (int e = tid; e < Q * D2; e += FT) {          // d Mm = (dS + dS^T) Mm
            const int i = e / D2, d = e - i * D2;
            float s = 0.f;
#pragma unroll 8
            for (int j = 0; j < Q; ++j) s = fmaf(T[i * QP + j] + T[j * QP + i], mm[j * DP + d], s);
            cM[(gi * Q + i) * D2 + d] = s;
        }
        __syncthreads();
    }
}

// The whole window block of the forward in the same wavefront: the mapping M = F W_map^T in front (result lane = node, register = mapped
// feature krow(r, h): exactly the operand form of S = M' M'^T, whose k order is free because both operands are the same registers) and
// the block's Linear z5 = AX W_theta^T + b with its BatchNorm statistics behind (AX^T is already the a-operand form).  Replaces
// [mapping GEMM, graph kernel, theta GEMM, bias + statistics kernel]: three launches and the Mm / AX round trips less per block.
template <int D2T, bool BF>
__global__ __launch_bounds__(64 * FC_MX_WAVES) void fc_block_mx_kernel(FcGeom g, const float* __restrict__ prm, const float* __restrict__ running,
                                                                       Cells* cells, int training, const float* __restrict__ F, Ptr2 Mmp, Ptr2 Pp,
                                                                       Ptr2 AXp, Ptr2 z5p) {
    const int blk = blockIdx.y;
    float* __restrict__ Mm = Mmp.p[blk];
    float* __restrict__ P = Pp.p[blk];
    float* __restrict__ AX = AXp.p[blk];
    float* __restrict__ z5 = z5p.p[blk];
    constexpr int HK = D2T / 2, HDT = D2T / 2;                         // hidden width of the block's Linear = D2 / 2
    __shared__ BnCoef cd[D2T];
    __shared__ float bm[D2T];
    if (threadIdx.x < D2T) {
        cd[threadIdx.x] = fbn(g, cells, prm, running, training, 3 + 2 * blk, threadIdx.x);
        bm[threadIdx.x] = prm[g.o_bmap[blk] + threadIdx.x];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5, Q = g.Q, N = g.N;
    const float sc = c < D2T ? cd[c < D2T ? c : 0].sc : 0.f, sh = c < D2T ? cd[c < D2T ? c : 0].sh : 0.f;
    const int cq = c < Q ? c : Q - 1;
    // constant operands: row c of W_map over this half's k, row c of W_theta over k = krow(step, half), the Linear's bias by register
    float wm[HK], wt[HK], bt[HDT / 2];
#pragma unroll
    for (int s = 0; s < HK; ++s) {
        wm[s] = c < D2T ? prm[g.o_map[blk] + c * D2T + HK * h + s] : 0.f;
        wt[s] = c < HDT ? prm[g.o_th[blk] + c * D2T + fc_krow(s, h)] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < HDT / 2; ++r) bt[r] = prm[g.o_thb[blk] + fc_krow(r, h)];
    float st_s[HDT / 2], st_q[HDT / 2];                                // this lane's share of the statistics: channel krow(r, h)
#pragma unroll
    for (int r = 0; r < HDT / 2; ++r) st_s[r] = st_q[r] = 0.f;
    const fc_f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t gi = (int64_t)blockIdx.x * FC_MX_WAVES + (threadIdx.x >> 6); gi < g.G[blk]; gi += (int64_t)gridDim.x * FC_MX_WAVES) {
        const int64_t b = gi / g.W[blk];
        const int64_t row0 = (b * g.NP + (gi - b * g.W[blk]) * g.S[blk]) * g.N;
        float fh[HK];                                                  // node c's half row of F (k = HK h + step)
        {
            const float4* src = reinterpret_cast<const float4*>(F + (row0 + cq) * D2T + HK * h);
#pragma unroll
            for (int v = 0; v < HK / 4; ++v) {
                const float4 t = src[v];
                fh[4 * v] = t.x; fh[4 * v + 1] = t.y; fh[4 * v + 2] = t.z; fh[4 * v + 3] = t.w;
            }
        }
        float xk[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int node = fc_krow(s, h);
            const float fv = F[(row0 + (node < Q ? node : Q - 1)) * D2T + (c < D2T ? c : 0)];      // (unconditional: the sixteen loads go out together)
            xk[s] = (c < D2T && node < Q) ? fmaf(fv, sc, sh) : 0.f;
        }
        // M^T = W_map F^T: lane = node, register = mapped feature krow(r, h)
        const fc_f32x16 M = fc_prod<BF, HK>(wm, fh, zero);
        if (c < Q) {                                                   // the backward reads the mapping (overlapping windows write the same values)
            float* mr = Mm + (row0 + c) * D2T + 4 * h;
#pragma unroll
            for (int m = 0; m < D2T / 8; ++m) *reinterpret_cast<float4*>(mr + 8 * m) = make_float4(M[4 * m], M[4 * m + 1], M[4 * m + 2], M[4 * m + 3]);
        }
        float mb[HK];
#pragma unroll
        for (int s = 0; s < HK; ++s) mb[s] = M[s] + bm[fc_krow(s, h)];
        const fc_f32x16 S = fc_prod<BF, HK>(mb, mb, zero);
        float t[16], mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = fc_krow(r, h);
            const float v = leaky(i == c ? S[r] - 1e8f : S[r]);
            t[r] = i < Q ? v : -INFINITY;
            mx = fmaxf(mx, t[r]);
        }
        mx = fmaxf(mx, fc_swap32(mx));
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            t[r] = __expf(t[r] - mx);
            sum += t[r];
        }
        sum += fc_swap32(sum);
        const float inv = 1.0f / sum;
        if (c < Q) {
            float* pr = P + (gi * Q + c) * Q + 4 * h;
#pragma unroll
            for (int m = 0; m < 4; ++m)
                if (8 * m + 4 * h < Q)
                    *reinterpret_cast<float4*>(pr + 8 * m) = make_float4(t[4 * m] * inv, t[4 * m + 1] * inv, t[4 * m + 2] * inv, t[4 * m + 3] * inv);
        }
        float adjr[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int j = fc_krow(s, h);
            const float adj = (t[s] * inv + (j == c ? 1.f : 0.f)) * (((c < N) == (j < N)) ? 1.f : DECAY);
            adjr[s] = j < Q ? adj : 0.f;
        }
        const fc_f32x16 A = fc_prod<BF, 16>(xk, adjr, zero);
        if (c < Q) {
            float* ar = AX + (gi * Q + c) * D2T + 4 * h;
#pragma unroll
            for (int m = 0; m < D2T / 8; ++m) *reinterpret_cast<float4*>(ar + 8 * m) = make_float4(A[4 * m], A[4 * m + 1], A[4 * m + 2], A[4 * m + 3]);
        }
        // z5^T = W_theta AX^T: lane = node, register = output channel krow(r, h)
        float ak[HK];
#pragma unroll
        for (int s = 0; s < HK; ++s) ak[s] = A[s];
        const fc_f32x16 Z = fc_prod<BF, HK>(wt, ak, zero);
        if (c < Q) {
            float zv[HDT / 2];
#pragma unroll
            for (int r = 0; r < HDT / 2; ++r) {
                zv[r] = Z[r] + bt[r];
                st_s[r] += zv[r];
                st_q[r] = fmaf(zv[r], zv[r], st_q[r]);
            }
            float* zr = z5 + (gi * Q + c) * HDT + 4 * h;
#pragma unroll
            for (int m = 0; m < HDT / 8; ++m) *reinterpret_cast<float4*>(zr + 8 * m) = make_float4(zv[4 * m], zv[4 * m + 1], zv[4 * m + 2], zv[4 * m + 3]);
        }
    }
    if (training) {
        // BatchNorm statistics of z5: the 32 node lanes of a half, the workgroup's wavefronts through LDS, then ONE atomic per channel and
        // workgroup (same-address fp64 atomics serialise in L2: per wavefront they cost more than the products above)
        __shared__ double red[FC_MX_WAVES][HDT][2];
#pragma unroll
        for (int r = 0; r < HDT / 2; ++r) {
            double a = (double)st_s[r], q2 = (double)st_q[r];
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) {
                a += __shfl_xor(a, o);
                q2 += __shfl_xor(q2, o);
            }
            if (c == 0) {
                red[threadIdx.x >> 6][fc_krow(r, h)][0] = a;
                red[threadIdx.x >> 6][fc_krow(r, h)][1] = q2;
            }
        }
        __syncthreads();
        if (threadIdx.x < 2 * HDT) {
            double v = 0.0;
#pragma unroll
            for (int wv = 0; wv < FC_MX_WAVES; ++wv) v += red[wv][threadIdx.x >> 1][threadIdx.x & 1];
            if (v != 0.0) atomicAdd(&cells[blockIdx.x % CELL_REP].fwd[4 + 2 * blk][threadIdx.x >> 1][threadIdx.x & 1], v);
        }
    }
}

// graph backward, one wavefront per graph on the fp32 matrix cores (see fc_graph_mx_kernel for the register forms):
//   T = (dAX X'^T) o mask, softmax backward by rows with the leaky slope of S = M' M'^T - 1e8 I, cX = Adj^T dAX, cM = (dS + dS^T) M'.
// The softmax reductions run inside a lane when the lane is the ROW: T is computed transposed (a-operand X', b-operand dAX), P is read
// in both orientations, and dS^T comes from one product with the identity (a-operand dS: the result's register <-> lane roles swap).
// FUSED: the gradient arrives as d z5 and d AX = d z5 W_theta is formed here in both register forms (eight more products, K = D2 / 2)
// instead of a GEMM launch and two reads of its result; the half rows of X' are then read in the order krow(step, half) of that form.
template <int D2T, bool FUSED, bool BF>
__global__ __launch_bounds__(64 * FC_MX_WAVES, 2) void fc_graph_bwd_mx_kernel(FcGeom g, const float* __restrict__ prm, const Cells* cells,
                                                                           const float* __restrict__ F, CPtr2 Mmp, CPtr2 Pp, CPtr2 dz5p,
                                                                           Ptr2 dAXp /* in (not FUSED): d AX; out: cX */, Ptr2 cMp) {
    const int blk = blockIdx.y;
    const float* __restrict__ Mm = Mmp.p[blk];
    const float* __restrict__ P = Pp.p[blk];
    const float* __restrict__ dz5 = dz5p.p[blk];
    float* dAX = dAXp.p[blk];
    float* __restrict__ cM = cMp.p[blk];
    constexpr int HK = D2T / 2, HO = D2T / 4;                          // HO: k of a half-wave in the product over the Linear's D2 / 2 outputs
    __shared__ BnCoef cd[D2T];
    __shared__ float bm[D2T];
    __shared__ float trt[FC_MX_WAVES][32 * 33];
    if (threadIdx.x < D2T) {
        cd[threadIdx.x] = fbn(g, cells, prm, nullptr, 1, 3 + 2 * blk, threadIdx.x);
        bm[threadIdx.x] = prm[g.o_bmap[blk] + threadIdx.x];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5, Q = g.Q, N = g.N;
    const int cf = c < D2T ? c : 0;
    const float bc = bm[cf];
    const int cq = c < Q ? c : Q - 1;
    const fc_f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float wth[HO];                                                     // FUSED: W_theta[HO h + step][c]
#pragma unroll
    for (int s = 0; s < HO; ++s) wth[s] = (FUSED && c < D2T) ? prm[g.o_th[blk] + (HO * h + s) * D2T + c] : 0.f;
    for (int64_t gi = (int64_t)blockIdx.x * FC_MX_WAVES + (threadIdx.x >> 6); gi < g.G[blk]; gi += (int64_t)gridDim.x * FC_MX_WAVES) {
        const int64_t b = gi / g.W[blk];
        const int64_t row0 = (b * g.NP + (gi - b * g.W[blk]) * g.S[blk]) * g.N;
        // half rows of node c: mapped features (k = HK h + step), normalised features and incoming gradient (k = HK h + step, or
        // krow(step, half) when the gradient is formed here)
        float mh[HK], xh[HK], dh[HK], dk[16];
        {
            const float4* ms = reinterpret_cast<const float4*>(Mm + (row0 + cq) * D2T + HK * h);
#pragma unroll
            for (int v = 0; v < HK / 4; ++v) {
                const int k0 = FUSED ? 8 * v + 4 * h : HK * h + 4 * v;
                const float4 m4 = ms[v], f4 = *reinterpret_cast<const float4*>(F + (row0 + cq) * D2T + k0);
                const float mv[4] = {m4.x, m4.y, m4.z, m4.w}, fv[4] = {f4.x, f4.y, f4.z, f4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    mh[4 * v + e] = mv[e] + bm[HK * h + 4 * v + e];
                    xh[4 * v + e] = fmaf(fv[e], cd[k0 + e].sc, cd[k0 + e].sh);
                }
            }
        }
        if constexpr (FUSED) {
            float zo[HO];                                              // node c's half row of d z5 (k = HO h + step)
            const float4* zs = reinterpret_cast<const float4*>(dz5 + (gi * Q + cq) * HK + HO * h);
#pragma unroll
            for (int v = 0; v < HO / 4; ++v) {
                const float4 z4 = zs[v];
                zo[4 * v] = z4.x; zo[4 * v + 1] = z4.y; zo[4 * v + 2] = z4.z; zo[4 * v + 3] = z4.w;
            }
            const fc_f32x16 DH = fc_prod<BF, HO>(wth, zo, zero);          // (register -> feature krow, lane -> node)
            const fc_f32x16 DK = fc_prod<BF, HO>(zo, wth, zero);          // (register -> node krow, lane -> feature)
#pragma unroll
            for (int s = 0; s < HK; ++s) dh[s] = DH[s];
#pragma unroll
            for (int s = 0; s < 16; ++s) dk[s] = DK[s];
        } else {
            const float4* ds = reinterpret_cast<const float4*>(dAX + (gi * Q + cq) * D2T + HK * h);
#pragma unroll
            for (int v = 0; v < HK / 4; ++v) {
                const float4 d4 = ds[v];
                dh[4 * v] = d4.x; dh[4 * v + 1] = d4.y; dh[4 * v + 2] = d4.z; dh[4 * v + 3] = d4.w;
            }
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int i = fc_krow(s, h);
                const float v = dAX[(gi * Q + (i < Q ? i : Q - 1)) * D2T + cf];
                dk[s] = (c < D2T && i < Q) ? v : 0.f;
            }
        }
        // The products below are ordered so that an operand is loaded right before its product and dies behind it, with a compiler
        // barrier between the stages: hoisted to the top (what the scheduler does on its own) the operands of all five products are
        // live at once -- 308 registers, one wavefront per SIMD and nothing to hide its ~70 loads per graph behind.  This way two fit
        // (205 registers; three spill and are slower): 0.692 -> 0.681 ms per step.
        const fc_f32x16 Tt = fc_prod<BF, HK>(xh, dh, zero);               // (register -> j, lane -> i): sum_d X'[j][d] dAX[i][d]
        const fc_f32x16 Sm = fc_prod<BF, HK>(mh, mh, zero);
        asm volatile("" ::: "memory");
        // softmax backward of row c (P by rows: this lane's row), then the leaky slope of the pre-activation
        float ds_[16];
        {
            float pl[16], dot = 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (8 * m + 4 * h < Q) t4 = *reinterpret_cast<const float4*>(P + (gi * Q + cq) * Q + 8 * m + 4 * h);
                pl[4 * m] = t4.x; pl[4 * m + 1] = t4.y; pl[4 * m + 2] = t4.z; pl[4 * m + 3] = t4.w;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = fc_krow(r, h);
                ds_[r] = Tt[r] * (((c < N) == (j < N)) ? 1.f : DECAY);
                dot = fmaf(ds_[r], pl[r], dot);                         // pl is 0 beyond the graph
            }
            dot += fc_swap32(dot);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = fc_krow(r, h);
                const float pre = j == c ? Sm[r] - 1e8f : Sm[r];
                ds_[r] = (j < Q && c < Q) ? pl[r] * (ds_[r] - dot) * (pre > 0.f ? 1.f : LEAKY) : 0.f;
            }
        }
        asm volatile("" ::: "memory");
        // cX^T = dAX^T Adj: a-operand the gradient by node (lane = feature), b-operand column c of the adjacency (P by columns) -> lane = node j
        {
            fc_f32x16 CX = zero;
            // (the sixteen loads unconditional -- row index clamped -- and ahead of the products: as `i < Q ? P[..] : 0` each one sat in
            // its own branch with a full wait between it and its product, sixteen dependent memory round trips per graph; same below)
            float pc[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int i = fc_krow(s, h);
                pc[s] = P[(gi * Q + (i < Q ? i : Q - 1)) * Q + cq];
            }
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int i = fc_krow(s, h);
                const float adj = (pc[s] + (i == c ? 1.f : 0.f)) * (((i < N) == (c < N)) ? 1.f : DECAY);
                pc[s] = i < Q ? adj : 0.f;
            }
            CX = fc_prod<BF, 16>(dk, pc, CX);
            if (c < Q) {                                                // (all loads of this graph's dAX block are behind us)
                float* xr = dAX + (gi * Q + c) * D2T + 4 * h;
#pragma unroll
                for (int m = 0; m < D2T / 8; ++m) *reinterpret_cast<float4*>(xr + 8 * m) = make_float4(CX[4 * m], CX[4 * m + 1], CX[4 * m + 2], CX[4 * m + 3]);
            }
        }
        asm volatile("" ::: "memory");
        // dS^T: register and lane swap roles through a wavefront-private LDS tile (16 writes + 16 reads, in order: no barrier; it was 16 matrix
        // instructions against the identity)
        fc_f32x16 St;
        {
            float* tt = &trt[threadIdx.x >> 6][0];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s = 0; s < 16; ++s) tt[fc_krow(s, h) * 33 + c] = ds_[s];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s = 0; s < 16; ++s) St[s] = tt[c * 33 + fc_krow(s, h)];
        }
        // cM^T = M'^T (dS + dS^T)^T: a-operand the mapped features by node (lane = feature), b-operand row c of dS + dS^T
        fc_f32x16 CM = zero;
        float mk[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int node = fc_krow(s, h);
            mk[s] = Mm[(row0 + (node < Q ? node : Q - 1)) * D2T + cf];
        }
        float dsum[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int node = fc_krow(s, h);
            mk[s] = (c < D2T && node < Q) ? mk[s] + bc : 0.f;
            dsum[s] = ds_[s] + St[s];
        }
        CM = fc_prod<BF, 16>(mk, dsum, CM);
        if (c < Q) {
            float* mr = cM + (gi * Q + c) * D2T + 4 * h;
#pragma unroll
            for (int m = 0; m < D2T / 8; ++m) *reinterpret_cast<float4*>(mr + 8 * m) = make_float4(CM[4 * m], CM[4 * m + 1], CM[4 * m + 2], CM[4 * m + 3]);
        }
    }
}

// gX[r][d] = sum over the graphs that contain row r = (b, t, n) of their contribution; the same for gM.  Windows hold two
// consecutive patches (tau = 0, 1) and start every S[blk] patches: row t is node tau*N + n of window w = (t - tau) / S.
__global__ __launch_bounds__(FB) void fc_graph_gather_kernel(FcGeom g, CPtr2 cXp, CPtr2 cMp, Ptr2 gXp, Ptr2 gMp) {
    const int blk = blockIdx.y;
    const float* __restrict__ cX = cXp.p[blk];
    const float* __restrict__ cM = cMp.p[blk];
    float* __restrict__ gX = gXp.p[blk];
    float* __restrict__ gM = gMp.p[blk];
    const int64_t total = g.M * g.D2;
    const int S = g.S[blk], W = g.W[blk];
    fc_walk(total, [&](auto e) {
        const int d = (int)(e % g.D2);
        const auto r = e / g.D2;
        const int n = (int)(r % g.N), t = (int)((r / g.N) % g.NP);
        const auto b = r / (g.N * g.NP);
        float ax = 0.f, am = 0.f;
#pragma unroll
        for (int tau = 0; tau < 2; ++tau) {
            const int tt = t - tau;
            if (tt >= 0 && tt % S == 0 && tt / S < W) {
                const auto src = ((b * W + tt / S) * g.Q + tau * g.N + n) * g.D2 + d;
                ax += cX[src];
                am += cM[src];
            }
        }
        gX[e] = ax;
        gM[e] = am;
    });
}

// sums for the window BatchNorms' backward from the row-accumulated gradients gX_b:  sum g, sum g * xhat
__global__ __launch_bounds__(FB) void fc_feat_stats_kernel(FcGeom g, const float* __restrict__ prm, Cells* cells,
                                                          const float* __restrict__ F, const float* __restrict__ gX0,
                                                          const float* __restrict__ gX1) {
    __shared__ double sl[2 * BS_DOUBLES];
    __shared__ BnCoef c0[MAXD], c1[MAXD];
    if (threadIdx.x < g.D2) {
        c0[threadIdx.x] = fbn(g, cells, prm, nullptr, 1, 3, threadIdx.x);
        c1[threadIdx.x] = fbn(g, cells, prm, nullptr, 1, 5, threadIdx.x);
    }
    BlockStats s0, s1;
    s0.init(sl, g.D2);
    s1.init(sl + BS_DOUBLES, g.D2);
    const int64_t total = g.M * g.D2;
    fc_walk(total, [&](auto e) {
        const int d = (int)(e % g.D2);
        const float f = F[e], a = gX0[e], b = gX1[e];
        s0.add(d, a, a * (f - c0[d].mean) * c0[d].inv);
        s1.add(d, b, b * (f - c1[d].mean) * c1[d].inv);
    });
    s0.flush(cells[blockIdx.x % CELL_REP].bwd[3], g.D2);
    s1.flush(cells[blockIdx.x % CELL_REP].bwd[5], g.D2);
}

// The window BatchNorms' backward, both blocks' d M W_map products and the positional-encoding / dropout backward with the BatchNorm-c
// sums in ONE launch (row-group mapping: D2 lanes per row): d F = bn'(gX_0) + bn'(gX_1) + gM_0 W_map0 + gM_1 W_map1, then the dropout
// mask and the sums of dy and dy * xhat(z3).  Four launches of 17 + 16 + 18 + 9 us before.
__global__ __launch_bounds__(FB) void fc_feat_pe_bwd_kernel(FcGeom g, const float* __restrict__ prm, Cells* cells, const float* __restrict__ F,
                                                           const float* __restrict__ gX0, const float* __restrict__ gX1,
                                                           const float* __restrict__ gM0, const float* __restrict__ gM1,
                                                           const float* __restrict__ z3, float* __restrict__ dF, uint32_t drop_thr,
                                                           float drop_scale, uint32_t drop_key, const uint32_t* key_dev, int64_t row_offset) {
    __shared__ double sl[BS_DOUBLES];
    __shared__ BnCoef c0[MAXD], c1[MAXD], cc[MAXD];
    __shared__ float m0[MAXD][2], m1[MAXD][2];
    __shared__ float wl[2][MAXD * MAXD];
    __shared__ float tile[2][2][FB];
    const int D2 = g.D2;
    if ((int)threadIdx.x < D2) {
        const int d = threadIdx.x;
        c0[d] = fbn(g, cells, prm, nullptr, 1, 3, d);
        c1[d] = fbn(g, cells, prm, nullptr, 1, 5, d);
        cc[d] = fbn(g, cells, prm, nullptr, 1, 2, d);
        m0[d][0] = (float)(cell_bwd(cells, 3, d, 0) / g.cnt[3]); m0[d][1] = (float)(cell_bwd(cells, 3, d, 1) / g.cnt[3]);
        m1[d][0] = (float)(cell_bwd(cells, 5, d, 0) / g.cnt[5]); m1[d][1] = (float)(cell_bwd(cells, 5, d, 1) / g.cnt[5]);
    }
    for (int e = threadIdx.x; e < D2 * D2; e += FB) { wl[0][e] = prm[g.o_map[0] + e]; wl[1][e] = prm[g.o_map[1] + e]; }
    BlockStats st;
    st.init(sl, D2);
    const uint32_t key = key_dev ? *key_dev : drop_key;
    const RowGroup rg(D2);
    const int d = rg.pos;
    const int64_t stride = (int64_t)gridDim.x * rg.rows_per;
    int buf = 0;
    for (int64_t mb = (int64_t)blockIdx.x * rg.rows_per; mb < g.M; mb += stride, buf ^= 1) {
        const int64_t m = mb + rg.sub;
        const bool row_on = rg.on && m < g.M;
        const int64_t e = m * D2 + d;
        float f = 0.f, a0 = 0.f, a1 = 0.f, zz = 0.f;
        if (row_on) {
            tile[buf][0][rg.sub * D2 + d] = gM0[e];
            tile[buf][1][rg.sub * D2 + d] = gM1[e];
            f = F[e]; a0 = gX0[e]; a1 = gX1[e]; zz = z3[e];
        }
        lds_barrier();
        if (row_on) {
            const int t = (int)((m / g.N) % g.NP);
            const float k0 = mult(g, 0, t), k1 = mult(g, 1, t);
            const float x0 = (f - c0[d].mean) * c0[d].inv, x1 = (f - c1[d].mean) * c1[d].inv;
            float v = c0[d].sc * (a0 - k0 * (m0[d][0] + x0 * m0[d][1])) + c1[d].sc * (a1 - k1 * (m1[d][0] + x1 * m1[d][1]));
            const float* r0 = &tile[buf][0][rg.sub * D2];
            const float* r1 = &tile[buf][1][rg.sub * D2];
            float p0 = 0.f, p1 = 0.f;
            for (int o = 0; o < D2; ++o) {
                p0 = fmaf(r0[o], wl[0][o * D2 + d], p0);
                p1 = fmaf(r1[o], wl[1][o * D2 + d], p1);
            }
            v = (v + p0) + p1;                                         // (the order of the unfused path: block 0's product, then block 1's)
            if (drop_thr) {
                const uint32_t ctr = (uint32_t)((m + row_offset) * D2 + d);
                v = lowbias32(ctr ^ key) >= drop_thr ? v * drop_scale : 0.f;
            }
            dF[e] = v;
            st.add(d, v, v * (zz - cc[d].mean) * cc[d].inv);
        }
    }
    st.flush(cells[blockIdx.x % CELL_REP].bwd[2], D2);
}

// d a2 = d z3 W3 and the activation's backward with the BatchNorm-b sums in one launch (the counterpart of fc_proj3_kernel):
// dy2 = d a2 [a2 > 0], sums of dy2 and dy2 * xhat(z2) per channel
__global__ __launch_bounds__(FB) void fc_proj3_bwd_kernel(FcGeom g, const float* __restrict__ prm, Cells* cells, const float* __restrict__ z2,
                                                         const float* __restrict__ a2, const float* __restrict__ dz3,
                                                         float* __restrict__ da2) {
    extern __shared__ float proj_lds[];                                // W3s[D2][CL + 1] | tile[2][FB]
    __shared__ double sl[BS_DOUBLES];
    __shared__ BnCoef cb[MAXC];
    const int CL = g.CL, D2 = g.D2, L2 = g.L2, WS = CL + 1;
    float* w3 = proj_lds;
    float* tile = proj_lds + D2 * WS;
    if (threadIdx.x < g.CO) cb[threadIdx.x] = fbn(g, cells, prm, nullptr, 1, 1, threadIdx.x);
    for (int e = threadIdx.x; e < D2 * CL; e += FB) w3[(e / CL) * WS + e % CL] = prm[g.o_W3 + e];
    BlockStats st;
    st.init(sl, g.CO);
    const RowGroup rg(CL > D2 ? CL : D2);
    const int co = rg.pos < CL ? rg.pos / L2 : 0;
    const float* wc = w3 + (rg.pos < CL ? rg.pos : 0);                 // column rg.pos of W3: lanes consecutive
    const int64_t stride = (int64_t)gridDim.x * rg.rows_per;
    auto fetch_d = [&](int64_t m0) {
        const int64_t m = m0 + rg.sub;
        return (rg.on && m < g.M && rg.pos < D2) ? dz3[m * D2 + rg.pos] : 0.f;
    };
    float nd = fetch_d((int64_t)blockIdx.x * rg.rows_per);
    int buf = 0;
    for (int64_t m0 = (int64_t)blockIdx.x * rg.rows_per; m0 < g.M; m0 += stride, buf ^= 1) {
        const int64_t m = m0 + rg.sub;
        const bool row_on = rg.on && m < g.M;
        const float dcur = nd;
        nd = fetch_d(m0 + stride);
        float av = 0.f, zv = 0.f;
        if (row_on && rg.pos < CL) { av = a2[m * CL + rg.pos]; zv = z2[m * CL + rg.pos]; }
        if (row_on && rg.pos < D2) tile[buf * FB + rg.sub * rg.tile + rg.pos] = dcur;
        lds_barrier();
        if (row_on && rg.pos < CL) {
            const float* dr = tile + buf * FB + rg.sub * rg.tile;
            float d = 0.f;
            for (int o = 0; o < D2; ++o) d = fmaf(dr[o], wc[o * WS], d);
            const float dy = av > 0.f ? d : 0.f;
            da2[m * CL + rg.pos] = dy;
            st.add(co, dy, dy * (zv - cb[co].mean) * cb[co].inv);
        }
    }
    st.flush(cells[blockIdx.x % CELL_REP].bwd[1], g.CO);
}

// dy2 = da2 * [a2 > 0] (in place); BatchNorm-b backward sums
__global__ __launch_bounds__(FB) void fc_act2_bwd_kernel(FcGeom g, const float* __restrict__ prm, Cells* cells,
                                                        const float* __restrict__ z2, const float* __restrict__ a2,
                                                        float* __restrict__ da2) {
    __shared__ double sl[BS_DOUBLES];
    __shared__ BnCoef cb[MAXC];
    if (threadIdx.x < g.CO) cb[threadIdx.x] = fbn(g, cells, prm, nullptr, 1, 1, threadIdx.x);
    BlockStats st;
    st.init(sl, g.CO);
    const int64_t total = g.M * g.CL;
    for (int64_t e = (int64_t)blockIdx.x * FB + threadIdx.x; e < total; e += (int64_t)gridDim.x * FB) {
        const int co = (int)((e / g.L2) % g.CO);
        const float dy = a2[e] > 0.f ? da2[e] : 0.f;
        da2[e] = dy;
        st.add(co, dy, dy * (z2[e] - cb[co].mean) * cb[co].inv);
    }
    st.flush(cells[blockIdx.x % CELL_REP].bwd[1], g.CO);
}

// dy1[m][ci][q] = [a1 > 0] * sum_co sum_k w2[co][ci][k] * dz2[m][co][q + 1 - k]; BatchNorm-a backward sums  (row-group mapping, see fc_conv2_kernel)
// (`z2` != nullptr: `dz2` is still d y2 -- the gradient in front of BatchNorm 1's backward, which is applied to every element as it is loaded)
// <SK, SL1, SL2, SH1, SCO>: the wiring's convolution shape as compile-time constants (0 = generic): the index arithmetic of a row -- divisions
// by L1 / L2, tap ranges, the CO x K product loop -- is most of this kernel's instructions at 2 x 3 x 4 x 8 x 6 (the reference's FD004 wiring)
template <int SK, int SL1, int SL2, int SH1, int SCO>
__global__ __launch_bounds__(FB) void fc_conv2_dx_kernel(FcGeom g, const float* __restrict__ prm, Cells* cells,
                                                        const float* __restrict__ z1, const float* __restrict__ dz2,
                                                        float* __restrict__ dy1, const float* __restrict__ z2) {
    __shared__ double sl[BS_DOUBLES];
    __shared__ BnCoef ca[16];
    __shared__ float wl[FC_W2_MAX];
    __shared__ BnBwdLds bb;
    if (z2) bb.init(g, cells, prm, 1);
    if (threadIdx.x < g.H1) ca[threadIdx.x] = fbn(g, cells, prm, nullptr, 1, 0, threadIdx.x);
    for (int e = threadIdx.x; e < g.CO * g.H1 * g.K; e += FB) wl[e] = prm[g.o_w2 + e];
    BlockStats st;
    st.init(sl, g.H1);
    const int H1 = SH1 ? SH1 : g.H1, K = SK ? SK : g.K, L1 = SL1 ? SL1 : g.L1, L2 = SL2 ? SL2 : g.L2, CO = SCO ? SCO : g.CO;
    const int CL = CO * L2, T1 = H1 * L1;
    auto one = [&](int64_t m, int pos) {
        const int ci = pos / L1, q = pos - ci * L1;
        const float zz = z1[m * T1 + pos];
        float dy = 0.f;
        if (fmaf(zz, ca[ci].sc, ca[ci].sh) > 0.f) {
            const float* dr = dz2 + m * CL;
            for (int co = 0; co < CO; ++co)
                for (int k = 0; k < K; ++k) {
                    const int p = q + 1 - k;
                    if (p >= 0 && p < L2) {
                        const float dv = z2 ? bb.apply(dr[co * L2 + p], z2[m * CL + co * L2 + p], co) : dr[co * L2 + p];
                        dy = fmaf(wl[(co * H1 + ci) * K + k], dv, dy);
                    }
                }
        }
        dy1[m * T1 + pos] = dy;
        st.add(ci, dy, dy * (zz - ca[ci].mean) * ca[ci].inv);
    };
    const RowGroup rg(T1 > CL ? T1 : CL);
    if (rg.wide()) {
        for (int64_t m = blockIdx.x; m < g.M; m += gridDim.x)
            for (int pos = threadIdx.x; pos < T1; pos += FB) one(m, pos);
    } else {
        __shared__ float tile[2][FB];                                  // the row of d z2, staged as in fc_conv2_kernel
        const int ci = rg.pos < T1 ? rg.pos / L1 : 0, q = rg.pos - ci * L1;
        const int64_t stride = (int64_t)gridDim.x * rg.rows_per;
        int buf = 0;
        auto fetch_d = [&](int64_t m0) {
            const int64_t m = m0 + rg.sub;
            if (!(rg.on && m < g.M && rg.pos < CL)) return 0.f;
            const float dv = dz2[m * CL + rg.pos];
            return z2 ? bb.apply(dv, z2[m * CL + rg.pos], rg.pos / L2) : dv;
        };
        auto fetch_z = [&](int64_t m0) {
            const int64_t m = m0 + rg.sub;
            return (rg.on && m < g.M && rg.pos < T1) ? z1[m * T1 + rg.pos] : 0.f;
        };
        float nd = fetch_d((int64_t)blockIdx.x * rg.rows_per), nz = fetch_z((int64_t)blockIdx.x * rg.rows_per);
        for (int64_t m0 = (int64_t)blockIdx.x * rg.rows_per; m0 < g.M; m0 += stride, buf ^= 1) {
            const int64_t m = m0 + rg.sub;
            const bool row_on = rg.on && m < g.M;
            const float dcur = nd, zz = nz;
            nd = fetch_d(m0 + stride);                                 // the next group's elements are in flight under this one's products
            nz = fetch_z(m0 + stride);
            if (row_on && rg.pos < CL) tile[buf][rg.sub * rg.tile + rg.pos] = dcur;
            lds_barrier();
            if (row_on && rg.pos < T1) {
                float dy = 0.f;
                if (fmaf(zz, ca[ci].sc, ca[ci].sh) > 0.f) {
                    const float* dr = &tile[buf][rg.sub * rg.tile];
                    for (int co = 0; co < CO; ++co)
                        for (int k = 0; k < K; ++k) {
                            const int p = q + 1 - k;
                            if (p >= 0 && p < L2) dy = fmaf(wl[(co * H1 + ci) * K + k], dr[co * L2 + p], dy);
                        }
                }
                dy1[m * T1 + rg.pos] = dy;
                st.add(ci, dy, dy * (zz - ca[ci].mean) * ca[ci].inv);
            }
        }
    }
    st.flush(cells[blockIdx.x % CELL_REP].bwd[0], g.H1);
}

// conv weight gradients: WHICH 2: dw2[co][ci][k] = sum_m sum_p dz2[m][co][p] * a1[m][ci][p + k - 1]
//                        WHICH 1: dw1[c][k]      = sum_m sum_p dz1[m][c][p]  * v[m][p + k - K/2]
// each workgroup reduces a contiguous chunk of rows; thread-owned outputs, one partial row per workgroup
// (`zbn` != nullptr: `dz` is still the gradient in FRONT of the convolution's BatchNorm backward -- BatchNorm WHICH - 1, pre-activations
// `zbn` = z2 | z1 --, applied to every element as it is loaded)
// (<SK, SL1, SL2, SH1, SCO>: the convolution shape as compile-time constants, 0 = generic -- see fc_conv2_dx_kernel)
template <int WHICH, int SK = 0, int SL1 = 0, int SL2 = 0, int SH1 = 0, int SCO = 0>
__device__ __forceinline__ void fc_conv_wgrad_body(const FcGeom& g_, const float* __restrict__ x, const float* __restrict__ prm,
                                                   const Cells* cells, const float* __restrict__ z1, const float* __restrict__ dz,
                                                   float* __restrict__ gpart, const float* __restrict__ zbn) {
    // the shape fields this kernel reads, constants where the instantiation fixes them
    struct Shape {
        const FcGeom& f;
        int K, L1, L2, H1, CO, CL;
        int64_t M;
        int N, NP, TL, PS;
    };
    const Shape g{g_, SK ? SK : g_.K, SL1 ? SL1 : g_.L1, SL2 ? SL2 : g_.L2, SH1 ? SH1 : g_.H1, SCO ? SCO : g_.CO,
                  (SCO ? SCO : g_.CO) * (SL2 ? SL2 : g_.L2), g_.M, g_.N, g_.NP, g_.TL, g_.PS};
    __shared__ BnCoef ca[16];
    __shared__ BnBwdLds bb;
    if (zbn) bb.init(g_, cells, prm, WHICH - 1);
    if (WHICH == 2 && threadIdx.x < g.H1) ca[threadIdx.x] = fbn(g_, cells, prm, nullptr, 1, 0, threadIdx.x);
    __syncthreads();
    const int nout = WHICH == 2 ? g.CO * g.H1 * g.K : g.H1 * g.K;
    const int64_t per = (g.M + gridDim.x - 1) / gridDim.x;
    const int64_t m0 = per * blockIdx.x, m1 = m0 + per < g.M ? m0 + per : g.M;
    // output o over rows m = mb, mb + step, ... of this workgroup's chunk
    auto partial = [&](int o, int64_t mb, int step) {
        float acc = 0.f;
        if (WHICH == 2) {
            const int k = o % g.K, ci = (o / g.K) % g.H1, co = o / (g.K * g.H1);
            for (int64_t m = mb; m < m1; m += step) {
                const float* dr = dz + m * g.CL + co * g.L2;
                const float* zb = zbn ? zbn + m * g.CL + co * g.L2 : nullptr;
                const float* zr = z1 + (m * g.H1 + ci) * g.L1;
                for (int p = 0; p < g.L2; ++p) {
                    const int q = p + k - 1;
                    if (q >= 0 && q < g.L1)
                        acc = fmaf(zb ? bb.apply(dr[p], zb[p], co) : dr[p], fmaxf(fmaf(zr[q], ca[ci].sc, ca[ci].sh), 0.f), acc);
                }
            }
        } else {
            const int k = o % g.K, c = o / g.K, pad = g.K / 2;
            for (int64_t m = mb; m < m1; m += step) {
                const int node = (int)(m % g.N), t = (int)((m / g.N) % g.NP);
                const int64_t b = m / ((int64_t)g.N * g.NP);
                const float* v = x + (b * g.N + node) * g.TL + t * g.PS;
                const float* dr = dz + (m * g.H1 + c) * g.L1;
                const float* zb = zbn ? zbn + (m * g.H1 + c) * g.L1 : nullptr;
                for (int p = 0; p < g.L1; ++p) {
                    const int j = p + k - pad;
                    if (j >= 0 && j < g.PS) acc = fmaf(zb ? bb.apply(dr[p], zb[p], c) : dr[p], v[j], acc);
                }
            }
        }
        return acc;
    };
    if (WHICH == 2 && nout * 2 <= FB && g.CL <= 64 && g.H1 * g.L1 <= 64) {
        // The second convolution's weight gradient with its operands staged through LDS: blocks of 32 rows of d z2 and of the activated
        // input a1 = relu(bn_a(z1)) arrive with coalesced loads, then the FB / nout thread slices of an output walk the block's rows in
        // LDS.  (Walking global memory row by row, four dependent loads deep, this kernel took 81 us.)
        constexpr int RB = 32;
        __shared__ float sd[RB * 64], sa[RB * 64];
        __shared__ float red[FB];
        const int CL = g.CL, T1 = g.H1 * g.L1, L1 = g.L1, L2 = g.L2;
        const int nsl = FB / nout, o = threadIdx.x % nout, sl = threadIdx.x / nout;
        const int k = o % g.K, ci = (o / g.K) % g.H1, co = o / (g.K * g.H1);
        float acc = 0.f;
        for (int64_t mb = m0; mb < m1; mb += RB) {
            const int nr = (int)(m1 - mb < RB ? m1 - mb : RB);
            for (int e = threadIdx.x; e < nr * CL; e += FB)
                sd[e] = zbn ? bb.apply(dz[mb * CL + e], zbn[mb * CL + e], (e % CL) / L2) : dz[mb * CL + e];
            for (int e = threadIdx.x; e < nr * T1; e += FB) {
                const int c = (e % T1) / L1;
                sa[e] = fmaxf(fmaf(z1[mb * T1 + e], ca[c].sc, ca[c].sh), 0.f);
            }
            lds_barrier();
            if (sl < nsl)
                for (int r = sl; r < nr; r += nsl) {
                    const float* dr = sd + r * CL + co * L2;
                    const float* ar = sa + r * T1 + ci * L1;
                    for (int p = 0; p < L2; ++p) {
                        const int q = p + k - 1;
                        if (q >= 0 && q < L1) acc = fmaf(dr[p], ar[q], acc);
                    }
                }
            lds_barrier();
        }
        red[threadIdx.x] = sl < nsl ? acc : 0.f;
        __syncthreads();
        if ((int)threadIdx.x < nout) {
            float a = 0.f;
            for (int q = 0; q < nsl; ++q) a += red[q * nout + threadIdx.x];
            gpart[(int64_t)blockIdx.x * nout + threadIdx.x] = a;
        }
    } else if (nout * 2 <= FB) {
        // few outputs: FB / nout thread slices share each output (rows interleaved), combined in a fixed order through LDS
        __shared__ float red[FB];
        const int nsl = FB / nout, o = threadIdx.x % nout, sl = threadIdx.x / nout;
        red[threadIdx.x] = sl < nsl ? partial(o, m0 + sl, nsl) : 0.f;
        __syncthreads();
        if ((int)threadIdx.x < nout) {
            float acc = 0.f;
            for (int q = 0; q < nsl; ++q) acc += red[q * nout + threadIdx.x];
            gpart[(int64_t)blockIdx.x * nout + threadIdx.x] = acc;
        }
    } else {
        for (int o = threadIdx.x; o < nout; o += FB) gpart[(int64_t)blockIdx.x * nout + o] = partial(o, m0, 1);
    }
}
template <int WHICH, int SK = 0, int SL1 = 0, int SL2 = 0, int SH1 = 0, int SCO = 0>
__global__ __launch_bounds__(FB) void fc_conv_wgrad_kernel(FcGeom g, const float* __restrict__ x, const float* __restrict__ prm,
                                                          const Cells* cells, const float* __restrict__ z1, const float* __restrict__ dz,
                                                          float* __restrict__ gpart, const float* __restrict__ zbn) {
    fc_conv_wgrad_body<WHICH, SK, SL1, SL2, SH1, SCO>(g, x, prm, cells, z1, dz, gpart, zbn);
}
// both convolutions' weight gradients in one launch (blockIdx.y = 0: the second convolution's, the longer one): as two launches the
// second convolution's sat at the end of the side stream, behind the five split-K pairs, and the step's last kernels waited ~40 us for it
template <int SK = 0, int SL1 = 0, int SL2 = 0, int SH1 = 0, int SCO = 0>
__global__ __launch_bounds__(FB) void fc_conv_wgrad_both_kernel(FcGeom g, const float* __restrict__ x, const float* __restrict__ prm,
                                                               const Cells* cells, const float* __restrict__ z1, const float* __restrict__ z2,
                                                               const float* __restrict__ dy1, const float* __restrict__ da2,
                                                               float* __restrict__ gp1, float* __restrict__ gp2) {
    if (blockIdx.y == 0) fc_conv_wgrad_body<2, SK, SL1, SL2, SH1, SCO>(g, x, prm, cells, z1, da2, gp2, z2);
    else fc_conv_wgrad_body<1, SK, SL1, SL2, SH1, SCO>(g, x, prm, cells, z1, dy1, gp1, z1);
}

// Synchronised BatchNorm: the 16 replicas of ONE reduction pair (forward: sum z, sum z^2; backward: sum dy, sum dy x-hat;
// 2 MAXC contiguous doubles) are collapsed into replica 0 and the others zeroed -- the readers' replica sum is unchanged -- so that
// the caller's all-reduce runs on one contiguous buffer.
__global__ void fc_cells_collapse_kernel(Cells* cells, int bwd, int id) {
    for (int i = threadIdx.x; i < 2 * MAXC; i += blockDim.x) {
        double v = 0.0;
        for (int r = 0; r < CELL_REP; ++r) {
            double* p = bwd ? &cells[r].bwd[id][0][0] : &cells[r].fwd[id][0][0];
            v += p[i];
            if (r) p[i] = 0.0;
        }
        (bwd ? &cells[0].bwd[id][0][0] : &cells[0].fwd[id][0][0])[i] = v;
    }
}

// conv partial rows -> gradients; BatchNorm gamma / beta gradients from the cells (x bn_scale: under synchronised BatchNorm the
// cells hold GLOBAL sums on every rank and only one rank may contribute them to the all-reduced gradient)
// (`fuse.p` != nullptr: the thread that finalises a gradient element applies Adam to that parameter on the spot, and the workgroups from
// `nfin` on apply it to every parameter whose gradient was final before this launch -- the GEMM-produced ones; the launch then sits behind
// the join with the side stream and the step has no optimizer launch)
__global__ __launch_bounds__(FB) void fc_finalize_kernel(FcGeom g, const float* __restrict__ gp1, const float* __restrict__ gp2, int rows,
                                                        const Cells* cells, float* __restrict__ grads, float bn_scale, AdamFuse fuse, int nfin) {
    const int n1 = g.H1 * g.K, n2 = g.CO * g.H1 * g.K;
    if ((int)blockIdx.x >= nfin) {
        // parameters this launch does not finalise itself: everything but the two convolution weights and the BatchNorm scales / shifts
        const int stride = ((int)gridDim.x - nfin) * FB;
        for (int i = ((int)blockIdx.x - nfin) * FB + threadIdx.x; i < g.nparam; i += stride) {
            bool mine = (i >= g.o_w1 && i < g.o_w1 + n1) || (i >= g.o_w2 && i < g.o_w2 + n2);
#pragma unroll
            for (int id = 0; id < NBN; ++id)
                mine = mine || (i >= g.bn_g[id] && i < g.bn_g[id] + g.bn_ch[id]) || (i >= g.bn_b[id] && i < g.bn_b[id] + g.bn_ch[id]);
            if (!mine) fuse(grads + i, grads[i]);
        }
        return;
    }
    // one wavefront per value: lanes stride over the partial rows, then a fixed-order butterfly (deterministic)
    const int e = (blockIdx.x * FB + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (e < n1 + n2) {
        const float* src = e < n1 ? gp1 + e : gp2 + (e - n1);
        const int n = e < n1 ? n1 : n2;
        float a = 0.f;
        for (int r = lane; r < rows; r += 64) a += src[(int64_t)r * n];
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) a += __shfl_xor(a, m, 64);
        if (lane == 0) {
            float* dst = grads + (e < n1 ? g.o_w1 + e : g.o_w2 + (e - n1));
            *dst = a;
            fuse(dst, a);
        }
    } else if (lane == 0) {
        int c = e - n1 - n2;
        for (int id = 0; id < NBN; ++id) {
            if (c < g.bn_ch[id]) {
                const float gg = bn_scale * (float)cell_bwd(cells, id, c, 1), gb = bn_scale * (float)cell_bwd(cells, id, c, 0);
                grads[g.bn_g[id] + c] = gg;
                grads[g.bn_b[id] + c] = gb;
                fuse(grads + g.bn_g[id] + c, gg);
                fuse(grads + g.bn_b[id] + c, gb);
                return;
            }
            c -= g.bn_ch[id];
        }
    }
}

// BatchNorm batch statistics out: (mean, biased var) per layer/channel, or weight * (E[z], E[z^2]) for data parallel
__global__ void fc_bn_batch_kernel(FcGeom g, const Cells* cells, float* __restrict__ bn_batch, float weight) {
    for (int id = 0; id < NBN; ++id)
        for (int c = threadIdx.x; c < g.bn_ch[id]; c += blockDim.x) {
            const double m = cell_fwd(cells, id, c, 0) / g.cnt[id], q = cell_fwd(cells, id, c, 1) / g.cnt[id];
            float* mean = bn_batch + g.bn_off[id] + c;
            float* var = mean + g.bn_ch[id];
            if (weight > 0.f) {
                *mean = (float)(weight * m);
                *var = (float)(weight * q);
            } else {
                const double v = q - m * m;
                *mean = (float)m;
                *var = (float)(v < 0.0 ? 0.0 : v);
            }
        }
}

// running statistics update for all seven layers; count[id] values per channel went into the batch statistics
__global__ void fc_bn_running_kernel(FcGeom g, float* __restrict__ bn, const float* __restrict__ batch, float momentum, int from_moments) {
    for (int id = 0; id < NBN; ++id)
        for (int c = threadIdx.x; c < g.bn_ch[id]; c += blockDim.x) {
            float mean = batch[g.bn_off[id] + c], var = batch[g.bn_off[id] + g.bn_ch[id] + c];
            if (from_moments) {
                var = var - mean * mean;
                if (var < 0.f) var = 0.f;
            }
            const double n = g.cnt[id];
            const float unbiased = n > 1.0 ? (float)(var * (n / (n - 1.0))) : var;
            float* rm = bn + g.bn_off[id] + c;
            float* rv = rm + g.bn_ch[id];
            *rm = (1.0f - momentum) * *rm + momentum * mean;
            *rv = (1.0f - momentum) * *rv + momentum * unbiased;
        }
}

__global__ void fc_fill_one_kernel(float* p) { p[0] = 1.f; }

// the head of the side stream in one launch (it was three): the constant for the bias column sums, the batch statistics for the caller's
// bucket, and -- when the caller keeps plain (mean, var) there -- the running-statistics update from the same cells
__global__ void fc_side_head_kernel(FcGeom g, const Cells* cells, float* __restrict__ one, float* __restrict__ bn_batch, float weight,
                                    float* __restrict__ bn_running, float momentum) {
    if (threadIdx.x == 0) one[0] = 1.f;
    for (int id = 0; id < NBN; ++id)
        for (int c = threadIdx.x; c < g.bn_ch[id]; c += blockDim.x) {
            const double m = cell_fwd(cells, id, c, 0) / g.cnt[id], q = cell_fwd(cells, id, c, 1) / g.cnt[id];
            float* mean = bn_batch + g.bn_off[id] + c;
            float* var = mean + g.bn_ch[id];
            if (weight > 0.f) {
                *mean = (float)(weight * m);
                *var = (float)(weight * q);
            } else {
                const double v = q - m * m;
                const float mf = (float)m, vf = (float)(v < 0.0 ? 0.0 : v);
                *mean = mf;
                *var = vf;
                if (bn_running) {                                   // (fc_bn_running_kernel's arithmetic on the values just written)
                    const double n = g.cnt[id];
                    const float unbiased = n > 1.0 ? (float)(vf * (n / (n - 1.0))) : vf;
                    float* rm = bn_running + g.bn_off[id] + c;
                    float* rv = rm + g.bn_ch[id];
                    *rm = (1.0f - momentum) * *rm + momentum * mf;
                    *rv = (1.0f - momentum) * *rv + momentum * unbiased;
                }
            }
        }
}

struct FcWs {
    size_t cells, one, z1, z2, a2, z3, F, Mm[2], P[2], AX[2], z5[2], feat, h1, h2, h3, dpred, sqerr;
    size_t dh3, dh2, dh1, dfeat, dAX[2], dz5[2], gX[2], gM[2], dMb[2], dF, da2, dy1, gp1, gp2, split, total, split_floats;
    int rows;
};

void fc_ws_layout(const FcGeom& g, FcWs* w) {
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t o = 0;
    auto take = [&](size_t floats) { const size_t r = o; o = al(o + floats * sizeof(float)); return r; };
    w->cells = o; o = al(o + sizeof(Cells) * CELL_REP);
    w->one = take(64);
    const size_t M = (size_t)g.M, B = (size_t)g.B;
    w->z1 = take(M * g.H1 * g.L1);
    w->z2 = take(M * g.CL);
    w->a2 = take(M * g.CL);
    w->z3 = take(M * g.D2);
    w->F = take(M * g.D2);
    for (int b = 0; b < 2; ++b) {
        const size_t GQ = (size_t)g.G[b] * g.Q;
        w->Mm[b] = take(M * g.D2);
        w->P[b] = take(GQ * g.Q);
        w->AX[b] = take(GQ * g.D2);
        w->z5[b] = take(GQ * g.HD);
        w->dAX[b] = take(GQ * g.D2);
        w->dz5[b] = take(GQ * g.HD);
        w->gX[b] = take(M * g.D2);
        w->gM[b] = take(M * g.D2);
        w->dMb[b] = take(GQ * g.D2);     // per-graph d mapping blocks (their own buffer: AX[b] is still being read by the theta gradient)
    }
    w->feat = take(B * g.FIN);
    w->h1 = take(B * g.D2);
    w->h2 = take(B * g.D2);
    w->h3 = take(B * g.HD);
    w->dpred = take(B);
    w->sqerr = take(B);
    w->dh3 = take(B * g.HD);
    w->dh2 = take(B * g.D2);
    w->dh1 = take(B * g.D2);
    w->dfeat = take(B * g.FIN);
    w->dF = take(M * g.D2);
    w->da2 = take(M * g.CL);
    w->dy1 = take(M * g.H1 * g.L1);
    w->rows = 512;
    w->gp1 = take((size_t)w->rows * g.H1 * g.K);
    w->gp2 = take((size_t)w->rows * g.CO * g.H1 * g.K);
    // split-K scratch: max over the weight-gradient GEMMs of slices * M * N
    size_t mx = 1;
    auto need = [&](int Mo, int No, int64_t K) {
        const size_t v = sgemm_splitk_need_floats(Mo, No, (int)K);
        if (v > mx) mx = v;
    };
    need(1, g.HD, g.B); need(g.HD, g.D2, g.B); need(g.D2, g.D2, g.B); need(g.D2, g.FIN, g.B); need(1, g.D2, g.B);
    for (int b = 0; b < 2; ++b) { need(g.HD, g.D2, g.G[b] * g.Q); need(g.HD, g.D2 + 1, g.G[b] * g.Q); need(1, g.HD, g.G[b] * g.Q); }
    need(g.D2, g.D2, g.M); need(g.D2, g.D2 + 1, g.M); need(1, g.D2, g.M); need(g.D2, g.CL, g.M); need(g.D2, g.CL + 1, g.M);
    need((int)g.B, g.D2, g.FIN);                        // fc1 forward
    {   // the five batched weight-gradient products of the backward keep their partial rows side by side
        SplitKColsumJob j[5] = {};
        for (int b = 0; b < 2; ++b) { j[b].M = g.HD; j[b].N = g.D2; j[b].K = (int)(g.G[b] * g.Q); }
        for (int b = 0; b < 2; ++b) { j[2 + b].M = g.D2; j[2 + b].N = g.D2; j[2 + b].K = (int)g.M; }
        j[4].M = g.D2; j[4].N = g.CL; j[4].K = (int)g.M;
        const size_t v = sgemm_splitk_colsum_batch_floats(j, 5);
        if (v > mx) mx = v;
    }
    w->split_floats = mx;
    w->split = take(mx);
    w->total = o;
}

// streaming kernels: at most this many workgroups (each ends in one fp64 atomic per channel onto the statistics cells; 4096 workgroups
// spent longer in those than 1024 spend looping: 0.755 -> 0.736 ms per step at batch 256)
constexpr int FC_GRID_CAP = 1024;
inline unsigned grid_for(int64_t total) {
    int64_t b = (total + FB - 1) / FB;
    if (b > FC_GRID_CAP) b = FC_GRID_CAP;              // (512: 0.368 ms, 2048: 0.359, 256: 0.49 against 0.336 at FD004 batch 256)
    return (unsigned)(b < 1 ? 1 : b);
}

}  // namespace

int64_t fcstgnn_param_count(const rulgnn_fcstgnn_shape* s) {
    FcGeom g;
    return fc_geometry(s, &g) == RULGNN_OK ? g.nparam : -1;
}

int64_t fcstgnn_bn_count(const rulgnn_fcstgnn_shape* s) {
    FcGeom g;
    return fc_geometry(s, &g) == RULGNN_OK ? g.bn_total : -1;
}

size_t fcstgnn_workspace_bytes(const rulgnn_fcstgnn_shape* s) {
    FcGeom g;
    if (fc_geometry(s, &g) != RULGNN_OK) return 0;
    FcWs w;
    fc_ws_layout(g, &w);
    return w.total;
}

#define FC_RC(call)                        \
    do {                                   \
        const int rc_ = (call);            \
        if (rc_ != RULGNN_OK) return rc_;  \
    } while (0)

// mode bit 0: forward (args->training selects batch / running statistics), bit 1: backward
int fcstgnn_run(const rulgnn_fcstgnn_shape* s, const rulgnn_fcstgnn_args* a, int mode, hipStream_t st, const FcstgnnSync* sync, float* bn_running_out,
                float bn_momentum, const AdamFuse* fuse) {
    FcGeom g;
    FC_RC(fc_geometry(s, &g));
    if (sync) {
        // every BatchNorm normalises by the statistics of the GLOBAL batch: the cells are all-reduced between the kernel that
        // completes a pair and its first reader, and the element counts are those of the global batch
        if (mode != 3 || !a->training || a->global_batch < g.B || g.B < 1 || a->bn_moment_weight > 0.f) return RULGNN_EINVAL;
        const double scale = (double)a->global_batch / (double)g.B;
        for (int i = 0; i < NBN; ++i) g.cnt[i] *= scale;
    }
    FcWs w;
    fc_ws_layout(g, &w);
    if (a->workspace_bytes < w.total) return RULGNN_EWORKSPACE;
    char* ws = static_cast<char*>(a->workspace);
    auto P_ = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    Cells* cells = reinterpret_cast<Cells*>(ws + w.cells);
    const float* prm = a->params;
    const float* run = a->bn_stats;
    const int training = a->training ? 1 : 0;
    const int bf = a->compute_dtype == RULGNN_DTYPE_BF16 ? 1 : 0;     // bf16 operands on the row projections that run as GEMM launches
    const int D2 = g.D2, HD = g.HD, CL = g.CL, FIN = g.FIN;
    const int Mi = (int)g.M, Bi = (int)g.B;
    const float inv_gb = 1.0f / (float)(a->global_batch > 0 ? a->global_batch : g.B);
    // the MLP behind fc1 in one launch (fc_mlp_tail_kernel) for the fp32 path at the widths it is instantiated for
    // the window graphs one wavefront each on the fp32 matrix cores where a graph fits the 32 x 32 tile
    const bool graph_mx = g.Q <= 32 && g.Q % 4 == 0 && (g.D2 == 16 || g.D2 == 32);
    // activation + projection + statistics of the encoder's last Linear in one launch (fc_proj3_kernel) where a row's tile fits a workgroup
    const size_t proj_lds = sizeof(float) * ((size_t)g.D2 * (g.CL + 1) + 2 * FB);
    const bool proj_fused = g.CL <= FB && g.D2 <= FB && proj_lds <= 40 * 1024;
    const bool mlp_fused = g.D2 == 2 * g.HD && (g.D2 == 16 || g.D2 == 32 || g.D2 == 64);
    auto mlp_tail = [&](int tail_mode, const float* y, const float* dpred_in) {
        auto go = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3((unsigned)((g.B + 63) / 64)), dim3(64), 0, st, g, prm, P_(w.h1), P_(w.h2), P_(w.h3), y, dpred_in, a->pred,
                               P_(w.dpred), P_(w.sqerr), P_(w.dh3), P_(w.dh2), P_(w.dh1), inv_gb, tail_mode);
        };
        if (g.D2 == 16) go(fc_mlp_tail_kernel<16>);
        else if (g.D2 == 32) go(fc_mlp_tail_kernel<32>);
        else go(fc_mlp_tail_kernel<64>);
    };
    // positional-encoding dropout (train mode only)
    const float p = training ? a->dropout_p : 0.f;
    uint32_t thr = 0;
    if (p > 0.f) {
        const double t = (double)p * 4294967296.0;
        const uint64_t ti = (uint64_t)(t + 0.5);
        thr = ti > 4294967295ull ? 4294967295u : (uint32_t)ti;
    }
    const float dscale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    const uint32_t key = dropout_layer_key(a->seed, a->step, 0);
    const uint32_t* key_dev = a->step_state ? static_cast<const StepState*>(a->step_state)->drop_key : nullptr;
    const int64_t row_off = a->sample_offset * g.NP * g.N;
    (void)hipGetLastError();
    auto sync_pair = [&](int bwd, int id) -> int {
        if (!sync) return RULGNN_OK;
        hipLaunchKernelGGL(fc_cells_collapse_kernel, dim3(1), dim3(128), 0, st, cells, bwd, id);
        if (hipGetLastError() != hipSuccess) return RULGNN_EHIP;
        double* buf = bwd ? &cells[0].bwd[id][0][0] : &cells[0].fwd[id][0][0];
        return sync->fn(sync->user, buf, 2 * MAXC, st) == 0 ? RULGNN_OK : RULGNN_ECALLBACK;
    };

    if (mode & 1) {
        if (training && a->step_state) FC_RC(step_prepare_dropout(a->step_state, a->seed, 1, st));
        if (hipMemsetAsync(cells, 0, sizeof(Cells) * CELL_REP, st) != hipSuccess) return RULGNN_EHIP;
        // (the reference's C-MAPSS wiring has its own instantiations: shapes as compile-time constants)
        const bool fd004_conv = g.K == 2 && g.L1 == 3 && g.L2 == 4 && g.H1 == 8 && g.CO == 6;
        const bool fd004_in = fd004_conv && g.N == 14 && g.NP == 25 && g.PS == 2 && g.TL == 50;
        if (fd004_in) hipLaunchKernelGGL(fc_conv1_kernel<true>, dim3(grid_for(g.M * g.H1 * g.L1)), dim3(FB), 0, st, g, a->x, prm, P_(w.z1), cells, training);
        else hipLaunchKernelGGL(fc_conv1_kernel<false>, dim3(grid_for(g.M * g.H1 * g.L1)), dim3(FB), 0, st, g, a->x, prm, P_(w.z1), cells, training);
        FC_RC(sync_pair(0, 0));
        if (fd004_conv)
            hipLaunchKernelGGL((fc_conv2_kernel<2, 3, 4, 8, 6>), dim3(grid_for(g.M * CL)), dim3(FB), 0, st, g, prm, run, (const float*)P_(w.z1), P_(w.z2),
                               cells, training);
        else
            hipLaunchKernelGGL((fc_conv2_kernel<0, 0, 0, 0, 0>), dim3(grid_for(g.M * CL)), dim3(FB), 0, st, g, prm, run, (const float*)P_(w.z1), P_(w.z2),
                               cells, training);
        FC_RC(sync_pair(0, 1));
        if (proj_fused) {
            hipLaunchKernelGGL(fc_proj3_kernel, dim3(grid_for(g.M * CL)), dim3(FB), proj_lds, st, g, prm, run, cells, training,
                               (const float*)P_(w.z2), P_(w.a2), P_(w.z3));
        } else {
            hipLaunchKernelGGL(fc_act2_kernel, dim3(grid_for(g.M * CL)), dim3(FB), 0, st, g, prm, run, (const Cells*)cells, training,
                               (const float*)P_(w.z2), P_(w.a2));
            FC_RC(sgemm(P_(w.a2), CL, 1, prm + g.o_W3, CL, 1, P_(w.z3), D2, Mi, D2, CL, false, st, bf));
            hipLaunchKernelGGL(fc_bias_stats_kernel, dim3(grid_for(g.M * D2)), dim3(FB), 0, st, P_(w.z3), prm + g.o_b3, g.M, D2, cells, 2, training);
        }
        FC_RC(sync_pair(0, 2));
        hipLaunchKernelGGL(fc_pe_kernel, dim3(grid_for(g.M * D2)), dim3(FB), 0, st, g, prm, run, cells, training, (const float*)P_(w.z3),
                           P_(w.F), thr, dscale, key, key_dev, row_off);
        FC_RC(sync_pair(0, 3));
        FC_RC(sync_pair(0, 5));
        const int64_t Gmax = g.G[0] > g.G[1] ? g.G[0] : g.G[1];
        const bool block_fused = graph_mx && D2 == 2 * HD;
        if (block_fused) {
            // mapping, window graphs, the block's Linear and its BatchNorm statistics: one launch for both window blocks
            const unsigned wgs = (unsigned)((Gmax + FC_MX_WAVES - 1) / FC_MX_WAVES);
            auto go = [&](auto kernel) {
                // (one workgroup per CU and window block: 2 x 256 = the resident count, each wavefront walks ~4.5 graphs.  With a workgroup per four
                // graphs -- 2304 workgroups, 4.5 rounds -- every wavefront paid the prologue (BatchNorm table, weights, a barrier) for ONE graph:
                // FD004 batch 256 0.352 -> 0.336 ms with both graph kernels capped; 128: 0.373, 192: 0.343, 384: 0.343, 512: 0.340)
                constexpr unsigned capf = 256u;
                hipLaunchKernelGGL(kernel, dim3(wgs < capf ? wgs : capf, 2), dim3(64 * FC_MX_WAVES), 0, st, g, prm, run, cells, training,
                                   (const float*)P_(w.F), Ptr2{{P_(w.Mm[0]), P_(w.Mm[1])}}, Ptr2{{P_(w.P[0]), P_(w.P[1])}},
                                   Ptr2{{P_(w.AX[0]), P_(w.AX[1])}}, Ptr2{{P_(w.z5[0]), P_(w.z5[1])}});
            };
            // (compute_dtype = bf16: the four products of a window graph on bf16 matrix instructions, fc_prod)
            if (D2 == 16) { if (bf) go(fc_block_mx_kernel<16, true>); else go(fc_block_mx_kernel<16, false>); }
            else { if (bf) go(fc_block_mx_kernel<32, true>); else go(fc_block_mx_kernel<32, false>); }
            FC_RC(sync_pair(0, 4));
            FC_RC(sync_pair(0, 6));
        }
        for (int b = 0; b < 2 && !block_fused; ++b) {
            const int GQ = (int)(g.G[b] * g.Q);
            const unsigned wgs = (unsigned)((g.G[b] + FC_MX_WAVES - 1) / FC_MX_WAVES);
            {
                FC_RC(sgemm(P_(w.F), D2, 1, prm + g.o_map[b], D2, 1, P_(w.Mm[b]), D2, Mi, D2, D2, false, st, bf));
                if (graph_mx) {
                    auto go = [&](auto kernel) {
                        hipLaunchKernelGGL(kernel, dim3(wgs < 4096 ? wgs : 4096), dim3(64 * FC_MX_WAVES), 0, st, g, b, prm, run, (const Cells*)cells, training,
                                           (const float*)P_(w.F), (const float*)P_(w.Mm[b]), P_(w.P[b]), P_(w.AX[b]));
                    };
                    if (D2 == 16) go(fc_graph_mx_kernel<16>);
                    else go(fc_graph_mx_kernel<32>);
                } else {
                    hipLaunchKernelGGL(fc_graph_kernel, dim3((unsigned)(g.G[b] < 8192 ? g.G[b] : 8192)), dim3(FC_GRAPH_FWD_THREADS),
                                       sizeof(float) * (2 * g.Q * (g.D2 + 1) + g.Q * (g.Q + 1)), st, g, b, prm, run, (const Cells*)cells, training,
                                       (const float*)P_(w.F), (const float*)P_(w.Mm[b]), P_(w.P[b]), P_(w.AX[b]));
                }
                FC_RC(sgemm(P_(w.AX[b]), D2, 1, prm + g.o_th[b], D2, 1, P_(w.z5[b]), HD, GQ, HD, D2, false, st, bf));
                hipLaunchKernelGGL(fc_bias_stats_kernel, dim3(grid_for((int64_t)GQ * HD)), dim3(FB), 0, st, P_(w.z5[b]), prm + g.o_thb[b],
                                   (int64_t)GQ, HD, cells, 4 + 2 * b, training);
            }
            FC_RC(sync_pair(0, 4 + 2 * b));
        }
        hipLaunchKernelGGL(fc_pool_kernel, dim3(grid_for(Gmax * g.N * HD), 2), dim3(FB), 0, st, g, prm, run, (const Cells*)cells, training,
                           CPtr2{{P_(w.z5[0]), P_(w.z5[1])}}, P_(w.feat));
        // K = FIN (4032 at FD004) against a [batch x 16] output: split the reduction, or four workgroups walk it alone (250 us)
        FC_RC(sgemm_splitk(P_(w.feat), FIN, 1, prm + g.o_f1w, FIN, 1, P_(w.h1), D2, Bi, D2, FIN, false, P_(w.split), st));
        if (mlp_fused) {
            mlp_tail(1, a->y, nullptr);
        } else {
            hipLaunchKernelGGL(fc_bias_relu_kernel, dim3(grid_for(g.B * D2)), dim3(FB), 0, st, P_(w.h1), prm + g.o_f1b, g.B, D2);
            FC_RC(sgemm(P_(w.h1), D2, 1, prm + g.o_f2w, D2, 1, P_(w.h2), D2, Bi, D2, D2, false, st, bf));
            hipLaunchKernelGGL(fc_bias_relu_kernel, dim3(grid_for(g.B * D2)), dim3(FB), 0, st, P_(w.h2), prm + g.o_f2b, g.B, D2);
            FC_RC(sgemm(P_(w.h2), D2, 1, prm + g.o_f3w, D2, 1, P_(w.h3), HD, Bi, HD, D2, false, st, bf));
            hipLaunchKernelGGL(fc_bias_relu_kernel, dim3(grid_for(g.B * HD)), dim3(FB), 0, st, P_(w.h3), prm + g.o_f3b, g.B, HD);
            hipLaunchKernelGGL(fc_head_kernel, dim3((unsigned)((g.B + FB - 1) / FB)), dim3(FB), 0, st, g, prm, (const float*)P_(w.h3), a->y,
                               (const float*)nullptr, a->pred, P_(w.dpred), P_(w.sqerr), P_(w.dh3), inv_gb, 0);
        }
        if (training && a->bn_batch && !(mode & 2))            // (with a backward in the same call: beside its chain, below)
            hipLaunchKernelGGL(fc_bn_batch_kernel, dim3(1), dim3(64), 0, st, g, (const Cells*)cells, a->bn_batch, a->bn_moment_weight);
    }
    if (mode & 2) {
        float* gr = a->grads;
        float* split = P_(w.split);
        float* one = P_(w.one);
        SplitKColsumJob wjobs[5];
        int nwj = 0;
        // Weight / bias gradient GEMMs: nothing in this call reads their results, and each is a ~6-19 us launch at its latency floor.
        // With a second stream of the caller (args->aux_stream, aux_stream.hpp) they leave the critical path; they share the split-K
        // scratch and therefore one stream.
        AuxFork fk(st, a->aux_stream);
        hipStream_t wst = fk.side();
        auto fork = [&]() { fk.fork(); };
        // ... and so do the other launches nothing on the chain waits for: the constant for the column sums, the batch moments for the
        // bucket, the loss sum
        // (one fork for everything that is ready when the backward starts: with an incoming gradient the MLP's d h first)
        if (mlp_fused && a->dpred) mlp_tail(2, nullptr, a->dpred);
        fork();
        // (the host enqueues in program order: the MAIN stream's next kernel goes out before the dozen side-stream launches behind this fork
        // -- with the side launches first the main queue sat empty for ~30 us while the host enqueued them; same at every fork below)
        if (mlp_fused) FC_RC(sgemm(P_(w.dh1), D2, 1, prm + g.o_f1w, 1, FIN, P_(w.dfeat), FIN, Bi, FIN, D2, false, st, bf));
        if ((mode & 1) && training && a->bn_batch)
            hipLaunchKernelGGL(fc_side_head_kernel, dim3(1), dim3(64), 0, wst, g, (const Cells*)cells, one, a->bn_batch, a->bn_moment_weight,
                               (bn_running_out && a->bn_moment_weight == 0.f) ? bn_running_out : (float*)nullptr, bn_momentum);
        else
            hipLaunchKernelGGL(fc_fill_one_kernel, dim3(1), dim3(1), 0, wst, one);
        // (batches of the reference protocol's size: every MLP parameter gradient and the loss sum in one launch, fc_mlp_wgrad_kernel)
        const bool mlp_wgrad_fused = mlp_fused && g.B <= FC_MLPW_MAXB && D2 <= 64;
        if (!a->dpred && a->loss && !mlp_wgrad_fused) (void)block_sum((const float*)P_(w.sqerr), (int64_t)g.B, a->loss, wst);
        auto colsum = [&](const float* src, int64_t rows, int C, float* dst) {      // dst[c] = sum_r src[r][c]
            if (cols_sum_small_ok(rows, C)) return cols_sum_small(src, (int)rows, C, dst, wst);
            return sgemm_splitk(one, 0, 0, src, 1, C, dst, C, 1, C, (int)rows, false, split, wst);
        };
        // ---- MLP ----
        if (mlp_wgrad_fused) {
            FcMlpW k;
            k.dpred = P_(w.dpred); k.dh3 = P_(w.dh3); k.dh2 = P_(w.dh2); k.dh1 = P_(w.dh1);
            k.h3 = P_(w.h3); k.h2 = P_(w.h2); k.h1 = P_(w.h1); k.feat = P_(w.feat); k.sqerr = P_(w.sqerr);
            k.f1w = gr + g.o_f1w; k.f1b = gr + g.o_f1b; k.f2w = gr + g.o_f2w; k.f2b = gr + g.o_f2b;
            k.f3w = gr + g.o_f3w; k.f3b = gr + g.o_f3b; k.f4w = gr + g.o_f4w; k.f4b = gr + g.o_f4b;
            k.loss = (!a->dpred && a->loss) ? a->loss : nullptr;
            // (d fc1.weight [2h x FIN] stays on the split-K matrix-core pair: as 64-column workgroups walking the batch in order it took
            // ~90 us at batch 256 -- a chain of 256 dependent-latency iterations -- and the side stream is nearly as long as the main one)
            FC_RC(sgemm_splitk(P_(w.dh1), 1, D2, P_(w.feat), 1, FIN, gr + g.o_f1w, FIN, D2, FIN, Bi, false, split, wst));
            k.B = Bi; k.D2 = D2; k.HD = HD; k.FIN = FIN; k.nfw = 0;
            const int small = D2 * D2 + HD * D2 + HD + D2 + D2 + HD + 2;
            hipLaunchKernelGGL(fc_mlp_wgrad_kernel, dim3((unsigned)(k.nfw + (small + 255) / 256)), dim3(256), 0, wst, k);
        } else if (mlp_fused) {
            // d h3, d h2, d h1 are there (formed with the loss in the forward's fused MLP kernel, or here from the incoming gradient):
            // one fork, every parameter gradient of the MLP on the side
            FC_RC(sgemm_splitk(P_(w.dpred), 0, 1, P_(w.h3), 1, HD, gr + g.o_f4w, HD, 1, HD, Bi, false, split, wst));
            FC_RC(colsum(P_(w.dpred), g.B, 1, gr + g.o_f4b));
            FC_RC(sgemm_splitk(P_(w.dh3), 1, HD, P_(w.h2), 1, D2, gr + g.o_f3w, D2, HD, D2, Bi, false, split, wst));
            FC_RC(colsum(P_(w.dh3), g.B, HD, gr + g.o_f3b));
            FC_RC(sgemm_splitk(P_(w.dh2), 1, D2, P_(w.h1), 1, D2, gr + g.o_f2w, D2, D2, D2, Bi, false, split, wst));
            FC_RC(colsum(P_(w.dh2), g.B, D2, gr + g.o_f2b));
            FC_RC(sgemm_splitk(P_(w.dh1), 1, D2, P_(w.feat), 1, FIN, gr + g.o_f1w, FIN, D2, FIN, Bi, false, split, wst));
            FC_RC(colsum(P_(w.dh1), g.B, D2, gr + g.o_f1b));
        } else {
        fork();
        if (a->dpred)
            hipLaunchKernelGGL(fc_head_kernel, dim3((unsigned)((g.B + FB - 1) / FB)), dim3(FB), 0, st, g, prm, (const float*)P_(w.h3),
                               (const float*)nullptr, a->dpred, a->pred, P_(w.dpred), P_(w.sqerr), P_(w.dh3), inv_gb, 1);
        FC_RC(sgemm_splitk(P_(w.dpred), 0, 1, P_(w.h3), 1, HD, gr + g.o_f4w, HD, 1, HD, Bi, false, split, wst));
        FC_RC(sgemm_splitk(P_(w.dpred), 0, 1, one, 0, 0, gr + g.o_f4b, 1, 1, 1, Bi, false, split, wst));
        FC_RC(sgemm_splitk(P_(w.dh3), 1, HD, P_(w.h2), 1, D2, gr + g.o_f3w, D2, HD, D2, Bi, false, split, wst));
        FC_RC(colsum(P_(w.dh3), g.B, HD, gr + g.o_f3b));
        FC_RC(sgemm(P_(w.dh3), HD, 1, prm + g.o_f3w, 1, D2, P_(w.dh2), D2, Bi, D2, HD, false, st, bf));
        hipLaunchKernelGGL(fc_relu_mask_kernel, dim3(grid_for(g.B * D2)), dim3(FB), 0, st, P_(w.dh2), (const float*)P_(w.h2), g.B * D2);
        fork();
        FC_RC(sgemm_splitk(P_(w.dh2), 1, D2, P_(w.h1), 1, D2, gr + g.o_f2w, D2, D2, D2, Bi, false, split, wst));
        FC_RC(colsum(P_(w.dh2), g.B, D2, gr + g.o_f2b));
        FC_RC(sgemm(P_(w.dh2), D2, 1, prm + g.o_f2w, 1, D2, P_(w.dh1), D2, Bi, D2, D2, false, st, bf));
        hipLaunchKernelGGL(fc_relu_mask_kernel, dim3(grid_for(g.B * D2)), dim3(FB), 0, st, P_(w.dh1), (const float*)P_(w.h1), g.B * D2);
        fork();
        FC_RC(sgemm_splitk(P_(w.dh1), 1, D2, P_(w.feat), 1, FIN, gr + g.o_f1w, FIN, D2, FIN, Bi, false, split, wst));
        FC_RC(colsum(P_(w.dh1), g.B, D2, gr + g.o_f1b));
        }
        if (!mlp_fused) FC_RC(sgemm(P_(w.dh1), D2, 1, prm + g.o_f1w, 1, FIN, P_(w.dfeat), FIN, Bi, FIN, D2, false, st, bf));
        // ---- graph blocks ----
        // (both window blocks in each launch: blockIdx.y)
        {
            const int64_t Gmax = g.G[0] > g.G[1] ? g.G[0] : g.G[1];
            const int64_t GQ0 = g.G[0] * g.Q, GQ1 = g.G[1] * g.Q;
            const CPtr2 z5p{{P_(w.z5[0]), P_(w.z5[1])}};
            const Ptr2 dz5p{{P_(w.dz5[0]), P_(w.dz5[1])}};
            hipLaunchKernelGGL(fc_pool_bwd_kernel, dim3(grid_for(Gmax * g.Q * HD), 2), dim3(FB), 0, st, g, prm, cells, z5p,
                               (const float*)P_(w.dfeat), dz5p);
            FC_RC(sync_pair(1, 4));
            FC_RC(sync_pair(1, 6));
            hipLaunchKernelGGL(fc_bn_rows_bwd_kernel, dim3(grid_for(Gmax * g.Q * HD), 2), dim3(FB), 0, st, g, 4, prm, (const Cells*)cells, z5p, dz5p,
                               GQ0, GQ1);
            const bool bwd_fused = graph_mx && D2 == 2 * HD;          // d AX = d z5 W_theta inside the graph kernel
            if (!bwd_fused)
                for (int b = 0; b < 2; ++b)
                    FC_RC(sgemm(P_(w.dz5[b]), HD, 1, prm + g.o_th[b], 1, D2, P_(w.dAX[b]), D2, (int)(g.G[b] * g.Q), D2, HD, false, st, bf));
            // (the per-graph d mapping blocks have their own buffer: the theta gradient, possibly on the other stream, still reads AX[b])
            if (graph_mx) {
                const unsigned wgs = (unsigned)((Gmax + FC_MX_WAVES - 1) / FC_MX_WAVES);
                auto go = [&](auto kernel) {
                    constexpr unsigned capb = 256u;                      // (as the forward block kernel: one workgroup per CU and window block)
                    hipLaunchKernelGGL(kernel, dim3(wgs < capb ? wgs : capb, 2), dim3(64 * FC_MX_WAVES), 0, st, g, prm, (const Cells*)cells,
                                       (const float*)P_(w.F), CPtr2{{P_(w.Mm[0]), P_(w.Mm[1])}}, CPtr2{{P_(w.P[0]), P_(w.P[1])}},
                                       CPtr2{{P_(w.dz5[0]), P_(w.dz5[1])}}, Ptr2{{P_(w.dAX[0]), P_(w.dAX[1])}},
                                       Ptr2{{P_(w.dMb[0]), P_(w.dMb[1])}});
                };
                if (bwd_fused) {
                    if (D2 == 16) { if (bf) go(fc_graph_bwd_mx_kernel<16, true, true>); else go(fc_graph_bwd_mx_kernel<16, true, false>); }
                    else { if (bf) go(fc_graph_bwd_mx_kernel<32, true, true>); else go(fc_graph_bwd_mx_kernel<32, true, false>); }
                } else {
                    if (D2 == 16) { if (bf) go(fc_graph_bwd_mx_kernel<16, false, true>); else go(fc_graph_bwd_mx_kernel<16, false, false>); }
                    else { if (bf) go(fc_graph_bwd_mx_kernel<32, false, true>); else go(fc_graph_bwd_mx_kernel<32, false, false>); }
                }
            } else {
            for (int b = 0; b < 2; ++b) {
                const size_t lds_b = sizeof(float) * (3 * g.Q * (g.D2 + 1) + 3 * g.Q * (g.Q + 1));
                if (lds_b > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(fc_graph_bwd_kernel),
                                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b) != hipSuccess)
                    return RULGNN_EHIP;
            hipLaunchKernelGGL(fc_graph_bwd_kernel, dim3((unsigned)(g.G[b] < 8192 ? g.G[b] : 8192)), dim3(FC_GRAPH_BWD_THREADS), sizeof(float) * (3 * g.Q * (g.D2 + 1) + 3 * g.Q * (g.Q + 1)), st, g, b, prm,
                               (const Cells*)cells, (const float*)P_(w.F), (const float*)P_(w.Mm[b]), (const float*)P_(w.P[b]),
                               P_(w.dAX[b]), P_(w.dMb[b]));
            }
            }
            hipLaunchKernelGGL(fc_graph_gather_kernel, dim3(grid_for(g.M * D2), 2), dim3(FB), 0, st, g, CPtr2{{P_(w.dAX[0]), P_(w.dAX[1])}},
                               CPtr2{{P_(w.dMb[0]), P_(w.dMb[1])}}, Ptr2{{P_(w.gX[0]), P_(w.gX[1])}}, Ptr2{{P_(w.gM[0]), P_(w.gM[1])}});
            // (weight gradient and the bias gradient over the same rows: one split-K pass with the bias sums as an extra column.  The five
            // such products of the backward -- theta and mapping of both window blocks, the projection -- are collected and go out as ONE
            // pair of launches behind the last of their inputs (sgemm_splitk_colsum_batch): they were ten launches, the tail of the side stream)
            for (int b = 0; b < 2; ++b)
                wjobs[nwj++] = SplitKColsumJob{P_(w.dz5[b]), 1, HD, P_(w.AX[b]), 1, D2, gr + g.o_th[b], D2, HD, D2, (int)(g.G[b] * g.Q), gr + g.o_thb[b]};
        }
        hipLaunchKernelGGL(fc_feat_stats_kernel, dim3(grid_for(g.M * D2)), dim3(FB), 0, st, g, prm, cells, (const float*)P_(w.F),
                           (const float*)P_(w.gX[0]), (const float*)P_(w.gX[1]));
        FC_RC(sync_pair(1, 3));
        FC_RC(sync_pair(1, 5));
        // window BatchNorms' backward + both d M W_map products + positional encoding / dropout backward: one launch
        hipLaunchKernelGGL(fc_feat_pe_bwd_kernel, dim3(grid_for(g.M * D2)), dim3(FB), 0, st, g, prm, cells, (const float*)P_(w.F),
                           (const float*)P_(w.gX[0]), (const float*)P_(w.gX[1]), (const float*)P_(w.gM[0]), (const float*)P_(w.gM[1]),
                           (const float*)P_(w.z3), P_(w.dF), thr, dscale, key, key_dev, row_off);
        for (int b = 0; b < 2; ++b)
            wjobs[nwj++] = SplitKColsumJob{P_(w.gM[b]), 1, D2, P_(w.F), 1, D2, gr + g.o_map[b], D2, D2, D2, Mi, gr + g.o_bmap[b]};
        FC_RC(sync_pair(1, 2));
        hipLaunchKernelGGL(fc_bn_rows_bwd_kernel, dim3(grid_for(g.M * D2)), dim3(FB), 0, st, g, 2, prm, (const Cells*)cells,
                           CPtr2{{P_(w.z3), nullptr}}, Ptr2{{P_(w.dF), nullptr}}, g.M, (int64_t)0);
        fork();
        // ---- encoder convolutions ----
        if (proj_fused) {
            hipLaunchKernelGGL(fc_proj3_bwd_kernel, dim3(grid_for(g.M * CL)), dim3(FB), proj_lds, st, g, prm, cells, (const float*)P_(w.z2),
                               (const float*)P_(w.a2), (const float*)P_(w.dF), P_(w.da2));
        } else {
            FC_RC(sgemm(P_(w.dF), D2, 1, prm + g.o_W3, 1, CL, P_(w.da2), CL, Mi, CL, D2, false, st, bf));
            hipLaunchKernelGGL(fc_act2_bwd_kernel, dim3(grid_for(g.M * CL)), dim3(FB), 0, st, g, prm, cells, (const float*)P_(w.z2),
                               (const float*)P_(w.a2), P_(w.da2));
        }
        wjobs[nwj++] = SplitKColsumJob{P_(w.dF), 1, D2, P_(w.a2), 1, CL, gr + g.o_W3, CL, D2, CL, Mi, gr + g.o_b3};
        FC_RC(sgemm_splitk_colsum_batch(wjobs, nwj, one, split, w.split_floats, wst));
        FC_RC(sync_pair(1, 1));
        // (BatchNorm 1's and BatchNorm 0's channel-major backward passes ride in the loads of the kernels that consume them)
        // the second convolution's weight gradient needs d z2 (final here) and the forward statistics only: beside the rest of the chain
        const int rows = (int)(g.M < w.rows ? g.M : w.rows);
        const bool fd004_conv = g.K == 2 && g.L1 == 3 && g.L2 == 4 && g.H1 == 8 && g.CO == 6;     // the reference's C-MAPSS wiring: constants
        if (g.K == 2 && g.L1 == 3 && g.L2 == 4 && g.H1 == 8 && g.CO == 6)
            hipLaunchKernelGGL((fc_conv2_dx_kernel<2, 3, 4, 8, 6>), dim3(grid_for(g.M * g.H1 * g.L1)), dim3(FB), 0, st, g, prm, cells,
                               (const float*)P_(w.z1), (const float*)P_(w.da2), P_(w.dy1), (const float*)P_(w.z2));
        else
            hipLaunchKernelGGL((fc_conv2_dx_kernel<0, 0, 0, 0, 0>), dim3(grid_for(g.M * g.H1 * g.L1)), dim3(FB), 0, st, g, prm, cells,
                               (const float*)P_(w.z1), (const float*)P_(w.da2), P_(w.dy1), (const float*)P_(w.z2));
        FC_RC(sync_pair(1, 0));
        // both convolutions' weight gradients (the second one's needs d z2 only, but the side stream is the longer one by then)
        if (fd004_conv)
            hipLaunchKernelGGL((fc_conv_wgrad_both_kernel<2, 3, 4, 8, 6>), dim3(rows, 2), dim3(FB), 0, st, g, a->x, prm, (const Cells*)cells,
                               (const float*)P_(w.z1), (const float*)P_(w.z2), (const float*)P_(w.dy1), (const float*)P_(w.da2), P_(w.gp1), P_(w.gp2));
        else
            hipLaunchKernelGGL((fc_conv_wgrad_both_kernel<>), dim3(rows, 2), dim3(FB), 0, st, g, a->x, prm, (const Cells*)cells,
                               (const float*)P_(w.z1), (const float*)P_(w.z2), (const float*)P_(w.dy1), (const float*)P_(w.da2), P_(w.gp1), P_(w.gp2));
        int nbn = 0;
        for (int i = 0; i < NBN; ++i) nbn += g.bn_ch[i];
        // (reads the convolutions' partial rows and the cells only: in front of the join, beside whatever the side stream still runs --
        // unless it also applies the optimizer: then every gradient must be final, and it is the step's last launch behind the join)
        const int nfin = (g.H1 * g.K + g.CO * g.H1 * g.K + nbn + 3) / 4;
        const bool with_adam = fuse && fuse->p && !sync;
        AdamFuse none{};
        none.p = nullptr;
        if (with_adam) FC_RC(fk.join());
        hipLaunchKernelGGL(fc_finalize_kernel, dim3(nfin + (with_adam ? (g.nparam + FB - 1) / FB : 0)), dim3(FB), 0, st, g,
                           (const float*)P_(w.gp1), (const float*)P_(w.gp2), rows, (const Cells*)cells, gr, sync ? sync->bn_param_grad_scale : 1.0f,
                           with_adam ? *fuse : none, nfin);
        FC_RC(fk.join());                                    // the gradient GEMMs are done when the call's work has drained
    }
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

int fcstgnn_bn_running_update(const rulgnn_fcstgnn_shape* s, float* bn_stats, const float* bn_batch, float momentum, int from_moments,
                              hipStream_t st) {
    FcGeom g;
    FC_RC(fc_geometry(s, &g));
    (void)hipGetLastError();
    hipLaunchKernelGGL(fc_bn_running_kernel, dim3(1), dim3(64), 0, st, g, bn_stats, bn_batch, momentum, from_moments);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

}  // namespace rulgnn
