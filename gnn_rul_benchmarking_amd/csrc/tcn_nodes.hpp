// The node-channel temporal convolution block shared by ASTGCNN and ST_Conv (both instantiate the reference's
// TemporalConvNet(num_nodes, [num_nodes, num_nodes], kernel_size=6): models/ASTGCNN/Model.py:72-146, models/ST_Conv/Model.py:81-155):
// two causal Conv1d(N -> N, k = 6, dilation 1 | 2, no bias) + BatchNorm1d(N) + ReLU blocks with residuals over [N nodes] x [T steps].
// One sample per workgroup iteration, tile in LDS, BatchNorm sums through fp64 cells.  `Geom` provides
// B, BG (the samples behind the BatchNorm statistics: B, or the global batch under synchronised BatchNorm), N, T and the parameter
// offsets o_w1, o_g1, o_b1, o_w2, o_g2, o_b2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rulgnn {
namespace tcn {

constexpr int AB = 256;            // threads per workgroup
constexpr int KT = 6;              // TCN kernel size
constexpr int MAXN = 25;           // nodes
constexpr int MAXT = 64;
constexpr int XP = MAXT + 16;        // row pitch of the convolution tiles: 16-byte row reads, room for the taps' reach on both sides
constexpr float BN_EPS = 1e-5f;
#ifndef TCN_UNROLL_N
#define TCN_UNROLL_N 4
#endif
// the (ci | co | four-step) loops: with the shapes as constants the compiler unrolled them fully -- > 256 registers, one workgroup per CU
constexpr int TCN_UNROLL = TCN_UNROLL_N;

// reduction cells (fp64): forward sums [2 blocks][N][sum, sumsq], backward sums [2][N][sum dy, sum dy*xhat]
struct Cells {
    double fwd[2][MAXN][2];
    double bwd[2][MAXN][2];
};

// Every workgroup adds its partial sums with one atomic per channel; with one workgroup per sample thousands of them hit
// the same addresses and serialise (~10 ns each), so the cells exist CELL_REP times: workgroup b adds into replica
// b % CELL_REP, the readers sum the replicas in a fixed order.
constexpr int CELL_REP = 16;
__device__ inline double cell_sum(const Cells* cells, double (Cells::*field)[2][MAXN][2], int blk, int c, int j) {
    double v = 0.0;
#pragma unroll
    for (int r = 0; r < CELL_REP; ++r) v += (cells[r].*field)[blk][c][j];
    return v;
}

// BatchNorm scale/shift of block `blk` for channel c: y = z * sc + sh; xhat = (z - mean) * inv
struct BnCoef {
    float mean, inv, sc, sh;
};
__device__ inline BnCoef bn_coef(const Cells* cells, const float* bn_running, int training, int blk, int c, int N, double count,
                                 float gamma, float beta) {
    BnCoef r;
    float var;
    if (training) {
        const double m = cell_sum(cells, &Cells::fwd, blk, c, 0) / count;
        double v = cell_sum(cells, &Cells::fwd, blk, c, 1) / count - m * m;
        if (v < 0.0) v = 0.0;
        r.mean = (float)m;
        var = (float)v;
    } else {
        r.mean = bn_running[(blk * 2 + 0) * N + c];
        var = bn_running[(blk * 2 + 1) * N + c];
    }
    r.inv = 1.0f / sqrtf(var + BN_EPS);
    r.sc = gamma * r.inv;
    r.sh = beta - r.mean * r.sc;
    return r;
}

// ---------------------------------------------------------------------------------------------------
// The convolutions on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulation).
// One sample is a small GEMM: z[co][t] = sum_k W[co][k] X[k][t], k = (ci, tap), X[(ci, tap)][t] = in[ci][t - (KT-1-tap) D] -- M = N nodes
// (two 16-row tiles), N = T steps (a 16-column tile per wavefront, four wavefronts), K = 6 N (steps of four).  The weights are the A
// operand and stay in registers for the whole launch (lane (i, kq) holds W[16 mt + i][4 s + kq]: contiguous in k in the parameter
// layout); the B operand is one LDS word per step, read at (ci XP + tap D + t) of the zero-padded input tile.  As thread-per-output FMA
// loops the two forward launches took 12 + 10 us and the two backward ones 23 + 19 us at 20 nodes x 50 steps x 512 samples -- bound by
// LDS reads (six or seven 8/16-byte reads per 24 multiply-adds), the longest kernels of ASTGCNN's chain after the graph stage.
// ---------------------------------------------------------------------------------------------------
typedef float tcn_f4 __attribute__((ext_vector_type(4)));
constexpr int MROWS = 32;                                  // tile rows padded to two 16-row matrix tiles
constexpr int KSMAX = (MAXN * KT + 3) / 4;                 // k steps of the (channel, tap) reductions
__device__ __forceinline__ float tcn_row16_sum(float v) {  // sum over the 16 lanes of a row (lanes l, l ^ 1, ^ 2, ^ 4, ^ 8)
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------------
// n <= CNT TBK values src[tid + j TBK] as ONE batch of loads: every load is issued before the first use.  A `for (e = tid; e < n; e += TBK)
// lds[...] = src[e]` loop waits for each load before its store: ten L2 round trips in a row for the [20 x 120] weight rows of a
// convolution, four more for a sample's tile -- half of a 12-us launch (profiles/r05_tiled_path_and_load_chains.md, section 1b).
template <int CNT, int TBK>
__device__ __forceinline__ void tcn_load_batch(float (&v)[CNT], const float* __restrict__ src, int n, int tid) {
#pragma unroll
    for (int j = 0; j < CNT; ++j) {
        const int e = tid + j * TBK;
        v[j] = src[e < n ? e : n - 1];
    }
}

// TCN forward.  STAGE 1: z1 = conv1(x).  STAGE 2: out0 = relu(relu(bn1(z1)) + x); z2 = conv2_dil2(out0).
// One sample per workgroup iteration; per-channel sums of z accumulate in registers and go to the cells once.
// ---------------------------------------------------------------------------------------------------
// (<SN, ST>: nodes and time steps as compile-time constants, 0 = generic; TBK: threads per workgroup, 256 = one 16-step column tile per
//  wavefront)
template <int STAGE, typename Geom, int SN = 0, int ST = 0, int TBK = AB>
static __global__ __launch_bounds__(TBK) void tcn_conv_kernel(Geom g, const float* __restrict__ x, const float* __restrict__ prm,
                                                     const float* __restrict__ bn_running, int training, const float* __restrict__ z1,
                                                     float* __restrict__ zout, float* __restrict__ out0, Cells* cells) {
    static_assert(TBK == 256, "four wavefronts: one 16-step column tile each");
    constexpr int D = STAGE == 1 ? 1 : 2;
    constexpr int PADL = (KT - 1) * D;
    constexpr int KSC = SN ? (SN * KT + 3) / 4 : KSMAX;
    __shared__ __attribute__((aligned(16))) float xs[MAXN][XP];           // left-padded with PADL zeros
    __shared__ float wl[MAXN * (MAXN * KT + 1)];                          // the weights, rows of NK + 1 words (odd pitch: conflict-free reads)
    __shared__ BnCoef co1[MAXN];
    __shared__ float red[4][2][MROWS];
    const int N = SN ? SN : g.N, T = ST ? ST : g.T, tid = threadIdx.x, NK = N * KT, NKP = NK + 1;
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    const float* wsrc = prm + (STAGE == 1 ? g.o_w1 : g.o_w2);
    // (through LDS with coalesced loads: fetched by the lanes straight into their operand registers every load instruction touched 16
    // rows of the matrix -- 60 such instructions per lane took 7 of the kernel's 17 us)
    constexpr int WLC = ((SN ? SN * SN : MAXN * MAXN) * KT + TBK - 1) / TBK;      // loads per thread: the weights ...
    constexpr int XLC = ((SN ? SN : MAXN) * (ST ? ST : MAXT) + TBK - 1) / TBK;     // ... a sample's [N x T] tile
    float xv[XLC], zv[STAGE == 2 ? XLC : 1];
    {
        float wv[WLC];
        tcn_load_batch<WLC, TBK>(wv, wsrc, N * NK, tid);
        tcn_load_batch<XLC, TBK>(xv, x + (int64_t)blockIdx.x * N * T, N * T, tid);          // (the first sample's tile with them)
        if constexpr (STAGE == 2) tcn_load_batch<XLC, TBK>(zv, z1 + (int64_t)blockIdx.x * N * T, N * T, tid);
#pragma unroll
        for (int j = 0; j < WLC; ++j) {
            const int e = tid + j * TBK;
            if (e < N * NK) wl[(e / NK) * NKP + e % NK] = wv[j];
        }
    }
    __syncthreads();
    float wa[2][KSC];
    int boff[KSC];
#pragma unroll
    for (int s = 0; s < KSC; ++s) {
        const int k = 4 * s + kq, kc = k < NK ? k : NK - 1;
        boff[s] = (kc / KT) * XP + (kc % KT) * D;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int co = 16 * mt + li;
            const float v = wl[(co < N ? co : N - 1) * NKP + kc];
            wa[mt][s] = (co < N && k < NK) ? v : 0.f;
        }
    }
    for (int e = tid; e < MAXN * XP; e += TBK) (&xs[0][0])[e] = 0.f;
    if (STAGE == 2 && tid < N)
        co1[tid] = bn_coef(cells, bn_running, training, 0, tid, N, (double)g.BG * T, prm[g.o_g1 + tid], prm[g.o_b1 + tid]);
    float s1[2][4], s2[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[mt][r] = s2[mt][r] = 0.f;
    const int t = 16 * wave + li;
    const float* xsf = &xs[0][0];
    __syncthreads();
    for (int64_t b = blockIdx.x; b < g.B; b += gridDim.x) {
        if (b != blockIdx.x) {
            tcn_load_batch<XLC, TBK>(xv, x + b * N * T, N * T, tid);
            if constexpr (STAGE == 2) tcn_load_batch<XLC, TBK>(zv, z1 + b * N * T, N * T, tid);
        }
#pragma unroll
        for (int j = 0; j < XLC; ++j) {
            const int e = tid + j * TBK;
            if (e < N * T) {
                const int c = e / T, tt = e - c * T;
                float v = xv[j];
                if constexpr (STAGE == 2) {
                    const float y = fmaf(zv[j], co1[c].sc, co1[c].sh);
                    v = fmaxf(fmaxf(y, 0.f) + v, 0.f);
                    out0[b * N * T + e] = v;
                }
                xs[c][PADL + tt] = v;
            }
        }
        __syncthreads();
        if (16 * wave < T) {                                             // (wave-uniform)
            tcn_f4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int s = 0; s < KSC; ++s) {
                const float bv = xsf[boff[s] + t];                        // in[ci][t - (KT-1-tap) D]: column PADL + that = t + tap D
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[0][s], bv, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[1][s], bv, acc[1], 0, 0, 0);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = 16 * mt + 4 * kq + r;
                    if (co < N && t < T) {
                        const float v = acc[mt][r];
                        zout[b * N * T + co * T + t] = v;
                        s1[mt][r] += v;
                        s2[mt][r] = fmaf(v, v, s2[mt][r]);
                    }
                }
        }
        __syncthreads();
    }
    if (training) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = tcn_row16_sum(s1[mt][r]), q = tcn_row16_sum(s2[mt][r]);
                if (li == 0) {
                    red[wave][0][16 * mt + 4 * kq + r] = a;
                    red[wave][1][16 * mt + 4 * kq + r] = q;
                }
            }
        __syncthreads();
        if (tid < N) {
            const float a = (red[0][0][tid] + red[1][0][tid]) + (red[2][0][tid] + red[3][0][tid]);
            const float q = (red[0][1][tid] + red[1][1][tid]) + (red[2][1][tid] + red[3][1][tid]);
            atomicAdd(&cells[blockIdx.x % CELL_REP].fwd[STAGE - 1][tid][0], (double)a);
            atomicAdd(&cells[blockIdx.x % CELL_REP].fwd[STAGE - 1][tid][1], (double)q);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// conv backward.  STAGE 2: dz2 = BN2'(dy2); dW2 += dz2 (*) out0; dout0 = ds1 + conv2^T(dz2); ds0 = dout0 [out0 > 0];
// dy1 = ds0 [bn1(z1) > 0]; BN1 backward sums.   STAGE 1: dz1 = BN1'(dy1); dW1 += dz1 (*) x.
// Two more small GEMMs per sample on the matrix cores:
//   d W[co][(ci, tap)] += sum_t dz[co][t] in[ci][t - (KT-1-tap) D]   M = co, N = (ci, tap) (two or three 16-column tiles per wavefront),
//                                                                    K = t; the accumulators live in registers across the samples
//   d in[ci][t]         = sum_(co, tap) W[co][ci][tap] dz[co][t + (KT-1-tap) D]   M = ci, N = t (a tile per wavefront), K = (co, tap);
//                                                                    the A operand (W seen by input channel) in registers
// One partial weight-gradient row per workgroup, as before.
// ---------------------------------------------------------------------------------------------------
// (Measured and dropped: two wavefronts per SIMD asked of the compiler, __launch_bounds__(TBK, 2).  STAGE 2 at 20 nodes needs 264-280
// registers, so ONE workgroup fits a CU and the 512 samples of the N-CMAPSS batch run as two rounds of 256; at 255 registers the kernel itself
// is 1 us faster and the ASTGCNN step 10 us SLOWER -- the parameter-gradient products of the side stream run beside this kernel and starve.)
template <int STAGE, typename Geom, int SN = 0, int ST = 0, int TBK = AB>
static __global__ __launch_bounds__(TBK) void tcn_conv_bwd_kernel(Geom g, const float* __restrict__ prm, Cells* cells,
                                                         const float* __restrict__ zin, const float* __restrict__ dyin,
                                                         const float* __restrict__ src, const float* __restrict__ ds1,
                                                         const float* __restrict__ z1, float* __restrict__ dy1,
                                                         float* __restrict__ gpart) {
    static_assert(TBK == 256, "four wavefronts");
    constexpr int D = STAGE == 1 ? 1 : 2;
    constexpr int PAD = (KT - 1) * D;
    constexpr int KSC = SN ? (SN * KT + 3) / 4 : KSMAX;                   // k steps over (co, tap)
    constexpr int TSC = ST ? (ST + 3) / 4 : MAXT / 4;                     // k steps over t
    constexpr int NTW = ((SN ? SN : MAXN) * KT + 63) / 64;                // 16-column tiles of (ci, tap) per wavefront
    __shared__ __attribute__((aligned(16))) float xs[MROWS][XP];       // conv input (x or out0), left-padded with zeros (zero behind the row too)
    __shared__ __attribute__((aligned(16))) float dz[MROWS][XP];       // d z, zero from column T on and in the rows behind N
    __shared__ BnCoef cz[MAXN], c1[MAXN];
    __shared__ float bsum[MAXN][2];
    __shared__ float red[4][2][MROWS];
    __shared__ float wl[STAGE == 2 ? MAXN * (MAXN * KT + 1) : 1];        // (see tcn_conv_kernel)
    const int N = SN ? SN : g.N, T = ST ? ST : g.T, tid = threadIdx.x, blk = STAGE - 1, NK = N * KT, NKP = NK + 1;
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    const double count = (double)g.BG * T;
    for (int e = tid; e < MROWS * XP; e += TBK) {
        (&xs[0][0])[e] = 0.f;
        (&dz[0][0])[e] = 0.f;
    }
    // STAGE 2: W by input channel, the A operand of the data gradient: lane (i, kq) holds W[co][16 mt + i][tap], (co, tap) = 4 s + kq
    float wt[STAGE == 2 ? 2 : 1][STAGE == 2 ? KSC : 1];
    int doff[STAGE == 2 ? KSC : 1];
    if constexpr (STAGE == 2) {
        constexpr int WLC = ((SN ? SN * SN : MAXN * MAXN) * KT + TBK - 1) / TBK;
        float wv[WLC];
        tcn_load_batch<WLC, TBK>(wv, prm + g.o_w2, N * NK, tid);
#pragma unroll
        for (int j = 0; j < WLC; ++j) {
            const int e = tid + j * TBK;
            if (e < N * NK) wl[(e / NK) * NKP + e % NK] = wv[j];
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < KSC; ++s) {
            const int k = 4 * s + kq, kc = k < NK ? k : NK - 1, co = kc / KT, tap = kc % KT;
            doff[s] = co * XP + (KT - 1 - tap) * D;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int ci = 16 * mt + li;
                const float v = wl[co * NKP + (ci < N ? ci : N - 1) * KT + tap];
                wt[mt][s] = (ci < N && k < NK) ? v : 0.f;
            }
        }
    }
    // this lane's (ci, tap) columns of the weight gradient: input rows at ci XP + tap D
    int xoff[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int n = 16 * (NTW * wave + j) + li, nc = n < NK ? n : NK - 1;
        xoff[j] = (nc / KT) * XP + (nc % KT) * D;
    }
    if (tid < N) {
        cz[tid] = bn_coef(cells, nullptr, 1, blk, tid, N, count, prm[(STAGE == 1 ? g.o_g1 : g.o_g2) + tid],
                          prm[(STAGE == 1 ? g.o_b1 : g.o_b2) + tid]);
        if (STAGE == 2) c1[tid] = bn_coef(cells, nullptr, 1, 0, tid, N, count, prm[g.o_g1 + tid], prm[g.o_b1 + tid]);
        bsum[tid][0] = (float)(cell_sum(cells, &Cells::bwd, blk, tid, 0) / count);
        bsum[tid][1] = (float)(cell_sum(cells, &Cells::bwd, blk, tid, 1) / count);
    }
    tcn_f4 accw[2][NTW];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < NTW; ++j) accw[mt][j] = (tcn_f4){0.f, 0.f, 0.f, 0.f};
    float a1[2][4], a2[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) a1[mt][r] = a2[mt][r] = 0.f;
    const float* xsf = &xs[0][0];
    const float* dzf = &dz[0][0];
    const int t = 16 * wave + li;
    __syncthreads();
    for (int64_t b = blockIdx.x; b < g.B; b += gridDim.x) {
        {   // the sample's three tiles as one batch of loads (they were a round trip per pass of this loop)
            constexpr int XLC = ((SN ? SN : MAXN) * (ST ? ST : MAXT) + TBK - 1) / TBK;
            float zi[XLC], dyv[XLC], sv[XLC];
            tcn_load_batch<XLC, TBK>(zi, zin + b * N * T, N * T, tid);
            tcn_load_batch<XLC, TBK>(dyv, dyin + b * N * T, N * T, tid);
            tcn_load_batch<XLC, TBK>(sv, src + b * N * T, N * T, tid);
#pragma unroll
            for (int j = 0; j < XLC; ++j) {
                const int e = tid + j * TBK;
                if (e < N * T) {
                    const int c = e / T, tt = e - c * T;
                    const float xh = (zi[j] - cz[c].mean) * cz[c].inv;
                    dz[c][tt] = cz[c].sc * (dyv[j] - bsum[c][0] - xh * bsum[c][1]);
                    xs[c][PAD + tt] = sv[j];
                }
            }
        }
        __syncthreads();
        // d W[co][(ci, tap)] += sum_t dz[co][t] * in[ci][t - (KT-1-tap) D]      (column PAD + that = t + tap D; d z = 0 behind T)
#pragma unroll
        for (int s = 0; s < TSC; ++s) {
            const int tk = 4 * s + kq;
            const float av0 = dzf[li * XP + tk], av1 = dzf[(16 + li) * XP + tk];
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const float bv = xsf[xoff[j] + tk];
                accw[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0, bv, accw[0][j], 0, 0, 0);
                accw[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1, bv, accw[1][j], 0, 0, 0);
            }
        }
        if constexpr (STAGE == 2) {
            // d out0[ci][t] = ds1 + sum_co sum_k W[co][ci][k] dz[co][t + (KT-1-k) D]
            if (16 * wave < T) {
                // (d s1 / z1 of this lane's eight outputs requested in front of the products: inside `if (ci < N && t < T)` behind them each
                // pair was a round trip of its own)
                float dsv[2][4], z1v[2][4];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ci = 16 * mt + 4 * kq + r;
                        const int64_t idx = b * N * T + (ci < N ? ci : N - 1) * T + (t < T ? t : T - 1);
                        dsv[mt][r] = ds1[idx];
                        z1v[mt][r] = z1[idx];
                    }
                tcn_f4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int s = 0; s < KSC; ++s) {
                    const float bv = dzf[doff[s] + t];
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[0][s], bv, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[1][s], bv, acc[1], 0, 0, 0);
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ci = 16 * mt + 4 * kq + r;
                        if (ci < N && t < T) {
                            const int64_t idx = b * N * T + ci * T + t;
                            const float a = acc[mt][r] + dsv[mt][r];
                            const float o0 = xs[ci][PAD + t];
                            const float s0 = o0 > 0.f ? a : 0.f;
                            const float zz = z1v[mt][r];
                            const float y = fmaf(zz, c1[ci].sc, c1[ci].sh);
                            const float dy = y > 0.f ? s0 : 0.f;
                            dy1[idx] = dy;
                            a1[mt][r] += dy;
                            a2[mt][r] = fmaf(dy, (zz - c1[ci].mean) * c1[ci].inv, a2[mt][r]);
                        }
                    }
            }
        }
        __syncthreads();
    }
    float* dst = gpart + (int64_t)blockIdx.x * N * NK;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = 16 * mt + 4 * kq + r, n = 16 * (NTW * wave + j) + li;
                if (co < N && n < NK) dst[co * NK + n] = accw[mt][j][r];
            }
    if constexpr (STAGE == 2) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = tcn_row16_sum(a1[mt][r]), q = tcn_row16_sum(a2[mt][r]);
                if (li == 0) {
                    red[wave][0][16 * mt + 4 * kq + r] = a;
                    red[wave][1][16 * mt + 4 * kq + r] = q;
                }
            }
        __syncthreads();
        if (tid < N) {
            const float a = (red[0][0][tid] + red[1][0][tid]) + (red[2][0][tid] + red[3][0][tid]);
            const float q = (red[0][1][tid] + red[1][1][tid]) + (red[2][1][tid] + red[3][1][tid]);
            atomicAdd(&cells[blockIdx.x % CELL_REP].bwd[0][tid][0], (double)a);
            atomicAdd(&cells[blockIdx.x % CELL_REP].bwd[0][tid][1], (double)q);
        }
    }
}


}  // namespace tcn
}  // namespace rulgnn
