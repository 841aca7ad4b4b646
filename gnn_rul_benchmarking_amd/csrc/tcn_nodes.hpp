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
// TCN forward.  STAGE 1: z1 = conv1(x).  STAGE 2: out0 = relu(relu(bn1(z1)) + x); z2 = conv2_dil2(out0).
// One sample per workgroup iteration; per-channel sums of z accumulate in registers and go to the cells once.
// ---------------------------------------------------------------------------------------------------
// (<SN, ST>: nodes and time steps as compile-time constants, 0 = generic: the element loops divide by T, the channel loops run to N;
//  TBK: threads per workgroup -- at 20 nodes x 50 steps the work lists are 260 (node, four steps) items and 400 (co, ci) pairs: 256 threads
//  walk them in two rounds, the second nearly empty; 448 threads take each in one)
template <int STAGE, typename Geom, int SN = 0, int ST = 0, int TBK = AB>
static __global__ __launch_bounds__(TBK) void tcn_conv_kernel(Geom g, const float* __restrict__ x, const float* __restrict__ prm,
                                                     const float* __restrict__ bn_running, int training, const float* __restrict__ z1,
                                                     float* __restrict__ zout, float* __restrict__ out0, Cells* cells) {
    constexpr int D = STAGE == 1 ? 1 : 2;
    constexpr int PADL = (KT - 1) * D;
    static_assert(KT == 6, "the taps of a (co, ci) pair are read as three 8-byte pieces");
    __shared__ __attribute__((aligned(16))) float w[MAXN * MAXN * KT];
    __shared__ __attribute__((aligned(16))) float xs[MAXN][XP];           // left-padded with PADL zeros
    __shared__ float zs[MAXN][MAXT + 1];
    __shared__ BnCoef co1[MAXN];
    const int N = SN ? SN : g.N, T = ST ? ST : g.T, tid = threadIdx.x;
    const float* wsrc = prm + (STAGE == 1 ? g.o_w1 : g.o_w2);
    for (int e = tid; e < N * N * KT; e += TBK) w[e] = wsrc[e];
    for (int e = tid; e < MAXN * XP; e += TBK) (&xs[0][0])[e] = 0.f;
    if (STAGE == 2 && tid < N)
        co1[tid] = bn_coef(cells, bn_running, training, 0, tid, N, (double)g.BG * T, prm[g.o_g1 + tid], prm[g.o_b1 + tid]);
    float s1 = 0.f, s2 = 0.f;
    __syncthreads();
    for (int64_t b = blockIdx.x; b < g.B; b += gridDim.x) {
        const float* xb = x + b * N * T;
        for (int e = tid; e < N * T; e += TBK) {
            const int c = e / T, t = e - c * T;
            float v = xb[e];
            if (STAGE == 2) {
                const float y = fmaf(z1[b * N * T + e], co1[c].sc, co1[c].sh);
                v = fmaxf(fmaxf(y, 0.f) + v, 0.f);
                out0[b * N * T + e] = v;
            }
            xs[c][PADL + t] = v;
        }
        __syncthreads();
        // four consecutive steps t per thread: the six taps of a (co, ci) pair and the 4 + 5 D inputs they meet come as 16- / 8-byte LDS
        // reads once per 24 multiply-adds (it was two 4-byte reads per multiply-add); same order of additions per output as before
        const int Q = (T + 3) / 4;
        for (int wi = tid; wi < N * Q; wi += TBK) {
            const int co = wi / Q, t0 = 4 * (wi - co * Q);
            float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll TCN_UNROLL
            for (int ci = 0; ci < N; ++ci) {
                const float2* wr = reinterpret_cast<const float2*>(w + (co * N + ci) * KT);
                const float2 w01 = wr[0], w23 = wr[1], w45 = wr[2];
                const float wk[KT] = {w01.x, w01.y, w23.x, w23.y, w45.x, w45.y};
                constexpr int NX = (4 + PADL + 3) / 4;                  // xs[ci][t0 + k D + i]: PADL - (KT - 1) D = 0
                float xv[4 * NX];
                const float4* xr = reinterpret_cast<const float4*>(&xs[ci][t0]);
#pragma unroll
                for (int q = 0; q < NX; ++q) {
                    const float4 v = xr[q];
                    xv[4 * q] = v.x; xv[4 * q + 1] = v.y; xv[4 * q + 2] = v.z; xv[4 * q + 3] = v.w;
                }
#pragma unroll
                for (int k = 0; k < KT; ++k)
#pragma unroll
                    for (int i = 0; i < 4; ++i) a[i] = fmaf(wk[k], xv[k * D + i], a[i]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (t0 + i < T) {
                    zout[b * N * T + co * T + t0 + i] = a[i];
                    zs[co][t0 + i] = a[i];
                }
        }
        __syncthreads();
        if (training && tid < N) {
            for (int t = 0; t < T; ++t) {
                const float v = zs[tid][t];
                s1 += v;
                s2 = fmaf(v, v, s2);
            }
        }
        __syncthreads();
    }
    if (training && tid < N) {
        atomicAdd(&cells[blockIdx.x % CELL_REP].fwd[STAGE - 1][tid][0], (double)s1);
        atomicAdd(&cells[blockIdx.x % CELL_REP].fwd[STAGE - 1][tid][1], (double)s2);
    }
}

// ---------------------------------------------------------------------------------------------------
// conv backward.  STAGE 2: dz2 = BN2'(dy2); dW2 += dz2 (*) out0; dout0 = ds1 + conv2^T(dz2); ds0 = dout0 [out0 > 0];
// dy1 = ds0 [bn1(z1) > 0]; BN1 backward sums.   STAGE 1: dz1 = BN1'(dy1); dW1 += dz1 (*) x.
// Weight-gradient accumulators are thread-owned registers (fixed order), one partial row per workgroup.
// ---------------------------------------------------------------------------------------------------
template <int STAGE, typename Geom, int SN = 0, int ST = 0, int TBK = AB>
static __global__ __launch_bounds__(TBK) void tcn_conv_bwd_kernel(Geom g, const float* __restrict__ prm, Cells* cells,
                                                         const float* __restrict__ zin, const float* __restrict__ dyin,
                                                         const float* __restrict__ src, const float* __restrict__ ds1,
                                                         const float* __restrict__ z1, float* __restrict__ dy1,
                                                         float* __restrict__ gpart) {
    constexpr int D = STAGE == 1 ? 1 : 2;
    constexpr int PAD = (KT - 1) * D;
    constexpr int NPAIR = ((SN ? SN * SN : MAXN * MAXN) + TBK - 1) / TBK;      // (co, ci) pairs per thread: all six taps of a pair in one thread
    static_assert(KT == 6, "the taps of a (co, ci) pair are read as three 8-byte pieces");
    __shared__ __attribute__((aligned(16))) float w[MAXN * MAXN * KT];
    __shared__ __attribute__((aligned(16))) float xs[MAXN][XP];        // conv input (x or out0), left-padded with zeros (zero behind the row too)
    __shared__ __attribute__((aligned(16))) float dz[MAXN][XP];        // d z, zero from column T on
    __shared__ float sy[MAXN][MAXT + 1];
    __shared__ float sx[MAXN][MAXT + 1];
    __shared__ BnCoef cz[MAXN], c1[MAXN];
    __shared__ float bsum[MAXN][2];
    const int N = SN ? SN : g.N, T = ST ? ST : g.T, tid = threadIdx.x, blk = STAGE - 1;
    const double count = (double)g.BG * T;
    const int nW = N * N * KT;
    if (STAGE == 2)
        for (int e = tid; e < nW; e += TBK) w[e] = prm[g.o_w2 + e];
    for (int e = tid; e < MAXN * XP; e += TBK) {
        (&xs[0][0])[e] = 0.f;
        (&dz[0][0])[e] = 0.f;
    }
    __syncthreads();
    if (tid < N) {
        cz[tid] = bn_coef(cells, nullptr, 1, blk, tid, N, count, prm[(STAGE == 1 ? g.o_g1 : g.o_g2) + tid],
                          prm[(STAGE == 1 ? g.o_b1 : g.o_b2) + tid]);
        if (STAGE == 2) c1[tid] = bn_coef(cells, nullptr, 1, 0, tid, N, count, prm[g.o_g1 + tid], prm[g.o_b1 + tid]);
        bsum[tid][0] = (float)(cell_sum(cells, &Cells::bwd, blk, tid, 0) / count);
        bsum[tid][1] = (float)(cell_sum(cells, &Cells::bwd, blk, tid, 1) / count);
    }
    float acc[NPAIR][KT];
#pragma unroll
    for (int r = 0; r < NPAIR; ++r)
#pragma unroll
        for (int k = 0; k < KT; ++k) acc[r][k] = 0.f;
    const int Q = (T + 3) / 4;
    constexpr int NX = (4 + PAD + 3) / 4;
    float a1 = 0.f, a2 = 0.f;
    __syncthreads();
    for (int64_t b = blockIdx.x; b < g.B; b += gridDim.x) {
        for (int e = tid; e < N * T; e += TBK) {
            const int c = e / T, t = e - c * T;
            const int64_t idx = b * N * T + e;
            const float xh = (zin[idx] - cz[c].mean) * cz[c].inv;
            dz[c][t] = cz[c].sc * (dyin[idx] - bsum[c][0] - xh * bsum[c][1]);
            xs[c][PAD + t] = src[idx];
        }
        __syncthreads();
        // d W[co][ci][k] += sum_t dz[co][t] * in[ci][t - (KT-1-k) D]
        // (a thread owns all six taps of its pairs: four steps of d z and the 4 + 5 D inputs they meet per 16-byte reads, 24 multiply-adds
        // per five or six LDS reads instead of 48; same order of additions per weight: t ascending, d z = 0 behind T)
#pragma unroll
        for (int r = 0; r < NPAIR; ++r) {
            const int p = tid + r * TBK;
            if (p < N * N) {
                const int ci = p % N, co = p / N;
                float a6[KT] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                const float4* dr = reinterpret_cast<const float4*>(&dz[co][0]);
#pragma unroll TCN_UNROLL
                for (int q4 = 0; q4 < Q; ++q4) {
                    const float4 dv = dr[q4];
                    const float dzv[4] = {dv.x, dv.y, dv.z, dv.w};
                    float xv[4 * NX];
                    const float4* xr = reinterpret_cast<const float4*>(&xs[ci][4 * q4]);       // xs[ci][t + k D]: PAD - (KT - 1 - k) D = k D
#pragma unroll
                    for (int q = 0; q < NX; ++q) {
                        const float4 v = xr[q];
                        xv[4 * q] = v.x; xv[4 * q + 1] = v.y; xv[4 * q + 2] = v.z; xv[4 * q + 3] = v.w;
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int k = 0; k < KT; ++k) a6[k] = fmaf(dzv[i], xv[k * D + i], a6[k]);
                }
#pragma unroll
                for (int k = 0; k < KT; ++k) acc[r][k] += a6[k];
            }
        }
        if (STAGE == 2) {
            // d out0[ci][t] = ds1 + sum_co sum_k W[co][ci][k] dz[co][t + (KT-1-k) D]
            for (int wi = tid; wi < N * Q; wi += TBK) {
                const int ci = wi / Q, t0 = 4 * (wi - ci * Q);
                float a4[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) a4[i] = t0 + i < T ? ds1[b * N * T + ci * T + t0 + i] : 0.f;
#pragma unroll TCN_UNROLL
                for (int co = 0; co < N; ++co) {
                    const float2* wr = reinterpret_cast<const float2*>(w + (co * N + ci) * KT);
                    const float2 w01 = wr[0], w23 = wr[1], w45 = wr[2];
                    const float wk[KT] = {w01.x, w01.y, w23.x, w23.y, w45.x, w45.y};
                    float dv[4 * NX];
                    const float4* dr = reinterpret_cast<const float4*>(&dz[co][t0]);
#pragma unroll
                    for (int q = 0; q < NX; ++q) {
                        const float4 v = dr[q];
                        dv[4 * q] = v.x; dv[4 * q + 1] = v.y; dv[4 * q + 2] = v.z; dv[4 * q + 3] = v.w;
                    }
#pragma unroll
                    for (int k = 0; k < KT; ++k)
#pragma unroll
                        for (int i = 0; i < 4; ++i) a4[i] = fmaf(wk[k], dv[(KT - 1 - k) * D + i], a4[i]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                if (t0 + i >= T) continue;
                const int t = t0 + i;
                const int64_t idx = b * N * T + ci * T + t;
                const float a = a4[i];
                const float o0 = xs[ci][PAD + t];
                const float s0 = o0 > 0.f ? a : 0.f;
                const float zz = z1[idx];
                const float y = fmaf(zz, c1[ci].sc, c1[ci].sh);
                const float dy = y > 0.f ? s0 : 0.f;
                dy1[idx] = dy;
                sy[ci][t] = dy;
                sx[ci][t] = dy * (zz - c1[ci].mean) * c1[ci].inv;
                }
            }
            __syncthreads();
            if (tid < N)
                for (int t = 0; t < T; ++t) {
                    a1 += sy[tid][t];
                    a2 += sx[tid][t];
                }
        }
        __syncthreads();
    }
    float* dst = gpart + (int64_t)blockIdx.x * nW;
#pragma unroll
    for (int r = 0; r < NPAIR; ++r) {
        const int p = tid + r * TBK;
        if (p < N * N) {
#pragma unroll
            for (int k = 0; k < KT; ++k) dst[p * KT + k] = acc[r][k];
        }
    }
    if (STAGE == 2 && tid < N) {
        atomicAdd(&cells[blockIdx.x % CELL_REP].bwd[0][tid][0], (double)a1);
        atomicAdd(&cells[blockIdx.x % CELL_REP].bwd[0][tid][1], (double)a2);
    }
}


}  // namespace tcn
}  // namespace rulgnn
