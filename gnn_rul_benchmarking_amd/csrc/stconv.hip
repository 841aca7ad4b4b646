// ST_Conv path for gfx950 (the reference's natively C-MAPSS-wired spatio-temporal convolution):
// Pearson graph over the sensor windows -> MPNN (A X W) -> Conv1d('same') + BatchNorm + ReLU, in parallel with the
// node-channel TCN, gated combination tanh(.) * sigmoid(.) + residual -> Linear; forward and backward.
//
// Reference: models/ST_Conv/Model.py (pcc_graph_construction :10-28, MPNN_mk :30-55, CNNLayer :58-71, TemporalConvNet :81-155,
// ST_Conv_model :173-222) and algorithms/algorithms.py:195-220.  The reference's forward uses the "_1" modules for both
// branches (Model.py:196-206): the two branches are equal, so one branch is computed and its gradient carries both uses; the
// BatchNorm running statistics are updated twice per training forward (stconv_bn_running_update applies the momentum twice).
//
// The TCN is the block shared with ASTGCNN (tcn_nodes.hpp); the theta projection and the weight gradients with a long
// reduction are MFMA GEMMs (sgemm_mfma.hpp); the rest is one workgroup per sample with the [nodes x time] tile in LDS.
#include "sgemm_mfma.hpp"
#include "stgcn_host.hpp"
#include "tcn_nodes.hpp"

namespace rulgnn {

namespace {

using namespace tcn;
constexpr float LEAKY = 0.01f;
constexpr int PADL = (KT - 1) / 2;          // Conv1d(padding='same'), even kernel: 2 left, 3 right
constexpr int PADR = KT - 1 - PADL;

struct ScGeom {
    int64_t B;
    int64_t BG;             // samples behind the BatchNorm statistics: B, or the GLOBAL batch under synchronised BatchNorm
    int N, T, NT;
    int o_th, o_gw, o_gb, o_cw, o_cb, o_gc, o_bc, o_w1, o_g1, o_b1, o_w2, o_g2, o_b2, o_fcw, o_fcb, nparam;
};

__host__ int sc_geometry(const rulgnn_stconv_shape* s, ScGeom* g) {
    if (!s) return RULGNN_EINVAL;
    if (s->batch < 0 || s->num_nodes < 1 || s->time_length < 1 || s->kernel_size < 1) return RULGNN_EINVAL;
    if (s->kernel_size != KT || s->num_nodes > MAXN || s->time_length > MAXT) return RULGNN_EUNSUPPORTED;
    if (s->batch * (int64_t)s->num_nodes > ((int64_t)1 << 30)) return RULGNN_EUNSUPPORTED;
    g->B = s->batch;
    g->BG = s->batch;
    g->N = s->num_nodes;
    g->T = s->time_length;
    g->NT = g->N * g->T;
    const int N = g->N, T = g->T;
    int o = 0;
    auto take = [&](int n) { const int r = o; o += n; return r; };
    g->o_th = take(4);
    g->o_gw = take(T * T); g->o_gb = take(T);
    g->o_cw = take(N * N * KT); g->o_cb = take(N); g->o_gc = take(N); g->o_bc = take(N);
    g->o_w1 = take(N * N * KT); g->o_g1 = take(N); g->o_b1 = take(N);
    g->o_w2 = take(N * N * KT); g->o_g2 = take(N); g->o_b2 = take(N);
    g->o_fcw = take(N * T); g->o_fcb = take(1);
    g->nparam = o;
    return RULGNN_OK;
}

struct Cells3 {                      // the CNN BatchNorm (slot 2 of the BatchNorm buffer), the four thetas
    double fwd[MAXN][2];
    double bwd[MAXN][2];
    double th[4];
};

__device__ inline double cell3_fwd(const Cells3* c3, int c, int j) {
    double v = 0.0;
    for (int r = 0; r < CELL_REP; ++r) v += c3[r].fwd[c][j];
    return v;
}
__device__ inline double cell3_bwd(const Cells3* c3, int c, int j) {
    double v = 0.0;
    for (int r = 0; r < CELL_REP; ++r) v += c3[r].bwd[c][j];
    return v;
}
__device__ inline double cell3_th(const Cells3* c3, int q) {
    double v = 0.0;
    for (int r = 0; r < CELL_REP; ++r) v += c3[r].th[q];
    return v;
}
__device__ inline BnCoef bnc_coef(const Cells3* c3, const float* bn_running, int training, int c, int N, double count, float gamma,
                                  float beta) {
    BnCoef r;
    float var;
    if (training) {
        const double m = cell3_fwd(c3, c, 0) / count;
        double v = cell3_fwd(c3, c, 1) / count - m * m;
        if (v < 0.0) v = 0.0;
        r.mean = (float)m;
        var = (float)v;
    } else {
        r.mean = bn_running[(2 * 2 + 0) * N + c];
        var = bn_running[(2 * 2 + 1) * N + c];
    }
    r.inv = 1.0f / sqrtf(var + tcn::BN_EPS);
    r.sc = gamma * r.inv;
    r.sh = beta - r.mean * r.sc;
    return r;
}

// Pearson adjacency between the nodes' windows and the aggregate A X   (Model.py:10-28, :47)
__global__ __launch_bounds__(AB) void sc_graph_kernel(ScGeom g, const float* __restrict__ x, float* __restrict__ ax) {
    __shared__ float X[MAXN][MAXT + 1];
    __shared__ float C[MAXN][MAXT + 1];
    __shared__ float A[MAXN][MAXN + 1];
    __shared__ float nrm[MAXN];
    const int N = g.N, T = g.T, tid = threadIdx.x;
    for (int64_t b = blockIdx.x; b < g.B; b += gridDim.x) {
        for (int e = tid; e < N * T; e += AB) X[e / T][e % T] = x[b * g.NT + e];
        __syncthreads();
        if (tid < N) {
            float m = 0.f;
            for (int t = 0; t < T; ++t) m += X[tid][t];
            m /= (float)T;
            float s = 0.f;
            for (int t = 0; t < T; ++t) {
                const float c = X[tid][t] - m;
                C[tid][t] = c;
                s = fmaf(c, c, s);
            }
            nrm[tid] = sqrtf(s);
        }
        __syncthreads();
        for (int e = tid; e < N * N; e += AB) {
            const int i = e / N, j = e - i * N;
            float s = 0.f;
            for (int t = 0; t < T; ++t) s = fmaf(C[i][t], C[j][t], s);
            A[i][j] = s / (nrm[i] * nrm[j]);
        }
        __syncthreads();
        for (int e = tid; e < N * T; e += AB) {
            const int i = e / T, t = e - i * T;
            float s = 0.f;
            for (int j = 0; j < N; ++j) s = fmaf(A[i][j], X[j][t], s);
            ax[b * g.NT + e] = s;
        }
        __syncthreads();
    }
}

// gpre += theta.bias (in place); g = leaky(gpre); zc = conv_same(g) + bias; BatchNorm-c sums   (Model.py:53-54, :66)
__global__ __launch_bounds__(AB) void sc_cnn_kernel(ScGeom g, const float* __restrict__ prm, float* __restrict__ gpre,
                                                   float* __restrict__ zc, Cells3* c3, int training) {
    __shared__ float w[MAXN * MAXN * KT];
    __shared__ float gs[MAXN][MAXT + KT - 1];
    __shared__ float zs[MAXN][MAXT + 1];
    const int N = g.N, T = g.T, tid = threadIdx.x;
    for (int e = tid; e < N * N * KT; e += AB) w[e] = prm[g.o_cw + e];
    for (int e = tid; e < N * (KT - 1); e += AB) {
        const int c = e / (KT - 1), q = e % (KT - 1);
        gs[c][q < PADL ? q : T + q] = 0.f;
    }
    float s1 = 0.f, s2 = 0.f;
    __syncthreads();
    for (int64_t b = blockIdx.x; b < g.B; b += gridDim.x) {
        for (int e = tid; e < N * T; e += AB) {
            const int c = e / T, t = e - c * T;
            const float v = gpre[b * g.NT + e] + prm[g.o_gb + t];
            gpre[b * g.NT + e] = v;
            gs[c][PADL + t] = v > 0.f ? v : LEAKY * v;
        }
        __syncthreads();
        for (int e = tid; e < N * T; e += AB) {
            const int co = e / T, t = e - co * T;
            float a = prm[g.o_cb + co];
            for (int ci = 0; ci < N; ++ci) {
                const float* wr = w + (co * N + ci) * KT;
#pragma unroll
                for (int k = 0; k < KT; ++k) a = fmaf(wr[k], gs[ci][t + k], a);
            }
            zc[b * g.NT + e] = a;
            zs[co][t] = a;
        }
        __syncthreads();
        if (training && tid < N)
            for (int t = 0; t < T; ++t) {
                const float v = zs[tid][t];
                s1 += v;
                s2 = fmaf(v, v, s2);
            }
        __syncthreads();
    }
    if (training && tid < N) {
        atomicAdd(&c3[blockIdx.x % CELL_REP].fwd[tid][0], (double)s1);
        atomicAdd(&c3[blockIdx.x % CELL_REP].fwd[tid][1], (double)s2);
    }
}

// combine (Model.py:210-218): res = tanh(th1 t + th2 c) * sigmoid(th3 t + th4 c) + x; pred = fc(res); MSE pieces.
// BACKWARD: instead d res = dpred * fc.weight -> d theta, ds1 / dy2 (TCN), dyc (CNN) and the BatchNorm backward sums.
template <int BACKWARD>
__global__ __launch_bounds__(AB) void sc_head_kernel(ScGeom g, const float* __restrict__ x, const float* __restrict__ prm,
                                                    const float* __restrict__ bn_running, int training, Cells* cells, Cells3* c3,
                                                    const float* __restrict__ zc, const float* __restrict__ z2,
                                                    const float* __restrict__ out0, const float* __restrict__ y,
                                                    float* __restrict__ res, float* __restrict__ pred, float* __restrict__ dpred,
                                                    float* __restrict__ sqerr, float* __restrict__ ds1, float* __restrict__ dy2,
                                                    float* __restrict__ dyc, float inv_gb) {
    __shared__ BnCoef c2[MAXN], cc[MAXN];
    __shared__ float red[AB];
    __shared__ float sums[BACKWARD ? 4 : 1][MAXN][MAXT + 1];     // per-element BatchNorm backward terms of one sample
    const int N = g.N, T = g.T, tid = threadIdx.x;
    const double count = (double)g.BG * T;
    if (tid < N) {
        c2[tid] = bn_coef(cells, bn_running, training, 1, tid, N, count, prm[g.o_g2 + tid], prm[g.o_b2 + tid]);
        cc[tid] = bnc_coef(c3, bn_running, training, tid, N, count, prm[g.o_gc + tid], prm[g.o_bc + tid]);
    }
    float a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;                 // theta gradients (per thread, over all its samples)
    float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;                 // BatchNorm backward sums of channel tid
    const float th1 = prm[g.o_th], th2 = prm[g.o_th + 1], th3 = prm[g.o_th + 2], th4 = prm[g.o_th + 3];
    __syncthreads();
    for (int64_t b = blockIdx.x; b < g.B; b += gridDim.x) {
        float part = 0.f;
        const float dp = BACKWARD ? dpred[b] : 0.f;
        for (int e = tid; e < N * T; e += AB) {
            const int c = e / T;
            const int64_t idx = b * g.NT + e;
            const float zcv = zc[idx], z2v = z2[idx];
            const float ycn = fmaf(zcv, cc[c].sc, cc[c].sh), y2 = fmaf(z2v, c2[c].sc, c2[c].sh);
            const float cv = ycn < 0.f ? 0.f : ycn;              // relu that keeps NaN (constant window -> NaN adjacency), like torch
            const float tv = fmaxf(fmaxf(y2, 0.f) + out0[idx], 0.f);
            const float u1 = fmaf(th1, tv, th2 * cv), u2 = fmaf(th3, tv, th4 * cv);
            const float th = tanhf(u1), sg = 1.0f / (1.0f + expf(-u2));
            if (!BACKWARD) {
                const float r = fmaf(th, sg, x[idx]);
                res[idx] = r;
                part = fmaf(r, prm[g.o_fcw + e], part);
            } else {
                const float dr = dp * prm[g.o_fcw + e];
                const float du1 = dr * sg * (1.0f - th * th), du2 = dr * th * sg * (1.0f - sg);
                a1 = fmaf(du1, tv, a1); a2 = fmaf(du1, cv, a2); a3 = fmaf(du2, tv, a3); a4 = fmaf(du2, cv, a4);
                const float dt = fmaf(du1, th1, du2 * th3), dc = fmaf(du1, th2, du2 * th4);
                const float s1 = tv > 0.f ? dt : 0.f;
                const float d2 = y2 > 0.f ? s1 : 0.f;
                const float dcn = ycn > 0.f ? dc : 0.f;
                ds1[idx] = s1;
                dy2[idx] = d2;
                dyc[idx] = dcn;
                const int t = e - c * T;
                sums[0][c][t] = d2;
                sums[BACKWARD ? 1 : 0][c][t] = d2 * (z2v - c2[c].mean) * c2[c].inv;
                sums[BACKWARD ? 2 : 0][c][t] = dcn;
                sums[BACKWARD ? 3 : 0][c][t] = dcn * (zcv - cc[c].mean) * cc[c].inv;
            }
        }
        if (!BACKWARD) {
            red[tid] = part;
            __syncthreads();
            for (int m = AB / 2; m > 0; m >>= 1) {
                if (tid < m) red[tid] += red[tid + m];
                __syncthreads();
            }
            if (tid == 0) {
                const float pr = red[0] + prm[g.o_fcb];
                pred[b] = pr;
                if (y) {
                    const float d = pr - y[b];
                    dpred[b] = 2.0f * d * inv_gb;
                    sqerr[b] = d * d * inv_gb;
                }
            }
            __syncthreads();
        } else {
            __syncthreads();
            if (tid < N)
                for (int t = 0; t < T; ++t) {
                    b0 += sums[0][tid][t];
                    b1 += sums[BACKWARD ? 1 : 0][tid][t];
                    b2 += sums[BACKWARD ? 2 : 0][tid][t];
                    b3 += sums[BACKWARD ? 3 : 0][tid][t];
                }
            __syncthreads();
        }
    }
    if (BACKWARD) {
        if (tid < N) {
            atomicAdd(&cells[blockIdx.x % CELL_REP].bwd[1][tid][0], (double)b0);
            atomicAdd(&cells[blockIdx.x % CELL_REP].bwd[1][tid][1], (double)b1);
            atomicAdd(&c3[blockIdx.x % CELL_REP].bwd[tid][0], (double)b2);
            atomicAdd(&c3[blockIdx.x % CELL_REP].bwd[tid][1], (double)b3);
        }
        const float av[4] = {a1, a2, a3, a4};
        for (int q = 0; q < 4; ++q) {                 // block tree reduction of each theta gradient, one atomic per workgroup
            __syncthreads();
            red[tid] = av[q];
            __syncthreads();
            for (int m = AB / 2; m > 0; m >>= 1) {
                if (tid < m) red[tid] += red[tid + m];
                __syncthreads();
            }
            if (tid == 0) atomicAdd(&c3[blockIdx.x % CELL_REP].th[q], (double)red[0]);
        }
    }
}

// CNN branch backward: dzc = BN'(dyc); dW += dzc (*) g, db += sum dzc; dg = conv^T(dzc); dgpre = dg * leaky'(gpre) (over gpre)
__global__ __launch_bounds__(AB) void sc_cnn_bwd_kernel(ScGeom g, const float* __restrict__ prm, const Cells3* c3,
                                                       const float* __restrict__ zc, const float* __restrict__ dyc,
                                                       float* __restrict__ gpre, float* __restrict__ gpart) {
    constexpr int NACC = (MAXN * MAXN * KT + MAXN + AB - 1) / AB;
    __shared__ float w[MAXN * MAXN * KT];
    __shared__ float gs[MAXN][MAXT + KT - 1];
    __shared__ float dz[MAXN][MAXT + KT - 1];       // dzc at [PADR + t], zeros around
    __shared__ BnCoef cc[MAXN];
    __shared__ float bsum[MAXN][2];
    const int N = g.N, T = g.T, tid = threadIdx.x;
    const double count = (double)g.BG * T;
    const int nW = N * N * KT, nOut = nW + N;
    for (int e = tid; e < nW; e += AB) w[e] = prm[g.o_cw + e];
    for (int e = tid; e < N * (KT - 1); e += AB) {
        const int c = e / (KT - 1), q = e % (KT - 1);
        gs[c][q < PADL ? q : T + q] = 0.f;
        dz[c][q < PADR ? q : T + q] = 0.f;
    }
    if (tid < N) {
        cc[tid] = bnc_coef(c3, nullptr, 1, tid, N, count, prm[g.o_gc + tid], prm[g.o_bc + tid]);
        bsum[tid][0] = (float)(cell3_bwd(c3, tid, 0) / count);
        bsum[tid][1] = (float)(cell3_bwd(c3, tid, 1) / count);
    }
    float acc[NACC];
#pragma unroll
    for (int r = 0; r < NACC; ++r) acc[r] = 0.f;
    __syncthreads();
    for (int64_t b = blockIdx.x; b < g.B; b += gridDim.x) {
        for (int e = tid; e < N * T; e += AB) {
            const int c = e / T, t = e - c * T;
            const int64_t idx = b * g.NT + e;
            const float xh = (zc[idx] - cc[c].mean) * cc[c].inv;
            dz[c][PADR + t] = cc[c].sc * (dyc[idx] - bsum[c][0] - xh * bsum[c][1]);
            const float v = gpre[idx];
            gs[c][PADL + t] = v > 0.f ? v : LEAKY * v;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < NACC; ++r) {
            const int e = tid + r * AB;
            if (e < nW) {                             // dW[co][ci][k] += sum_t dzc[co][t] g[ci][t + k - PADL]
                const int k = e % KT, ci = (e / KT) % N, co = e / (KT * N);
                float a = 0.f;
                for (int t = 0; t < T; ++t) a = fmaf(dz[co][PADR + t], gs[ci][t + k], a);
                acc[r] += a;
            } else if (e < nOut) {                    // d bias[co] += sum_t dzc[co][t]
                const int co = e - nW;
                float a = 0.f;
                for (int t = 0; t < T; ++t) a += dz[co][PADR + t];
                acc[r] += a;
            }
        }
        // dg[ci][s] = sum_co sum_k W[co][ci][k] dzc[co][s - k + PADL]
        for (int e = tid; e < N * T; e += AB) {
            const int ci = e / T, s = e - ci * T;
            float a = 0.f;
            for (int co = 0; co < N; ++co) {
                const float* wr = w + (co * N + ci) * KT;
#pragma unroll
                for (int k = 0; k < KT; ++k) a = fmaf(wr[k], dz[co][PADR + s - k + PADL], a);
            }
            const int64_t idx = b * g.NT + e;
            gpre[idx] = gpre[idx] > 0.f ? a : LEAKY * a;
        }
        __syncthreads();
    }
    float* dst = gpart + (int64_t)blockIdx.x * nOut;
#pragma unroll
    for (int r = 0; r < NACC; ++r) {
        const int e = tid + r * AB;
        if (e < nOut) dst[e] = acc[r];
    }
}

__global__ __launch_bounds__(AB) void sc_finalize_kernel(ScGeom g, const Cells* cells, const Cells3* c3, float* __restrict__ grads) {
    const int c = blockIdx.x * AB + threadIdx.x, N = g.N;
    if (c < N) {            // the conv weight / bias rows are summed by rows_sum (sgemm_mfma.hpp)
        grads[g.o_g1 + c] = (float)cell_sum(cells, &Cells::bwd, 0, c, 1);
        grads[g.o_b1 + c] = (float)cell_sum(cells, &Cells::bwd, 0, c, 0);
        grads[g.o_g2 + c] = (float)cell_sum(cells, &Cells::bwd, 1, c, 1);
        grads[g.o_b2 + c] = (float)cell_sum(cells, &Cells::bwd, 1, c, 0);
        grads[g.o_gc + c] = (float)cell3_bwd(c3, c, 1);
        grads[g.o_bc + c] = (float)cell3_bwd(c3, c, 0);
    } else if (c < N + 4) {
        grads[g.o_th + (c - N)] = (float)cell3_th(c3, c - N);
    }
}

// batch statistics of the three BatchNorms: [tcn1 | tcn2 | cnn] x (mean, biased var) x N, or weight * (E z, E z^2)
__global__ void sc_bn_batch_kernel(ScGeom g, const Cells* cells, const Cells3* c3, float* __restrict__ bn_batch, float weight) {
    const int e = threadIdx.x;
    if (e >= 3 * g.N) return;
    const int blk = e / g.N, c = e % g.N;
    const double count = (double)g.BG * g.T;
    const double s = blk < 2 ? cell_sum(cells, &Cells::fwd, blk, c, 0) : cell3_fwd(c3, c, 0), q2 = blk < 2 ? cell_sum(cells, &Cells::fwd, blk, c, 1) : cell3_fwd(c3, c, 1);
    const double m = s / count, q = q2 / count;
    if (weight > 0.f) {
        bn_batch[(blk * 2 + 0) * g.N + c] = (float)(weight * m);
        bn_batch[(blk * 2 + 1) * g.N + c] = (float)(weight * q);
    } else {
        const double v = q - m * m;
        bn_batch[(blk * 2 + 0) * g.N + c] = (float)m;
        bn_batch[(blk * 2 + 1) * g.N + c] = (float)(v < 0.0 ? 0.0 : v);
    }
}

// every BatchNorm module runs twice per training forward in the reference (Model.py:196-206): the momentum update twice
__global__ void sc_bn_running_kernel(float* __restrict__ bn, const float* __restrict__ batch, int N, double count, float momentum,
                                     int from_moments) {
    const int e = threadIdx.x;
    if (e >= 3 * N) return;
    const int blk = e / N, c = e % N;
    float mean = batch[(blk * 2 + 0) * N + c], var = batch[(blk * 2 + 1) * N + c];
    if (from_moments) {
        var = var - mean * mean;
        if (var < 0.f) var = 0.f;
    }
    const float unbiased = count > 1.0 ? (float)(var * (count / (count - 1.0))) : var;
    float rm = bn[(blk * 2 + 0) * N + c], rv = bn[(blk * 2 + 1) * N + c];
    for (int k = 0; k < 2; ++k) {
        rm = (1.0f - momentum) * rm + momentum * mean;
        rv = (1.0f - momentum) * rv + momentum * unbiased;
    }
    bn[(blk * 2 + 0) * N + c] = rm;
    bn[(blk * 2 + 1) * N + c] = rv;
}

__global__ void sc_fill_one_kernel(float* p) { p[0] = 1.f; }

struct ScWs {
    size_t cells, c3, one, ax, gpre, zc, z1, out0, z2, res, dpred, sqerr, ds1, dy2, dyc, dy1, gp1, gp2, gp3, split, total;
    int rows;
};

void sc_ws_layout(const ScGeom& g, ScWs* w) {
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t BNT = (size_t)g.B * g.NT * sizeof(float);
    size_t o = 0;
    w->cells = o; o = al(o + sizeof(Cells) * CELL_REP);
    w->c3 = o; o = al(o + sizeof(Cells3) * CELL_REP);
    w->one = o; o = al(o + 256);
    for (size_t* p : {&w->ax, &w->gpre, &w->zc, &w->z1, &w->out0, &w->z2, &w->res, &w->ds1, &w->dy2, &w->dyc, &w->dy1}) {
        *p = o;
        o = al(o + BNT);
    }
    w->dpred = o; o = al(o + (size_t)g.B * sizeof(float));
    w->sqerr = o; o = al(o + (size_t)g.B * sizeof(float));
    w->rows = 1024;
    const size_t nW = (size_t)g.N * g.N * KT;
    w->gp1 = o; o = al(o + w->rows * nW * sizeof(float));
    w->gp2 = o; o = al(o + w->rows * nW * sizeof(float));
    w->gp3 = o; o = al(o + w->rows * (nW + g.N) * sizeof(float));
    size_t mx = 1;
    auto need = [&](int M, int Nn, int64_t K) {
        const size_t v = sgemm_splitk_need_floats(M, Nn, (int)K);
        if (v > mx) mx = v;
    };
    need(g.T, g.T, g.B * g.N); need(1, g.T, g.B * g.N); need(1, g.NT, g.B); need(1, 1, g.B);
    w->split = o; o = al(o + mx * sizeof(float));
    w->total = o;
}

template <typename K>
int sc_rows(K kernel, int64_t items, int cap) {
    int dev = 0, cus = 256, per_cu = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, AB, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    int64_t want = (int64_t)cus * per_cu;
    if (want > items) want = items;
    if (want > cap) want = cap;
    return want < 1 ? 1 : (int)want;
}

}  // namespace

int64_t stconv_param_count(const rulgnn_stconv_shape* s) {
    ScGeom g;
    return sc_geometry(s, &g) == RULGNN_OK ? g.nparam : -1;
}

size_t stconv_workspace_bytes(const rulgnn_stconv_shape* s) {
    ScGeom g;
    if (sc_geometry(s, &g) != RULGNN_OK) return 0;
    ScWs w;
    sc_ws_layout(g, &w);
    return w.total;
}

#define SC_RC(call)                        \
    do {                                   \
        const int rc_ = (call);            \
        if (rc_ != RULGNN_OK) return rc_;  \
    } while (0)

// mode bit 0: forward (args->training: batch / running statistics), bit 1: backward.  The args struct is ASTGCNN's.
int stconv_run(const rulgnn_stconv_shape* s, const rulgnn_astgcnn_args* a, int mode, hipStream_t st) {
    ScGeom g;
    SC_RC(sc_geometry(s, &g));
    ScWs w;
    sc_ws_layout(g, &w);
    if (a->workspace_bytes < w.total) return RULGNN_EWORKSPACE;
    char* ws = static_cast<char*>(a->workspace);
    auto F = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    Cells* cells = reinterpret_cast<Cells*>(ws + w.cells);
    Cells3* c3 = reinterpret_cast<Cells3*>(ws + w.c3);
    const float* prm = a->params;
    const int training = a->training ? 1 : 0;
    const int N = g.N, T = g.T, M = (int)(g.B * N);
    const float inv_gb = 1.0f / (float)(a->global_batch > 0 ? a->global_batch : g.B);
    (void)hipGetLastError();
    const int rows = sc_rows(sc_cnn_bwd_kernel, g.B, w.rows);
    if (mode & 1) {
        if (hipMemsetAsync(cells, 0, w.one - w.cells, st) != hipSuccess) return RULGNN_EHIP;     // both cell blocks
        hipLaunchKernelGGL(sc_graph_kernel, dim3(rows), dim3(AB), 0, st, g, a->x, F(w.ax));
        SC_RC(sgemm(F(w.ax), T, 1, prm + g.o_gw, T, 1, F(w.gpre), T, M, T, T, false, st));
        hipLaunchKernelGGL(sc_cnn_kernel, dim3(rows), dim3(AB), 0, st, g, prm, F(w.gpre), F(w.zc), c3, training);
        hipLaunchKernelGGL((tcn_conv_kernel<1, ScGeom>), dim3(rows), dim3(AB), 0, st, g, a->x, prm, a->bn_stats, training,
                           (const float*)nullptr, F(w.z1), (float*)nullptr, cells);
        hipLaunchKernelGGL((tcn_conv_kernel<2, ScGeom>), dim3(rows), dim3(AB), 0, st, g, a->x, prm, a->bn_stats, training,
                           (const float*)F(w.z1), F(w.z2), F(w.out0), cells);
        hipLaunchKernelGGL(sc_head_kernel<0>, dim3(rows), dim3(AB), 0, st, g, a->x, prm, a->bn_stats, training, cells, c3,
                           (const float*)F(w.zc), (const float*)F(w.z2), (const float*)F(w.out0), a->y, F(w.res), a->pred, F(w.dpred),
                           F(w.sqerr), (float*)nullptr, (float*)nullptr, (float*)nullptr, inv_gb);
        if (training && a->bn_batch)
            hipLaunchKernelGGL(sc_bn_batch_kernel, dim3(1), dim3(128), 0, st, g, (const Cells*)cells, (const Cells3*)c3, a->bn_batch,
                               a->bn_moment_weight);
    }
    if (mode & 2) {
        float* gr = a->grads;
        float* split = F(w.split);
        float* one = F(w.one);
        hipLaunchKernelGGL(sc_fill_one_kernel, dim3(1), dim3(1), 0, st, one);
        if (a->dpred && hipMemcpyAsync(F(w.dpred), a->dpred, sizeof(float) * g.B, hipMemcpyDeviceToDevice, st) != hipSuccess)
            return RULGNN_EHIP;
        // fc: d weight = dpred^T res ; d bias = sum dpred
        SC_RC(sgemm_splitk(F(w.dpred), 0, 1, F(w.res), 1, g.NT, gr + g.o_fcw, g.NT, 1, g.NT, (int)g.B, false, split, st));
        SC_RC(sgemm_splitk(F(w.dpred), 0, 1, one, 0, 0, gr + g.o_fcb, 1, 1, 1, (int)g.B, false, split, st));
        hipLaunchKernelGGL(sc_head_kernel<1>, dim3(rows), dim3(AB), 0, st, g, a->x, prm, a->bn_stats, 1, cells, c3, (const float*)F(w.zc),
                           (const float*)F(w.z2), (const float*)F(w.out0), (const float*)nullptr, (float*)nullptr, (float*)nullptr,
                           F(w.dpred), (float*)nullptr, F(w.ds1), F(w.dy2), F(w.dyc), inv_gb);
        hipLaunchKernelGGL((tcn_conv_bwd_kernel<2, ScGeom>), dim3(rows), dim3(AB), 0, st, g, prm, cells, (const float*)F(w.z2),
                           (const float*)F(w.dy2), (const float*)F(w.out0), (const float*)F(w.ds1), (const float*)F(w.z1), F(w.dy1),
                           F(w.gp2));
        hipLaunchKernelGGL((tcn_conv_bwd_kernel<1, ScGeom>), dim3(rows), dim3(AB), 0, st, g, prm, cells, (const float*)F(w.z1),
                           (const float*)F(w.dy1), a->x, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, F(w.gp1));
        hipLaunchKernelGGL(sc_cnn_bwd_kernel, dim3(rows), dim3(AB), 0, st, g, prm, (const Cells3*)c3, (const float*)F(w.zc),
                           (const float*)F(w.dyc), F(w.gpre), F(w.gp3));
        // theta of the MPNN: d weight = dgpre^T (A X) ; d bias = column sums
        SC_RC(sgemm_splitk(F(w.gpre), 1, T, F(w.ax), 1, T, gr + g.o_gw, T, T, T, M, false, split, st));
        SC_RC(sgemm_splitk(one, 0, 0, F(w.gpre), 1, T, gr + g.o_gb, T, 1, T, M, false, split, st));
        {
            const int nW = N * N * KT;
            if (g.o_cb == g.o_cw + nW) {       // the convolution's weight and bias are neighbours in the flat buffer as in the partial rows: one launch for all
                SC_RC(rows_sum3(F(w.gp1), gr + g.o_w1, F(w.gp2), gr + g.o_w2, rows, nW, nW, F(w.gp3), gr + g.o_cw, nullptr, rows, nW + N, nW + N, st));
            } else {
                SC_RC(rows_sum(F(w.gp1), rows, nW, nW, gr + g.o_w1, st));
                SC_RC(rows_sum(F(w.gp2), rows, nW, nW, gr + g.o_w2, st));
                SC_RC(rows_sum(F(w.gp3), rows, nW + N, nW, gr + g.o_cw, st));
                SC_RC(rows_sum(F(w.gp3) + nW, rows, nW + N, N, gr + g.o_cb, st));
            }
        }
        hipLaunchKernelGGL(sc_finalize_kernel, dim3((N + 4 + AB - 1) / AB), dim3(AB), 0, st, g, (const Cells*)cells, (const Cells3*)c3, gr);
        if (!a->dpred && a->loss)
            (void)block_sum((const float*)F(w.sqerr), (int64_t)g.B, a->loss, st);
    }
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

int stconv_bn_running_update(const rulgnn_stconv_shape* s, float* bn_stats, const float* bn_batch, int64_t count, float momentum,
                             int from_moments, hipStream_t st) {
    ScGeom g;
    SC_RC(sc_geometry(s, &g));
    (void)hipGetLastError();
    hipLaunchKernelGGL(sc_bn_running_kernel, dim3(1), dim3(128), 0, st, bn_stats, bn_batch, g.N, (double)count, momentum, from_moments);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

}  // namespace rulgnn
