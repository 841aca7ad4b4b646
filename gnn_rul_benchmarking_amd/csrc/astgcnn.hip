// ASTGCNN path for gfx950: TCN (two causal Conv1d k=6 + BatchNorm + ReLU blocks with residuals) -> tanh gate ->
// exp(-cdist) graph -> Chebyshev graph convolution (K <= 3) -> node mean -> Linear; forward and backward.
//
// Reference: models/ASTGCNN/Model.py (TemporalConvNet :72-146, GatingMechanism :169-181, construct_graph :184-195,
// ChebNet :198-230, ASTGCNN_model :233-254) and algorithms/algorithms.py:139-163 (MSE + Adam).
//
// Decomposition.  Per sample the operands are [nodes<=25] x [time<=64] tiles; the dense
// projections (gate theta, P, Chebyshev filters and their weight gradients) are batched over all samples as
// [batch*nodes, .] GEMMs on the matrix cores (sgemm_mfma.hpp); everything else is one workgroup per sample with the
// tile in LDS.  BatchNorm batch statistics cut the step into phases (fp64 reduction cells, like the ST_GCN chain):
//   conv1 | conv2 | gate -> graph -> head | gate_bwd (BN2 sums) | conv2_bwd (BN1 sums) | conv1_bwd | finalize.
// Node mean and Chebyshev projection commute, so the [batch*nodes, K*E] x [K*E, O] product is done on the node sums
// ([batch, K*E]) -- nodes times less work, same result up to summation order.
#include "adam_device.hpp"
#include "aux_stream.hpp"
#include "reduce_device.hpp"
#include "sgemm_mfma.hpp"
#include "stgcn_host.hpp"
#include "tcn_nodes.hpp"

namespace rulgnn {

namespace {

using namespace tcn;               // AB, KT (Model.py:236), MAXN (torch.cdist's exact path: <= 25 rows), MAXT, Cells, bn_coef, conv kernels

struct AstGeom {
    int64_t B;
    int64_t BG;             // samples behind the BatchNorm statistics: B, or the GLOBAL batch under synchronised BatchNorm
    int N, T, E, O, K, KE;
    int o_w1, o_g1, o_b1, o_w2, o_g2, o_b2, o_thw, o_thb, o_gb, o_pw, o_f, o_fcw, o_fcb, nparam;
};

__host__ int ast_geometry(const rulgnn_astgcnn_shape* s, AstGeom* g) {
    if (!s) return RULGNN_EINVAL;
    if (s->batch < 0 || s->num_nodes < 1 || s->time_length < 1 || s->output_dim < 1 || s->K < 1) return RULGNN_EINVAL;
    if (s->num_nodes > MAXN || s->time_length > MAXT || s->output_dim > 256 || s->K > 3) return RULGNN_EUNSUPPORTED;
    if (s->batch * (int64_t)s->num_nodes > ((int64_t)1 << 30)) return RULGNN_EUNSUPPORTED;
    g->B = s->batch;
    g->BG = s->batch;
    g->N = s->num_nodes;
    g->T = g->E = s->time_length;          // the gate multiplies [N, E] by [N, T] elementwise: E == T (Model.py:181)
    g->O = s->output_dim;
    g->K = s->K;
    g->KE = g->K * g->E;
    const int N = g->N, T = g->T, E = g->E;
    int o = 0;
    g->o_w1 = o; o += N * N * KT;
    g->o_g1 = o; o += N;
    g->o_b1 = o; o += N;
    g->o_w2 = o; o += N * N * KT;
    g->o_g2 = o; o += N;
    g->o_b2 = o; o += N;
    g->o_thw = o; o += E * T;
    g->o_thb = o; o += E;
    g->o_gb = o; o += E;
    g->o_pw = o; o += E * E;
    g->o_f = o; o += g->K * E * g->O;
    g->o_fcw = o; o += g->O;
    g->o_fcb = o; o += 1;
    g->nparam = o;
    return RULGNN_OK;
}

// ---------------------------------------------------------------------------------------------------
// front: gate -> P projection -> graph -> Chebyshev terms -> node sums -> filter product, ONE sample per workgroup iteration (round 4:
// the three [batch*nodes, .] GEMM launches and the gate launch around the graph kernel were 11 + 9 + 11 + 11 us of a 260-us step for 100 k
// multiply-adds per sample; here they are loops over the LDS tiles the graph stage needs anyway):
//   out1 = relu(relu(bn2(z2)) + out0);  zg = tanh(x theta^T + theta.bias + gate.bias);  G = zg * out1          (Model.py:169-181)
//   PX = G P^T;  A = exp(-cdist(PX, PX));  T1 = A G;  T2 = 2 A T1 - G;  node sums of [G | T1 | T2]               (Model.py:184-230)
//   pooled N = Scat Fcat   (Fcat = filters viewed as [K E, O]; the head kernel divides by N)
// Writes what the backward reads: out1, zg (`zg_out`), Tcat = [G | T1 | T2], PX, A, dist, Scat.
// ---------------------------------------------------------------------------------------------------
// Every product runs four output columns per thread: the operand along the output's contiguous dimension comes as one 16-byte LDS read
// per four multiply-adds (tiles at a row pitch of MAXT + 4 floats, the weight tables stored k-major).
constexpr int TP = MAXT + 4;
__device__ __forceinline__ void fma4(float (&acc)[4], float a, const float4& b) {
    acc[0] = fmaf(a, b.x, acc[0]); acc[1] = fmaf(a, b.y, acc[1]); acc[2] = fmaf(a, b.z, acc[2]); acc[3] = fmaf(a, b.w, acc[3]);
}
// n <= CNT AB values src[tid + j AB] as ONE batch of loads: every load is issued before the first use.  A `for (e = tid; e < n; e += AB)
// lds[...] = src[e]` loop waits for each load before its store -- ten L2 round trips of ~0.7 us in a row for a [50 x 50] weight matrix,
// twenty in ast_front_kernel's prologue: half of that kernel's 27 us (profiles/r05_tiled_path_and_load_chains.md, section 1b).
template <int CNT>
__device__ __forceinline__ void load_batch(float (&v)[CNT], const float* __restrict__ src, int n, int tid) {
#pragma unroll
    for (int j = 0; j < CNT; ++j) {
        const int e = tid + j * AB;
        v[j] = src[e < n ? e : n - 1];
    }
}
// (<SN, SE, SO>: nodes, time steps / features (E == T) and output width as compile-time constants, 0 = generic)
template <int SN, int SE, int SO>
__global__ __launch_bounds__(AB) void ast_front_kernel(AstGeom g, const float* __restrict__ x, const float* __restrict__ prm,
                                                      const float* __restrict__ bn_running, int training, const Cells* cells,
                                                      const float* __restrict__ z2, const float* __restrict__ out0,
                                                      float* __restrict__ zg_out, float* __restrict__ out1, float* __restrict__ tcat,
                                                      float* __restrict__ px, float* __restrict__ adj, float* __restrict__ distm,
                                                      float* __restrict__ scat, float* __restrict__ pooled, const float* __restrict__ y,
                                                      float* __restrict__ pred, float* __restrict__ dpred, float* __restrict__ sqerr,
                                                      float* __restrict__ dmat, float inv_gb) {
    __shared__ __attribute__((aligned(16))) float THt[MAXT][TP];         // theta.weight^T: [k = time][t = gate column]
    __shared__ __attribute__((aligned(16))) float PWt[MAXT][TP];         // P.weight^T: [k][c]
    __shared__ __attribute__((aligned(16))) float P[MAXN][TP];
    __shared__ __attribute__((aligned(16))) float G[MAXN][TP];
    __shared__ __attribute__((aligned(16))) float T1[MAXN][TP];          // the x tile first, T1 later
    __shared__ float A[MAXN][MAXN + 1];
    __shared__ float SC[3 * MAXT];
    __shared__ float gbias[MAXT];
    __shared__ BnCoef co2[MAXN];
    const int N = SN ? SN : g.N, T = SE ? SE : g.T, E = SE ? SE : g.E, KE = g.K * E, O = SO ? SO : g.O, tid = threadIdx.x;
    const int Q = (E + 3) / 4;                    // column quads (E == T)
    constexpr int WL = ((SE ? SE * SE : MAXT * MAXT) + AB - 1) / AB;           // loads per thread of a weight matrix
    constexpr int XL = ((SN ? SN : MAXN) * (SE ? SE : MAXT) + AB - 1) / AB;     // ... of a sample's [N x T] tile
    // The filter column of this thread's (k stripe, output) and the head's weights: requested with the prologue's batch and kept in
    // registers -- in the pooled stage they were two more round trips per sample, in the head three (one instantiated output width only).
    constexpr bool PRE = SO != 0 && 4 * SO <= AB && SO <= 64;
    constexpr int FK = PRE ? (3 * (SE ? SE : MAXT) + 3) / 4 : 1;
    float fv[FK];
    float fcw0 = 0.f, fcb0 = 0.f;
    if constexpr (PRE) {
        const int st = tid / O, o = tid - st * O;
#pragma unroll
        for (int q = 0; q < FK; ++q) {
            const int k = st + 4 * q;
            fv[q] = prm[g.o_f + (k < KE ? k : KE - 1) * O + (tid < 4 * O ? o : 0)];
        }
        fcw0 = prm[g.o_fcw + (tid < O ? tid : O - 1)];
        fcb0 = prm[g.o_fcb];
    }
    {
        float wv[WL], pv[WL], xv[XL];
        load_batch(wv, prm + g.o_thw, E * T, tid);
        load_batch(pv, prm + g.o_pw, E * E, tid);
        load_batch(xv, x + (int64_t)blockIdx.x * N * T, N * T, tid);            // (the first sample's tile with them)
        // (... and the gate bias, the BatchNorm coefficients' sums: one round trip for the whole prologue)
        const float gb0 = prm[g.o_thb + (tid < E ? tid : E - 1)] + prm[g.o_gb + (tid < E ? tid : E - 1)];
        if (tid < N) co2[tid] = bn_coef(cells, bn_running, training, 1, tid, N, (double)g.BG * T, prm[g.o_g2 + tid], prm[g.o_b2 + tid]);
        if (tid < E) gbias[tid] = gb0;
        for (int e = tid + AB; e < E; e += AB) gbias[e] = prm[g.o_thb + e] + prm[g.o_gb + e];
        for (int e = tid; e < MAXT * TP; e += AB) { (&THt[0][0])[e] = 0.f; (&PWt[0][0])[e] = 0.f; }
        for (int e = tid; e < MAXN * TP; e += AB) { (&P[0][0])[e] = 0.f; (&G[0][0])[e] = 0.f; (&T1[0][0])[e] = 0.f; }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < WL; ++j) {
            const int e = tid + j * AB;
            if (e < E * T) THt[e % T][e / T] = wv[j];
            if (e < E * E) PWt[e % E][e / E] = pv[j];
        }
#pragma unroll
        for (int j = 0; j < XL; ++j) {
            const int e = tid + j * AB;
            if (e < N * T) T1[e / T][e % T] = xv[j];
        }
    }
    __syncthreads();
    for (int64_t b = blockIdx.x; b < g.B; b += gridDim.x) {
        float* tc = tcat + b * N * KE;
        if (b != blockIdx.x) {
            float xv[XL];
            load_batch(xv, x + b * N * T, N * T, tid);
#pragma unroll
            for (int j = 0; j < XL; ++j) {
                const int e = tid + j * AB;
                if (e < N * T) T1[e / T][e % T] = xv[j];
            }
            __syncthreads();
        }
        for (int w = tid; w < N * Q; w += AB) {                  // gate: four columns t per thread
            const int c = w / Q, t0 = 4 * (w - c * Q);
            // (z2 / out0 of the four columns requested in front of the product: behind it, inside `if (t < T)`, they were four round trips)
            float zv[4], ov[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t idx = b * N * T + c * T + (t0 + r < T ? t0 + r : T - 1);
                zv[r] = z2[idx];
                ov[r] = out0[idx];
            }
            float zp[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 10
            for (int k = 0; k < T; ++k) fma4(zp, T1[c][k], *reinterpret_cast<const float4*>(&THt[k][t0]));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = t0 + r;
                if (t < T) {
                    const int64_t idx = b * N * T + c * T + t;
                    const float y = fmaf(zv[r], co2[c].sc, co2[c].sh);
                    const float o1 = fmaxf(fmaxf(y, 0.f) + ov[r], 0.f);
                    const float zg = tanhf(zp[r] + gbias[t]);
                    out1[idx] = o1;
                    zg_out[idx] = zg;
                    const float gv = zg * o1;
                    G[c][t] = gv;
                    tc[c * KE + t] = gv;
                }
            }
        }
        __syncthreads();
        for (int w = tid; w < N * Q; w += AB) {                  // PX = G P^T
            const int i = w / Q, c0 = 4 * (w - i * Q);
            float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 10
            for (int k = 0; k < E; ++k) fma4(a, G[i][k], *reinterpret_cast<const float4*>(&PWt[k][c0]));
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (c0 + r < E) {
                    P[i][c0 + r] = a[r];
                    px[b * N * E + i * E + c0 + r] = a[r];
                }
        }
        __syncthreads();
        for (int e = tid; e < N * N; e += AB) {                  // (columns >= E of the P rows are zero: whole quads)
            const int i = e / N, j = e - i * N;
            float d2 = 0.f;
            for (int q = 0; q < Q; ++q) {
                const float4 pi = *reinterpret_cast<const float4*>(&P[i][4 * q]), pj = *reinterpret_cast<const float4*>(&P[j][4 * q]);
                const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z, dw = pi.w - pj.w;
                d2 = fmaf(dx, dx, d2); d2 = fmaf(dy, dy, d2); d2 = fmaf(dz, dz, d2); d2 = fmaf(dw, dw, d2);
            }
            const float ds = sqrtf(d2);
            const float a = expf(-ds);
            A[i][j] = a;
            adj[b * N * N + e] = a;
            distm[b * N * N + e] = ds;
        }
        __syncthreads();
        if (g.K > 1) {
            for (int w = tid; w < N * Q; w += AB) {
                const int i = w / Q, c0 = 4 * (w - i * Q);
                float a[4] = {0.f, 0.f, 0.f, 0.f};
                for (int j = 0; j < N; ++j) fma4(a, A[i][j], *reinterpret_cast<const float4*>(&G[j][c0]));
                // (T1 aliases the x tile, which nobody reads any more)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (c0 + r < E) {
                        T1[i][c0 + r] = a[r];
                        tc[i * KE + E + c0 + r] = a[r];
                    } else {
                        T1[i][c0 + r] = 0.f;
                    }
            }
            __syncthreads();
        }
        if (g.K > 2) {
            for (int w = tid; w < N * Q; w += AB) {
                const int i = w / Q, c0 = 4 * (w - i * Q);
                float a[4] = {0.f, 0.f, 0.f, 0.f};
                for (int j = 0; j < N; ++j) fma4(a, A[i][j], *reinterpret_cast<const float4*>(&T1[j][c0]));
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (c0 + r < E) {
                        const float v = 2.0f * a[r] - G[i][c0 + r];
                        tc[i * KE + 2 * E + c0 + r] = v;
                        P[i][c0 + r] = v;                        // (P is dead behind the adjacency: the third Chebyshev term's LDS copy)
                    }
            }
            __syncthreads();
        }
        for (int e = tid; e < KE; e += AB) {                    // node sums over the LDS copies of the three terms (G | T1 | P): the rows
            const int part = e / E, c = e - part * E;           // of tc this workgroup just wrote were a store drain + a round trip away
            const float (*src)[TP] = part == 0 ? G : (part == 1 ? T1 : P);
            float sum = 0.f;
            for (int i = 0; i < N; ++i) sum += src[i][c];
            scat[b * KE + e] = sum;
            SC[e] = sum;
        }
        __syncthreads();
        {                                                        // pooled N = Scat Fcat: four k-stripes per output, combined in fixed order
            float* PL = &T1[0][0];                               // [4][O] partial sums over the T1 tile (dead by now; O <= 256 <= MAXN TP / 4)
            if constexpr (PRE) {
                if (tid < 4 * O) {
                    const int st = tid / O;
                    float a = 0.f;
#pragma unroll
                    for (int q = 0; q < FK; ++q) {
                        const int k = st + 4 * q;
                        if (k < KE) a = fmaf(SC[k], fv[q], a);
                    }
                    PL[tid] = a;
                }
            } else {
            for (int it = tid; it < 4 * O; it += AB) {
                const int st = it / O, o = it - st * O;
                float a = 0.f;
#pragma unroll 8
                for (int k = st; k < KE; k += 4) a = fmaf(SC[k], prm[g.o_f + k * O + o], a);
                PL[st * O + o] = a;
            }
            }
            __syncthreads();
            // head (ast_head_kernel's arithmetic, wavefront 0): pooled / N, pred = pooled fc^T + b, MSE pieces, D = dpred fc.weight / N
            if (tid < 64) {
                const float inv_n = 1.0f / (float)N;
                float a = 0.f;
                for (int o = tid; o < O; o += 64) {
                    const float v = ((PL[o] + PL[O + o]) + (PL[2 * O + o] + PL[3 * O + o])) / (float)N;
                    pooled[b * O + o] = v;
                    a = fmaf(v, PRE ? fcw0 : prm[g.o_fcw + o], a);
                }
                for (int m = 32; m > 0; m >>= 1) a += __shfl_xor(a, m, 64);
                const float pr = a + (PRE ? fcb0 : prm[g.o_fcb]);
                if (tid == 0) pred[b] = pr;
                if (y) {
                    const float d = pr - y[b];
                    const float dp = 2.0f * d * inv_gb;
                    if (tid == 0) {
                        dpred[b] = dp;
                        sqerr[b] = d * d * inv_gb;
                    }
                    for (int o = tid; o < O; o += 64) dmat[b * O + o] = dp * (PRE ? fcw0 : prm[g.o_fcw + o]) * inv_n;
                }
            }
        }
        __syncthreads();
    }
}

// head: pooled = (Scat Fcat) / N (from the GEMM, in `pooled`), pred = pooled fc^T + b; MSE pieces; D = dpred * fc.weight / N
__global__ __launch_bounds__(AB) void ast_head_kernel(AstGeom g, const float* __restrict__ prm, float* __restrict__ pooled,
                                                     const float* __restrict__ y, const float* __restrict__ dpred_in,
                                                     float* __restrict__ pred, float* __restrict__ dpred, float* __restrict__ sqerr,
                                                     float* __restrict__ dmat, float inv_gb, int backward_only) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, O = g.O;
    const float inv_n = 1.0f / (float)g.N;
    for (int64_t b = (int64_t)blockIdx.x * (AB / 64) + wave; b < g.B; b += (int64_t)gridDim.x * (AB / 64)) {
        float dp;
        if (!backward_only) {
            float a = 0.f;
            for (int o = lane; o < O; o += 64) {
                const float v = pooled[b * O + o] / (float)g.N;
                pooled[b * O + o] = v;
                a = fmaf(v, prm[g.o_fcw + o], a);
            }
            for (int m = 32; m > 0; m >>= 1) a += __shfl_xor(a, m, 64);
            const float pr = a + prm[g.o_fcb];
            if (lane == 0) pred[b] = pr;
            if (!y) continue;
            const float d = pr - y[b];
            dp = 2.0f * d * inv_gb;
            if (lane == 0) {
                dpred[b] = dp;
                sqerr[b] = d * d * inv_gb;
            }
        } else {
            dp = dpred_in[b];
            if (lane == 0) dpred[b] = dp;
        }
        for (int o = lane; o < O; o += 64) dmat[b * O + o] = dp * prm[g.o_fcw + o] * inv_n;
    }
}

// ---------------------------------------------------------------------------------------------------
// graph backward (one sample per workgroup).  d cheb is the same row dm/N for every node, so the three d T_k are
// rows dt0, dt1, dt2 (from DT = D Fcat^T) broadcast over the nodes, which collapses most products to vectors:
//   u_j = dt2.T1_j, v_j = dt1.G_j, w_j = dt2.G_j, cs_i = sum_j A[j][i]
//   dA[i][j] = 2 u_j + v_j + 2 cs_i w_j
//   dG_cheb[i] = dt0 - dt2 + cs_i dt1 + 2 (sum_j A[j][i] cs_j) dt2
//   d dist = -A dA;  dPX_i = sum_j (d dist_ij + d dist_ji) (PX_i - PX_j) / dist_ij   (0 where dist = 0, as torch)
// Round 4: the two GEMM launches around it live here -- DT = D Fcat^T (one row of K E values per sample) in front, and behind it the
// projection's share of the gate gradient, dG = dG_cheb + dPX P, from the dPX tile in LDS.
// ---------------------------------------------------------------------------------------------------
template <int SN, int SE, int SO>
__global__ __launch_bounds__(AB) void ast_graph_bwd_kernel(AstGeom g, const float* __restrict__ prm, const float* __restrict__ px,
                                                          const float* __restrict__ tcat, const float* __restrict__ adj,
                                                          const float* __restrict__ distm, const float* __restrict__ dmat,
                                                          float* __restrict__ dpx, Cells* cells, const float* __restrict__ z2,
                                                          const float* __restrict__ out1, float* __restrict__ zg_dzpre,
                                                          float* __restrict__ ds1, float* __restrict__ dy2, float* __restrict__ thb_part) {
    // ... and the gate's bias gradient d theta.bias = d gate.bias = column sums of d Zpre: one partial row per workgroup (summed by the
    // finalize kernel; it was a split-K launch pair and a copy at the END of the side chain, which had become the step's critical path)
    // ... and the gate backward with the tail of the TCN backward (it consumed d G element by element, one sample per workgroup as well):
    //   dzg = dG out1; dZpre = dzg (1 - zg^2) (over zg); dout1 = dG zg; ds1 = dout1 [out1 > 0]; dy2 = ds1 [bn2(z2) > 0]; BN2 backward sums
    __shared__ float sy[MAXN][MAXT + 1];
    __shared__ float sx[MAXN][MAXT + 1];
    __shared__ float DPX2[MAXN][MAXT + 1];       // d Zpre of the sample (for its column sums)
    __shared__ BnCoef co2[MAXN];
    __shared__ __attribute__((aligned(16))) float PW[MAXT][TP];           // P.weight [k][c]
    __shared__ __attribute__((aligned(16))) float DPX[MAXN][TP];
    __shared__ __attribute__((aligned(16))) float P[MAXN][TP];
    __shared__ float DM[256];
    __shared__ float DTP[4][3 * MAXT];
    __shared__ float A[MAXN][MAXN + 1];
    __shared__ float Cs[MAXN][MAXN + 1];          // (d dist_ij + d dist_ji) / dist_ij
    __shared__ float Cf[MAXN][MAXN + 1];
    __shared__ float d0[MAXT], d1[MAXT], d2v[MAXT];
    __shared__ float u[MAXN], v[MAXN], w[MAXN], cs[MAXN], acs[MAXN];
    const int N = SN ? SN : g.N, E = SE ? SE : g.E, KE = g.K * E, O = SO ? SO : g.O, tid = threadIdx.x;
    const int Q = (E + 3) / 4;
    constexpr int WL = ((SE ? SE * SE : MAXT * MAXT) + AB - 1) / AB;           // loads per thread of a weight matrix (load_batch)
    constexpr int XL = ((SN ? SN : MAXN) * (SE ? SE : MAXT) + AB - 1) / AB;     // ... of a sample's [N x E] tile
    constexpr int AL = ((SN ? SN * SN : MAXN * MAXN) + AB - 1) / AB;            // ... of its [N x N] matrices
    // the filter values of this thread's (o stripe, row) items of the D Fcat^T stage: requested with the prologue's batch and kept in
    // registers (they were two round trips per pass of that stage; instantiated shapes only)
    constexpr bool PRE = SO != 0 && SE != 0 && SO % 4 == 0;
    constexpr int DTN = PRE ? (4 * 3 * SE + AB - 1) / AB : 1, DTQ = PRE ? SO / 4 : 1;
    float frv[DTN][DTQ];
    if constexpr (PRE) {
#pragma unroll
        for (int ps = 0; ps < DTN; ++ps) {
            const int it = tid + ps * AB < 4 * KE ? tid + ps * AB : 4 * KE - 1;
            const int st = it / KE, e = it - st * KE;
#pragma unroll
            for (int q = 0; q < DTQ; ++q) frv[ps][q] = prm[g.o_f + e * O + st + 4 * q];
        }
    }
    {
        float pv[WL];
        load_batch(pv, prm + g.o_pw, E * E, tid);
        if (tid < N) co2[tid] = bn_coef(cells, nullptr, 1, 1, tid, N, (double)g.BG * g.T, prm[g.o_g2 + tid], prm[g.o_b2 + tid]);
        for (int e = tid; e < MAXT * TP; e += AB) (&PW[0][0])[e] = 0.f;
        for (int e = tid; e < MAXN * TP; e += AB) { (&P[0][0])[e] = 0.f; (&DPX[0][0])[e] = 0.f; }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < WL; ++j) {
            const int e = tid + j * AB;
            if (e < E * E) PW[e / E][e % E] = pv[j];
        }
    }
    float a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int64_t b = blockIdx.x; b < g.B; b += gridDim.x) {
        const float* tc = tcat + b * N * KE;
        {   // the sample's tiles: one batch of loads (they were nine round trips in a row)
            float pv[XL], av[AL], cv[AL];
            load_batch(pv, px + b * N * E, N * E, tid);
            load_batch(av, adj + b * N * N, N * N, tid);
            load_batch(cv, distm + b * N * N, N * N, tid);
            const float dm = dmat[b * O + (tid < O ? tid : O - 1)];
#pragma unroll
            for (int j = 0; j < XL; ++j) {
                const int e = tid + j * AB;
                if (e < N * E) P[e / E][e % E] = pv[j];
            }
#pragma unroll
            for (int j = 0; j < AL; ++j) {
                const int e = tid + j * AB;
                if (e < N * N) {
                    A[e / N][e % N] = av[j];
                    Cs[e / N][e % N] = cv[j];
                }
            }
            if (tid < O) DM[tid] = dm;
            for (int e = tid + AB; e < O; e += AB) DM[e] = dmat[b * O + e];
        }
        __syncthreads();
        if constexpr (PRE) {                                     // DT = D Fcat^T: row e of the filters viewed as [K E, O], four o-stripes
#pragma unroll
            for (int ps = 0; ps < DTN; ++ps) {
                const int it = tid + ps * AB;
                if (it < 4 * KE) {
                    const int st = it / KE, e = it - st * KE;
                    float a = 0.f;
#pragma unroll
                    for (int q = 0; q < DTQ; ++q) a = fmaf(DM[st + 4 * q], frv[ps][q], a);
                    DTP[st][e] = a;
                }
            }
        } else {
        for (int it = tid; it < 4 * KE; it += AB) {
            const int st = it / KE, e = it - st * KE;
            const float* fr = prm + g.o_f + e * O;
            float a = 0.f;
#pragma unroll 8
            for (int o = st; o < O; o += 4) a = fmaf(DM[o], fr[o], a);
            DTP[st][e] = a;
        }
        }
        __syncthreads();
        for (int e = tid; e < 3 * E; e += AB) {
            const float a = e < KE ? (DTP[0][e] + DTP[1][e]) + (DTP[2][e] + DTP[3][e]) : 0.f;
            if (e < E) d0[e] = a;
            else if (e < 2 * E) d1[e - E] = a;
            else d2v[e - 2 * E] = a;
        }
        __syncthreads();
        {   // per node j: three products over the E channels of its (global) Chebyshev rows and a column sum of A -- eight lanes per node and
            // xor-shuffles (N threads walked 2-3 E dependent global loads each while the other 236 waited)
            const int j = tid >> 3, l8 = tid & 7;
            if (j < N) {
                float uu = 0.f, vv = 0.f, ww = 0.f, c = 0.f;
                constexpr int EL = ((SE ? SE : MAXT) + 7) / 8;                 // (every load of the lane first: one round trip, not EL)
                float gj[EL], g1[EL];
#pragma unroll
                for (int q = 0; q < EL; ++q) {
                    const int e = l8 + 8 * q < E ? l8 + 8 * q : E - 1;
                    gj[q] = tc[j * KE + e];
                    g1[q] = g.K > 2 ? tc[j * KE + E + e] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < EL; ++q) {
                    const int e = l8 + 8 * q;
                    if (e < E) {
                        vv = fmaf(d1[e], gj[q], vv);
                        ww = fmaf(d2v[e], gj[q], ww);
                        if (g.K > 2) uu = fmaf(d2v[e], g1[q], uu);
                    }
                }
                for (int i = l8; i < N; i += 8) c += A[i][j];
#pragma unroll
                for (int o = 1; o < 8; o <<= 1) {
                    uu += __shfl_xor(uu, o, 64); vv += __shfl_xor(vv, o, 64); ww += __shfl_xor(ww, o, 64); c += __shfl_xor(c, o, 64);
                }
                if (l8 == 0) { u[j] = uu; v[j] = vv; w[j] = ww; cs[j] = c; }
            }
        }
        __syncthreads();
        if (tid < N) {
            float a = 0.f;
            for (int j = 0; j < N; ++j) a = fmaf(A[j][tid], cs[j], a);
            acs[tid] = a;
        }
        // d dist (unsymmetrised) into Cf
        for (int e = tid; e < N * N; e += AB) {
            const int i = e / N, j = e - i * N;
            const float dA = g.K > 1 ? 2.0f * u[j] + v[j] + 2.0f * cs[i] * w[j] : 0.f;
            Cf[i][j] = -A[i][j] * dA;
        }
        __syncthreads();
        for (int e = tid; e < N * N; e += AB) {                  // the symmetrised coefficient once per pair (it used to be once per output)
            const int i = e / N, j = e - i * N;
            const float dist = Cs[i][j];
            A[i][j] = dist > 0.f ? (Cf[i][j] + Cf[j][i]) / dist : 0.f;        // (A itself is no longer needed)
        }
        __syncthreads();
        for (int wk = tid; wk < N * Q; wk += AB) {
            const int i = wk / Q, c0 = 4 * (wk - i * Q);
            const float4 pi = *reinterpret_cast<const float4*>(&P[i][c0]);
            float a[4] = {0.f, 0.f, 0.f, 0.f};
            for (int j = 0; j < N; ++j) {
                const float cf = A[i][j];
                const float4 pj = *reinterpret_cast<const float4*>(&P[j][c0]);
                a[0] = fmaf(cf, pi.x - pj.x, a[0]); a[1] = fmaf(cf, pi.y - pj.y, a[1]);
                a[2] = fmaf(cf, pi.z - pj.z, a[2]); a[3] = fmaf(cf, pi.w - pj.w, a[3]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (c0 + r < E) {
                    dpx[b * N * E + i * E + c0 + r] = a[r];
                    DPX[i][c0 + r] = a[r];
                }
        }
        __syncthreads();
        for (int wk = tid; wk < N * Q; wk += AB) {               // dG = dG_cheb + dPX P
            const int i = wk / Q, c0 = 4 * (wk - i * Q);
            // (the saved values of the four columns requested in front of the product, not inside `if (c < E)` behind it: every load there
            // was a round trip of its own, and the stores in between kept the next ones from being issued early)
            float zgv[4], o1v[4], zzv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t idx = b * N * E + i * E + (c0 + r < E ? c0 + r : E - 1);
                zgv[r] = zg_dzpre[idx];
                o1v[r] = out1[idx];
                zzv[r] = z2[idx];
            }
            float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 10
            for (int k = 0; k < E; ++k) fma4(a, DPX[i][k], *reinterpret_cast<const float4*>(&PW[k][c0]));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = c0 + r;
                if (c < E) {
                    float gch = d0[c];
                    if (g.K > 1) gch += cs[i] * d1[c];
                    if (g.K > 2) gch += (2.0f * acs[i] - 1.0f) * d2v[c];
                    const float gg = gch + a[r];
                    const int64_t idx = b * N * E + i * E + c;               // (E == T: node i is the BatchNorm channel, c the time step)
                    const float zg = zgv[r], o1 = o1v[r];
                    const float dzp = gg * o1 * (1.0f - zg * zg);
                    zg_dzpre[idx] = dzp;
                    DPX2[i][c] = dzp;
                    const float sv = o1 > 0.f ? gg * zg : 0.f;
                    ds1[idx] = sv;
                    const float zz = zzv[r];
                    const float yv = fmaf(zz, co2[i].sc, co2[i].sh);
                    const float dy = yv > 0.f ? sv : 0.f;
                    dy2[idx] = dy;
                    sy[i][c] = dy;
                    sx[i][c] = dy * (zz - co2[i].mean) * co2[i].inv;
                }
            }
        }
        __syncthreads();
        if (tid < N)
            for (int t = 0; t < E; ++t) {
                a1 += sy[tid][t];
                a2 += sx[tid][t];
            }
        if (tid >= 64 && tid < 64 + E)
            for (int i = 0; i < N; ++i) a3 += DPX2[i][tid - 64];
        __syncthreads();
    }
    if (tid < N) {
        atomicAdd(&cells[blockIdx.x % CELL_REP].bwd[1][tid][0], (double)a1);
        atomicAdd(&cells[blockIdx.x % CELL_REP].bwd[1][tid][1], (double)a2);
    }
    if (tid >= 64 && tid < 64 + E) thb_part[(int64_t)blockIdx.x * E + tid - 64] = a3;
}

// finalize: conv partial rows -> gradient; BatchNorm gamma/beta gradients and batch statistics from the cells; loss
// (x bn_scale: under synchronised BatchNorm the cells hold GLOBAL sums on every rank and only one rank may contribute them)
// (one workgroup; round 4: also the batch statistics out and the loss sum -- two launches less on the chain)
__device__ __forceinline__ void ast_bn_batch_body(const AstGeom& g, const Cells* cells, float* __restrict__ bn_batch, float weight, int e);
struct AstFin {
    const Cells* cells;
    float* grads;
    float bn_scale;
    float* bn_batch;
    float bn_weight;
    const float* sqerr;
    float* loss;
    float* bn_running;
    float bn_momentum;
};
// (the first AB threads of the workgroup work; every thread of it must call: barriers inside.  epi(dst, value): behind every gradient
// element stored -- the optimizer update of that element in a whole step, adam_device.hpp)
template <class Epi>
__device__ __forceinline__ void ast_finalize_body(const AstGeom& g, const AstFin& f, float (&red)[AB], const Epi& epi) {
    const Cells* cells = f.cells;
    float* __restrict__ grads = f.grads;
    float* __restrict__ bn_batch = f.bn_batch;
    float* __restrict__ bn_running = f.bn_running;
    const float* __restrict__ sqerr = f.sqerr;
    float* __restrict__ loss = f.loss;
    const float bn_scale = f.bn_scale, bn_weight = f.bn_weight, bn_momentum = f.bn_momentum;
    const int c = threadIdx.x;
    if (c < g.N) {          // the conv weight rows are summed by rows_sum (sgemm_mfma.hpp)
        const float v4[4] = {bn_scale * (float)cell_sum(cells, &Cells::bwd, 0, c, 1), bn_scale * (float)cell_sum(cells, &Cells::bwd, 0, c, 0),
                             bn_scale * (float)cell_sum(cells, &Cells::bwd, 1, c, 1), bn_scale * (float)cell_sum(cells, &Cells::bwd, 1, c, 0)};
        const int o4[4] = {g.o_g1, g.o_b1, g.o_g2, g.o_b2};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            grads[o4[q] + c] = v4[q];
            epi(grads + o4[q] + c, v4[q]);
        }
    }
    if (bn_batch && c >= 64 && c < 64 + 2 * g.N) {
        ast_bn_batch_body(g, cells, bn_batch, bn_weight, c - 64);
        if (bn_running) {       // nn.BatchNorm1d's running statistics (ast_bn_running_kernel's arithmetic on the values just written)
            const int blk = (c - 64) / g.N, ch = (c - 64) % g.N;
            const double count = (double)g.BG * g.T;
            const float mean = bn_batch[(blk * 2 + 0) * g.N + ch], var = bn_batch[(blk * 2 + 1) * g.N + ch];
            const float unbiased = count > 1.0 ? (float)(var * (count / (count - 1.0))) : var;
            float* rm = bn_running + (blk * 2 + 0) * g.N + ch;
            float* rv = bn_running + (blk * 2 + 1) * g.N + ch;
            *rm = (1.0f - bn_momentum) * *rm + bn_momentum * mean;
            *rv = (1.0f - bn_momentum) * *rv + bn_momentum * unbiased;
        }
    }
    if (loss) {             // strided partial sums, then a fixed-order tree (block_sum's arithmetic at 256 threads)
        if (threadIdx.x < AB) {
            float a = 0.f;
            for (int64_t i = threadIdx.x; i < g.B; i += AB) a += sqerr[i];
            red[threadIdx.x] = a;
        }
        __syncthreads();
        for (int m = AB / 2; m > 0; m >>= 1) {
            if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
            __syncthreads();
        }
        if (threadIdx.x == 0) loss[0] = red[0];
    }
}
__global__ __launch_bounds__(AB) void ast_finalize_kernel(AstGeom g, AstFin f) {
    __shared__ float red[AB];
    ast_finalize_body(g, f, red, NoEpilogue{});
}
// The end of a one-stream step as ONE launch: the slice sums of the five parameter-gradient products (reduce_slices_batch_body), the
// partial-row sums of both convolutions' weights and of the gate bias (rows_sum_job_body) and, in one more workgroup, the finalize body --
// three dependent launches of ~5 us each on a chain whose every launch counts.  In a whole step also the optimizer: every element of the
// gradient (13 parameter groups) is finalised here by exactly one thread, which applies Adam to that parameter on the spot (`ad`,
// adam_device.hpp: the arithmetic of adam_step_kernel) -- no optimizer launch.
__global__ __launch_bounds__(1024) void ast_tail_kernel(AstGeom g, AstFin f, RowsSumJobs jb, int nb0, int nb1, int nb2, ReduceBatch rb, AdamFuse ad) {
    __shared__ float red[32][33];
    __shared__ float part[16][64];
    __shared__ float fred[AB];
    const int nr = rb.first[rb.n];
    const int b = blockIdx.x;
    if (b < nr) reduce_slices_batch_body(rb, b, part, ad);
    else if (b - nr < nb0) rows_sum_job_body(jb, 0, b - nr, red, ad);
    else if (b - nr - nb0 < nb1) rows_sum_job_body(jb, 1, b - nr - nb0, red, ad);
    else if (b - nr - nb0 - nb1 < nb2) rows_sum_job_body(jb, 2, b - nr - nb0 - nb1, red, ad);
    else ast_finalize_body(g, f, fred, ad);
}

// Synchronised BatchNorm: the 16 replicas of one reduction pair (2 MAXN contiguous doubles) collapsed into replica 0, the
// others zeroed (the readers' replica sum is unchanged): the caller's all-reduce runs on one contiguous buffer.
__global__ void ast_cells_collapse_kernel(Cells* cells, int bwd, int blk) {
    for (int i = threadIdx.x; i < 2 * MAXN; i += blockDim.x) {
        double v = 0.0;
        for (int r = 0; r < CELL_REP; ++r) {
            double* p = bwd ? &cells[r].bwd[blk][0][0] : &cells[r].fwd[blk][0][0];
            v += p[i];
            if (r) p[i] = 0.0;
        }
        (bwd ? &cells[0].bwd[blk][0][0] : &cells[0].fwd[blk][0][0])[i] = v;
    }
}

// BatchNorm batch statistics out: (mean, biased var) per block/channel, or weight * (E[z], E[z^2]) for data parallel
__device__ __forceinline__ void ast_bn_batch_body(const AstGeom& g, const Cells* cells, float* __restrict__ bn_batch, float weight, int e) {
    const int blk = e / g.N, c = e % g.N;
    const double count = (double)g.BG * g.T;
    const double m = cell_sum(cells, &Cells::fwd, blk, c, 0) / count, q = cell_sum(cells, &Cells::fwd, blk, c, 1) / count;
    if (weight > 0.f) {
        bn_batch[(blk * 2 + 0) * g.N + c] = (float)(weight * m);
        bn_batch[(blk * 2 + 1) * g.N + c] = (float)(weight * q);
    } else {
        double v = q - m * m;
        bn_batch[(blk * 2 + 0) * g.N + c] = (float)m;
        bn_batch[(blk * 2 + 1) * g.N + c] = (float)(v < 0.0 ? 0.0 : v);
    }
}
__global__ void ast_bn_batch_kernel(AstGeom g, const Cells* cells, float* __restrict__ bn_batch, float weight) {
    if ((int)threadIdx.x < 2 * g.N) ast_bn_batch_body(g, cells, bn_batch, weight, threadIdx.x);
}

__global__ void ast_bn_running_kernel(float* __restrict__ bn, const float* __restrict__ batch, int N, double count, float momentum,
                                      int from_moments) {
    const int e = threadIdx.x;
    if (e >= 2 * N) return;
    const int blk = e / N, c = e % N;
    float mean = batch[(blk * 2 + 0) * N + c], var = batch[(blk * 2 + 1) * N + c];
    if (from_moments) {
        var = var - mean * mean;
        if (var < 0.f) var = 0.f;
    }
    const float unbiased = count > 1.0 ? (float)(var * (count / (count - 1.0))) : var;
    float* rm = bn + (blk * 2 + 0) * N + c;
    float* rv = bn + (blk * 2 + 1) * N + c;
    *rm = (1.0f - momentum) * *rm + momentum * mean;
    *rv = (1.0f - momentum) * *rv + momentum * unbiased;
}

constexpr int AST_BWD_ROWS = 4096;      // workgroups of the graph backward at most (one partial row of the gate's bias gradient each)
__global__ void ast_fill_one_kernel(float* p) { p[0] = 1.f; }
// head of a call with a forward: the reduction cells cleared and the constant 1 of the bias reductions, one launch
__global__ void ast_prepare_kernel(Cells* cells, float* one) {
    double* p = reinterpret_cast<double*>(cells);
    for (int i = threadIdx.x; i < (int)(sizeof(Cells) * CELL_REP / sizeof(double)); i += blockDim.x) p[i] = 0.0;
    if (threadIdx.x == 0) one[0] = 1.f;
}

struct AstWs {
    size_t cells, one, z1, out0, z2, zpre, out1, tcat, px, adj, dist, scat, pooled, dpred, sqerr, dmat, dpx, ds1, dy2, dy1, thb, gp1, gp2,
        split, total;
    size_t split_floats;
    int rows;
};
// output shapes and reduction lengths of the step's parameter-gradient products (fc.weight, fc.bias, filters, P.weight, theta.weight)
static void ast_pgrad_dims(const AstGeom& g, SplitKJob* j) {
    const int M = (int)(g.B * g.N), B = (int)g.B;
    const int mnk[5][3] = {{1, g.O, B}, {1, 1, B}, {g.KE, g.O, B}, {g.E, g.E, M}, {g.E, g.T, M}};
    for (int i = 0; i < 5; ++i) {
        j[i] = SplitKJob{};
        j[i].M = mnk[i][0]; j[i].N = mnk[i][1]; j[i].K = mnk[i][2] > 0 ? mnk[i][2] : 1;
    }
}

void ast_ws_layout(const AstGeom& g, AstWs* w) {
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t BNT = (size_t)g.B * g.N * g.T * sizeof(float);
    size_t o = 0;
    w->cells = o; o = al(o + sizeof(Cells) * CELL_REP);
    w->one = o; o = al(o + 256);
    w->z1 = o; o = al(o + BNT);
    w->out0 = o; o = al(o + BNT);
    w->z2 = o; o = al(o + BNT);
    w->zpre = o; o = al(o + BNT);
    w->out1 = o; o = al(o + BNT);
    w->tcat = o; o = al(o + BNT * g.K);
    w->px = o; o = al(o + BNT);
    w->adj = o; o = al(o + (size_t)g.B * g.N * g.N * sizeof(float));
    w->dist = o; o = al(o + (size_t)g.B * g.N * g.N * sizeof(float));
    w->scat = o; o = al(o + (size_t)g.B * g.KE * sizeof(float));
    w->pooled = o; o = al(o + (size_t)g.B * g.O * sizeof(float));
    w->dpred = o; o = al(o + (size_t)g.B * sizeof(float));
    w->sqerr = o; o = al(o + (size_t)g.B * sizeof(float));
    w->dmat = o; o = al(o + (size_t)g.B * g.O * sizeof(float));
    w->dpx = o; o = al(o + BNT);
    w->ds1 = o; o = al(o + BNT);
    w->dy2 = o; o = al(o + BNT);
    w->dy1 = o; o = al(o + BNT);
    w->rows = 1024;
    w->thb = o; o = al(o + (size_t)AST_BWD_ROWS * g.E * sizeof(float));
    w->gp1 = o; o = al(o + (size_t)w->rows * g.N * g.N * KT * sizeof(float));
    w->gp2 = o; o = al(o + (size_t)w->rows * g.N * g.N * KT * sizeof(float));
    const int mx = g.KE > g.E ? g.KE : g.E;
    // (... or the five parameter-gradient products of a step at once: ast_pgrad_jobs)
    SplitKJob dims[5];
    ast_pgrad_dims(g, dims);
    size_t sf = sgemm_splitk_partial_floats(mx, g.O > g.E ? g.O : g.E);
    const size_t bf = sgemm_splitk_batch_floats(dims, 5);
    sf = sf > bf ? sf : bf;
    w->split_floats = sf;
    w->split = o; o = al(o + sf * sizeof(float));
    w->total = o;
}

template <typename K>
int resident_rows(K kernel, int64_t items, int cap, int block = AB) {
    int dev = 0, cus = 256, per_cu = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    int64_t want = (int64_t)cus * per_cu;
    if (want > items) want = items;
    if (want > cap) want = cap;
    return want < 1 ? 1 : (int)want;
}

}  // namespace

int64_t astgcnn_param_count(const rulgnn_astgcnn_shape* s) {
    AstGeom g;
    return ast_geometry(s, &g) == RULGNN_OK ? g.nparam : -1;
}

size_t astgcnn_workspace_bytes(const rulgnn_astgcnn_shape* s) {
    AstGeom g;
    if (ast_geometry(s, &g) != RULGNN_OK) return 0;
    AstWs w;
    ast_ws_layout(g, &w);
    return w.total;
}

#define AST_RC(call)                  \
    do {                              \
        const int rc_ = (call);       \
        if (rc_ != RULGNN_OK) return rc_; \
    } while (0)

// mode bit 0: forward (training != 0: batch statistics), bit 1: backward
// <SN, SE, SO>: the kernels' instantiation -- the reference's two wirings (N-CMAPSS: 20 nodes, C-MAPSS: 14; 50 steps, 64 outputs) have their
// shapes as compile-time constants, anything else runs the generic <0, 0, 0>
template <int SN, int SE, int SO>
static int astgcnn_run_t(const rulgnn_astgcnn_shape* s, const rulgnn_astgcnn_args* a, int mode, hipStream_t st, const BnSyncHook* sync,
                         float* bn_running_out, float bn_momentum, AdamFuse* adam) {
    // threads of the TCN kernels: 20 nodes x 50 steps are 260 / 400 work items per sample -- one round of 448 threads (tcn_nodes.hpp)
    constexpr int TTB = AB;             // (the matrix-core convolution kernels: four wavefronts, a 16-step column tile each)
    AstGeom g;
    AST_RC(ast_geometry(s, &g));
    if (sync) {          // both BatchNorm layers normalise by the statistics of the GLOBAL batch (cells all-reduced between the kernels)
        if (mode != 3 || !a->training || a->global_batch < g.B || g.B < 1 || a->bn_moment_weight > 0.f) return RULGNN_EINVAL;
        g.BG = a->global_batch;
    }
    AstWs w;
    ast_ws_layout(g, &w);
    if (a->workspace_bytes < w.total) return RULGNN_EWORKSPACE;
    char* ws = static_cast<char*>(a->workspace);
    auto F = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    Cells* cells = reinterpret_cast<Cells*>(ws + w.cells);
    const float* prm = a->params;
    const int training = a->training ? 1 : 0;
    const int N = g.N, T = g.T, E = g.E, O = g.O, KE = g.KE;
    const int M = (int)(g.B * N);
    const float inv_gb = 1.0f / (float)(a->global_batch > 0 ? a->global_batch : g.B);
    (void)hipGetLastError();
    auto sync_pair = [&](int bwd, int blk) -> int {
        if (!sync) return RULGNN_OK;
        hipLaunchKernelGGL(ast_cells_collapse_kernel, dim3(1), dim3(64), 0, st, cells, bwd, blk);
        if (hipGetLastError() != hipSuccess) return RULGNN_EHIP;
        double* buf = bwd ? &cells[0].bwd[blk][0][0] : &cells[0].fwd[blk][0][0];
        return sync->fn(sync->user, buf, 2 * MAXN, st) == 0 ? RULGNN_OK : RULGNN_ECALLBACK;
    };
    // The parameter-gradient products of the backward feed nothing in this call: with a second stream of the caller (args->aux_stream,
    // aux_stream.hpp) they run beside the TCN backward, forked behind the graph-backward kernel's own completion signal.
    AuxFork fk(st, (mode & 2) ? a->aux_stream : nullptr);
    if (mode & 1) {
        hipLaunchKernelGGL(ast_prepare_kernel, dim3(1), dim3(1024), 0, st, cells, F(w.one));
        const int rows = resident_rows((tcn_conv_kernel<1, AstGeom, SN, SE, TTB>), g.B, 1 << 20, TTB);
        hipLaunchKernelGGL((tcn_conv_kernel<1, AstGeom, SN, SE, TTB>), dim3(rows), dim3(TTB), 0, st, g, a->x, prm, a->bn_stats, training, (const float*)nullptr,
                           F(w.z1), (float*)nullptr, cells);
        AST_RC(sync_pair(0, 0));
        hipLaunchKernelGGL((tcn_conv_kernel<2, AstGeom, SN, SE, TTB>), dim3(rows), dim3(TTB), 0, st, g, a->x, prm, a->bn_stats, training, (const float*)F(w.z1),
                           F(w.z2), F(w.out0), cells);
        AST_RC(sync_pair(0, 1));
        // gate, P projection, graph, Chebyshev terms, node sums, the filter product and the head: one launch (ast_front_kernel)
        hipLaunchKernelGGL((ast_front_kernel<SN, SE, SO>), dim3(resident_rows((ast_front_kernel<SN, SE, SO>), g.B, 1 << 20)), dim3(AB), 0, st, g, a->x, prm, a->bn_stats, training,
                           (const Cells*)cells, (const float*)F(w.z2), (const float*)F(w.out0), F(w.zpre), F(w.out1), F(w.tcat), F(w.px), F(w.adj),
                           F(w.dist), F(w.scat), F(w.pooled), a->y, a->pred, F(w.dpred), F(w.sqerr), F(w.dmat), inv_gb);
        if (training && a->bn_batch && !(mode & 2))            // (with a backward in the same call: beside its chain, below)
            hipLaunchKernelGGL(ast_bn_batch_kernel, dim3(1), dim3(64), 0, st, g, (const Cells*)cells, a->bn_batch, a->bn_moment_weight);
    }
    if (mode & 2) {
        if (a->dpred)          // external d loss / d pred (autograd): D = dpred fc.weight / N
            hipLaunchKernelGGL(ast_head_kernel, dim3((unsigned)((g.B + 3) / 4 > 2048 ? 2048 : (g.B + 3) / 4)), dim3(AB), 0, st, g, prm,
                               F(w.pooled), (const float*)nullptr, a->dpred, a->pred, F(w.dpred), F(w.sqerr), F(w.dmat), inv_gb, 1);
        float* gr = a->grads;
        float* split = F(w.split);
        hipStream_t wst = fk.side();
        const bool mse = a->dpred == nullptr;
        // the constant 1 the bias reduction reads: written IN FRONT of the fork (behind it the side stream's product over `one` was
        // ordered against nothing that wrote it -- garbage from a fresh workspace on the first step)
        // (with a forward in the same call ast_prepare_kernel wrote it; batch statistics and the loss sum ride in the finalize kernel)
        if (!(mode & 1)) hipLaunchKernelGGL(ast_fill_one_kernel, dim3(1), dim3(1), 0, st, F(w.one));
        // DT = D Fcat^T, the graph backward, dG = dG_cheb + dPX P and the gate backward (BatchNorm-2 sums): one launch
        const int bwd_rows = resident_rows((ast_graph_bwd_kernel<SN, SE, SO>), g.B, AST_BWD_ROWS);
        hipEvent_t bwd_done = fk.stop_event();                       // (the fork point: behind this kernel)
        RULGNN_LAUNCH_EV(bwd_done, (ast_graph_bwd_kernel<SN, SE, SO>), dim3(bwd_rows), dim3(AB), 0, st, g, prm,
                         (const float*)F(w.px), (const float*)F(w.tcat), (const float*)F(w.adj), (const float*)F(w.dist), (const float*)F(w.dmat),
                         F(w.dpx), cells, (const float*)F(w.z2), (const float*)F(w.out1), F(w.zpre), F(w.ds1), F(w.dy2), F(w.thb));
        // The five parameter-gradient products -- d fc.weight[o] = sum_b dpred[b] pooled[b][o], d fc.bias = sum_b dpred[b],
        // d filters = Scat^T D, d P = dPX^T G, d theta.weight = dZpre^T x -- as ONE split-K launch + ONE reduction (sgemm_splitk_batch).
        // As nine launches (5-8 us each, at their latency floor) they needed a side stream from the front kernel on -- two forks and a join,
        // each a 5-8 us bubble in the main queue, and 0.10 ms of host time per step to enqueue it all.  One stream, the pair behind the
        // backward chain: 0.119 ms per step; the pair on a side stream beside the TCN backward: 0.122; the nine launches: 0.130.
        SplitKJob jobs[5];
        ast_pgrad_dims(g, jobs);
        jobs[0].A = F(w.dpred); jobs[0].sAm = 0; jobs[0].sAk = 1; jobs[0].B = F(w.pooled); jobs[0].sBn = 1; jobs[0].sBk = O; jobs[0].C = gr + g.o_fcw; jobs[0].ldc = O;
        jobs[1].A = F(w.dpred); jobs[1].sAm = 0; jobs[1].sAk = 1; jobs[1].B = F(w.one); jobs[1].sBn = 0; jobs[1].sBk = 0; jobs[1].C = gr + g.o_fcb; jobs[1].ldc = 1;
        jobs[2].A = F(w.scat); jobs[2].sAm = 1; jobs[2].sAk = KE; jobs[2].B = F(w.dmat); jobs[2].sBn = 1; jobs[2].sBk = O; jobs[2].C = gr + g.o_f; jobs[2].ldc = O;
        jobs[3].A = F(w.dpx); jobs[3].sAm = 1; jobs[3].sAk = E; jobs[3].B = F(w.tcat); jobs[3].sBn = 1; jobs[3].sBk = KE; jobs[3].C = gr + g.o_pw; jobs[3].ldc = E;
        jobs[4].A = F(w.zpre); jobs[4].sAm = 1; jobs[4].sAk = E; jobs[4].B = a->x; jobs[4].sBn = 1; jobs[4].sBk = T; jobs[4].C = gr + g.o_thw; jobs[4].ldc = T;
        if (fk.active()) {
            fk.fork_after(bwd_done);
            AST_RC(sgemm_splitk_batch(jobs, 5, split, w.split_floats, wst));
        }
        const int rows = resident_rows((tcn_conv_bwd_kernel<2, AstGeom, SN, SE, TTB>), g.B, w.rows, TTB);
        AST_RC(sync_pair(1, 1));
        hipLaunchKernelGGL((tcn_conv_bwd_kernel<2, AstGeom, SN, SE, TTB>), dim3(rows), dim3(TTB), 0, st, g, prm, cells, (const float*)F(w.z2), (const float*)F(w.dy2),
                           (const float*)F(w.out0), (const float*)F(w.ds1), (const float*)F(w.z1), F(w.dy1), F(w.gp2));
        AST_RC(sync_pair(1, 0));
        hipLaunchKernelGGL((tcn_conv_bwd_kernel<1, AstGeom, SN, SE, TTB>), dim3(rows), dim3(TTB), 0, st, g, prm, cells, (const float*)F(w.z1), (const float*)F(w.dy1),
                           a->x, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, F(w.gp1));
        AstFin fin;
        fin.cells = cells; fin.grads = gr; fin.bn_scale = sync ? sync->bn_param_grad_scale : 1.0f;
        fin.bn_batch = ((mode & 1) && training) ? a->bn_batch : (float*)nullptr;
        fin.bn_weight = a->bn_moment_weight; fin.sqerr = F(w.sqerr);
        fin.loss = (mse && a->loss) ? a->loss : (float*)nullptr;
        fin.bn_running = ((mode & 1) && training && a->bn_batch && a->bn_moment_weight == 0.f) ? bn_running_out : (float*)nullptr;
        fin.bn_momentum = bn_momentum;
        if (!fk.active()) {
            // one stream: the products' slice sums, both convolutions' partial weight rows, the gate's bias gradient (the graph backward's
            // partial rows; theta.bias and gate.bias share it) and the finalize body in ONE launch behind the product launch
            ReduceBatch rb;
            AST_RC(sgemm_splitk_batch_products(jobs, 5, split, w.split_floats, st, &rb));
            RowsSumJobs jb{};
            jb.part[0] = F(w.gp1); jb.out[0] = gr + g.o_w1; jb.rows[0] = rows; jb.n[0] = N * N * KT; jb.ld[0] = (int64_t)N * N * KT;
            jb.part[1] = F(w.gp2); jb.out[1] = gr + g.o_w2; jb.rows[1] = rows; jb.n[1] = N * N * KT; jb.ld[1] = (int64_t)N * N * KT;
            jb.part[2] = F(w.thb); jb.out[2] = gr + g.o_thb; jb.out2[2] = gr + g.o_gb; jb.rows[2] = bwd_rows; jb.n[2] = E; jb.ld[2] = E;
            const int nb01 = (N * N * KT + 31) / 32, nb2 = (E + 31) / 32;
            AdamFuse ad{};
            if (adam && mode == 3 && !sync) {             // (a whole step: applied here; the caller launches no optimizer kernel)
                ad = *adam;
                ad.gbase = gr;
                adam->gbase = gr;
            }
            hipLaunchKernelGGL(ast_tail_kernel, dim3(rb.first[rb.n] + 2 * nb01 + nb2 + 1), dim3(1024), 0, st, g, fin, jb, nb01, nb01, nb2, rb, ad);
            return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
        }
        // both convolutions' partial weight rows in one launch (the second used to sit between the two backward kernels)
        // ... and the gate's bias gradient (the graph backward's partial rows; theta.bias and gate.bias share it)
        AST_RC(rows_sum3(F(w.gp1), gr + g.o_w1, F(w.gp2), gr + g.o_w2, rows, (int64_t)N * N * KT, N * N * KT, F(w.thb), gr + g.o_thb, gr + g.o_gb,
                         bwd_rows, (int64_t)E, E, st));
        // (the finalize kernel reads the cells and the squared errors only -- nothing the side stream writes: it runs in front of the join,
        // beside the side stream's last product, instead of behind the wake-up of a stream that sat waiting)
        hipLaunchKernelGGL(ast_finalize_kernel, dim3(1), dim3(AB), 0, st, g, fin);
        AST_RC(fk.join());
    }
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

int astgcnn_run(const rulgnn_astgcnn_shape* s, const rulgnn_astgcnn_args* a, int mode, hipStream_t st, const BnSyncHook* sync, float* bn_running_out,
                float bn_momentum, AdamFuse* adam) {
    if (adam) adam->gbase = nullptr;          // (set by the path that applies the update: see stgcn_host.hpp)
    if (s && s->time_length == 50 && s->output_dim == 64) {
        if (s->num_nodes == 20) return astgcnn_run_t<20, 50, 64>(s, a, mode, st, sync, bn_running_out, bn_momentum, adam);
        if (s->num_nodes == 14) return astgcnn_run_t<14, 50, 64>(s, a, mode, st, sync, bn_running_out, bn_momentum, adam);
    }
    return astgcnn_run_t<0, 0, 0>(s, a, mode, st, sync, bn_running_out, bn_momentum, adam);
}

int astgcnn_bn_running_update(const rulgnn_astgcnn_shape* s, float* bn_stats, const float* bn_batch, int64_t count, float momentum,
                              int from_moments, hipStream_t st) {
    AstGeom g;
    AST_RC(ast_geometry(s, &g));
    (void)hipGetLastError();
    hipLaunchKernelGGL(ast_bn_running_kernel, dim3(1), dim3(64), 0, st, bn_stats, bn_batch, g.N, (double)count, momentum, from_moments);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

}  // namespace rulgnn
