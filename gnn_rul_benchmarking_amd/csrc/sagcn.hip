// SAGCN on gfx950.
// Reference path replaced: SAGCN_model.forward -- models/SAGCN/Model.py:127-156 (features :6-70, cosine_distance :73-79, GCNLayer
// :81-96, GraphProjectionLayer :99-112, SelfAttentionLayer :115-125) -- and SAGCN.update, algorithms/algorithms.py:427-436.
//
//   x [bs, P * n] -> per patch 12 temporal + 8 spectral statistics, their cumulative form along the patches, the whole [P, 40]
//   block scaled to unit Frobenius norm -> cosine adjacency, D^-1/2 (A + I) D^-1/2 -> gcn1 (40 -> H, ReLU) -> two projection layers
//   (Linear over the NODE axis, then Linear over the feature axis, ReLU) -> attention over the nodes (tanh layer P -> Ah, softmax
//   layer Ah -> P, per feature channel) -> x * attention -> Linear(H * P -> 1).
//
// Everything up to A_hat X depends on the input alone (no parameter in front of it): two forward-only kernels.  One workgroup per
// patch computes the statistics -- the spectrum as a direct DFT over the half spectrum with an exact table of twiddles (patch
// lengths 16 / 20 / 1024 in the reference's rows: 20 is not a power of two), mirrored like torch.fft.fft mirrors a real input --;
// one workgroup per sample builds the 40 columns, the adjacency and A_hat X without ever storing the [P, P] matrix.
// The trained path keeps its activations NODE-MAJOR, [P][bs][H]: the node-axis Linear layers are then single GEMMs
// [P, P] x [P, bs * H] on the matrix cores, the feature-axis ones [P * bs, H] x [H, H], the attention two more, and every weight
// gradient a split-K GEMM with a fixed summation order.
//
// The two index-valued statistics sit on exact ties of the mirrored spectrum: the bin of the largest amplitude is the FIRST one
// (torch.argmax's documented rule); the bin at the median rank of the sorted power is taken in STABLE order (equal powers keep
// their bin order), which is what the reference's CPU sort does for patches of up to 16 points; for longer patches its order among
// equal keys is unspecified (torch.argsort, stable=False) and differs between its own CPU and GPU sorts.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sgemm_mfma.hpp"
#include "stgcn_host.hpp"

namespace rulgnn {

namespace {

constexpr int GB = 256;
constexpr int SG_F = 40, SG_RAW = 20;
constexpr int SG_MAXP = 256, SG_MAXN = 2048, SG_MAXH = 4096;

struct SgGeom {
    int64_t B, R, BH;                           // samples, node rows = P * B, columns of the node-major matrices = B * H
    int P, n, H, Ah, nh;                        // nh = n / 2 + 1 bins of the half spectrum
    int o_w1, o_b1, o_wl[2], o_bl[2], o_wp[2], o_bp[2], o_wt, o_bt, o_ws, o_bs, o_wfc, o_bfc, pcount;
    int64_t w_raw, w_feat, w_ax, w_h1, w_u[2], w_h[2], w_s, w_attn, w_dpred, w_sq, w_one, w_dA, w_dB, w_ds, w_split, w_rowpart, w_headpart, total;
};

int sg_geometry(const rulgnn_sagcn_shape* s, SgGeom* g) {
    if (!s) return RULGNN_EINVAL;
    if (s->batch < 0 || s->num_patch < 1 || s->patch_size < 2 || s->gcn_hidden_dim < 1 || s->attention_hidden_dim < 1) return RULGNN_EINVAL;
    if (s->num_patch > SG_MAXP || s->patch_size > SG_MAXN || s->gcn_hidden_dim > SG_MAXH || s->attention_hidden_dim > SG_MAXH)
        return RULGNN_EUNSUPPORTED;
    if (s->batch * (int64_t)s->gcn_hidden_dim * s->num_patch >= (int64_t)1 << 31) return RULGNN_EUNSUPPORTED;
    g->B = s->batch; g->P = s->num_patch; g->n = s->patch_size; g->H = s->gcn_hidden_dim; g->Ah = s->attention_hidden_dim;
    g->nh = g->n / 2 + 1;
    g->R = g->B * g->P;
    g->BH = g->B * g->H;
    const int P = g->P, H = g->H, Ah = g->Ah;
    int o = 0;
    g->o_w1 = o; o += H * SG_F;
    g->o_b1 = o; o += H;
    for (int i = 0; i < 2; ++i) {
        g->o_wl[i] = o; o += H * H;
        g->o_bl[i] = o; o += H;
        g->o_wp[i] = o; o += P * P;
        g->o_bp[i] = o; o += P;
    }
    g->o_wt = o; o += Ah * P;
    g->o_bt = o; o += Ah;
    g->o_ws = o; o += P * Ah;
    g->o_bs = o; o += P;
    g->o_wfc = o; o += H * P;
    g->o_bfc = o; o += 1;
    g->pcount = o;
    int64_t w = 0;
    auto take = [&w](int64_t nfl) { const int64_t at = w; w += (nfl + 63) & ~(int64_t)63; return at; };
    const int64_t RH = g->R * H;
    g->w_raw = take(g->R * SG_RAW);
    g->w_feat = take(g->R * SG_F);
    g->w_ax = take(g->R * SG_F);
    g->w_h1 = take(RH);
    for (int i = 0; i < 2; ++i) { g->w_u[i] = take(RH); g->w_h[i] = take(RH); }
    g->w_s = take((int64_t)Ah * g->BH);
    g->w_attn = take(RH);
    g->w_dpred = take(g->B);
    g->w_sq = take(g->B);
    g->w_one = take(64);
    g->w_dA = take(RH);
    g->w_dB = take(RH);
    g->w_ds = take((int64_t)Ah * g->BH);
    size_t sp = 1024;
    if (g->B > 0) {
        const int R = (int)g->R, BH = (int)g->BH;
        const size_t need[] = {sgemm_splitk_need_floats(P, Ah, BH), sgemm_splitk_need_floats(Ah, P, BH), sgemm_splitk_need_floats(H, H, R),
                               sgemm_splitk_need_floats(1, H, R), sgemm_splitk_need_floats(P, P, BH), sgemm_splitk_need_floats(H, SG_F, R),
                               sgemm_splitk_need_floats(1, 1, (int)g->B)};
        for (size_t v : need) sp = v > sp ? v : sp;
    }
    g->w_split = take((int64_t)sp);
    g->w_rowpart = take((int64_t)SG_MAXH * 16);
    g->w_headpart = take(g->B * ((H + 63) / 64));
    g->total = w;
    return RULGNN_OK;
}

// ---- block reductions over GB threads, K values at once, fixed order ------------------------------------------------------------
template <int K, int TBS = GB>
__device__ __forceinline__ void block_sum(float (&v)[K], float* red) {          // red: K * TBS floats
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) red[k * TBS + threadIdx.x] = v[k];
    __syncthreads();
    for (int m = TBS / 2; m > 0; m >>= 1) {
        if ((int)threadIdx.x < m) {
#pragma unroll
            for (int k = 0; k < K; ++k) red[k * TBS + threadIdx.x] += red[k * TBS + threadIdx.x + m];
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = red[k * TBS];
    __syncthreads();
}

// ---- kernel 1: the 20 statistics of every patch (Model.py:17-52) -----------------------------------------------------------------
// dynamic LDS: s[n] | cos[n] | sin[n] | power of the half spectrum [nh] | red[7 * GB]
// (TBS: threads per workgroup -- a patch is n <= 64 values at the reference's wirings (20): one wavefront per patch keeps four times the patches
// in flight per CU and its barriers cost nothing; 256 threads per patch left 236 of them idle in every loop)
template <int TBS>
__global__ __launch_bounds__(TBS) void sg_patch_features_kernel(SgGeom g, const float* __restrict__ x, float* __restrict__ raw) {
    extern __shared__ float lds[];
    const int n = g.n, nh = g.nh, tid = threadIdx.x;
    float* s = lds;
    float* tc = s + n;
    float* ts = tc + n;
    float* pw = ts + n;
    float* red = pw + nh;
    int* ired = reinterpret_cast<int*>(red + 6 * TBS);
    for (int j = tid; j < n; j += TBS) {
        const double a = 2.0 * (double)j / (double)n;
        tc[j] = (float)cospi(a);
        ts[j] = (float)sinpi(a);
    }
    const float fn = (float)n, inv_n = 1.0f / fn;
    for (int64_t patch = blockIdx.x; patch < g.R; patch += gridDim.x) {
        __syncthreads();
        const float* xp = x + patch * n;
        for (int j = tid; j < n; j += TBS) s[j] = xp[j];
        __syncthreads();
        // pass A: extrema and first moments
        float mx = -INFINITY, mn = INFINITY;
        float a4[4] = {0.f, 0.f, 0.f, 0.f};                          // sum s, sum s^2, sum asin, sum atan
        for (int j = tid; j < n; j += TBS) {
            const float v = s[j];
            mx = fmaxf(mx, v); mn = fminf(mn, v);
            a4[0] += v; a4[1] = fmaf(v, v, a4[1]);
            a4[2] += asinf(fminf(fmaxf(v, -1.f + 1e-7f), 1.f - 1e-7f));
            a4[3] += atanf(v);
        }
        red[tid] = mx; red[TBS + tid] = mn;
        __syncthreads();
        for (int m = TBS / 2; m > 0; m >>= 1) {
            if (tid < m) { red[tid] = fmaxf(red[tid], red[tid + m]); red[TBS + tid] = fminf(red[TBS + tid], red[TBS + tid + m]); }
            __syncthreads();
        }
        mx = red[0]; mn = red[TBS];
        block_sum<4, TBS>(a4, red);
        const float mean = a4[0] * inv_n, ma = a4[2] * inv_n, mt = a4[3] * inv_n;
        // pass B: central moments, entropy of softmax(s), spread of asin / atan
        float b7[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int j = tid; j < n; j += TBS) {
            const float v = s[j], d = v - mean, d2 = d * d, z = v - mx, e = expf(z);
            b7[0] += d2; b7[1] = fmaf(d2, d, b7[1]); b7[2] = fmaf(d2, d2, b7[2]);
            b7[3] += e; b7[4] = fmaf(e, z, b7[4]);
            const float da = asinf(fminf(fmaxf(v, -1.f + 1e-7f), 1.f - 1e-7f)) - ma, dt = atanf(v) - mt;
            b7[5] = fmaf(da, da, b7[5]); b7[6] = fmaf(dt, dt, b7[6]);
        }
        {
            float lo[4] = {b7[0], b7[1], b7[2], b7[3]};
            block_sum<4, TBS>(lo, red);
            float hi[3] = {b7[4], b7[5], b7[6]};
            block_sum<3, TBS>(hi, red);
            b7[0] = lo[0]; b7[1] = lo[1]; b7[2] = lo[2]; b7[3] = lo[3]; b7[4] = hi[0]; b7[5] = hi[1]; b7[6] = hi[2];
        }
        const float var = b7[0] / (fn - 1.f), sd = sqrtf(var);
        // half spectrum by direct DFT: F[k] = sum_t s[t] (cos - i sin)(2 pi k t / n)
        float amax = -1.f;
        int kmax = 0;
        float c3[3] = {0.f, 0.f, 0.f};                               // sum of the mirrored power, of its square, and the Nyquist bin's power
        __syncthreads();
        for (int k = tid; k < nh; k += TBS) {
            float re = 0.f, im = 0.f;
            int idx = 0;
            for (int t = 0; t < n; ++t) {
                const float v = s[t];
                re = fmaf(v, tc[idx], re);
                im = fmaf(-v, ts[idx], im);
                idx += k;
                if (idx >= n) idx -= n;
            }
            const float amp = sqrtf(re * re + im * im);
            const float p = amp * amp * inv_n;
            pw[k] = p;
            const bool single = k == 0 || 2 * k == n;                // DC and Nyquist are their own mirror image
            const float mult = single ? 1.f : 2.f;
            c3[0] = fmaf(mult, p, c3[0]);
            c3[1] = fmaf(mult * p, p, c3[1]);
            if (2 * k == n) c3[2] = p;
            if (amp > amax) { amax = amp; kmax = k; }
        }
        block_sum<3, TBS>(c3, red);
        red[tid] = amax; ired[tid] = kmax;
        __syncthreads();
        for (int m = TBS / 2; m > 0; m >>= 1) {
            if (tid < m) {
                const float o = red[tid + m];
                const int ko = ired[tid + m];
                if (o > red[tid] || (o == red[tid] && ko < ired[tid])) { red[tid] = o; ired[tid] = ko; }
            }
            __syncthreads();
        }
        amax = red[0]; kmax = ired[0];
        __syncthreads();
        // the bin at rank n / 2 of the stably sorted mirrored power
        int median_bin = -1;
        for (int k = tid; k < n; k += TBS) {
            const float p = pw[k < nh ? k : n - k];
            int rank = 0;
            for (int j = 0; j < n; ++j) {
                const float q = pw[j < nh ? j : n - j];
                rank += (q < p || (q == p && j < k)) ? 1 : 0;
            }
            if (rank == n / 2) median_bin = k;
        }
        ired[tid] = median_bin;
        __syncthreads();
        for (int m = TBS / 2; m > 0; m >>= 1) {
            if (tid < m) ired[tid] = max(ired[tid], ired[tid + m]);
            __syncthreads();
        }
        median_bin = ired[0];
        if (tid == 0) {
            auto freq = [n, inv_n](int k) { return (k <= (n - 1) / 2 ? (float)k : (float)(k - n)) * inv_n; };      // torch.fft.fftfreq
            float* o = raw + patch * SG_RAW;
            const float sd2 = var, tot = c3[0];
            o[0] = mx; o[1] = mn; o[2] = sd; o[3] = sqrtf(a4[1] * inv_n); o[4] = mean; o[5] = mx - mn; o[6] = var;
            o[7] = logf(b7[3]) - b7[4] / b7[3];
            o[8] = sqrtf(b7[5] / (fn - 1.f)); o[9] = sqrtf(b7[6] / (fn - 1.f));
            o[10] = (b7[2] * inv_n) / (sd2 * sd2) - 3.f;
            o[11] = (b7[1] * inv_n) / (sd2 * sd);
            // mirrored bins cancel in sum(freq * power); an even length keeps its Nyquist bin at -1/2
            o[12] = (n % 2 == 0 ? -0.5f * c3[2] : 0.f) / tot;
            o[13] = median_bin >= 0 ? freq(median_bin) : NAN;
            o[14] = tot;
            o[15] = tot / tot;                                        // every fftfreq bin is below fs / 2
            o[16] = sqrtf(c3[1] / tot);
            o[17] = amax * amax * inv_n; o[18] = amax; o[19] = freq(kmax);
        }
    }
}

// ---- kernel 2: one workgroup per sample: cumulative columns, unit norm, cosine adjacency, A_hat X (Model.py:6-14, 66-79, 86-91) --
// dynamic LDS: F[P][41] | nrm[P] | dinv[P] | red[GB]
__global__ __launch_bounds__(GB) void sg_graph_kernel(SgGeom g, const float* __restrict__ raw, float* __restrict__ feat,
                                                      float* __restrict__ ax) {
    extern __shared__ float lds[];
    constexpr int LD = SG_F + 1;
    const int P = g.P, tid = threadIdx.x;
    float* F = lds;
    float* nrm = F + P * LD;
    float* dinv = nrm + P;
    float* red = dinv + P;
    for (int64_t b = blockIdx.x; b < g.B; b += gridDim.x) {
        __syncthreads();
        if (tid < SG_RAW) {
            double c = 0.0;                                           // c / sqrt|c| amplifies the running sum's rounding near its zero crossings
            for (int p = 0; p < P; ++p) {
                const float v = raw[(b * P + p) * SG_RAW + tid];
                c += (double)v;
                F[p * LD + tid] = v;
                F[p * LD + SG_RAW + tid] = (float)(c / sqrt(fmax(fabs(c), 1e-12)));
            }
        }
        __syncthreads();
        float q[1] = {0.f};
        for (int i = tid; i < P * SG_F; i += GB) { const float v = F[(i / SG_F) * LD + i % SG_F]; q[0] = fmaf(v, v, q[0]); }
        block_sum<1>(q, red);
        const float inv = 1.0f / sqrtf(q[0]);
        for (int i = tid; i < P * SG_F; i += GB) {
            const float v = F[(i / SG_F) * LD + i % SG_F] * inv;
            F[(i / SG_F) * LD + i % SG_F] = v;
            feat[b * P * SG_F + i] = v;
        }
        __syncthreads();
        for (int p = tid; p < P; p += GB) {
            float a = 0.f;
            for (int f = 0; f < SG_F; ++f) a = fmaf(F[p * LD + f], F[p * LD + f], a);
            nrm[p] = sqrtf(a);
        }
        __syncthreads();
        for (int p = tid; p < P; p += GB) {
            float rs = 1.f;                                           // the self loop
            for (int o = 0; o < P; ++o) {
                float d = 0.f;
                for (int f = 0; f < SG_F; ++f) d = fmaf(F[p * LD + f], F[o * LD + f], d);
                rs += d / (nrm[p] * nrm[o]);
            }
            dinv[p] = 1.0f / sqrtf(rs);                               // a non-positive degree gives NaN, like ** -0.5
        }
        __syncthreads();
        for (int p = tid; p < P; p += GB) {
            float acc[SG_F];
#pragma unroll
            for (int f = 0; f < SG_F; ++f) acc[f] = 0.f;
            for (int o = 0; o < P; ++o) {
                float d = 0.f;
#pragma unroll
                for (int f = 0; f < SG_F; ++f) d = fmaf(F[p * LD + f], F[o * LD + f], d);
                const float w = (d / (nrm[p] * nrm[o]) + (o == p ? 1.f : 0.f)) * dinv[p] * dinv[o];
#pragma unroll
                for (int f = 0; f < SG_F; ++f) acc[f] = fmaf(w, F[o * LD + f], acc[f]);
            }
            float* out = ax + ((int64_t)p * g.B + b) * SG_F;
#pragma unroll
            for (int f = 0; f < SG_F; ++f) out[f] = acc[f];
        }
    }
}

// v[r][c] = act(v[r][c] + bias[c])     act: 0 none, 1 ReLU
// The same kernel with its two [P x P] passes on the fp32 matrix cores (P a multiple of 16, P <= 128: the reference's 128-patch wirings).
// Above, 128 of the 256 threads each walked 128 x 40 multiply-adds for the degree and 128 x 80 for the aggregation, recomputing the cosine
// matrix in between (101 us for 100 samples).  Here C = Fh Fh^T (Fh: rows scaled to unit length) is formed once, as 16 x 16 tiles of
// v_mfma_f32_16x16x4f32, and kept in LDS; the aggregation A_hat F is a second product whose A operand is formed from C and the degrees on the
// fly.  Lane (kq, li) = (lane / 16, lane % 16) feeds A[m = li][k = 4 s + kq] and B[k = 4 s + kq][n = li] and receives C[m = 4 kq + r][n = li].
// LDS: F[P][41] | Fh[P][41] (+ 8 floats so that the last row's columns 40..47 exist) | nrm[P] | dinv[P] | red[GB] | C[P][P + 1]
__device__ __forceinline__ f32x4t sg_mfma(float a, float b, f32x4t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
inline size_t sg_graph_mx_lds(int P) {
    const size_t ctile = (size_t)P * (P + 1), csums = (size_t)2 * P * SG_RAW;      // the C tile also holds the fp64 running sums [P][SG_RAW] first
    return sizeof(float) * ((size_t)2 * P * (SG_F + 1) + 8 + 2 * P + GB + (ctile > csums ? ctile : csums));
}

__global__ __launch_bounds__(GB) void sg_graph_mx_kernel(SgGeom g, const float* __restrict__ raw, float* __restrict__ feat,
                                                         float* __restrict__ ax) {
    extern __shared__ float lds[];
    constexpr int LD = SG_F + 1;
    const int P = g.P, PT = P / 16, CP = P + 1, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    float* F = lds;
    float* Fh = F + P * LD;
    float* nrm = Fh + P * LD + 8;
    float* dinv = nrm + P;
    float* red = dinv + P;
    float* Cm = red + GB;
    for (int64_t b = blockIdx.x; b < g.B; b += gridDim.x) {
        __syncthreads();
        // the running sums stay a sequential fp64 chain per feature (c / sqrt|c| amplifies their rounding near the zero crossings), but only the
        // additions: the square roots and divisions -- 128 dependent fp64 pairs on 20 threads before -- are taken by all threads afterwards
        double* cs = reinterpret_cast<double*>(Cm);                   // [P][SG_RAW], over the (not yet used) C tile; 8-byte aligned: every term above is even
        if (tid < SG_RAW) {
            double c = 0.0;
            for (int p = 0; p < P; ++p) {
                const float v = raw[(b * P + p) * SG_RAW + tid];
                c += (double)v;
                F[p * LD + tid] = v;
                cs[p * SG_RAW + tid] = c;
            }
        }
        __syncthreads();
        for (int i = tid; i < P * SG_RAW; i += GB) {
            const double c = cs[i];
            F[(i / SG_RAW) * LD + SG_RAW + i % SG_RAW] = (float)(c / sqrt(fmax(fabs(c), 1e-12)));
        }
        __syncthreads();
        float q[1] = {0.f};
        for (int i = tid; i < P * SG_F; i += GB) { const float v = F[(i / SG_F) * LD + i % SG_F]; q[0] = fmaf(v, v, q[0]); }
        block_sum<1>(q, red);
        const float inv = 1.0f / sqrtf(q[0]);
        for (int i = tid; i < P * SG_F; i += GB) {
            const float v = F[(i / SG_F) * LD + i % SG_F] * inv;
            F[(i / SG_F) * LD + i % SG_F] = v;
            feat[b * P * SG_F + i] = v;
        }
        __syncthreads();
        for (int p = tid; p < P; p += GB) {
            float a = 0.f;
            for (int f = 0; f < SG_F; ++f) a = fmaf(F[p * LD + f], F[p * LD + f], a);
            nrm[p] = sqrtf(a);
        }
        __syncthreads();
        for (int i = tid; i < P * LD + 8; i += GB) {
            const int p = i / LD, f = i - p * LD;
            Fh[i] = p < P && f < SG_F ? F[p * LD + f] / nrm[p] : 0.f;
        }
        __syncthreads();
        // C = Fh Fh^T: row tiles wave, wave + 4 of this wavefront, every column tile
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = wave + 4 * ii;
            if (i < PT) {
                f32x4t acc[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = (f32x4t){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
                for (int s4 = 0; s4 < SG_F / 4; ++s4) {
                    const float av = Fh[(16 * i + li) * LD + 4 * s4 + kq];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (j < PT) acc[j] = sg_mfma(av, Fh[(16 * j + li) * LD + 4 * s4 + kq], acc[j]);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j < PT) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) Cm[(16 * i + 4 * kq + r) * CP + 16 * j + li] = acc[j][r];
                    }
            }
        }
        __syncthreads();
        for (int p = tid; p < P; p += GB) {
            float rs = 1.f;                                           // the self loop
            for (int o = 0; o < P; ++o) rs += Cm[p * CP + o];
            dinv[p] = 1.0f / sqrtf(rs);                               // a non-positive degree gives NaN, like ** -0.5
        }
        __syncthreads();
        // AX = A_hat F, A_hat[p][o] = (C[p][o] + [p == o]) dinv[p] dinv[o]; feature columns 40..47 of the third tile are discarded
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = wave + 4 * ii;
            if (i < PT) {
                f32x4t acc[3] = {(f32x4t){0.f, 0.f, 0.f, 0.f}, (f32x4t){0.f, 0.f, 0.f, 0.f}, (f32x4t){0.f, 0.f, 0.f, 0.f}};
                const int pr = 16 * i + li;
                const float dp = dinv[pr];
                for (int s4 = 0; s4 < P / 4; ++s4) {
                    const int o = 4 * s4 + kq;
                    const float av = (Cm[pr * CP + o] + (o == pr ? 1.f : 0.f)) * dp * dinv[o];
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[j] = sg_mfma(av, F[o * LD + 16 * j + li], acc[j]);
                }
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int pp = 16 * i + 4 * kq + r, f = 16 * j + li;
                        if (f < SG_F) ax[((int64_t)pp * g.B + b) * SG_F + f] = acc[j][r];
                    }
            }
        }
    }
}

__global__ void sg_bias_cols_kernel(float* __restrict__ v, const float* __restrict__ bias, int64_t rows, int cols, int act) {
    const int64_t tot = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (int64_t)gridDim.x * blockDim.x) {
        const float t = v[i] + bias[i % cols];
        v[i] = act == 1 ? fmaxf(t, 0.f) : t;
    }
}

// v[r][c] = act(v[r][c] + bias[r])     act: 0 none, 2 tanh
__global__ void sg_bias_rows_kernel(float* __restrict__ v, const float* __restrict__ bias, int rows, int64_t cols, int act) {
    const int64_t tot = (int64_t)rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (int64_t)gridDim.x * blockDim.x) {
        const float t = v[i] + bias[i / cols];
        v[i] = act == 2 ? tanhf(t) : t;
    }
}

// d[i] *= (h[i] > 0)
__global__ void sg_relu_bwd_kernel(float* __restrict__ d, const float* __restrict__ h, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        d[i] = h[i] > 0.f ? d[i] : 0.f;
}

// d[i] *= 1 - s[i]^2
__global__ void sg_tanh_bwd_kernel(float* __restrict__ d, const float* __restrict__ s, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        d[i] *= 1.f - s[i] * s[i];
}

// out[r] = sum_c v[r][c], fixed order: workgroup (r, s) adds slice s of row r, sg_rowsum_finish adds the slices in order
// (one workgroup per row left 128-200 workgroups walking 400 KB each: 100 us per call)
constexpr int SG_ROW_SLICES = 16;
__global__ __launch_bounds__(GB) void sg_rowsum_kernel(const float* __restrict__ v, int64_t cols, float* __restrict__ part) {
    __shared__ float red[GB];
    const int64_t per = (cols + SG_ROW_SLICES - 1) / SG_ROW_SLICES;
    const int64_t c0 = (int64_t)blockIdx.y * per, c1 = c0 + per < cols ? c0 + per : cols;
    const float* row = v + (int64_t)blockIdx.x * cols;
    float a[1] = {0.f};
    for (int64_t c = c0 + threadIdx.x; c < c1; c += GB) a[0] += row[c];
    block_sum<1>(a, red);
    if (threadIdx.x == 0) part[blockIdx.x * SG_ROW_SLICES + blockIdx.y] = a[0];
}
__global__ void sg_rowsum_finish_kernel(const float* __restrict__ part, int rows, float* __restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    float a = 0.f;
    for (int s = 0; s < SG_ROW_SLICES; ++s) a += part[r * SG_ROW_SLICES + s];
    out[r] = a;
}

// Softmax over the nodes of every feature channel (logits [P][B*H] + bias[p], in place -> attention), x * attention and the final
// Linear: one wavefront per (sample, 64 channels), one thread per column -- an online max / sum pass, then one pass that writes the
// attention and accumulates the column's share of the prediction (two strided passes over the logits instead of three, 1600 wavefronts
// instead of 100 workgroups).  The per-chunk shares are added per sample in a fixed order by sg_head_finish_kernel.
__global__ __launch_bounds__(64) void sg_attn_head_kernel(SgGeom g, float* __restrict__ attn, const float* __restrict__ h3,
                                                         const float* __restrict__ prm, float* __restrict__ part) {
    __shared__ float red[64];
    const int P = g.P, H = g.H, chunks = (H + 63) / 64;
    const float* bs = prm + g.o_bs;
    const float* wfc = prm + g.o_wfc;
    for (int64_t w = blockIdx.x; w < g.B * chunks; w += gridDim.x) {
        const int64_t b = w / chunks;
        const int h = (int)(w % chunks) * 64 + threadIdx.x;
        float acc = 0.f;
        if (h < H) {
            const int64_t c = b * H + h;
            // online softmax over the patches, eight rows per rescale: the eight loads are in flight together and the running sum is rescaled
            // once per group (it was one load, two exponentials and a dependent rescale per patch)
            float m = -INFINITY, sum = 0.f;
            int p0 = 0;
            for (; p0 + 7 < P; p0 += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = attn[(p0 + u) * g.BH + c] + bs[p0 + u];
                float cm = v[0];
#pragma unroll
                for (int u = 1; u < 8; ++u) cm = fmaxf(cm, v[u]);
                const float mn = fmaxf(m, cm);
                float e = 0.f;
#pragma unroll
                for (int u = 0; u < 8; ++u) e += expf(v[u] - mn);
                sum = sum * expf(m - mn) + e;
                m = mn;
            }
            for (; p0 < P; ++p0) {
                const float v = attn[p0 * g.BH + c] + bs[p0];
                const float mn = fmaxf(m, v);
                sum = sum * expf(m - mn) + expf(v - mn);
                m = mn;
            }
            const float inv = 1.0f / sum;
#pragma unroll 8
            for (int p = 0; p < P; ++p) {
                const float a = expf(attn[p * g.BH + c] + bs[p] - m) * inv;
                attn[p * g.BH + c] = a;
                acc = fmaf(h3[p * g.BH + c] * a, wfc[p * H + h], acc);
            }
        }
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int k = 32; k > 0; k >>= 1) {
            if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
            __syncthreads();
        }
        if (threadIdx.x == 0) part[w] = red[0];
        __syncthreads();
    }
}

__global__ void sg_head_finish_kernel(SgGeom g, const float* __restrict__ part, const float* __restrict__ prm, const float* __restrict__ y,
                                      float* __restrict__ pred, float* __restrict__ ws, float inv_gb) {
    const int chunks = (g.H + 63) / 64;
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < g.B; b += (int64_t)gridDim.x * blockDim.x) {
        float a = prm[g.o_bfc];
        for (int k = 0; k < chunks; ++k) a += part[b * chunks + k];
        pred[b] = a;
        if (y) {
            const float d = a - y[b];
            ws[g.w_sq + b] = d * d * inv_gb;
            ws[g.w_dpred + b] = 2.f * d * inv_gb;
        }
    }
}

// backward of the head and the softmax, one thread per (sample, channel) column:
//   dout = dpred w_fc ; d h3 = dout * attn -> dh3 ; d attn = dout * h3 ; d logits = attn * (d attn - sum_p d attn * attn) -> dlg
__global__ void sg_attn_head_bwd_kernel(SgGeom g, const float* __restrict__ attn, const float* __restrict__ h3, const float* __restrict__ prm,
                                        const float* __restrict__ dpred, float* __restrict__ dh3, float* __restrict__ dlg) {
    const int P = g.P, H = g.H;
    const float* wfc = prm + g.o_wfc;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < g.BH; c += (int64_t)gridDim.x * blockDim.x) {
        const int h = (int)(c % H);
        const float dp = dpred[c / H];
        float dot = 0.f;
        for (int p = 0; p < P; ++p) {
            const float a = attn[p * g.BH + c], dout = dp * wfc[p * H + h];
            dh3[p * g.BH + c] = dout * a;
            dot = fmaf(dout * h3[p * g.BH + c], a, dot);
        }
        for (int p = 0; p < P; ++p) {
            const float a = attn[p * g.BH + c], dout = dp * wfc[p * H + h];
            dlg[p * g.BH + c] = a * (dout * h3[p * g.BH + c] - dot);
        }
    }
}

// g fc.weight[p * H + h] = sum_b dpred[b] h3[p][b][h] attn[p][b][h]
__global__ void sg_fcw_kernel(SgGeom g, const float* __restrict__ attn, const float* __restrict__ h3, const float* __restrict__ dpred,
                              float* __restrict__ gw) {
    const int64_t tot = (int64_t)g.P * g.H;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / g.H, h = i % g.H;
        float a = 0.f;
        for (int64_t b = 0; b < g.B; ++b) {
            const int64_t at = p * g.BH + b * g.H + h;
            a = fmaf(dpred[b], h3[at] * attn[at], a);
        }
        gw[i] = a;
    }
}

__global__ void sg_fill_kernel(float* p, int n, float v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

inline unsigned sg_grid(int64_t n) {
    int64_t b = (n + GB - 1) / GB;
    return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

int64_t sagcn_param_count(const rulgnn_sagcn_shape* s) {
    SgGeom g;
    return sg_geometry(s, &g) == RULGNN_OK ? g.pcount : -1;
}

size_t sagcn_workspace_bytes(const rulgnn_sagcn_shape* s) {
    SgGeom g;
    return sg_geometry(s, &g) == RULGNN_OK ? (size_t)g.total * sizeof(float) : 0;
}

int64_t sagcn_tap_offset(const rulgnn_sagcn_shape* s, int which) {
    SgGeom g;
    if (sg_geometry(s, &g) != RULGNN_OK) return -1;
    switch (which) {
        case 0: return g.w_feat;
        case 1: return g.w_ax;
        case 2: return g.w_h[1];
        case 3: return g.w_attn;
        default: return -1;
    }
}

#define SG_RC(call)                        \
    do {                                   \
        const int rc_ = (call);            \
        if (rc_ != RULGNN_OK) return rc_;  \
    } while (0)
#define SG_LAUNCH_OK()                                           \
    do {                                                         \
        if (hipGetLastError() != hipSuccess) return RULGNN_EHIP; \
    } while (0)

// mode bit 0: forward, bit 1: backward (after a forward with the same args / workspace)
int sagcn_run(const rulgnn_sagcn_shape* s, const rulgnn_sagcn_args* a, int mode, hipStream_t st) {
    SgGeom g;
    SG_RC(sg_geometry(s, &g));
    if (a->workspace_bytes < (size_t)g.total * sizeof(float)) return RULGNN_EWORKSPACE;
    if (g.B == 0) return RULGNN_OK;
    float* ws = static_cast<float*>(a->workspace);
    const float* prm = a->params;
    const int64_t gb = a->global_batch > 0 ? a->global_batch : g.B;
    const float inv_gb = 1.0f / (float)gb;
    const int P = g.P, H = g.H, Ah = g.Ah, R = (int)g.R, BH = (int)g.BH;
    const int64_t RH = (int64_t)R * H;
    float* split = ws + g.w_split;
    float* h1 = ws + g.w_h1;
    float* u[2] = {ws + g.w_u[0], ws + g.w_u[1]};
    float* hh[3] = {h1, ws + g.w_h[0], ws + g.w_h[1]};                 // h1, h2, h3
    float* S = ws + g.w_s;
    float* attn = ws + g.w_attn;
    (void)hipGetLastError();
    if (mode & 1) {
        const size_t lds1 = sizeof(float) * ((size_t)3 * g.n + g.nh + 7 * GB);
        const size_t lds2 = sizeof(float) * ((size_t)P * (SG_F + 1) + 2 * P + GB);
        if (lds1 > 64 * 1024 || lds2 > 64 * 1024) return RULGNN_EUNSUPPORTED;
        if (g.n <= 64) {
            const size_t lw = sizeof(float) * ((size_t)3 * g.n + g.nh + 7 * 64);
            hipLaunchKernelGGL(sg_patch_features_kernel<64>, dim3((unsigned)(g.R < 65536 ? g.R : 65536)), dim3(64), lw, st, g, a->x, ws + g.w_raw);
        } else
        hipLaunchKernelGGL(sg_patch_features_kernel<GB>, dim3((unsigned)(g.R < 16384 ? g.R : 16384)), dim3(GB), lds1, st, g, a->x, ws + g.w_raw);
        if (P % 16 == 0 && P <= 128) {
            const size_t lm = sg_graph_mx_lds(P);
            static bool raised = false;
            if (lm > 48 * 1024 && !raised) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sg_graph_mx_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                raised = true;
            }
            hipLaunchKernelGGL(sg_graph_mx_kernel, dim3((unsigned)(g.B < 4096 ? g.B : 4096)), dim3(GB), lm, st, g, (const float*)(ws + g.w_raw),
                           ws + g.w_feat, ws + g.w_ax);
        } else
        hipLaunchKernelGGL(sg_graph_kernel, dim3((unsigned)(g.B < 4096 ? g.B : 4096)), dim3(GB), lds2, st, g, (const float*)(ws + g.w_raw),
                           ws + g.w_feat, ws + g.w_ax);
        SG_LAUNCH_OK();
        // gcn1: [R, 40] x W1^T
        SG_RC(sgemm(ws + g.w_ax, SG_F, 1, prm + g.o_w1, SG_F, 1, h1, H, R, H, SG_F, false, st));
        hipLaunchKernelGGL(sg_bias_cols_kernel, dim3(sg_grid(RH)), dim3(GB), 0, st, h1, prm + g.o_b1, (int64_t)R, H, 1);
        for (int i = 0; i < 2; ++i) {
            // node axis: U [P, B*H] = Wp [P, P] x h [P, B*H] + bp[p]
            SG_RC(sgemm(prm + g.o_wp[i], P, 1, hh[i], 1, BH, u[i], BH, P, BH, P, false, st));
            hipLaunchKernelGGL(sg_bias_rows_kernel, dim3(sg_grid(RH)), dim3(GB), 0, st, u[i], prm + g.o_bp[i], P, (int64_t)BH, 0);
            // feature axis: [R, H] x Wl^T + bl, ReLU
            SG_RC(sgemm(u[i], H, 1, prm + g.o_wl[i], H, 1, hh[i + 1], H, R, H, H, false, st));
            hipLaunchKernelGGL(sg_bias_cols_kernel, dim3(sg_grid(RH)), dim3(GB), 0, st, hh[i + 1], prm + g.o_bl[i], (int64_t)R, H, 1);
        }
        SG_LAUNCH_OK();
        // attention: S [Ah, B*H] = tanh(Wt [Ah, P] x h3 + bt[a]) ; logits [P, B*H] = Ws [P, Ah] x S (+ bs[p] in the head kernel)
        SG_RC(sgemm(prm + g.o_wt, P, 1, hh[2], 1, BH, S, BH, Ah, BH, P, false, st));
        hipLaunchKernelGGL(sg_bias_rows_kernel, dim3(sg_grid((int64_t)Ah * BH)), dim3(GB), 0, st, S, prm + g.o_bt, Ah, (int64_t)BH, 2);
        SG_RC(sgemm(prm + g.o_ws, Ah, 1, S, 1, BH, attn, BH, P, BH, Ah, false, st));
        {
            const int64_t waves = g.B * ((H + 63) / 64);
            hipLaunchKernelGGL(sg_attn_head_kernel, dim3((unsigned)(waves < 65536 ? waves : 65536)), dim3(64), 0, st, g, attn, (const float*)hh[2], prm,
                               ws + g.w_headpart);
            hipLaunchKernelGGL(sg_head_finish_kernel, dim3(sg_grid(g.B)), dim3(GB), 0, st, g, (const float*)(ws + g.w_headpart), prm, a->y, a->pred, ws,
                               inv_gb);
        }
        if (a->y && a->loss) (void)block_sum((const float*)(ws + g.w_sq), g.B, a->loss, st);
        SG_LAUNCH_OK();
    }
    if (mode & 2) {
        if (!a->grads) return RULGNN_EINVAL;
        float* gr = a->grads;
        const float* dpred = a->dpred ? a->dpred : ws + g.w_dpred;
        float* one = ws + g.w_one;
        float* dA = ws + g.w_dA;
        float* dB = ws + g.w_dB;
        float* ds = ws + g.w_ds;
        hipLaunchKernelGGL(sg_fill_kernel, dim3(1), dim3(64), 0, st, one, 1, 1.0f);
        // head + softmax: dA = d h3 (direct path), dB = d logits
        hipLaunchKernelGGL(sg_fcw_kernel, dim3(sg_grid((int64_t)P * H)), dim3(GB), 0, st, g, (const float*)attn, (const float*)hh[2], dpred,
                           gr + g.o_wfc);
        SG_RC(sgemm_splitk(dpred, 0, 1, one, 0, 0, gr + g.o_bfc, 1, 1, 1, (int)g.B, false, split, st));
        hipLaunchKernelGGL(sg_attn_head_bwd_kernel, dim3(sg_grid(BH)), dim3(GB), 0, st, g, (const float*)attn, (const float*)hh[2], prm, dpred, dA, dB);
        SG_LAUNCH_OK();
        SG_RC(sgemm_splitk(dB, BH, 1, S, BH, 1, gr + g.o_ws, Ah, P, Ah, BH, false, split, st));
        hipLaunchKernelGGL(sg_rowsum_kernel, dim3(P, SG_ROW_SLICES), dim3(GB), 0, st, (const float*)dB, (int64_t)BH, ws + g.w_rowpart);
        hipLaunchKernelGGL(sg_rowsum_finish_kernel, dim3((P + GB - 1) / GB), dim3(GB), 0, st, (const float*)(ws + g.w_rowpart), P, gr + g.o_bs);
        // d S [Ah, B*H] = Ws^T x d logits ; through tanh
        SG_RC(sgemm(prm + g.o_ws, 1, Ah, dB, 1, BH, ds, BH, Ah, BH, P, false, st));
        hipLaunchKernelGGL(sg_tanh_bwd_kernel, dim3(sg_grid((int64_t)Ah * BH)), dim3(GB), 0, st, ds, (const float*)S, (int64_t)Ah * BH);
        SG_RC(sgemm_splitk(ds, BH, 1, hh[2], BH, 1, gr + g.o_wt, P, Ah, P, BH, false, split, st));
        hipLaunchKernelGGL(sg_rowsum_kernel, dim3(Ah, SG_ROW_SLICES), dim3(GB), 0, st, (const float*)ds, (int64_t)BH, ws + g.w_rowpart);
        hipLaunchKernelGGL(sg_rowsum_finish_kernel, dim3((Ah + GB - 1) / GB), dim3(GB), 0, st, (const float*)(ws + g.w_rowpart), Ah, gr + g.o_bt);
        // d h3 += Wt^T x d(tanh input)
        SG_RC(sgemm(prm + g.o_wt, 1, P, ds, 1, BH, dA, BH, P, BH, Ah, true, st));
        SG_LAUNCH_OK();
        float* d = dA;
        float* other = dB;
        for (int i = 1; i >= 0; --i) {
            hipLaunchKernelGGL(sg_relu_bwd_kernel, dim3(sg_grid(RH)), dim3(GB), 0, st, d, (const float*)hh[i + 1], RH);
            SG_RC(sgemm_splitk(d, 1, H, u[i], 1, H, gr + g.o_wl[i], H, H, H, R, false, split, st));
            SG_RC(sgemm_splitk(one, 0, 0, d, 1, H, gr + g.o_bl[i], H, 1, H, R, false, split, st));
            // d U [R, H] = d V [R, H] x Wl
            SG_RC(sgemm(d, H, 1, prm + g.o_wl[i], 1, H, other, H, R, H, H, false, st));
            SG_RC(sgemm_splitk(other, BH, 1, hh[i], BH, 1, gr + g.o_wp[i], P, P, P, BH, false, split, st));
            hipLaunchKernelGGL(sg_rowsum_kernel, dim3(P, SG_ROW_SLICES), dim3(GB), 0, st, (const float*)other, (int64_t)BH, ws + g.w_rowpart);
            hipLaunchKernelGGL(sg_rowsum_finish_kernel, dim3((P + GB - 1) / GB), dim3(GB), 0, st, (const float*)(ws + g.w_rowpart), P, gr + g.o_bp[i]);
            // d h_in [P, B*H] = Wp^T x d U
            SG_RC(sgemm(prm + g.o_wp[i], 1, P, other, 1, BH, d, BH, P, BH, P, false, st));
            SG_LAUNCH_OK();
        }
        hipLaunchKernelGGL(sg_relu_bwd_kernel, dim3(sg_grid(RH)), dim3(GB), 0, st, d, (const float*)h1, RH);
        SG_RC(sgemm_splitk(d, 1, H, ws + g.w_ax, 1, SG_F, gr + g.o_w1, SG_F, H, SG_F, R, false, split, st));
        SG_RC(sgemm_splitk(one, 0, 0, d, 1, H, gr + g.o_b1, H, 1, H, R, false, split, st));
        SG_LAUNCH_OK();
    }
    return RULGNN_OK;
}

}  // namespace rulgnn
