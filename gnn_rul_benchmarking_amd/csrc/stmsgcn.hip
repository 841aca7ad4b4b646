// STMSGCN path for gfx950: SED features -> GCN stack -> GRU -> node mean -> Linear, forward and backward.
//
// Reference: models/STMSGCN/Model.py (SED_features :7-31, GCNLayer :34-49, GRULayer :52-60,
// STMSGCN_model.forward :84-112) and algorithms/algorithms.py:546-571 (MSE + Adam).
//
// Decomposition.  A sample is num_patch independent small graphs (nodes = spectral
// bands of one patch, <= 32) followed by a GRU that runs over the patches for every (sample, node) pair:
//   features   one workgroup per graph: direct DFT of the patch in LDS -> SED -> the GCN layers with all
//              operands (features, adjacency, weights) in LDS -> writes the concatenated features `cat`
//              [graph][node][C] and, fused, the GRU input projection gi = cat W_ih^T + b_ih;
//   gru_fwd    one lane per (sequence, hidden unit): the recurrence over num_patch steps;
//   head       one workgroup per sample: mean over nodes, Linear, squared error and d loss / d pred;
//   gru_bwd    BPTT with the gates recomputed from gi and h; writes d gi;
//   gcn_bwd    one workgroup per graph: d cat = d gi W_ih, then the GCN layers backwards (through the
//              linear map, both D^-1/2 factors, the row sums and the Gram matrix), weight gradients
//              accumulated in LDS per workgroup;
//   finalize   fixed-order reduction of the per-workgroup partial gradients, fc gradients, loss.
// Everything is fp32; reductions have a fixed order, so results are run-to-run reproducible.
#include "async_mem.hpp"
#include "sgemm_mfma.hpp"
#include "stgcn_host.hpp"

namespace rulgnn {

namespace {

constexpr int MB = 256;             // threads per workgroup
constexpr int MAXN = 32;            // graph nodes
constexpr int MAXL = RULGNN_STMSGCN_MAX_LAYERS;
constexpr float LEAKY = 0.01f;      // F.leaky_relu default slope (Model.py:47)

struct MsgGeom {
    int64_t B, G;                   // samples, graphs = B * NP
    int NP, P, interval, bw, n, L, C, H, H3;
    int dims[MAXL + 1];             // [1, gcn_dims...]
    int coff[MAXL + 2];             // column offset of each layer's features inside cat
    int woff[MAXL], boff[MAXL];     // parameter offsets
    int gcn_params;                 // floats in the GCN layers (prefix of the flat buffer)
    int off_wih, off_whh, off_bih, off_bhh, off_fcw, off_fcb, nparam;
    int CS, AXS, maxf;              // LDS strides (odd) and the widest layer input
};

__host__ int msg_geometry(const rulgnn_stmsgcn_shape* s, MsgGeom* g) {
    if (!s) return RULGNN_EINVAL;
    if (s->batch < 0 || s->num_patch < 1 || s->patch_size < 2 || s->interval < 1 || s->band_width < 1 ||
        s->num_gcn_layers < 1 || s->gru_hidden < 1)
        return RULGNN_EINVAL;
    if (s->interval >= s->patch_size || (s->patch_size - s->interval) % s->band_width != 0) return RULGNN_EINVAL;
    if (s->num_gcn_layers > MAXL || s->patch_size > 512 || s->num_patch > 4096 || s->gru_hidden > 16)
        return RULGNN_EUNSUPPORTED;
    g->n = (s->patch_size - s->interval) / s->band_width;
    if (g->n > MAXN) return RULGNN_EUNSUPPORTED;
    g->B = s->batch;
    g->NP = s->num_patch;
    g->G = s->batch * (int64_t)s->num_patch;
    if (g->G * g->n > ((int64_t)1 << 30)) return RULGNN_EUNSUPPORTED;
    g->P = s->patch_size;
    g->interval = s->interval;
    g->bw = s->band_width;
    g->L = s->num_gcn_layers;
    g->H = s->gru_hidden;
    g->H3 = 3 * g->H;
    g->dims[0] = 1;
    g->coff[0] = 0;
    g->coff[1] = 1;
    int off = 0, maxf = 1;
    for (int l = 0; l < g->L; ++l) {
        const int f = s->gcn_dims[l];
        if (f < 1) return RULGNN_EINVAL;
        if (f > 64) return RULGNN_EUNSUPPORTED;
        g->dims[l + 1] = f;
        g->coff[l + 2] = g->coff[l + 1] + f;
        g->woff[l] = off;
        off += f * g->dims[l];
        g->boff[l] = off;
        off += f;
        if (g->dims[l] > maxf) maxf = g->dims[l];
    }
    g->C = g->coff[g->L + 1];
    if (g->C > 128) return RULGNN_EUNSUPPORTED;
    g->gcn_params = off;
    g->off_wih = off; off += g->H3 * g->C;
    g->off_whh = off; off += g->H3 * g->H;
    g->off_bih = off; off += g->H3;
    g->off_bhh = off; off += g->H3;
    g->off_fcw = off; off += g->NP * g->H;
    g->off_fcb = off; off += 1;
    g->nparam = off;
    g->CS = g->C | 1;
    g->maxf = maxf;
    g->AXS = maxf | 1;
    return RULGNN_OK;
}

// ---------------------------------------------------------------------------------------------------
// shared per-graph stages (block-cooperative; all operands in LDS)
// ---------------------------------------------------------------------------------------------------
// A' = x x^T + I for the feature slice [off, off+f) of cat  (Model.py:98 and :41).
// (All per-graph products below keep a 2x2 or 1x4 block of outputs per thread: on the vector ALUs every FMA of a
// one-output-per-thread loop costs two LDS reads, and the LDS pipe is what binds these kernels.)
template <int TW>
__device__ inline void gram_plus_identity(const float* cat, int CS, int off, int f, int n, float* araw) {
    constexpr int TS = TW > 1 ? 2 : 1;
    const int nb = TW > 1 ? (n + 1) >> 1 : n;
    for (int e = threadIdx.x; e < nb * nb; e += MB) {
        const int i = TS * (e / nb), j = TS * (e % nb);
        const int i1 = (TS > 1 && i + 1 < n) ? i + 1 : i, j1 = (TS > 1 && j + 1 < n) ? j + 1 : j;
        const float* xi = cat + i * CS + off;
        const float* xi1 = cat + i1 * CS + off;
        const float* xj = cat + j * CS + off;
        const float* xj1 = cat + j1 * CS + off;
        float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;
#pragma unroll 8
        for (int c = 0; c < f; ++c) {
            const float u0 = xi[c], u1 = xi1[c], v0 = xj[c], v1 = xj1[c];
            a00 = fmaf(u0, v0, a00); a01 = fmaf(u0, v1, a01);
            a10 = fmaf(u1, v0, a10); a11 = fmaf(u1, v1, a11);
        }
        araw[i * (n + 1) + j] = a00 + (i == j ? 1.f : 0.f);
        if (TS > 1 && j + 1 < n) araw[i * (n + 1) + j + 1] = a01 + (i == j + 1 ? 1.f : 0.f);
        if (TS > 1 && i + 1 < n) araw[(i + 1) * (n + 1) + j] = a10 + (i + 1 == j ? 1.f : 0.f);
        if (TS > 1 && i + 1 < n && j + 1 < n) araw[(i + 1) * (n + 1) + j + 1] = a11 + (i == j ? 1.f : 0.f);
    }
}
// r = rowsum(A')^-1/2 (Model.py:43; a negative row sum gives NaN exactly like torch's pow).
__device__ inline void inv_sqrt_degree(const float* araw, int n, float* rv) {
    if (threadIdx.x < n) {
        const float* row = araw + threadIdx.x * (n + 1);
        float d = 0.f;
        for (int j = 0; j < n; ++j) d += row[j];
        rv[threadIdx.x] = 1.0f / sqrtf(d);
    }
}
// ahat = D (A' D)  (Model.py:44)
__device__ inline void normalise(const float* araw, const float* rv, int n, float* ahat) {
    for (int e = threadIdx.x; e < n * n; e += MB) {
        const int i = e / n, j = e - i * n;
        ahat[i * (n + 1) + j] = (araw[i * (n + 1) + j] * rv[j]) * rv[i];
    }
}
// AX = ahat x  (Model.py:45)
template <int TW>
__device__ inline void aggregate(const float* ahat, const float* cat, int CS, int off, int f, int n, float* ax, int AXS) {
    const int fq = (f + TW - 1) / TW;
    for (int e = threadIdx.x; e < n * fq; e += MB) {
        const int i = e / fq, c = TW * (e - i * fq);
        const float* ar = ahat + i * (n + 1);
        const float* xc = cat + off + c;
        const bool k1 = TW > 1 && c + 1 < f, k2 = TW > 1 && c + 2 < f, k3 = TW > 1 && c + 3 < f;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 8
        for (int j = 0; j < n; ++j) {
            const float w = ar[j];
            const float* xr = xc + j * CS;
            a0 = fmaf(w, xr[0], a0);
            a1 = fmaf(w, k1 ? xr[1] : 0.f, a1);
            a2 = fmaf(w, k2 ? xr[2] : 0.f, a2);
            a3 = fmaf(w, k3 ? xr[3] : 0.f, a3);
        }
        float* o = ax + i * AXS + c;
        o[0] = a0;
        if (k1) o[1] = a1;
        if (k2) o[2] = a2;
        if (k3) o[3] = a3;
    }
}

// ---------------------------------------------------------------------------------------------------
// features: SED + GCN stack (+ GRU input projection)
// ---------------------------------------------------------------------------------------------------
// TW: register-block width of the per-graph products (4 for graphs of >= 12 nodes, 1 = one output per thread for small ones)
template <int TW>
__global__ __launch_bounds__(MB) void msg_features_kernel(MsgGeom g, const float* __restrict__ x,
                                                          const float* __restrict__ prm, float* __restrict__ cat_out,
                                                          float* __restrict__ gi_out) {
    extern __shared__ float smem[];
    const int n = g.n, P = g.P, C = g.C, CS = g.CS, AXS = g.AXS, H3 = g.H3;
    float* wt = smem;                              // per layer: Wt[k][o] (transposed) | b[o]
    float* wih = wt + g.gcn_params;                // WihT[c][3H] | b_ih[3H]
    float* tw = wih + (C + 1) * H3;                // (cos, sin)(2 pi k / P)
    float* xs = tw + 2 * P;
    float* fre = xs + P;
    float* fim = fre + P;
    float* cat = fim + P;                          // [n][CS]
    float* araw = cat + n * CS;                    // [n][n+1]
    float* rv = araw + n * (n + 1);                // [32]
    float* ax = rv + MAXN;                         // [n][AXS]
    float* pre = ax;                               // partial DFT sums (real | imaginary) borrow the AX tile: it is idle until the GCN stack
    float* pim = ax + MB;
    const int tid = threadIdx.x;

    for (int l = 0; l < g.L; ++l) {
        const int fi = g.dims[l], fo = g.dims[l + 1];
        for (int e = tid; e < fi * fo; e += MB) {
            const int o = e / fi, k = e - o * fi;
            wt[g.woff[l] + k * fo + o] = prm[g.woff[l] + e];
        }
        for (int e = tid; e < fo; e += MB) wt[g.boff[l] + e] = prm[g.boff[l] + e];
    }
    if (gi_out) {
        for (int e = tid; e < H3 * C; e += MB) {
            const int q = e / C, c = e - q * C;
            wih[c * H3 + q] = prm[g.off_wih + e];
        }
        for (int e = tid; e < H3; e += MB) wih[C * H3 + e] = prm[g.off_bih + e];
    }
    for (int k = tid; k < P; k += MB) {
        float sn, cs;
        sincospif(2.0f * (float)k / (float)P, &sn, &cs);
        tw[2 * k] = cs;
        tw[2 * k + 1] = sn;
    }
    __syncthreads();

    for (int64_t gi = blockIdx.x; gi < g.G; gi += gridDim.x) {
        // ---- SED_features (Model.py:7-31): full DFT of the patch, lagged difference, band energies ----
        const float* xp = x + gi * P;
        for (int k = tid; k < P; k += MB) xs[k] = xp[k];
        __syncthreads();
        if (P * 2 <= MB && n * AXS >= 2 * MB) {
            // short patches: MB / P threads share one frequency, each summing a contiguous run of the time index (the patch of
            // 128 points left half of the workgroup idle for 128 dependent iterations); partial sums meet in LDS
            const int S = MB / P, j = tid % P, h = tid / P;
            const int per = (P + S - 1) / S, t0 = h * per, t1 = t0 + per < P ? t0 + per : P;
            float re = 0.f, im = 0.f;
            if (h < S) {
                int idx = (int)(((int64_t)j * t0) % P);
                for (int t = t0; t < t1; ++t) {
                    const float v = xs[t];
                    re = fmaf(v, tw[2 * idx], re);
                    im = fmaf(-v, tw[2 * idx + 1], im);
                    idx += j;
                    if (idx >= P) idx -= P;
                }
            }
            pre[tid] = re;
            pim[tid] = im;
            __syncthreads();
            if (tid < P) {
                float a = 0.f, b = 0.f;
                for (int q = 0; q < S; ++q) { a += pre[q * P + tid]; b += pim[q * P + tid]; }
                fre[tid] = a;
                fim[tid] = b;
            }
        } else {
            for (int j = tid; j < P; j += MB) {
                float re = 0.f, im = 0.f;
                int idx = 0;
#pragma unroll 8
                for (int t = 0; t < P; ++t) {
                    const float v = xs[t];
                    re = fmaf(v, tw[2 * idx], re);
                    im = fmaf(-v, tw[2 * idx + 1], im);
                    idx += j;
                    if (idx >= P) idx -= P;
                }
                fre[j] = re;
                fim[j] = im;
            }
        }
        __syncthreads();
        float en[2] = {0.f, 0.f};
        for (int j = tid, r = 0; j < P - g.interval; j += MB, ++r) {
            const float dr = fre[j + g.interval] - fre[j], di = fim[j + g.interval] - fim[j];
            en[r] = dr * dr + di * di;
        }
        __syncthreads();
        for (int j = tid, r = 0; j < P - g.interval; j += MB, ++r) fre[j] = en[r];
        __syncthreads();
        if (tid < n) {
            float s = 0.f;
            for (int q = 0; q < g.bw; ++q) s += fre[tid * g.bw + q];
            cat[tid * CS] = s;
        }
        __syncthreads();
        // ---- GCN stack (Model.py:96-100) ----
        for (int l = 0; l < g.L; ++l) {
            const int fi = g.dims[l], fo = g.dims[l + 1], off = g.coff[l], offo = g.coff[l + 1];
            gram_plus_identity<TW>(cat, CS, off, fi, n, araw);
            __syncthreads();
            inv_sqrt_degree(araw, n, rv);
            __syncthreads();
            normalise(araw, rv, n, araw);
            __syncthreads();
            aggregate<TW>(araw, cat, CS, off, fi, n, ax, AXS);
            __syncthreads();
            const float* w = wt + g.woff[l];
            const float* b = wt + g.boff[l];
            const int foq = (fo + TW - 1) / TW;
            for (int e = tid; e < n * foq; e += MB) {
                const int i = e / foq, o = TW * (e - i * foq);
                const bool k1 = TW > 1 && o + 1 < fo, k2 = TW > 1 && o + 2 < fo, k3 = TW > 1 && o + 3 < fo;
                float z0 = 0.f, z1 = 0.f, z2 = 0.f, z3 = 0.f;
#pragma unroll 8
                for (int k = 0; k < fi; ++k) {
                    const float av = ax[i * AXS + k];
                    const float* wr = w + k * fo + o;
                    z0 = fmaf(av, wr[0], z0);
                    z1 = fmaf(av, k1 ? wr[1] : 0.f, z1);
                    z2 = fmaf(av, k2 ? wr[2] : 0.f, z2);
                    z3 = fmaf(av, k3 ? wr[3] : 0.f, z3);
                }
                float* dst = cat + i * CS + offo + o;
                z0 += b[o];
                dst[0] = z0 > 0.f ? z0 : LEAKY * z0;
                if (k1) { z1 += b[o + 1]; dst[1] = z1 > 0.f ? z1 : LEAKY * z1; }
                if (k2) { z2 += b[o + 2]; dst[2] = z2 > 0.f ? z2 : LEAKY * z2; }
                if (k3) { z3 += b[o + 3]; dst[3] = z3 > 0.f ? z3 : LEAKY * z3; }
            }
            __syncthreads();
        }
        if (cat_out) {
            float* dst = cat_out + gi * (int64_t)(n * C);
            for (int e = tid; e < n * C; e += MB) {
                const int i = e / C, c = e - i * C;
                dst[e] = cat[i * CS + c];
            }
        }
        if (gi_out) {                              // GRU input projection, fused (nn.GRU: x W_ih^T + b_ih)
            float* dst = gi_out + gi * (int64_t)(n * H3);
            for (int e = tid; e < n * H3; e += MB) {
                const int i = e / H3, q = e - i * H3;
                float a = 0.f;
                for (int c = 0; c < C; ++c) a = fmaf(cat[i * CS + c], wih[c * H3 + q], a);
                dst[e] = a + wih[C * H3 + q];
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// features on the fp32 matrix cores: ONE WAVEFRONT PER GRAPH (graphs of >= 12 nodes; round 3)
// ---------------------------------------------------------------------------------------------------
// msg_features_kernel above spends ~97 % of a graph's 70 us waiting: 256 threads share one graph, every product is a loop of
// dependent LDS reads behind a workgroup barrier (~30 barriers per graph).  Here a graph belongs to one wavefront and lives in its
// registers; every product of the GCN stack is a chain of v_mfma_f32_32x32x2_f32 (exact fp32: an fmaf chain, bitwise) on the graph
// padded to 32 nodes, and two register forms of a [node, feature] matrix are enough to chain them without ever transposing:
//   NL ("node on lanes"):    register m, lane (h, i)  = M[node i][feature f(m, h)]     f(m, h) = 32 (m / 16) + krow(m % 16, h)
//   FL ("feature on lanes"): register r, lane (h, c)  = M[node krow(r, h)][feature c]  (one set of 16 registers per 32 features)
// with krow(r, h) = 8 (r / 4) + 4 h + r % 4 -- the row a 32x32 MFMA result keeps in accumulator register r of lane half h, so that
// an MFMA result IS one of the two forms: rows = features gives NL, rows = nodes gives FL.  Per layer (Model.py:34-49, :96-100):
//   A' = X X^T + I            A = B = X in NL form (the contraction runs over the features)                    f_in / 2 MFMAs
//   r = rowsum(A')^-1/2, A^ = r_i A'_ij r_j     in the result layout; the row factors come back through 32 floats of LDS
//   (A^ X)^T = X^T A^         A = X in FL form, B = A^ as it stands (symmetric): the result is A^ X in NL form         16 per 32 features
//   Z = (A^ X) W^T            A = A^X (NL), B = W as an operand table -> FL form;  the SAME two registers swapped -> NL form
// so a layer hands the next one both forms (the Linear layer runs twice; no transposes, no barriers), the FL form goes out to `cat`
// with coalesced stores, and the GRU input projection accumulates from every layer's NL form as it appears.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int MXW = 4;                   // wavefronts = graphs in flight per workgroup
__host__ __device__ constexpr int krow(int r, int h) { return 8 * (r >> 2) + 4 * h + (r & 3); }
__host__ __device__ constexpr int nl_feat(int m, int h) { return 32 * (m >> 4) + krow(m & 15, h); }

struct MxLds {                           // float offsets of the workgroup's LDS
    int tw, bias, bih, wop[MAXL], wih[MAXL + 1], wave, wave_floats, tr, total;
};
__host__ __device__ inline void mx_lds_layout(const MsgGeom& g, MxLds* o) {
    int p = 0;
    o->tw = p; p += 2 * g.P;
    o->bias = p; p += 64 * g.L;
    o->bih = p; p += 32;
    for (int l = 0; l < g.L; ++l) {
        const int nbi = (g.dims[l] + 31) / 32, nbo = (g.dims[l + 1] + 31) / 32;
        o->wop[l] = p; p += nbo * nbi * 16 * 64;
    }
    o->wih[0] = p; p += 64;
    for (int l = 0; l < g.L; ++l) { o->wih[l + 1] = p; p += ((g.dims[l + 1] + 31) / 32) * 16 * 64; }
    o->wave = p;
    o->wave_floats = 3 * g.P + 64;       // xs | re | im | row-factor scratch, one set per wavefront
    p += MXW * o->wave_floats;
    o->tr = p; p += MXW * 32 * 33;       // per-wavefront transpose tile (FL -> NL form of a layer's output block)
    o->total = p;
}
static bool mx_features_ok(const MsgGeom& g, size_t* lds_bytes) {
    if (g.n < 12 || g.H3 > 32) return false;
    MxLds o;
    mx_lds_layout(g, &o);
    *lds_bytes = sizeof(float) * (size_t)o.total;
    return *lds_bytes <= 80 * 1024;              // (two workgroups per CU)
}

__device__ __forceinline__ f32x16 mfma32(float a, float b, const f32x16& c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
// the 16 values v[krow(r, h)] of a per-node vector in this wavefront's LDS scratch (four 16-byte reads: krow runs in fours)
__device__ __forceinline__ void row_values(const float* vec, int h, float (&out)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 t = *reinterpret_cast<const float4*>(vec + 8 * q + 4 * h);
        out[4 * q] = t.x; out[4 * q + 1] = t.y; out[4 * q + 2] = t.z; out[4 * q + 3] = t.w;
    }
}

// XJ: the reference's XJTU-SY wiring (gcn_dims 16-64-16-1 on 25-node graphs) with its layer shapes as compile-time constants: the layer loop
// unrolls, block counts and column offsets fold, the guards around the matrix instructions of absent feature blocks disappear
struct MsgXJ {
    static constexpr int L = 4, n = 25, C = 98, gcn_params = 2177;
    __host__ __device__ static constexpr int dim(int l) { return l == 0 ? 1 : l == 1 ? 16 : l == 2 ? 64 : l == 3 ? 16 : 1; }
    __host__ __device__ static constexpr int coff(int l) { return l == 0 ? 0 : l == 1 ? 1 : l == 2 ? 17 : l == 3 ? 81 : l == 4 ? 97 : 98; }
    __host__ __device__ static constexpr int woff(int l) { return l == 0 ? 0 : l == 1 ? 32 : l == 2 ? 1120 : 2160; }
    __host__ __device__ static constexpr int boff(int l) { return l == 0 ? 16 : l == 1 ? 1056 : l == 2 ? 2144 : 2176; }
    static bool matches(const MsgGeom& g) {
        if (g.L != L || g.n != n || g.C != C || g.gcn_params != gcn_params) return false;
        for (int l = 0; l <= L; ++l)
            if (g.dims[l] != dim(l) || g.coff[l] != coff(l)) return false;
        for (int l = 0; l < L; ++l)
            if (g.woff[l] != woff(l) || g.boff[l] != boff(l)) return false;
        return true;
    }
};
template <bool XJ>
__global__ __launch_bounds__(64 * MXW, 2) void msg_features_mx_kernel(MsgGeom g, const float* __restrict__ x, const float* __restrict__ prm,
                                                                    float* __restrict__ cat_out, float* __restrict__ gi_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    MxLds L_;
    mx_lds_layout(g, &L_);
    const int n = XJ ? MsgXJ::n : g.n, P = g.P, C = XJ ? MsgXJ::C : g.C, H3 = g.H3;
    const int NL = XJ ? MsgXJ::L : g.L;
    auto gdim = [&](int l) { return XJ ? MsgXJ::dim(l) : g.dims[l]; };
    auto gcoff = [&](int l) { return XJ ? MsgXJ::coff(l) : g.coff[l]; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, j = lane & 31;
    float* tw = smem + L_.tw;

    // ---- operand tables (once per workgroup) ----------------------------------------------------------------------------
    for (int k = tid; k < P; k += 64 * MXW) {
        float sn, cs;
        sincospif(2.0f * (float)k / (float)P, &sn, &cs);
        tw[2 * k] = cs;
        tw[2 * k + 1] = sn;
    }
    for (int e = tid; e < 64 * g.L; e += 64 * MXW) {
        const int l = e >> 6, o = e & 63;
        smem[L_.bias + e] = o < g.dims[l + 1] ? prm[g.boff[l] + o] : 0.f;
    }
    for (int e = tid; e < 32; e += 64 * MXW) smem[L_.bih + e] = gi_out && e < H3 ? prm[g.off_bih + e] : 0.f;
    for (int l = 0; l < g.L; ++l) {
        const int fi = g.dims[l], fo = g.dims[l + 1], nbi = (fi + 31) / 32, nbo = (fo + 31) / 32;
        for (int e = tid; e < nbo * nbi * 16 * 64; e += 64 * MXW) {          // Linear: lane (hh, o) of register m: W[32 ob + o][f(m, hh)]
            const int ln = e & 63, m = (e >> 6) % (nbi * 16), ob = (e >> 6) / (nbi * 16);
            const int o = 32 * ob + (ln & 31), f = nl_feat(m, ln >> 5);
            smem[L_.wop[l] + e] = (o < fo && f < fi) ? prm[g.woff[l] + o * fi + f] : 0.f;
        }
        for (int e = tid; e < nbo * 16 * 64; e += 64 * MXW) {                 // GRU input projection of this layer's output columns
            const int ln = e & 63, m = e >> 6;
            const int q = ln & 31, f = nl_feat(m, ln >> 5);
            smem[L_.wih[l + 1] + e] = (gi_out && q < H3 && f < fo) ? prm[g.off_wih + q * C + g.coff[l + 1] + f] : 0.f;
        }
    }
    for (int e = tid; e < 64; e += 64 * MXW)                                   // ... and of the input column (the SED feature)
        smem[L_.wih[0] + e] = (gi_out && (e & 31) < H3 && (e >> 5) == 0) ? prm[g.off_wih + (e & 31) * C] : 0.f;
    __syncthreads();

    float* xs = smem + L_.wave + wave * L_.wave_floats;        // this wavefront's scratch
    float* fre = xs + P;
    float* fim = fre + P;
    float* rowv = fim + P;                                     // [64]: two per-node vectors for row_values()
    const bool node_ok = j < n;

    for (int64_t gi = (int64_t)blockIdx.x * MXW + wave; gi < g.G; gi += (int64_t)gridDim.x * MXW) {
        // ---- SED_features (Model.py:7-31): DFT of the patch, lagged difference, band energies -------------------------------
        const float* xp = x + gi * P;
        __builtin_amdgcn_wave_barrier();
        for (int k = lane; k < P; k += 64) xs[k] = xp[k];
        __builtin_amdgcn_wave_barrier();
        // The patch is real: F[P - k] = conj(F[k]).  Frequencies 0 .. P/2 - 1 are summed directly (two per pass share the reads of
        // the patch), F[P/2] = sum (-1)^t x[t] is a wavefront reduction, the upper half is mirrored.  (Odd P: all frequencies directly.)
        const int half = (P & 1) ? P : P / 2;
        if (half <= 64) {                                    // one frequency per lane
            if (lane < half) {
                float re0 = 0.f, im0 = 0.f;
                int i0 = 0;
#pragma unroll 8
                for (int t = 0; t < P; ++t) {
                    const float v = xs[t];
                    const float2 w0 = *reinterpret_cast<const float2*>(tw + 2 * i0);
                    re0 = fmaf(v, w0.x, re0); im0 = fmaf(-v, w0.y, im0);
                    i0 += lane; if (i0 >= P) i0 -= P;
                }
                fre[lane] = re0; fim[lane] = im0;
            }
        } else {
            for (int j0 = lane; j0 < half; j0 += 128) {
                const int j1 = j0 + 64 < half ? j0 + 64 : j0;
                float re0 = 0.f, im0 = 0.f, re1 = 0.f, im1 = 0.f;
                int i0 = 0, i1 = 0;
#pragma unroll 4
                for (int t = 0; t < P; ++t) {
                    const float v = xs[t];
                    const float2 w0 = *reinterpret_cast<const float2*>(tw + 2 * i0), w1 = *reinterpret_cast<const float2*>(tw + 2 * i1);
                    re0 = fmaf(v, w0.x, re0); im0 = fmaf(-v, w0.y, im0);
                    re1 = fmaf(v, w1.x, re1); im1 = fmaf(-v, w1.y, im1);
                    i0 += j0; if (i0 >= P) i0 -= P;
                    i1 += j1; if (i1 >= P) i1 -= P;
                }
                fre[j0] = re0; fim[j0] = im0;
                if (j0 + 64 < half) { fre[j1] = re1; fim[j1] = im1; }
            }
        }
        if (!(P & 1)) {
            float alt = 0.f;
            for (int t = lane; t < P; t += 64) alt += (t & 1) ? -xs[t] : xs[t];
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) alt += __shfl_xor(alt, m, 64);
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) { fre[P / 2] = alt; fim[P / 2] = 0.f; }
            for (int k = P / 2 + 1 + lane; k < P; k += 64) { fre[k] = fre[P - k]; fim[k] = -fim[P - k]; }
        }
        __builtin_amdgcn_wave_barrier();
        for (int k = lane; k < P - g.interval; k += 64) {
            const float dr = fre[k + g.interval] - fre[k], di = fim[k + g.interval] - fim[k];
            xs[k] = dr * dr + di * di;                         // (the patch itself is no longer needed)
        }
        __builtin_amdgcn_wave_barrier();
        float x0 = 0.f;
        if (node_ok)
            for (int q = 0; q < g.bw; ++q) x0 += xs[j * g.bw + q];
        if (cat_out && node_ok && h == 0) cat_out[gi * (int64_t)(n * C) + j * C] = x0;

        // ---- GCN stack (Model.py:96-100) ---------------------------------------------------------------------------------
        float Xn[32], Xf[2][16];
#pragma unroll
        for (int m = 0; m < 32; ++m) Xn[m] = 0.f;
        Xn[0] = h == 0 ? x0 : 0.f;                             // NL form of the one input feature (f(0, 0) = 0, f(0, 1) = 4)
        f32x16 giacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        giacc = mfma32(Xn[0], smem[L_.wih[0] + lane], giacc);
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const int fi = gdim(l), fo = gdim(l + 1), nbi = (fi + 31) / 32, nbo = (fo + 31) / 32;
            // A' = X X^T + I
            f32x16 G = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < 32; ++m)
                if (nl_feat(m, 0) < fi) G = mfma32(Xn[m], Xn[m], G);
            float d = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                G[r] += (krow(r, 0) + 4 * h == j) ? 1.f : 0.f;
                d += G[r];
            }
            d += __shfl_xor(d, 32, 64);
            const float rj = 1.0f / sqrtf(d);                  // rowsum^-1/2 (Model.py:43); A' is symmetric: column sum = row sum
            __builtin_amdgcn_wave_barrier();
            if (h == 0) { rowv[j] = rj; rowv[32 + j] = Xn[0]; }
            __builtin_amdgcn_wave_barrier();
            float rr[16];
            row_values(rowv, h, rr);
            float Ah[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) Ah[r] = (G[r] * rj) * rr[r];          // A^[krow(r, h)][j]
            // A^ X in NL form
            float AXn[32];
#pragma unroll
            for (int m = 0; m < 32; ++m) AXn[m] = 0.f;
            if (fi == 1) {                                     // one feature: a matrix-vector product on the vector ALUs
                float xr[16], s1 = 0.f;
                row_values(rowv + 32, h, xr);
#pragma unroll
                for (int r = 0; r < 16; ++r) s1 = fmaf(Ah[r], xr[r], s1);
                s1 += __shfl_xor(s1, 32, 64);
                AXn[0] = h == 0 ? s1 : 0.f;
            } else {
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    if (b < nbi) {
                        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc = mfma32(Xf[b][r], Ah[r], acc);
#pragma unroll
                        for (int r = 0; r < 16; ++r) AXn[16 * b + r] = acc[r];
                    }
                }
            }
            // Z = leaky((A^ X) W^T + b) in both forms; the FL form goes out to cat, the NL form feeds the GRU projection
            float Zn[32], Zf[2][16];
#pragma unroll
            for (int m = 0; m < 32; ++m) Zn[m] = 0.f;
            const float* wl = smem + L_.wop[l];
            const float* bl = smem + L_.bias + 64 * l;
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) {
#pragma unroll
                for (int r = 0; r < 16; ++r) Zf[ob][r] = 0.f;
                if (ob < nbo) {
                    // (round 4: the Linear layer ran twice to leave its result in both forms -- the second form now comes from the first
                    // through a wavefront-private LDS tile: 32 LDS operations instead of up to 32 matrix instructions per block)
                    f32x16 aF = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int m = 0; m < 32; ++m) {
                        if (nl_feat(m, 0) < fi) {
                            const float w = wl[(ob * nbi * 16 + m) * 64 + lane];
                            aF = mfma32(AXn[m], w, aF);
                        }
                    }
                    const float bF = bl[32 * ob + j];
                    float* dst = cat_out ? cat_out + gi * (int64_t)(n * C) + gcoff(l + 1) + 32 * ob + j : nullptr;
                    float* trt = smem + L_.tr + wave * (32 * 33);
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int node = krow(r, 0) + 4 * h;
                        float zf = aF[r] + bF;
                        zf = zf > 0.f ? zf : LEAKY * zf;
                        const bool okf = node < n && 32 * ob + j < fo;
                        Zf[ob][r] = okf ? zf : 0.f;
                        trt[node * 33 + j] = okf ? zf : 0.f;
                        if (dst && okf) dst[node * C] = zf;
                    }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int m = 0; m < 16; ++m) Zn[16 * ob + m] = trt[j * 33 + krow(m, 0) + 4 * h];
                }
            }
            if (gi_out) {
                const float* wq = smem + L_.wih[l + 1];
#pragma unroll
                for (int m = 0; m < 32; ++m)
                    if (nl_feat(m, 0) < fo) giacc = mfma32(Zn[m], wq[m * 64 + lane], giacc);
            }
#pragma unroll
            for (int m = 0; m < 32; ++m) Xn[m] = Zn[m];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) Xf[b][r] = Zf[b][r];
        }
        if (gi_out && j < H3) {                                // nn.GRU: x W_ih^T + b_ih, [graph][node][3H]
            float* dst = gi_out + gi * (int64_t)(n * H3) + j;
            const float bq = smem[L_.bih + j];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int node = krow(r, 0) + 4 * h;
                if (node < n) dst[node * H3] = giacc[r] + bq;
            }
        }
    }
}

static size_t features_lds_bytes(const MsgGeom& g) {
    return sizeof(float) * ((size_t)g.gcn_params + (g.C + 1) * g.H3 + 5 * g.P + g.n * g.CS + g.n * (g.n + 1) + MAXN +
                            g.n * g.AXS);
}

// ---------------------------------------------------------------------------------------------------
// GRU (nn.GRU, one layer, batch_first, h0 = 0; gate order r, z, n)
// ---------------------------------------------------------------------------------------------------
// hardware exp2 / reciprocal (v_exp_f32, v_rcp_f32: ~1 ulp each) instead of libm expf / tanhf + IEEE division: the gate
// non-linearities sit on the critical path of every sequential GRU step (same finding as in the HAGCN LSTM)
// (__frcp_rn compiles to the IEEE division sequence with this compiler: the builtin is the single v_rcp_f32)
__device__ inline float sigmoidf_(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ inline float tanhf_(float v) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * v)); }     // exp -> inf: 1; -> 0: -1

constexpr int GRU_AHEAD = 4;             // steps a step's inputs are requested ahead
// One lane per (sequence, hidden unit); HG = lanes per sequence (power of two >= H).
template <int HG>
__global__ __launch_bounds__(MB) void msg_gru_forward_kernel(MsgGeom g, const float* __restrict__ gi,
                                                             const float* __restrict__ prm, float* __restrict__ hseq) {
    __shared__ float hs[MB];
    const int tid = threadIdx.x, j = tid % HG, H = g.H, n = g.n;
    const int64_t s = ((int64_t)blockIdx.x * MB + tid) / HG;           // sequence = b * n + node
    const bool live = s < g.B * n && j < H;
    const int64_t b = live ? s / n : 0;
    const int node = live ? (int)(s - b * n) : 0;
    const int jj = j < H ? j : 0;
    float wr[HG], wz[HG], wn[HG];
#pragma unroll
    for (int k = 0; k < HG; ++k) {
        const bool ok = k < H;
        wr[k] = ok ? prm[g.off_whh + (jj) * H + k] : 0.f;
        wz[k] = ok ? prm[g.off_whh + (H + jj) * H + k] : 0.f;
        wn[k] = ok ? prm[g.off_whh + (2 * H + jj) * H + k] : 0.f;
    }
    float br = prm[g.off_bhh + jj], bz = prm[g.off_bhh + H + jj], bn = prm[g.off_bhh + 2 * H + jj];
    asm volatile("" : "+v"(br), "+v"(bz), "+v"(bn));
    const int base = tid - j;
    float h = 0.f;
    hs[tid] = 0.f;
#pragma unroll
    for (int k = 0; k < HG; ++k) asm volatile("" : "+v"(wr[k]), "+v"(wz[k]), "+v"(wn[k]));   // the compiler waits for its weight loads here
    // The HG lanes of a sequence sit in ONE wavefront: its LDS operations execute in order, so the exchange of h needs no workgroup
    // barrier -- and __syncthreads() also waits for vmcnt(0), i.e. for the step's store and the prefetched inputs: two memory round
    // trips per sequential step (0.75 us per step of 256).  Global traffic is asynchronous and counted by hand (async_mem.hpp): the
    // three gate inputs of a step are requested GRU_AHEAD steps ahead into a ring the wavefront reads back itself.
    __shared__ float ring[GRU_AHEAD][3][MB];
    const unsigned ring_wave = lds_address(&ring[0][0][tid & ~63]);
    const float* gsrc = gi + ((b * g.NP) * n + node) * g.H3 + jj;      // (lanes without work request sequence 0: the count must not vary)
    const int64_t gstep = (int64_t)n * g.H3;
    float* hdst = hseq + ((b * g.NP) * n + node) * H + jj;
    const int64_t hstep = (int64_t)n * H;
    int requested = 0;
    auto request = [&](int slot) {
        const unsigned dst = ring_wave + (unsigned)slot * (3 * MB * 4);
        dma_dword(gsrc, dst);
        dma_dword(gsrc + H, dst + MB * 4);
        dma_dword(gsrc + 2 * H, dst + 2 * MB * 4);
        if (++requested < g.NP) gsrc += gstep;
    };
    asm volatile("" : "+v"(h));
    wait_vm<0>();                                                      // (the weights above: the compiler's own loads)
#pragma unroll
    for (int i = 0; i < GRU_AHEAD; ++i) request(i);
    wait_vm<0>();
    for (int t0 = 0; t0 < g.NP; t0 += GRU_AHEAD) {
#pragma unroll
        for (int i = 0; i < GRU_AHEAD; ++i) {
            if (t0 + i >= g.NP) break;
            // issued after this step's requests: that step's store and (three requests, store) of the GRU_AHEAD - 1 steps since
            wait_vm<1 + 4 * (GRU_AHEAD - 1)>();
            const float gr = ring[i][0][tid], gz = ring[i][1][tid], gn = ring[i][2][tid];
            float ar = br, az = bz, an = bn;
#pragma unroll
            for (int k = 0; k < HG; ++k) {
                const float hk = hs[base + k];
                ar = fmaf(wr[k], hk, ar);
                az = fmaf(wz[k], hk, az);
                an = fmaf(wn[k], hk, an);
            }
            const float r = sigmoidf_(gr + ar), z = sigmoidf_(gz + az);
            const float c = tanhf_(gn + r * an);
            h = (1.0f - z) * c + z * h;
            hs[tid] = j < H ? h : 0.f;
            request(i);
            if (live) store_async(hdst, h);
            hdst += hstep;
        }
    }
}
template <int HG>
__global__ __launch_bounds__(MB) void msg_gru_backward_kernel(MsgGeom g, const float* __restrict__ gi,
                                                              const float* __restrict__ hseq, const float* __restrict__ prm,
                                                              const float* __restrict__ dpred, float* __restrict__ dgi,
                                                              float* __restrict__ gpart) {
    __shared__ float hp[MB];
    __shared__ float dg[3][MB];
    __shared__ float red[MB / 64][3 * 16 * 16 + 3 * 16];
    const int tid = threadIdx.x, j = tid % HG, H = g.H, n = g.n;
    const int64_t s = ((int64_t)blockIdx.x * MB + tid) / HG;
    const bool live = s < g.B * n && j < H;
    const int64_t b = live ? s / n : 0;
    const int node = live ? (int)(s - b * n) : 0;
    const int jj = j < H ? j : 0;
    float wr[HG], wz[HG], wn[HG], tr[HG], tz[HG], tn[HG], ar_[HG], az_[HG], an_[HG];
#pragma unroll
    for (int k = 0; k < HG; ++k) {
        const bool ok = k < H;
        wr[k] = ok ? prm[g.off_whh + (jj) * H + k] : 0.f;             // row jj of each gate block
        wz[k] = ok ? prm[g.off_whh + (H + jj) * H + k] : 0.f;
        wn[k] = ok ? prm[g.off_whh + (2 * H + jj) * H + k] : 0.f;
        tr[k] = ok ? prm[g.off_whh + (k) * H + jj] : 0.f;              // column jj
        tz[k] = ok ? prm[g.off_whh + (H + k) * H + jj] : 0.f;
        tn[k] = ok ? prm[g.off_whh + (2 * H + k) * H + jj] : 0.f;
        ar_[k] = az_[k] = an_[k] = 0.f;
    }
    float br = prm[g.off_bhh + jj], bz = prm[g.off_bhh + H + jj], bn = prm[g.off_bhh + 2 * H + jj];
    float abr = 0.f, abz = 0.f, abn = 0.f;
    const int base = tid - j;
    float dscale = live ? dpred[b] / (float)n : 0.f;
    float dh = 0.f;
    // (see the forward: the lanes of a sequence share a wavefront, no workgroup barrier; global traffic counted by hand.)  The tape of a
    // step -- h entering it, its three gate inputs, the head's weight for this unit -- is requested GRU_AHEAD steps ahead.
    __shared__ float ring[GRU_AHEAD][5][MB];
    const unsigned ring_wave = lds_address(&ring[0][0][tid & ~63]);
    const int64_t seq0 = (b * g.NP) * n + node;                        // row of (sequence, t = 0); lanes without work use sequence 0
    const int64_t gstep = (int64_t)n * g.H3, hstep = (int64_t)n * H;
    const float* gsrc = gi + (seq0 + (int64_t)(g.NP - 1) * n) * g.H3 + jj;
    const float* hsrc = hseq + (seq0 + (int64_t)(g.NP > 1 ? g.NP - 2 : 0) * n) * H + jj;      // h entering step t = h of step t - 1 (t = 0: unused)
    const float* fsrc = prm + g.off_fcw + (int64_t)(g.NP - 1) * H + jj;
    float* ddst = dgi + (seq0 + (int64_t)(g.NP - 1) * n) * g.H3 + jj;
    int requested = 0;
    auto request = [&](int slot) {                                     // the tape of step NP - 1 - requested (clamped at the end)
        const unsigned dst = ring_wave + (unsigned)slot * (5 * MB * 4);
        dma_dword(hsrc, dst);
        dma_dword(gsrc, dst + MB * 4);
        dma_dword(gsrc + H, dst + 2 * MB * 4);
        dma_dword(gsrc + 2 * H, dst + 3 * MB * 4);
        dma_dword(fsrc, dst + 4 * MB * 4);
        ++requested;
        if (requested < g.NP) { gsrc -= gstep; fsrc -= H; }
        if (requested + 1 < g.NP) hsrc -= hstep;
    };
#pragma unroll
    for (int k = 0; k < HG; ++k) asm volatile("" : "+v"(wr[k]), "+v"(wz[k]), "+v"(wn[k]), "+v"(tr[k]), "+v"(tz[k]), "+v"(tn[k]));
    asm volatile("" : "+v"(br), "+v"(bz), "+v"(bn), "+v"(dscale));       // the compiler waits for its own loads here
#pragma unroll
    for (int i = 0; i < GRU_AHEAD; ++i) request(i);
    wait_vm<0>();
    for (int t0 = g.NP - 1; t0 >= 0; t0 -= GRU_AHEAD) {
#pragma unroll
        for (int i = 0; i < GRU_AHEAD; ++i) {
            const int t = t0 - i;
            if (t < 0) break;
            // issued after this step's requests: that step's three stores and (five requests, three stores) of the GRU_AHEAD - 1 steps since
            wait_vm<3 + 8 * (GRU_AHEAD - 1)>();
            const float hprev = t > 0 ? ring[i][0][tid] : 0.f, gr = ring[i][1][tid], gz = ring[i][2][tid], gn = ring[i][3][tid];
            const float fw = ring[i][4][tid];
            hp[tid] = hprev;
            float ar = br, az = bz, an = bn;
#pragma unroll
            for (int k = 0; k < HG; ++k) {
                const float hk = hp[base + k];
                ar = fmaf(wr[k], hk, ar);
                az = fmaf(wz[k], hk, az);
                an = fmaf(wn[k], hk, an);
            }
            const float r = sigmoidf_(gr + ar), z = sigmoidf_(gz + az);
            const float c = tanhf_(gn + r * an);
            dh += live ? dscale * fw : 0.f;
            const float dn_pre = dh * (1.0f - z) * (1.0f - c * c);
            const float dz_pre = dh * (hprev - c) * z * (1.0f - z);
            const float dr_pre = dn_pre * an * r * (1.0f - r);
            const float dgn = dn_pre * r;
            request(i);
            if (live) {
                store_async(ddst, dr_pre);
                store_async(ddst + H, dz_pre);
                store_async(ddst + 2 * H, dn_pre);
            }
            ddst -= gstep;
            dg[0][tid] = live ? dr_pre : 0.f;
            dg[1][tid] = live ? dz_pre : 0.f;
            dg[2][tid] = live ? dgn : 0.f;
            float acc = dh * z;
#pragma unroll
            for (int k = 0; k < HG; ++k) {
                const float hk = hp[base + k];
                ar_[k] = fmaf(live ? dr_pre : 0.f, hk, ar_[k]);
                az_[k] = fmaf(live ? dz_pre : 0.f, hk, az_[k]);
                an_[k] = fmaf(live ? dgn : 0.f, hk, an_[k]);
                acc = fmaf(dg[0][base + k], tr[k], acc);
                acc = fmaf(dg[1][base + k], tz[k], acc);
                acc = fmaf(dg[2][base + k], tn[k], acc);
            }
            abr += live ? dr_pre : 0.f;
            abz += live ? dz_pre : 0.f;
            abn += live ? dgn : 0.f;
            dh = acc;
        }
    }
    wait_vm<0>();
    // reduce the per-lane accumulators over the sequences of this workgroup: butterfly across the lanes that hold the
    // same unit, then the four wavefronts in order.
    const int wave = tid / 64, lane = tid % 64;
    auto wave_sum = [&](float v) {
#pragma unroll
        for (int m = HG; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
        return v;
    };
    const int nW = 3 * H * H;
#pragma unroll
    for (int k = 0; k < HG; ++k) {
        const float a = wave_sum(ar_[k]), c = wave_sum(az_[k]), d = wave_sum(an_[k]);
        if (lane < HG && j < H && k < H) {
            red[wave][(j) * H + k] = a;
            red[wave][(H + j) * H + k] = c;
            red[wave][(2 * H + j) * H + k] = d;
        }
    }
    {
        const float a = wave_sum(abr), c = wave_sum(abz), d = wave_sum(abn);
        if (lane < HG && j < H) {
            red[wave][nW + j] = a;
            red[wave][nW + H + j] = c;
            red[wave][nW + 2 * H + j] = d;
        }
    }
    __syncthreads();
    float* dst = gpart + (int64_t)blockIdx.x * (nW + 3 * H);
    for (int e = tid; e < nW + 3 * H; e += MB) dst[e] = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
}

// ---------------------------------------------------------------------------------------------------
// head: mean over nodes, fc, squared error
// ---------------------------------------------------------------------------------------------------
constexpr int HEAD_MAX_PARTS = 16;
__global__ __launch_bounds__(MB) void msg_head_kernel(MsgGeom g, const float* __restrict__ hseq, const float* __restrict__ prm,
                                                      const float* __restrict__ y, float* __restrict__ pred,
                                                      float* __restrict__ pooled, float* __restrict__ dpred,
                                                      float* __restrict__ sqerr, float inv_gb, int parts, float* __restrict__ hpart) {
    // `parts` workgroups per sample (a sample's 26 MB / batch of hidden states walked by ONE workgroup was 57 us at batch 128): each
    // takes a slice of the (patch, unit) pairs and leaves its share of the dot product; msg_head_finish_kernel adds them in order
    __shared__ float red[MB];
    const int64_t b = blockIdx.x / parts;
    const int part = blockIdx.x - (int)b * parts;
    const int H = g.H, n = g.n, Q = g.NP * H;
    const int qper = (Q + parts - 1) / parts, q0 = part * qper, q1 = q0 + qper < Q ? q0 + qper : Q;
    float acc = 0.f;
    for (int q = q0 + threadIdx.x; q < q1; q += MB) {
        const int p = q / H, j = q - p * H;
        const float* src = hseq + ((b * g.NP + p) * n) * H + j;
        float s = 0.f;
        for (int i = 0; i < n; ++i) s += src[i * H];
        s = s / (float)n;
        pooled[b * Q + q] = s;
        acc = fmaf(s, prm[g.off_fcw + q], acc);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int m = MB / 2; m > 0; m >>= 1) {
        if (threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (parts > 1) {
            hpart[blockIdx.x] = red[0];
            return;
        }
        const float pr = red[0] + prm[g.off_fcb];
        pred[b] = pr;
        if (y) {
            const float d = pr - y[b];
            dpred[b] = 2.0f * d * inv_gb;
            sqerr[b] = d * d * inv_gb;
        }
    }
}
__global__ void msg_head_finish_kernel(MsgGeom g, const float* __restrict__ prm, const float* __restrict__ hpart, int parts,
                                       const float* __restrict__ y, float* __restrict__ pred, float* __restrict__ dpred,
                                       float* __restrict__ sqerr, float inv_gb) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= g.B) return;
    float pr = 0.f;
    for (int k = 0; k < parts; ++k) pr += hpart[b * parts + k];
    pr += prm[g.off_fcb];
    pred[b] = pr;
    if (y) {
        const float d = pr - y[b];
        dpred[b] = 2.0f * d * inv_gb;
        sqerr[b] = d * d * inv_gb;
    }
}

// ---------------------------------------------------------------------------------------------------
// GCN stack backward (+ GRU input projection backward)
// ---------------------------------------------------------------------------------------------------
template <int TW>
__global__ __launch_bounds__(MB) void msg_gcn_backward_kernel(MsgGeom g, const float* __restrict__ cat_in,
                                                              const float* __restrict__ dcat_in, const float* __restrict__ prm,
                                                              float* __restrict__ gpart) {
    // d cat = d gi W_ih arrives from a GEMM over all graphs (and d W_ih, d b_ih are GEMMs too, see msg_run): that keeps 19 KB
    // of weights / accumulators out of the LDS -- three workgroups per CU instead of two -- and a quarter of the per-graph FMAs.
    extern __shared__ float smem[];
    const int n = g.n, C = g.C, CS = g.CS, AXS = g.AXS, n1 = n + 1;
    const int nacc = g.gcn_params;
    float* w = smem;                               // GCN weights, row-major as in the flat buffer
    float* acc = w + g.gcn_params;                 // gradient accumulators of the GCN layers
    float* cat = acc + nacc;                       // [n][CS]
    float* dcat = cat + n * CS;                    // [n][CS]
    float* araw = dcat + n * CS;                   // [n][n+1]
    float* ahat = araw + n * n1;
    float* dah = ahat + n * n1;
    float* rv = dah + n * n1;                      // [32]
    float* ddv = rv + MAXN;                        // [32]
    float* ax = ddv + MAXN;                        // [n][AXS]  (AX, then dAX)
    const int tid = threadIdx.x;

    for (int e = tid; e < g.gcn_params; e += MB) w[e] = prm[e];
    for (int e = tid; e < nacc; e += MB) acc[e] = 0.f;
    __syncthreads();

    for (int64_t gi = blockIdx.x; gi < g.G; gi += gridDim.x) {
        const float* csrc = cat_in + gi * (int64_t)(n * C);
        const float* dsrc = dcat_in + gi * (int64_t)(n * C);
        for (int e = tid; e < n * C; e += MB) {
            const int i = e / C, c = e - i * C;
            cat[i * CS + c] = csrc[e];
            dcat[i * CS + c] = dsrc[e];
        }
        __syncthreads();
        for (int l = g.L - 1; l >= 0; --l) {
            const int fi = g.dims[l], fo = g.dims[l + 1], off = g.coff[l], offo = g.coff[l + 1];
            // dz = d out * leaky'(out), in place in the d cat slice of this layer's output
            for (int e = tid; e < n * fo; e += MB) {
                const int i = e / fo, o = e - i * fo;
                const float out = cat[i * CS + offo + o];
                dcat[i * CS + offo + o] *= (out > 0.f ? 1.f : LEAKY);
            }
            gram_plus_identity<TW>(cat, CS, off, fi, n, araw);
            __syncthreads();
            inv_sqrt_degree(araw, n, rv);
            __syncthreads();
            normalise(araw, rv, n, ahat);
            __syncthreads();
            aggregate<TW>(ahat, cat, CS, off, fi, n, ax, AXS);
            __syncthreads();
            // d W[o][k] += sum_i dz[i][o] AX[i][k] ; d b[o] += sum_i dz[i][o]
            {
                const int fq = (fi + TW - 1) / TW;
                for (int e = tid; e < fo * fq; e += MB) {
                    const int o = e / fq, k = TW * (e - o * fq);
                    const bool k1 = TW > 1 && k + 1 < fi, k2 = TW > 1 && k + 2 < fi, k3 = TW > 1 && k + 3 < fi;
                    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 8
                    for (int i = 0; i < n; ++i) {
                        const float d = dcat[i * CS + offo + o];
                        const float* xr = ax + i * AXS + k;
                        a0 = fmaf(d, xr[0], a0);
                        a1 = fmaf(d, k1 ? xr[1] : 0.f, a1);
                        a2 = fmaf(d, k2 ? xr[2] : 0.f, a2);
                        a3 = fmaf(d, k3 ? xr[3] : 0.f, a3);
                    }
                    float* dst = acc + g.woff[l] + o * fi + k;
                    dst[0] += a0;
                    if (k1) dst[1] += a1;
                    if (k2) dst[2] += a2;
                    if (k3) dst[3] += a3;
                }
            }
            for (int o = tid; o < fo; o += MB) {
                float a = 0.f;
                for (int i = 0; i < n; ++i) a += dcat[i * CS + offo + o];
                acc[g.boff[l] + o] += a;
            }
            if (l == 0) {                           // the SED features carry no gradient
                __syncthreads();
                break;
            }
            __syncthreads();
            // d AX = dz W   (overwrites AX)
            {
                const int fq = (fi + TW - 1) / TW;
                const float* wl = w + g.woff[l];
                for (int e = tid; e < n * fq; e += MB) {
                    const int i = e / fq, k = TW * (e - i * fq);
                    const bool k1 = TW > 1 && k + 1 < fi, k2 = TW > 1 && k + 2 < fi, k3 = TW > 1 && k + 3 < fi;
                    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 8
                    for (int o = 0; o < fo; ++o) {
                        const float d = dcat[i * CS + offo + o];
                        const float* wr = wl + o * fi + k;
                        a0 = fmaf(d, wr[0], a0);
                        a1 = fmaf(d, k1 ? wr[1] : 0.f, a1);
                        a2 = fmaf(d, k2 ? wr[2] : 0.f, a2);
                        a3 = fmaf(d, k3 ? wr[3] : 0.f, a3);
                    }
                    // AX of this graph is no longer needed (the d W loop above finished before the barrier)
                    float* dst = ax + i * AXS + k;
                    dst[0] = a0;
                    if (k1) dst[1] = a1;
                    if (k2) dst[2] = a2;
                    if (k3) dst[3] = a3;
                }
            }
            __syncthreads();
            // d ahat = d AX x^T
            {
                constexpr int TS = TW > 1 ? 2 : 1;
    const int nb = TW > 1 ? (n + 1) >> 1 : n;
                for (int e = tid; e < nb * nb; e += MB) {
                    const int i = TS * (e / nb), j = TS * (e % nb);
                    const int i1 = (TS > 1 && i + 1 < n) ? i + 1 : i, j1 = (TS > 1 && j + 1 < n) ? j + 1 : j;
                    const float* u0p = ax + i * AXS;
                    const float* u1p = ax + i1 * AXS;
                    const float* v0p = cat + j * CS + off;
                    const float* v1p = cat + j1 * CS + off;
                    float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;
#pragma unroll 8
                    for (int k = 0; k < fi; ++k) {
                        const float u0 = u0p[k], u1 = u1p[k], v0 = v0p[k], v1 = v1p[k];
                        a00 = fmaf(u0, v0, a00); a01 = fmaf(u0, v1, a01);
                        a10 = fmaf(u1, v0, a10); a11 = fmaf(u1, v1, a11);
                    }
                    dah[i * n1 + j] = a00;
                    if (TS > 1 && j + 1 < n) dah[i * n1 + j + 1] = a01;
                    if (TS > 1 && i + 1 < n) dah[(i + 1) * n1 + j] = a10;
                    if (TS > 1 && i + 1 < n && j + 1 < n) dah[(i + 1) * n1 + j + 1] = a11;
                }
            }
            __syncthreads();
            // d r (both D factors), then d(row sum): dd = -1/2 d^-3/2 dr = -1/2 r^3 dr
            if (tid < n) {
                const int i = tid;
                float a = 0.f;
                for (int j = 0; j < n; ++j) {
                    a = fmaf(dah[i * n1 + j] * araw[i * n1 + j], rv[j], a);
                    a = fmaf(dah[j * n1 + i] * araw[j * n1 + i], rv[j], a);
                }
                const float r = rv[i];
                ddv[i] = -0.5f * a * r * r * r;
            }
            __syncthreads();
            // M = dA' + dA'^T with dA'[i][j] = r_i d ahat[i][j] r_j + dd_i   (stored over A')
            for (int e = tid; e < n * n; e += MB) {
                const int i = e / n, j = e - i * n;
                araw[i * n1 + j] = rv[i] * rv[j] * (dah[i * n1 + j] + dah[j * n1 + i]) + ddv[i] + ddv[j];
            }
            __syncthreads();
            // d x = ahat^T d AX + M x, accumulated into the d cat slice of this layer's input
            {
                const int fq = (fi + TW - 1) / TW;
                for (int e = tid; e < n * fq; e += MB) {
                    const int i = e / fq, c = TW * (e - i * fq);
                    const bool k1 = TW > 1 && c + 1 < fi, k2 = TW > 1 && c + 2 < fi, k3 = TW > 1 && c + 3 < fi;
                    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 8
                    for (int j = 0; j < n; ++j) {
                        const float h = ahat[j * n1 + i], m = araw[i * n1 + j];
                        const float* dr = ax + j * AXS + c;
                        const float* xr = cat + j * CS + off + c;
                        a0 = fmaf(h, dr[0], a0);               a0 = fmaf(m, xr[0], a0);
                        a1 = fmaf(h, k1 ? dr[1] : 0.f, a1);    a1 = fmaf(m, k1 ? xr[1] : 0.f, a1);
                        a2 = fmaf(h, k2 ? dr[2] : 0.f, a2);    a2 = fmaf(m, k2 ? xr[2] : 0.f, a2);
                        a3 = fmaf(h, k3 ? dr[3] : 0.f, a3);    a3 = fmaf(m, k3 ? xr[3] : 0.f, a3);
                    }
                    float* dst = dcat + i * CS + off + c;
                    dst[0] += a0;
                    if (k1) dst[1] += a1;
                    if (k2) dst[2] += a2;
                    if (k3) dst[3] += a3;
                }
            }
            __syncthreads();
        }
    }
    float* dst = gpart + (int64_t)blockIdx.x * nacc;
    for (int e = tid; e < nacc; e += MB) dst[e] = acc[e];
}

// ---------------------------------------------------------------------------------------------------
// GCN backward on the fp32 matrix cores: ONE WAVEFRONT PER GRAPH (graphs of >= 12 nodes; round 3)
// ---------------------------------------------------------------------------------------------------
// Same two register forms as msg_features_mx_kernel.  What arrives from memory arrives in the FL form (coalesced rows of cat / d cat);
// the NL form of a matrix is its product with the identity on the matrix cores (exact: x 1 + 0s).  Per layer, top down:
//   dz = d out * leaky'(out)                          FL (loaded), NL = dz^T-as-rows x I
//   A', r, A^ recomputed from X (NL: Gram)            as in the forward
//   AX = A^ X                                          A = A^ (symmetric: rows on lanes), B = X (FL)          -> FL
//   d W = dz^T AX, d b = column sums                   A = dz (FL: rows = out features, k = nodes), B = AX (FL) -> into this wavefront's
//                                                      LDS accumulator (plain read-add-write: fixed order, reproducible)
//   d AX = dz W                                        A/B = dz (NL) and the W^T operand table, both orders   -> NL and FL
//   d A^ = d AX X^T and its transpose                  A/B = d AX (NL), X (NL), both orders
//   d r, d(row sum), M = d A' + d A'^T                 in the result layout, row factors through 32 floats of LDS
//   d x = A^ d AX + M X                                A = A^ / M (symmetric), B = d AX / X (FL)               -> FL, added to the d cat
//                                                      slice of the layer below when that layer loads it
struct MxBwdLds {
    int wt[MAXL], acc, wave, tr, total;  // W^T operand tables (layers >= 1), per-wavefront gradient accumulators, per-wavefront scratch
};
__host__ __device__ inline void mx_bwd_lds_layout(const MsgGeom& g, MxBwdLds* o) {
    int p = 0;
    for (int l = 0; l < g.L; ++l) {
        o->wt[l] = p;
        if (l > 0) p += ((g.dims[l] + 31) / 32) * ((g.dims[l + 1] + 31) / 32) * 16 * 64;
    }
    o->acc = p; p += MXW * g.gcn_params;
    o->wave = p; p += MXW * 64;
    o->tr = p; p += MXW * 32 * 33;                   // per-wavefront transpose tile (FL -> NL form of a 32 x 32 block)
    o->total = p;
}
static bool mx_backward_ok(const MsgGeom& g, size_t* lds_bytes) {
    if (g.n < 12) return false;
    MxBwdLds o;
    mx_bwd_lds_layout(g, &o);
    *lds_bytes = sizeof(float) * (size_t)o.total;
    return *lds_bytes <= 96 * 1024;
}

// (one workgroup per CU: the accumulator half of the unified register file then takes the spills instead of scratch memory, 3.21 -> 3.04 ms
// per step; requesting the next layer's tiles a layer ahead needs 96 more live registers and is 1.7 x SLOWER: 865 accvgpr moves per layer)
template <bool XJ>
__global__ __launch_bounds__(64 * MXW, 1) void msg_gcn_backward_mx_kernel(MsgGeom g, const float* __restrict__ cat_in,
                                                                        const float* __restrict__ dcat_in, const float* __restrict__ prm,
                                                                        float* __restrict__ gpart) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    MxBwdLds L_;
    mx_bwd_lds_layout(g, &L_);
    const int n = XJ ? MsgXJ::n : g.n, C = XJ ? MsgXJ::C : g.C;
    const int NL = XJ ? MsgXJ::L : g.L;
    auto gdim = [&](int l) { return XJ ? MsgXJ::dim(l) : g.dims[l]; };
    auto gcoff = [&](int l) { return XJ ? MsgXJ::coff(l) : g.coff[l]; };
    auto gwoff = [&](int l) { return XJ ? MsgXJ::woff(l) : g.woff[l]; };
    auto gboff = [&](int l) { return XJ ? MsgXJ::boff(l) : g.boff[l]; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, j = lane & 31;
    for (int l = 1; l < g.L; ++l) {                // d AX = dz W: lane (hh, k) of register m: W[o = f(m, hh)][32 ib + k]
        const int fi = g.dims[l], fo = g.dims[l + 1], nbi = (fi + 31) / 32, nbo = (fo + 31) / 32;
        for (int e = tid; e < nbi * nbo * 16 * 64; e += 64 * MXW) {
            const int ln = e & 63, m = (e >> 6) % (nbo * 16), ib = (e >> 6) / (nbo * 16);
            const int o = nl_feat(m, ln >> 5), k = 32 * ib + (ln & 31);
            smem[L_.wt[l] + e] = (o < fo && k < fi) ? prm[g.woff[l] + o * fi + k] : 0.f;
        }
    }
    for (int e = tid; e < MXW * g.gcn_params; e += 64 * MXW) smem[L_.acc + e] = 0.f;
    __syncthreads();
    float* acc = smem + L_.acc + wave * g.gcn_params;      // this wavefront's gradient accumulator
    float* rowv = smem + L_.wave + wave * 64;
    float* trt = smem + L_.tr + wave * (32 * 33);            // [node][33]: the FL -> NL form of a block goes through LDS (16 writes + 16
    // reads of a wavefront-private tile, in order: no barrier) instead of 16 matrix instructions against the identity (1024 cycles a block,
    // nine blocks a graph at the XJTU-SY shapes)
    auto fl_to_nl = [&](const float (&FL)[16], float* NLdst) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) trt[(krow(r, 0) + 4 * h) * 33 + j] = FL[r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int m = 0; m < 16; ++m) NLdst[m] = trt[j * 33 + krow(m, 0) + 4 * h];
    };

    float ident[16];                                         // I[krow(r, h)][j]
#pragma unroll
    for (int r = 0; r < 16; ++r) ident[r] = (krow(r, 0) + 4 * h == j) ? 1.f : 0.f;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    for (int64_t gi = (int64_t)blockIdx.x * MXW + wave; gi < g.G; gi += (int64_t)gridDim.x * MXW) {
        const float* cg = cat_in + gi * (int64_t)(n * C);
        const float* dg = dcat_in + gi * (int64_t)(n * C);
        // FL-form load of `width` columns starting at column `col0` of a [n][C] matrix (zero outside)
        // (every load is unconditional -- index clamped into the matrix, value selected afterwards: as `cond ? load : 0` each of the
        // 32 loads sat behind its own branch and paid its own memory latency, 2.2 ms for the whole kernel instead of 1.5)
        auto load_fl = [&](const float* base, int col0, int width, float (&dst)[2][16]) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                if (32 * b < width) {
                    const int c = 32 * b + j, cc = c < width ? c : width - 1;
                    float v[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int node = krow(r, 0) + 4 * h;
                        v[r] = base[(node < n ? node : n - 1) * C + col0 + cc];
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) dst[b][r] = (krow(r, 0) + 4 * h < n && c < width) ? v[r] : 0.f;
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) dst[b][r] = 0.f;
                }
            }
        };
        float dxF[2][16];                                   // d x of the layer above, FL form of the current layer's OUTPUT columns
        float outF[2][16];                                  // the current layer's output (= the X of the layer above)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) dxF[b][r] = 0.f;

#pragma unroll
        for (int l = NL - 1; l >= 0; --l) {
            const int fi = gdim(l), fo = gdim(l + 1), off = gcoff(l), offo = gcoff(l + 1);
            const int nbi = (fi + 31) / 32, nbo = (fo + 31) / 32;
            // dz (FL) = (d cat slice + d x from above) * leaky'(out)
            float dzF[2][16];
            {
                if (l == NL - 1) load_fl(cg, offo, fo, outF);        // (below the top layer: the X of the layer above, already here)
                load_fl(dg, offo, fo, dzF);
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        // (d x of a PADDED node row is not zero -- M[pad][j] = dd_j -- so the mask is applied here, not only in the loads)
                        const bool ok = krow(r, 0) + 4 * h < n && 32 * b + j < fo;
                        dzF[b][r] = ok ? (dzF[b][r] + dxF[b][r]) * (outF[b][r] > 0.f ? 1.f : LEAKY) : 0.f;
                    }
            }
            // X in both forms
            float XF[2][16], XN[32];
            load_fl(cg, off, fi, XF);
#pragma unroll
            for (int m = 0; m < 32; ++m) XN[m] = 0.f;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                if (b < nbi) fl_to_nl(XF[b], &XN[16 * b]);
            }
            // A' = X X^T + I, r, A^
            f32x16 G = zero16;
#pragma unroll
            for (int m = 0; m < 32; ++m)
                if (nl_feat(m, 0) < fi) G = mfma32(XN[m], XN[m], G);
            float d = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                G[r] += ident[r];
                d += G[r];
            }
            d += __shfl_xor(d, 32, 64);
            const float rj = 1.0f / sqrtf(d);
            __builtin_amdgcn_wave_barrier();
            if (h == 0) rowv[j] = rj;
            __builtin_amdgcn_wave_barrier();
            float rr[16], Ah[16];
            row_values(rowv, h, rr);
#pragma unroll
            for (int r = 0; r < 16; ++r) Ah[r] = (G[r] * rj) * rr[r];
            // AX (FL), d W, d b
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) {
                if (ib < nbi) {
                    f32x16 ax = zero16;
#pragma unroll
                    for (int r = 0; r < 16; ++r) ax = mfma32(Ah[r], XF[ib][r], ax);          // rows = nodes: FL form
#pragma unroll
                    for (int ob = 0; ob < 2; ++ob) {
                        if (ob < nbo) {
                            f32x16 dw = zero16;
#pragma unroll
                            for (int r = 0; r < 16; ++r) dw = mfma32(dzF[ob][r], ax[r], dw);  // rows = out features, columns = in features
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int o = 32 * ob + krow(r, 0) + 4 * h, k = 32 * ib + j;
                                if (o < fo && k < fi) acc[gwoff(l) + o * fi + k] += dw[r];
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) {
                if (ob < nbo) {
                    float sb = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) sb += dzF[ob][r];
                    sb += __shfl_xor(sb, 32, 64);
                    if (h == 0 && 32 * ob + j < fo) acc[gboff(l) + 32 * ob + j] += sb;
                }
            }
            if (l == 0) break;                               // the SED features carry no gradient
            // dz in NL form
            float dzN[32];
#pragma unroll
            for (int m = 0; m < 32; ++m) dzN[m] = 0.f;
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) {
                if (ob < nbo) fl_to_nl(dzF[ob], &dzN[16 * ob]);
            }
            // d AX = dz W in both forms
            float dAXN[32], dAXF[2][16];
#pragma unroll
            for (int m = 0; m < 32; ++m) dAXN[m] = 0.f;
            const float* wt = smem + L_.wt[l];
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) {
#pragma unroll
                for (int r = 0; r < 16; ++r) dAXF[ib][r] = 0.f;
                if (ib < nbi) {
                    f32x16 aF = zero16;
#pragma unroll
                    for (int m = 0; m < 32; ++m) {
                        if (nl_feat(m, 0) < fo) {
                            const float w = wt[(ib * nbo * 16 + m) * 64 + lane];
                            aF = mfma32(dzN[m], w, aF);        // rows = nodes, columns = in features: FL
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) dAXF[ib][r] = aF[r];
                    fl_to_nl(dAXF[ib], &dAXN[16 * ib]);       // (the NL form through the LDS tile: it was the same products a second time)
                }
            }
            // S = d A^ + d A^^T
            // (S = T + T^T with T = d AX X^T: the transpose of the 32 x 32 result through the LDS tile -- result layout in, result layout
            // out -- instead of the same products a second time with the operands swapped)
            f32x16 S = zero16;
#pragma unroll
            for (int m = 0; m < 32; ++m)
                if (nl_feat(m, 0) < fi) S = mfma32(dAXN[m], XN[m], S);
            {
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 16; ++r) trt[(krow(r, 0) + 4 * h) * 33 + j] = S[r];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 16; ++r) S[r] += trt[j * 33 + krow(r, 0) + 4 * h];
            }
            // d r (both D factors) and d(row sum): dd = -1/2 r^3 sum_j A'[i][j] r_j S[i][j]   (A', S symmetric)
            float a1 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) a1 = fmaf(G[r] * rr[r], S[r], a1);
            a1 += __shfl_xor(a1, 32, 64);
            const float ddj = -0.5f * a1 * rj * rj * rj;
            __builtin_amdgcn_wave_barrier();
            if (h == 0) rowv[32 + j] = ddj;
            __builtin_amdgcn_wave_barrier();
            float ddr[16], Mm[16];
            row_values(rowv + 32, h, ddr);
#pragma unroll
            for (int r = 0; r < 16; ++r) Mm[r] = (rr[r] * rj) * S[r] + ddr[r] + ddj;
            // d x = A^ d AX + M X  (FL), handed to the layer below
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) {
#pragma unroll
                for (int r = 0; r < 16; ++r) dxF[ib][r] = 0.f;
                if (ib < nbi) {
                    f32x16 t = zero16;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        t = mfma32(Ah[r], dAXF[ib][r], t);
                        t = mfma32(Mm[r], XF[ib][r], t);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) dxF[ib][r] = t[r];
                }
            }
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) outF[b][r] = XF[b][r];
        }
    }
    __syncthreads();
    float* dst = gpart + (int64_t)blockIdx.x * g.gcn_params;
    for (int e = tid; e < g.gcn_params; e += 64 * MXW) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < MXW; ++w) v += smem[L_.acc + w * g.gcn_params + e];
        dst[e] = v;
    }
}

static size_t gcn_backward_lds_bytes(const MsgGeom& g) {
    return sizeof(float) * ((size_t)2 * g.gcn_params + 2 * g.n * g.CS + 3 * g.n * (g.n + 1) + 2 * MAXN + g.n * g.AXS);
}

// ---------------------------------------------------------------------------------------------------
// backward of the GRU input projection gi = cat W_ih^T + b_ih over all (graph, node) rows in ONE pass:
//   d cat = d gi W_ih,   d W_ih = d gi^T cat,   d b_ih = column sums of d gi.
// Two generic GEMM launches read d gi and the 318-MB cat twice at 2.3 TB/s (172 us each + 80 us of reductions at XJTU-SY c1 / batch
// 128).  Here a lane owns TWO columns of cat (W_ih's columns in 2 x 3H registers for the whole kernel, coalesced row loads / stores,
// packed FMAs) and a wavefront walks rows; the row of d gi every lane needs comes as LDS broadcast reads from a block the workgroup
// staged with coalesced loads (a first version fed it through the scalar cache: 412 us -- streaming data misses that cache line by
// line).  The column behind the last carries the constant 1: its accumulators are d b_ih.  One partial row per workgroup, reduced in
// fixed order by msg_finalize.
// ---------------------------------------------------------------------------------------------------
typedef float gi_f2 __attribute__((ext_vector_type(2)));
// acc.xy += a.xy * s.x  /  a.xy * s.y : the packed FMA with one half of its second source feeding both lanes (op_sel) -- written out
// because the compiler builds the (s, s) pair with moves instead
__device__ __forceinline__ void pk_fma_lo(gi_f2& acc, gi_f2 a, gi_f2 s) { asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(a), "v"(s)); }
__device__ __forceinline__ void pk_fma_hi(gi_f2& acc, gi_f2 a, gi_f2 s) { asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(a), "v"(s)); }
constexpr int GIB = 256;                 // four wavefronts = four row streams
constexpr int GI_ROWS = 64;              // rows of d gi per staged block (16 per wavefront)
constexpr int GI_MAX_PARTS = 1024;
template <int H3T>
__global__ __launch_bounds__(GIB) void msg_gi_backward_kernel(int C, const float* __restrict__ dgi, const float* __restrict__ cat,
                                                              const float* __restrict__ wih, float* __restrict__ dcat, float* __restrict__ gpart,
                                                              int64_t rows) {
    constexpr int BLK4 = GI_ROWS * H3T / 4;                            // float4s per block
    __shared__ float4 blk[2][BLK4];
    __shared__ float red[3][2 * H3T][64];                              // the other wavefronts' accumulators at the end
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (uniform: row addresses stay scalar)
    const int c0 = 2 * lane, c1 = c0 + 1;
    // columns beyond the matrix: loads are clamped to a valid column and x = v * keep + one  (the column behind the last is the constant 1)
    const int l0 = c0 < C ? c0 : C - 1, l1 = c1 < C ? c1 : C - 1;
    const float k0 = c0 < C ? 1.f : 0.f, k1 = c1 < C ? 1.f : 0.f, o0 = c0 == C ? 1.f : 0.f, o1 = c1 == C ? 1.f : 0.f;
    gi_f2 w[H3T], acc[H3T];
#pragma unroll
    for (int h = 0; h < H3T; ++h) {
        w[h] = gi_f2{c0 < C ? wih[h * C + c0] : 0.f, c1 < C ? wih[h * C + c1] : 0.f};
        acc[h] = gi_f2{0.f, 0.f};
    }
    const int64_t per = ((rows + gridDim.x - 1) / gridDim.x + GI_ROWS - 1) / GI_ROWS * GI_ROWS;       // whole blocks per workgroup
    const int64_t R0 = (int64_t)blockIdx.x * per, R1 = R0 + per < rows ? R0 + per : rows;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch = [&](int64_t base, int k) {                            // float4 number tid + 256 k of the block that starts at row `base`
        const int i = tid + GIB * k;
        const int64_t e = base * H3T + 4 * (int64_t)i;                 // rows are contiguous: the block is one span of floats
        return (i < BLK4 && e + 3 < R1 * H3T) ? *reinterpret_cast<const float4*>(dgi + e)
                                              : (i < BLK4 && e < R1 * H3T ? make_float4(dgi[e], e + 1 < R1 * H3T ? dgi[e + 1] : 0.f,
                                                                                       e + 2 < R1 * H3T ? dgi[e + 2] : 0.f, 0.f) : zero4);
    };
    constexpr int NF = (BLK4 + GIB - 1) / GIB;
    float4 nxt[NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) nxt[k] = fetch(R0, k);
    int buf = 0;
    for (int64_t base = R0; base < R1; base += GI_ROWS, buf ^= 1) {
#pragma unroll
        for (int k = 0; k < NF; ++k)
            if (tid + GIB * k < BLK4) blk[buf][tid + GIB * k] = nxt[k];
        __syncthreads();                                               // (the other buffer was read two blocks ago: one barrier per block)
        if (base + GI_ROWS < R1) {
#pragma unroll
            for (int k = 0; k < NF; ++k) nxt[k] = fetch(base + GI_ROWS, k);
        }
        // this wavefront's 16 rows: their cat pairs first (16 loads in flight), then row by row
        const int64_t r0 = base + 16 * wave;
        gi_f2 x[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float* rp = cat + (r0 + j < R1 ? r0 + j : R1 - 1) * C;        // uniform
            x[j] = gi_f2{fmaf(rp[l0], k0, o0), fmaf(rp[l1], k1, o1)};
        }
        // the broadcast reads of row j + 1 are issued in front of row j's products (a wavefront issues in order and only two fit a
        // SIMD: read-then-multiply per row left the LDS and the VALU taking turns, 144 us of the kernel's 240)
        float4 sq[2][H3T / 4];
        {
            const float4* sr = &blk[buf][(16 * wave) * (H3T / 4)];
#pragma unroll
            for (int q = 0; q < H3T / 4; ++q) sq[0][q] = sr[q];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (r0 + j >= R1) break;
            if (j + 1 < 16) {
                const float4* sr = &blk[buf][(16 * wave + j + 1) * (H3T / 4)];
#pragma unroll
                for (int q = 0; q < H3T / 4; ++q) sq[(j + 1) & 1][q] = sr[q];
            }
            gi_f2 d = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < H3T / 4; ++q) {
                const float4 s4 = sq[j & 1][q];                        // the same address in every lane: a broadcast read
                const gi_f2 s01 = {s4.x, s4.y}, s23 = {s4.z, s4.w};
                pk_fma_lo(d, w[4 * q], s01);      pk_fma_lo(acc[4 * q], x[j], s01);
                pk_fma_hi(d, w[4 * q + 1], s01);  pk_fma_hi(acc[4 * q + 1], x[j], s01);
                pk_fma_lo(d, w[4 * q + 2], s23);  pk_fma_lo(acc[4 * q + 2], x[j], s23);
                pk_fma_hi(d, w[4 * q + 3], s23);  pk_fma_hi(acc[4 * q + 3], x[j], s23);
            }
            float* dr = dcat + (r0 + j) * C;
            if (c0 < C) dr[c0] = d.x;
            if (c1 < C) dr[c1] = d.y;
            asm volatile("" ::: "memory");                             // (keeps the reads from being hoisted further: 24 registers a row)
        }
    }
    // the four row streams of the workgroup -> one partial row  [d W_ih (3H x C) | d b_ih (3H)]
    __syncthreads();
    if (wave > 0) {
#pragma unroll
        for (int h = 0; h < H3T; ++h) {
            red[wave - 1][2 * h][lane] = acc[h].x;
            red[wave - 1][2 * h + 1][lane] = acc[h].y;
        }
    }
    __syncthreads();
    if (wave == 0) {
        float* dst = gpart + (int64_t)blockIdx.x * (H3T * C + H3T);
#pragma unroll
        for (int h = 0; h < H3T; ++h) {
            const float ax = ((acc[h].x + red[0][2 * h][lane]) + red[1][2 * h][lane]) + red[2][2 * h][lane];
            const float ay = ((acc[h].y + red[0][2 * h + 1][lane]) + red[1][2 * h + 1][lane]) + red[2][2 * h + 1][lane];
            if (c0 < C) dst[h * C + c0] = ax;
            else if (c0 == C) dst[H3T * C + h] = ax;
            if (c1 < C) dst[h * C + c1] = ay;
            else if (c1 == C) dst[H3T * C + h] = ay;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// finalize: partial rows -> flat gradient; fc gradients; loss
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(MB) void msg_finalize_kernel(MsgGeom g, const float* __restrict__ gpart_gcn, int rows_gcn,
                                                          const float* __restrict__ gpart_gru, int rows_gru,
                                                          const float* __restrict__ gpart_gi, int rows_gi,
                                                          const float* __restrict__ dpred, const float* __restrict__ pooled,
                                                          float* __restrict__ grads) {
    // one wavefront per value: lanes stride over the partial rows (or the batch), fixed-order butterfly (deterministic).
    // W_ih and b_ih come from msg_gi_backward_kernel's partial rows, or (rows_gi == 0) were written by the GEMMs of msg_run.
    const int e = (blockIdx.x * MB + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int ngru = g.H3 * g.H + g.H3, ngi = g.H3 * g.C + g.H3;
    const bool is_wih = e >= g.off_wih && e < g.off_whh, is_bih = e >= g.off_bih && e < g.off_bhh;
    if (e >= g.nparam || ((is_wih || is_bih) && rows_gi == 0)) return;
    float a = 0.f;
    if (e < g.off_wih) {                                       // GCN layers
        for (int r = lane; r < rows_gcn; r += 64) a += gpart_gcn[(int64_t)r * g.gcn_params + e];
    } else if (is_wih) {
        for (int r = lane; r < rows_gi; r += 64) a += gpart_gi[(int64_t)r * ngi + (e - g.off_wih)];
    } else if (is_bih) {
        for (int r = lane; r < rows_gi; r += 64) a += gpart_gi[(int64_t)r * ngi + g.H3 * g.C + (e - g.off_bih)];
    } else if (e < g.off_bih) {                                // W_hh
        for (int r = lane; r < rows_gru; r += 64) a += gpart_gru[(int64_t)r * ngru + (e - g.off_whh)];
    } else if (e < g.off_fcw) {                                // b_hh
        for (int r = lane; r < rows_gru; r += 64) a += gpart_gru[(int64_t)r * ngru + g.H3 * g.H + (e - g.off_bhh)];
    } else if (e < g.off_fcb) {                                // fc.weight
        const int q = e - g.off_fcw, Q = g.NP * g.H;
        for (int64_t b = lane; b < g.B; b += 64) a = fmaf(dpred[b], pooled[b * Q + q], a);
    } else {                                                   // fc.bias
        for (int64_t b = lane; b < g.B; b += 64) a += dpred[b];
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) a += __shfl_xor(a, m, 64);
    if (lane == 0) grads[e] = a;
}

__global__ void msg_fill_kernel(float* p, int n, float v) {
    if ((int)threadIdx.x < n) p[threadIdx.x] = v;
}

struct MsgWs {
    size_t cat, gi, hseq, dgi, dcat, one, split, pooled, dpred, sqerr, hpart, gpart_gcn, gpart_gru, gpart_gi, total;
    int rows_gcn_max, rows_gru, HG;
};

static void msg_ws_layout(const MsgGeom& g, MsgWs* w) {
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t rows = (size_t)g.G * g.n;
    size_t o = 0;
    w->cat = o; o = al(o + rows * g.C * sizeof(float));
    w->gi = o; o = al(o + rows * g.H3 * sizeof(float));
    w->hseq = o; o = al(o + rows * g.H * sizeof(float));
    w->dgi = o; o = al(o + rows * g.H3 * sizeof(float));
    w->dcat = o; o = al(o + rows * g.C * sizeof(float));
    w->one = o; o = al(o + 64 * sizeof(float));
    {
        const size_t s1 = sgemm_splitk_need_floats(g.H3, g.C, (int)rows), s2 = sgemm_splitk_need_floats(g.H3, 1, (int)rows);
        w->split = o; o = al(o + (s1 > s2 ? s1 : s2) * sizeof(float));
    }
    w->pooled = o; o = al(o + (size_t)g.B * g.NP * g.H * sizeof(float));
    w->dpred = o; o = al(o + (size_t)g.B * sizeof(float));
    w->sqerr = o; o = al(o + (size_t)g.B * sizeof(float));
    w->hpart = o; o = al(o + (size_t)g.B * HEAD_MAX_PARTS * sizeof(float));
    w->HG = g.H <= 4 ? 4 : (g.H <= 8 ? 8 : 16);
    w->rows_gcn_max = 1024;
    w->rows_gru = (int)(((size_t)g.B * g.n * w->HG + MB - 1) / MB);
    if (w->rows_gru < 1) w->rows_gru = 1;
    w->gpart_gcn = o; o = al(o + (size_t)w->rows_gcn_max * g.gcn_params * sizeof(float));
    w->gpart_gru = o; o = al(o + (size_t)w->rows_gru * (g.H3 * g.H + g.H3) * sizeof(float));
    w->gpart_gi = o; o = al(o + (size_t)GI_MAX_PARTS * (g.H3 * g.C + g.H3) * sizeof(float));
    w->total = o;
}

template <typename K>
static int resident_grid(K kernel, int64_t items, size_t lds, int cap_rows) {
    int dev = 0, cus = 256, per_cu = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, MB, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    int64_t want = (int64_t)cus * per_cu;
    if (want > items) want = items;
    if (want > cap_rows) want = cap_rows;
    if (want < 1) want = 1;
    return (int)want;
}

template <int TW>
static int launch_features_tw(const MsgGeom& g, const float* x, const float* prm, float* cat, float* gi, hipStream_t st) {
    const size_t lds = features_lds_bytes(g);
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(msg_features_kernel<TW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
        return RULGNN_EHIP;
    const int grid = resident_grid(msg_features_kernel<TW>, g.G, lds, 1 << 20);
    hipLaunchKernelGGL(msg_features_kernel<TW>, dim3(grid), dim3(MB), lds, st, g, x, prm, cat, gi);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}
static int launch_features(const MsgGeom& g, const float* x, const float* prm, float* cat, float* gi, hipStream_t st) {
    size_t lds = 0;
    if (mx_features_ok(g, &lds)) {                  // graphs of >= 12 nodes: one wavefront per graph on the fp32 matrix cores
        auto go = [&](auto kern) -> int {
            if (lds > 48 * 1024 &&
                hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return RULGNN_EHIP;
            int dev = 0, cus = 256, per_cu = 0;
            if (hipGetDevice(&dev) == hipSuccess) {
                int v = 0;
                if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
            }
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 64 * MXW, lds) != hipSuccess || per_cu < 1) per_cu = 1;
            int64_t grid = (int64_t)cus * per_cu;
            const int64_t need = (g.G + MXW - 1) / MXW;
            if (grid > need) grid = need;
            hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * MXW), lds, st, g, x, prm, cat, gi);
            return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
        };
        return MsgXJ::matches(g) ? go(&msg_features_mx_kernel<true>) : go(&msg_features_mx_kernel<false>);   // (the XJTU-SY layer shapes as constants)
    }
    return g.n >= 12 ? launch_features_tw<4>(g, x, prm, cat, gi, st) : launch_features_tw<1>(g, x, prm, cat, gi, st);
}

template <int TW>
static int launch_gcn_backward(const MsgGeom& g, int rows_max, const float* cat, const float* dcat, const float* prm, float* gpart,
                               hipStream_t st, int* rows_out) {
    const size_t lds = gcn_backward_lds_bytes(g);
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(msg_gcn_backward_kernel<TW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
        return RULGNN_EHIP;
    const int rows = resident_grid(msg_gcn_backward_kernel<TW>, g.G, lds, rows_max);
    hipLaunchKernelGGL(msg_gcn_backward_kernel<TW>, dim3(rows), dim3(MB), lds, st, g, cat, dcat, prm, gpart);
    *rows_out = rows;
    return RULGNN_OK;
}

template <int HG>
static void launch_gru(const MsgGeom& g, const MsgWs& w, char* ws, const float* prm, bool backward, hipStream_t st) {
    const int grid = w.rows_gru;
    if (!backward)
        hipLaunchKernelGGL(msg_gru_forward_kernel<HG>, dim3(grid), dim3(MB), 0, st, g, (const float*)(ws + w.gi), prm,
                           (float*)(ws + w.hseq));
    else
        hipLaunchKernelGGL(msg_gru_backward_kernel<HG>, dim3(grid), dim3(MB), 0, st, g, (const float*)(ws + w.gi),
                           (const float*)(ws + w.hseq), prm, (const float*)(ws + w.dpred), (float*)(ws + w.dgi),
                           (float*)(ws + w.gpart_gru));
}

static void dispatch_gru(const MsgGeom& g, const MsgWs& w, char* ws, const float* prm, bool backward, hipStream_t st) {
    if (w.HG == 4) launch_gru<4>(g, w, ws, prm, backward, st);
    else if (w.HG == 8) launch_gru<8>(g, w, ws, prm, backward, st);
    else launch_gru<16>(g, w, ws, prm, backward, st);
}

}  // namespace

int64_t stmsgcn_param_count(const rulgnn_stmsgcn_shape* s) {
    MsgGeom g;
    return msg_geometry(s, &g) == RULGNN_OK ? g.nparam : -1;
}

size_t stmsgcn_workspace_bytes(const rulgnn_stmsgcn_shape* s) {
    MsgGeom g;
    if (msg_geometry(s, &g) != RULGNN_OK) return 0;
    MsgWs w;
    msg_ws_layout(g, &w);
    return w.total;
}

int stmsgcn_features(const rulgnn_stmsgcn_shape* s, const float* x, const float* prm, float* features, hipStream_t st) {
    MsgGeom g;
    const int rc = msg_geometry(s, &g);
    if (rc != RULGNN_OK) return rc;
    if (g.B == 0) return RULGNN_OK;
    (void)hipGetLastError();
    return launch_features(g, x, prm, features, nullptr, st);
}

// mode bit 0: forward, bit 1: backward
int stmsgcn_run(const rulgnn_stmsgcn_shape* s, const rulgnn_stmsgcn_args* a, int mode, hipStream_t st) {
    MsgGeom g;
    int rc = msg_geometry(s, &g);
    if (rc != RULGNN_OK) return rc;
    MsgWs w;
    msg_ws_layout(g, &w);
    if (a->workspace_bytes < w.total) return RULGNN_EWORKSPACE;
    char* ws = static_cast<char*>(a->workspace);
    (void)hipGetLastError();
    const float inv_gb = 1.0f / (float)(a->global_batch > 0 ? a->global_batch : g.B);
    if (mode & 1) {
        rc = launch_features(g, a->x, a->params, (float*)(ws + w.cat), (float*)(ws + w.gi), st);
        if (rc != RULGNN_OK) return rc;
        dispatch_gru(g, w, ws, a->params, false, st);
        {
            // enough workgroups to fill the chip: a sample's (patch, unit) pairs in up to HEAD_MAX_PARTS slices
            int parts = 1;
            while (parts < HEAD_MAX_PARTS && g.B * parts < 1024 && g.NP * g.H / (2 * parts) >= MB) parts *= 2;
            hipLaunchKernelGGL(msg_head_kernel, dim3((unsigned)(g.B * parts)), dim3(MB), 0, st, g, (const float*)(ws + w.hseq), a->params,
                               a->y, a->pred, (float*)(ws + w.pooled), (float*)(ws + w.dpred), (float*)(ws + w.sqerr), inv_gb, parts,
                               (float*)(ws + w.hpart));
            if (parts > 1)
                hipLaunchKernelGGL(msg_head_finish_kernel, dim3((unsigned)((g.B + 255) / 256)), dim3(256), 0, st, g, a->params,
                                   (const float*)(ws + w.hpart), parts, a->y, a->pred, (float*)(ws + w.dpred), (float*)(ws + w.sqerr), inv_gb);
        }
    }
    if (mode & 2) {
        if (a->dpred) {
            if (hipMemcpyAsync(ws + w.dpred, a->dpred, sizeof(float) * g.B, hipMemcpyDeviceToDevice, st) != hipSuccess)
                return RULGNN_EHIP;
        }
        dispatch_gru(g, w, ws, a->params, true, st);
        // GRU input projection backward over all (graph, node) rows: d cat = d gi W_ih, d W_ih = d gi^T cat, d b_ih = column sums of d gi
        int rows_gi = 0;
        {
            const int64_t R = g.G * g.n;
            const float* dgi = (const float*)(ws + w.dgi);
            if (g.H3 == 24 && g.C < 128) {                          // the reference's GRU width (hidden 8): one streaming pass
                int64_t parts = (R + 4 * GI_ROWS - 1) / (4 * GI_ROWS);
                if (parts > GI_MAX_PARTS) parts = GI_MAX_PARTS;
                rows_gi = (int)parts;
                hipLaunchKernelGGL(msg_gi_backward_kernel<24>, dim3((unsigned)parts), dim3(GIB), 0, st, g.C, dgi, (const float*)(ws + w.cat),
                                   a->params + g.off_wih, (float*)(ws + w.dcat), (float*)(ws + w.gpart_gi), R);
            } else {
                const float* wih = a->params + g.off_wih;
                float* one = (float*)(ws + w.one);
                float* split = (float*)(ws + w.split);
                hipLaunchKernelGGL(msg_fill_kernel, dim3(1), dim3(64), 0, st, one, 64, 1.0f);
                rc = sgemm(dgi, g.H3, 1, wih, 1, g.C, (float*)(ws + w.dcat), g.C, (int)R, g.C, g.H3, false, st);
                if (rc != RULGNN_OK) return rc;
                rc = sgemm_splitk(dgi, 1, g.H3, (const float*)(ws + w.cat), 1, g.C, a->grads + g.off_wih, g.C, g.H3, g.C, (int)R, false, split, st);
                if (rc != RULGNN_OK) return rc;
                rc = sgemm_splitk(dgi, 1, g.H3, one, 0, 0, a->grads + g.off_bih, 1, g.H3, 1, (int)R, false, split, st);
                if (rc != RULGNN_OK) return rc;
            }
        }
        int rows = 0;
        size_t lds_mx = 0;
        int rcb;
        if (mx_backward_ok(g, &lds_mx)) {                  // graphs of >= 12 nodes: one wavefront per graph on the fp32 matrix cores
            auto launch_mx = [&](auto kern) -> int {
                if (lds_mx > 48 * 1024 &&
                    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mx) != hipSuccess)
                    return RULGNN_EHIP;
                int dev = 0, cus = 256, per_cu = 0;
                if (hipGetDevice(&dev) == hipSuccess) {
                    int v = 0;
                    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
                }
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 64 * MXW, lds_mx) != hipSuccess || per_cu < 1) per_cu = 1;
                int64_t grid = (int64_t)cus * per_cu;
                const int64_t need = (g.G + MXW - 1) / MXW;
                if (grid > need) grid = need;
                if (grid > w.rows_gcn_max) grid = w.rows_gcn_max;
                rows = (int)grid;
                hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * MXW), lds_mx, st, g, (const float*)(ws + w.cat),
                                   (const float*)(ws + w.dcat), a->params, (float*)(ws + w.gpart_gcn));
                return RULGNN_OK;
            };
            const int rl = MsgXJ::matches(g) ? launch_mx(&msg_gcn_backward_mx_kernel<true>) : launch_mx(&msg_gcn_backward_mx_kernel<false>);
            if (rl != RULGNN_OK) return rl;
            rcb = hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
        } else
        rcb = g.n >= 12 ? launch_gcn_backward<4>(g, w.rows_gcn_max, (const float*)(ws + w.cat), (const float*)(ws + w.dcat), a->params,
                                                           (float*)(ws + w.gpart_gcn), st, &rows)
                                  : launch_gcn_backward<1>(g, w.rows_gcn_max, (const float*)(ws + w.cat), (const float*)(ws + w.dcat), a->params,
                                                           (float*)(ws + w.gpart_gcn), st, &rows);
        if (rcb != RULGNN_OK) return rcb;
        const bool mse = a->dpred == nullptr;
        hipLaunchKernelGGL(msg_finalize_kernel, dim3((g.nparam + 3) / 4), dim3(MB), 0, st, g,
                           (const float*)(ws + w.gpart_gcn), rows, (const float*)(ws + w.gpart_gru), w.rows_gru,
                           (const float*)(ws + w.gpart_gi), rows_gi, (const float*)(ws + w.dpred), (const float*)(ws + w.pooled), a->grads);
        if (mse && a->loss)
            (void)block_sum((const float*)(ws + w.sqerr), (int64_t)g.B, a->loss, st);
    }
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

}  // namespace rulgnn
