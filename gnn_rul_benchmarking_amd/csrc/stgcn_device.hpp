// Device-side building blocks of the fused ST_GCN kernels (gfx950 / CDNA4, wave64).
//
// Work decomposition ("row" mapping): a wavefront holds 64/RW samples, one per RW-lane row;
// lane t of a row owns patch t (the axis the TCN convolves over and theta mixes).  The ten
// statistic channels of a patch live in ten VGPRs of that lane.  With RW = 16 every cross-lane
// step is a DPP modifier on a VALU instruction: the causal shift is `row_shr`, the A.X.W
// projection broadcasts lane k with `row_newbcast`, and row sums are quad_perm / row_mirror
// butterflies -- no LDS traffic and no barriers inside a sample.
//
// Math follows reference models/ST_GCN/Model.py (line numbers cited at each block).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rulgnn {

constexpr int F = 10;          // statistic channels == graph nodes (Model.py:201 in_features = 10)
constexpr int NPAIR = 55;      // upper triangle of the symmetric 10x10 Pearson adjacency
constexpr int CONVW = 2 * F * F;   // Conv1d(10,10,k=2) weights
constexpr float LEAKY = 0.01f;
constexpr float BN_EPS = 1e-5f;
constexpr int WAVES_PER_BLOCK = 4;
constexpr int BLOCK = 64 * WAVES_PER_BLOCK;

// packed upper-triangular index of a symmetric 10x10 matrix
__host__ __device__ constexpr int sym(int a, int b) {
    return (a < b ? a : b) * F - (a < b ? a : b) * ((a < b ? a : b) - 1) / 2 + ((a < b ? b : a) - (a < b ? a : b));
}

// flat parameter layout (see include/rulgnn.h).  K = MPNN order (Model.py:74-79: theta is a ModuleList of K Linear(N, N), stored as
// weight | bias per order, in named_parameters() order); every K defaults to 1, the reference's wiring and the only order the
// matrix-core and the tiled kernels take.
constexpr int MAX_MPNN_ORDER = 3;
__host__ __device__ constexpr int layer_stride(int N, int K = 1) { return K * (N * N + N) + 2 * (CONVW + 2 * F); }
__host__ __device__ constexpr int off_theta_w(int N, int kk = 0) { return kk * (N * N + N); }
__host__ __device__ constexpr int off_theta_b(int N, int kk = 0) { return kk * (N * N + N) + N * N; }
__host__ __device__ constexpr int off_conv_w(int N, int blk, int K = 1) { return K * (N * N + N) + blk * (CONVW + 2 * F); }
__host__ __device__ constexpr int off_bn_g(int N, int blk, int K = 1) { return off_conv_w(N, blk, K) + CONVW; }
__host__ __device__ constexpr int off_bn_b(int N, int blk, int K = 1) { return off_conv_w(N, blk, K) + CONVW + F; }
__host__ __device__ constexpr int off_fc1_w(int N, int L, int K = 1) { return L * layer_stride(N, K); }
__host__ __device__ constexpr int off_fc1_b(int N, int L, int K = 1) { return off_fc1_w(N, L, K) + N * N; }
__host__ __device__ constexpr int off_fc2_w(int N, int L, int K = 1) { return off_fc1_b(N, L, K) + N; }
__host__ __device__ constexpr int off_fc2_b(int N, int L, int K = 1) { return off_fc2_w(N, L, K) + N; }
__host__ __device__ constexpr int param_count(int N, int L, int K = 1) { return off_fc2_b(N, L, K) + 1; }

// ---------------------------------------------------------------------------------------------
// DPP primitives.  bound_ctrl = true: lanes shifted in from outside the 16-lane row read 0,
// which is exactly the causal zero padding of Conv1d(padding=(k-1)d)+Chomp1d (Model.py:92-98).
// ---------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
constexpr int DPP_QUAD_XOR1 = 0xB1;      // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;      // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_SHL = 0x100;
constexpr int DPP_ROW_SHR = 0x110;
constexpr int DPP_ROW_MIRROR = 0x140;
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_NEWBCAST = 0x150;  // gfx90a+: broadcast lane n of each row to the row

// acc[c] += src[c]@(lane K of my row) * w  for the ten channels: ten v_fmac_f32_dpp.
// hipcc's DPP combiner folds v_mov_dpp into add/mul but not into fmac, hence the asm.
// The leading s_nop covers the VALU-write -> DPP-read hazard (2 wait states) for whatever
// instruction the scheduler placed in front (hipcc pads nothing inside an asm statement).
template <int K>
__device__ __forceinline__ void fmac10_rowbcast(float (&acc)[F], const float (&src)[F], float w) {
    asm("s_nop 1\n\t"
        "v_fmac_f32_dpp %0, %10, %20 row_newbcast:%21 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %11, %20 row_newbcast:%21 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %12, %20 row_newbcast:%21 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %3, %13, %20 row_newbcast:%21 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %4, %14, %20 row_newbcast:%21 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %5, %15, %20 row_newbcast:%21 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %6, %16, %20 row_newbcast:%21 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %7, %17, %20 row_newbcast:%21 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %8, %18, %20 row_newbcast:%21 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %9, %19, %20 row_newbcast:%21 row_mask:0xf bank_mask:0xf"
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]),
          "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9])
        : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]), "v"(src[4]), "v"(src[5]), "v"(src[6]), "v"(src[7]),
          "v"(src[8]), "v"(src[9]), "v"(w), "n"(K));
}
// single-channel form (fc1: pooled[t] broadcast)
template <int K>
__device__ __forceinline__ void fmac1_rowbcast(float& acc, float src, float w) {
    asm("s_nop 1\n\t"
        "v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
        : "+v"(acc)
        : "v"(src), "v"(w), "n"(K));
}

// ---------------------------------------------------------------------------------------------
// Row<RW>: cross-lane operations inside one RW-lane row (= one sample)
// ---------------------------------------------------------------------------------------------
template <int RW>
struct Row {
    static_assert(RW == 16 || RW == 32 || RW == 64, "row width");
    static constexpr int SPW = 64 / RW;   // samples per wavefront

    // sum over the row, result in every lane of the row
    static __device__ __forceinline__ float allsum(float v) {
        v += dpp<DPP_QUAD_XOR1>(v);
        v += dpp<DPP_QUAD_XOR2>(v);
        v += dpp<DPP_ROW_HALF_MIRROR>(v);
        v += dpp<DPP_ROW_MIRROR>(v);
        if constexpr (RW >= 32) v += __shfl_xor(v, 16, 64);
        if constexpr (RW >= 64) v += __shfl_xor(v, 32, 64);
        return v;
    }
    // value of lane (t - D) of my row, 0 for t < D      (causal tap, Model.py:134-160)
    template <int D>
    static __device__ __forceinline__ float shr(float v, int t) {
        if constexpr (RW == 16) {
            return dpp<DPP_ROW_SHR + D>(v);
        } else {
            float r = __shfl_up(v, D, RW);
            return t >= D ? r : 0.f;
        }
    }
    // value of lane (t + D) of my row, 0 past the end of the row  (transpose of the causal tap)
    template <int D>
    static __device__ __forceinline__ float shl(float v, int t) {
        if constexpr (RW == 16) {
            return dpp<DPP_ROW_SHL + D>(v);
        } else {
            float r = __shfl_down(v, D, RW);
            return t + D < RW ? r : 0.f;
        }
    }
    // acc[c] += (src[c] of lane k) * wrow[k], k = 0..RW-1 (RW == 16: unrolled DPP broadcast;
    // wider rows: ds_bpermute).  wrow points at this lane's zero-padded weight row in LDS.
    static __device__ __forceinline__ void project10(float (&acc)[F], const float (&src)[F], const float* wrow, int n) {
        if constexpr (RW == 16) {
            const float4* w4 = reinterpret_cast<const float4*>(wrow);
            float4 a = w4[0], b = w4[1], c = w4[2], d = w4[3];
            fmac10_rowbcast<0>(acc, src, a.x);
            fmac10_rowbcast<1>(acc, src, a.y);
            fmac10_rowbcast<2>(acc, src, a.z);
            fmac10_rowbcast<3>(acc, src, a.w);
            fmac10_rowbcast<4>(acc, src, b.x);
            fmac10_rowbcast<5>(acc, src, b.y);
            fmac10_rowbcast<6>(acc, src, b.z);
            fmac10_rowbcast<7>(acc, src, b.w);
            fmac10_rowbcast<8>(acc, src, c.x);
            fmac10_rowbcast<9>(acc, src, c.y);
            fmac10_rowbcast<10>(acc, src, c.z);
            fmac10_rowbcast<11>(acc, src, c.w);
            fmac10_rowbcast<12>(acc, src, d.x);
            fmac10_rowbcast<13>(acc, src, d.y);
            fmac10_rowbcast<14>(acc, src, d.z);
            fmac10_rowbcast<15>(acc, src, d.w);
        } else {
            for (int k = 0; k < n; ++k) {
                const float w = wrow[k];
#pragma unroll
                for (int c = 0; c < F; ++c) acc[c] = fmaf(__shfl(src[c], k, RW), w, acc[c]);
            }
        }
    }
    static __device__ __forceinline__ void project1(float& acc, float src, const float* wrow, int n) {
        if constexpr (RW == 16) {
            const float4* w4 = reinterpret_cast<const float4*>(wrow);
            float4 a = w4[0], b = w4[1], c = w4[2], d = w4[3];
            fmac1_rowbcast<0>(acc, src, a.x);
            fmac1_rowbcast<1>(acc, src, a.y);
            fmac1_rowbcast<2>(acc, src, a.z);
            fmac1_rowbcast<3>(acc, src, a.w);
            fmac1_rowbcast<4>(acc, src, b.x);
            fmac1_rowbcast<5>(acc, src, b.y);
            fmac1_rowbcast<6>(acc, src, b.z);
            fmac1_rowbcast<7>(acc, src, b.w);
            fmac1_rowbcast<8>(acc, src, c.x);
            fmac1_rowbcast<9>(acc, src, c.y);
            fmac1_rowbcast<10>(acc, src, c.z);
            fmac1_rowbcast<11>(acc, src, c.w);
            fmac1_rowbcast<12>(acc, src, d.x);
            fmac1_rowbcast<13>(acc, src, d.y);
            fmac1_rowbcast<14>(acc, src, d.z);
            fmac1_rowbcast<15>(acc, src, d.w);
        } else {
            for (int k = 0; k < n; ++k) acc = fmaf(__shfl(src, k, RW), wrow[k], acc);
        }
    }
};

// LDS row stride (floats) of a zero-padded [RW][RW] weight matrix: RW + 4 keeps 16-B alignment
// and makes the per-lane ds_read_b128 of consecutive rows hit distinct bank groups.
template <int RW>
__host__ __device__ constexpr int wstride() { return RW + 4; }

// NaN-propagating activations (torch.relu / F.leaky_relu propagate NaN; fmaxf would not)
__device__ __forceinline__ float relu(float v) { return v < 0.f ? 0.f : v; }
__device__ __forceinline__ float leaky(float v) { return v > 0.f ? v : v * LEAKY; }

// ---------------------------------------------------------------------------------------------
// Patch statistics -- Model.py:7-52.  `pp` = this lane's patch in LDS (P floats).
// Two passes like the reference (mean first, then central moments).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void patch_statistics(const float* pp, int P, float (&st)[F]) {
    float s = 0.f, sq = 0.f, sa = 0.f, mx = -INFINITY, mn = INFINITY;
    if ((P & 1) == 0) {
        const float2* p2 = reinterpret_cast<const float2*>(pp);
#pragma unroll 5
        for (int i = 0; i < P / 2; ++i) {
            const float2 v = p2[i];
            s += v.x + v.y;
            sq = fmaf(v.x, v.x, sq);
            sq = fmaf(v.y, v.y, sq);
            sa += fabsf(v.x) + fabsf(v.y);
            mx = fmaxf(mx, fmaxf(v.x, v.y));
            mn = fminf(mn, fminf(v.x, v.y));
        }
    } else {
        for (int i = 0; i < P; ++i) {
            const float v = pp[i];
            s += v;
            sq = fmaf(v, v, sq);
            sa += fabsf(v);
            mx = fmaxf(mx, v);
            mn = fminf(mn, v);
        }
    }
    const float invP = 1.0f / (float)P;
    const float mean = s * invP;
    float m2 = 0.f, m3 = 0.f, m4 = 0.f;
    if ((P & 1) == 0) {
        const float2* p2 = reinterpret_cast<const float2*>(pp);
#pragma unroll 5
        for (int i = 0; i < P / 2; ++i) {
            const float2 v = p2[i];
            const float d0 = v.x - mean, d1 = v.y - mean;
            const float q0 = d0 * d0, q1 = d1 * d1;
            m2 += q0 + q1;
            m3 = fmaf(q0, d0, m3);
            m3 = fmaf(q1, d1, m3);
            m4 = fmaf(q0, q0, m4);
            m4 = fmaf(q1, q1, m4);
        }
    } else {
        for (int i = 0; i < P; ++i) {
            const float d = pp[i] - mean;
            const float q = d * d;
            m2 += q;
            m3 = fmaf(q, d, m3);
            m4 = fmaf(q, q, m4);
        }
    }
    const float var = m2 / (float)(P - 1);      // torch.var default: unbiased
    const float sd = sqrtf(var);
    const float isd = 1.0f / sd;                // sd == 0 -> inf; 0 * inf = NaN like the reference's 0/0
    const float isd2 = isd * isd;
    st[0] = mx;
    st[1] = mn;
    st[2] = mx - mn;
    st[3] = var;
    st[4] = sd;
    st[5] = mean;
    st[6] = sqrtf(sq * invP);
    st[7] = sa * invP;
    st[8] = (m3 * invP) * (isd2 * isd);            // mean(((x-mu)/sd)^3)
    st[9] = (m4 * invP) * (isd2 * isd2) - 3.0f;    // mean(((x-mu)/sd)^4) - 3
}

// v_max_f32 / v_max3_f32 as written: fmaxf() adds a canonicalising v_max x, x, x per operand (IEEE mode), and these run at
// half rate.  NaN is dropped by the hardware max unless every operand is NaN; callers deal with that where it matters.
__device__ __forceinline__ float vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ float vmin3(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Patch statistics (Model.py:7-52) with the patch held in registers: one pass of LDS reads serves both passes of the
// arithmetic, and max / min take two elements per (half-rate) instruction.  Same formulas as patch_statistics().
// FAST: hardware sqrt / rcp (1 ulp; denormal inputs flush, so a variance below 1e-38 gives sd = 0 -> non-finite skew / kurtosis --
// only for callers that recompute non-finite results exactly); otherwise the IEEE sequences of patch_statistics().
template <int P, bool FAST>
__device__ __forceinline__ void patch_statistics_regs(const float* pp, float (&st)[F]) {
    // No contraction: fused into fma(-s, 1/P, x) the deviation from the mean of a CONSTANT patch is a rounding residue instead of
    // the exact 0 that makes the reference's skew / kurtosis 0/0 = NaN (Model.py:41-52) -- caught by the NaN fixture.
#pragma clang fp contract(off)
    static_assert(P % 2 == 0 && P <= 64, "even patch sizes that fit the register budget");
    float v[P];
    const float2* p2 = reinterpret_cast<const float2*>(pp);
#pragma unroll
    for (int i = 0; i < P / 2; ++i) { const float2 q = p2[i]; v[2 * i] = q.x; v[2 * i + 1] = q.y; }
    float s = 0.f, sq = 0.f, sa = 0.f, mx = -INFINITY, mn = INFINITY;
#pragma unroll
    for (int i = 0; i < P; i += 2) {
        s += v[i] + v[i + 1];
        sq = fmaf(v[i], v[i], sq);
        sq = fmaf(v[i + 1], v[i + 1], sq);
        sa += __builtin_fabsf(v[i]) + __builtin_fabsf(v[i + 1]);
        mx = vmax3(mx, v[i], v[i + 1]);
        mn = vmin3(mn, v[i], v[i + 1]);
    }
    const float invP = 1.0f / (float)P;
    const float mean = s * invP;
    float m2 = 0.f, m3 = 0.f, m4 = 0.f;
#pragma unroll
    for (int i = 0; i < P; i += 2) {
        const float d0 = v[i] - mean, d1 = v[i + 1] - mean;
        const float q0 = d0 * d0, q1 = d1 * d1;
        m2 += q0 + q1;
        m3 = fmaf(q0, d0, m3);
        m3 = fmaf(q1, d1, m3);
        m4 = fmaf(q0, q0, m4);
        m4 = fmaf(q1, q1, m4);
    }
    float var, sd, isd;                            // sd == 0 -> isd = inf; 0 * inf = NaN like the reference's 0/0
    if constexpr (FAST) {
        var = m2 * (1.0f / (float)(P - 1));
        sd = __builtin_amdgcn_sqrtf(var);
        isd = __builtin_amdgcn_rcpf(sd);
    } else {
        var = m2 / (float)(P - 1);
        sd = sqrtf(var);
        isd = 1.0f / sd;
    }
    const float isd2 = isd * isd;
    st[0] = mx;
    st[1] = mn;
    st[2] = mx - mn;
    st[3] = var;
    st[4] = sd;
    st[5] = mean;
    st[6] = FAST ? __builtin_amdgcn_sqrtf(sq * invP) : sqrtf(sq * invP);
    st[7] = sa * invP;
    st[8] = (m3 * invP) * (isd2 * isd);
    st[9] = (m4 * invP) * (isd2 * isd2) - 3.0f;
}


// ---------------------------------------------------------------------------------------------
// Pearson adjacency between the ten statistic rows -- Model.py:53-71.
// X0[c] is zero in padded lanes (t >= N).  Output: packed symmetric A, uniform over the row.
// ---------------------------------------------------------------------------------------------
template <int RW>
__device__ __forceinline__ void pearson_adjacency(const float (&X0)[F], bool lane_valid, int N, float (&A)[NPAIR]) {
    float C[F];
    const float invN = 1.0f / (float)N;
#pragma unroll
    for (int c = 0; c < F; ++c) {
        const float mean = Row<RW>::allsum(X0[c]) * invN;
        C[c] = lane_valid ? X0[c] - mean : 0.f;
    }
#pragma unroll
    for (int a = 0; a < F; ++a)
#pragma unroll
        for (int b = a; b < F; ++b) A[sym(a, b)] = Row<RW>::allsum(C[a] * C[b]);
    float nrm[F];
#pragma unroll
    for (int c = 0; c < F; ++c) nrm[c] = sqrtf(A[sym(c, c)]);
#pragma unroll
    for (int a = 0; a < F; ++a)
#pragma unroll
        for (int b = a; b < F; ++b)      // dot / (|a||b|): v_rcp_f32 (1 ulp); 0 * rcp(0) = 0 * inf = NaN like the reference's 0/0
            A[sym(a, b)] = A[sym(a, b)] * __builtin_amdgcn_rcpf(nrm[a] * nrm[b]);
}

// AX[c] = sum_c' A[c][c'] X[c']   (torch.bmm(A, X), Model.py:87)
__device__ __forceinline__ void adj_aggregate(const float (&A)[NPAIR], const float (&X)[F], float (&AX)[F]) {
#pragma unroll
    for (int a = 0; a < F; ++a) {
        float acc = 0.f;
#pragma unroll
        for (int b = 0; b < F; ++b) acc = fmaf(A[sym(a, b)], X[b], acc);
        AX[a] = acc;
    }
}

// Wave-uniform weights are read with scalar loads.  Left alone the compiler hoists those loads out of the persistent tile loop
// and shares them between the forward recompute and the transposed convolution of a G phase; 200 live scalars do not fit the
// 106-SGPR file, so they end up in VGPR lanes (v_writelane / v_readlane: up to 20 % of a tile iteration).  scalar_fresh makes one
// use site re-load instead: the pointer passes through an opaque scalar asm that depends on a loop-variant token (no hoisting;
// the ID keeps two sites from being merged) and is read in the constant address space, which keeps the loads on the scalar
// unit without the read-only proof that the laundering throws away.
typedef const float __attribute__((address_space(4)))* scalar_f32p;
template <int ID>
__device__ __forceinline__ scalar_f32p scalar_fresh(const float* p, int token) {
    unsigned long long a = reinterpret_cast<unsigned long long>(p);
    asm("; scalar_fresh %2" : "+s"(a) : "v"(token), "n"(ID));
    return (scalar_f32p)a;
}

template <bool FRESH, int ID>
__device__ __forceinline__ auto conv_weights(const float* p, int token) {
    if constexpr (FRESH) return scalar_fresh<ID>(p, token); else return p;
}

// z[co] = sum_ci w[co][ci][0] h[ci][t-D] + w[co][ci][1] h[ci][t]; weights are wave-uniform
// (scalar loads -> SGPR operands of v_fmac).
template <int RW, int D, typename WP>
__device__ __forceinline__ void causal_conv(const float (&h)[F], WP w, int t, float (&z)[F]) {
    float hs[F];
#pragma unroll
    for (int c = 0; c < F; ++c) hs[c] = Row<RW>::template shr<D>(h[c], t);
#pragma unroll
    for (int co = 0; co < F; ++co) {
        float acc = 0.f;
#pragma unroll
        for (int ci = 0; ci < F; ++ci) {
            acc = fmaf(w[(co * F + ci) * 2 + 0], hs[ci], acc);
            acc = fmaf(w[(co * F + ci) * 2 + 1], h[ci], acc);
        }
        z[co] = acc;
    }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Convolution rows with the weights in LDS: out[r] = sum_q w[r][2q] * sh[q] + w[r][2q + 1] * pl[q], 20 weights per row.
//   forward conv (block's natural layout [co][ci][tap]):  r = co, sh = input shifted right by the dilation, pl = input;
//   transposed conv (layout [ci][co][tap], built in the prologue): r = ci, sh = dz shifted left, pl = dz.
// The 20 weights of a row arrive through one inline-asm batch of five uniform-address ds_read_b128 + one wait, chained on the
// previous row's result.  Asm on purpose: as plain loads the compiler lifts all 50 loop-invariant reads out of the persistent
// tile loop (+200 live VGPRs -> scratch), and unchained batches bunch up with the same effect (profiles/r01_ubench_gfx950.md).
// Versus scalar-operand weights: no SGPR pressure (a third of the G_{2l} tile loop was v_readlane / v_writelane spill traffic)
// and VGPR-operand FMAs issue at twice the rate of SGPR-operand ones.
__device__ __forceinline__ void conv_rows_lds(const float (&sh)[F], const float (&pl)[F], const float* wl, float (&out)[F]) {
    static_assert(F == 10, "five 16-byte reads per row");
    const uint32_t base = (uint32_t)(uintptr_t)wl;        // LDS byte address (low 32 bits of the shared-window pointer)
    float chain = 0.f;
#pragma unroll
    for (int r = 0; r < F; ++r) {
        // two batches per row (12 + 8 weights): 8 fewer live VGPRs than one batch of 20 -- what G2 needed to fit 168 registers
        f32x4 w0, w1, w2, w3, w4;
        const uint32_t addr = base + r * 2 * F * 4;
        asm volatile("ds_read_b128 %[w0], %[ad]\n\t"
                     "ds_read_b128 %[w1], %[ad] offset:16\n\t"
                     "ds_read_b128 %[w2], %[ad] offset:32\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : [w0] "=&v"(w0), [w1] "=&v"(w1), [w2] "=&v"(w2), [ch] "+v"(chain)
                     : [ad] "v"(addr));
        float acc = 0.f;
        acc = fmaf(w0[0], sh[0], acc); acc = fmaf(w0[1], pl[0], acc); acc = fmaf(w0[2], sh[1], acc); acc = fmaf(w0[3], pl[1], acc);
        acc = fmaf(w1[0], sh[2], acc); acc = fmaf(w1[1], pl[2], acc); acc = fmaf(w1[2], sh[3], acc); acc = fmaf(w1[3], pl[3], acc);
        acc = fmaf(w2[0], sh[4], acc); acc = fmaf(w2[1], pl[4], acc); acc = fmaf(w2[2], sh[5], acc); acc = fmaf(w2[3], pl[5], acc);
        asm volatile("ds_read_b128 %[w3], %[ad] offset:48\n\t"
                     "ds_read_b128 %[w4], %[ad] offset:64\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : [w3] "=&v"(w3), [w4] "=&v"(w4), [ac] "+v"(acc)
                     : [ad] "v"(addr));
        acc = fmaf(w3[0], sh[6], acc); acc = fmaf(w3[1], pl[6], acc); acc = fmaf(w3[2], sh[7], acc); acc = fmaf(w3[3], pl[7], acc);
        acc = fmaf(w4[0], sh[8], acc); acc = fmaf(w4[1], pl[8], acc); acc = fmaf(w4[2], sh[9], acc); acc = fmaf(w4[3], pl[9], acc);
        out[r] = acc;
        chain = acc;
    }
    out[F - 1] = chain;
}
template <int RW, int D>
__device__ __forceinline__ void causal_conv_lds(const float (&h)[F], const float* wl, int t, float (&z)[F]) {
    float hs[F];
#pragma unroll
    for (int c = 0; c < F; ++c) hs[c] = Row<RW>::template shr<D>(h[c], t);
    conv_rows_lds(hs, h, wl, z);
}
// ---------------------------------------------------------------------------------------------
// 4-block MFMA building blocks (row width 16 only).
//
// v_mfma_f32_16x16x1_4b_f32 computes, for each of the four 16-lane blocks b INDEPENDENTLY,
// D_b[i][j] += A_b[i] * B_b[j]: with one sample per 16-lane row, one instruction is an outer-product
// update for all four samples of the wavefront at once.  Measured layout (tools/probe_mfma4b.hip):
// acc[4b + r] in lane (g = lane>>4, j = lane&15) is D_b[4g + r][j]; bringing it back to the row mapping
// (sample b in lane row b, matrix row in the register index) is a 4x4 transpose between register index
// and lane row, done with v_permlane32_swap + v_permlane16_swap (4 swaps per group of four registers).
// (A conv and a theta projection built on this were measured slower than the DPP/scalar-weight
// versions and are not kept: profiles/r01_ubench_gfx950.md.)
// ---------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void transpose_rows4(float& v0, float& v1, float& v2, float& v3) {
    // in: register b, lane row g  ->  out: register g, lane row b.
    // Inline asm on purpose: with __builtin_amdgcn_permlane{16,32}_swap hipcc (ROCm 7.2) returned the
    // FIRST result for both elements of the second-stage swaps (tools/probe: stores v2 twice).  The
    // s_nops cover the VALU-write -> permlane-read wait states hipcc itself inserts around these ops.
    asm("s_nop 1\n\t"
        "v_permlane32_swap_b32 %0, %2\n\t"
        "v_permlane32_swap_b32 %1, %3\n\t"
        "s_nop 1\n\t"
        "v_permlane16_swap_b32 %0, %1\n\t"
        "v_permlane16_swap_b32 %2, %3\n\t"
        "s_nop 1"
        : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
}

// ---------------------------------------------------------------------------------------------
// Pearson adjacency and A.X on the matrix cores (row width 16).
//
// The adjacency of a sample is kept LANE-DISTRIBUTED: ten registers, Arow[c] holding A[c][c'] in lane
// c' of the sample's 16-lane row (instead of 55 row-uniform registers).  That form is exactly the
// A operand of v_mfma_f32_16x16x1_4b_f32, so
//     AX[c][t] = sum_c' A[c'][c] X[c'][t]   (A symmetric)
// is ten 4-block MFMAs (A = Arow[c'], B = the row-mapped X[c']) plus the register<->lane-row transpose,
// and the Pearson Gram matrix itself is G[c][c'] = sum_t C[c][t] C[c'][t] = sixteen MFMAs with
// A = B = CT[t], the centred statistics TRANSPOSED inside the row (lane = channel, register = patch;
// one pass through a per-wave LDS tile).  Row means become in-lane sums.  This replaces 65 four-step
// DPP butterflies (4 cycles per DPP op on gfx950) and frees 45 VGPRs.
// ---------------------------------------------------------------------------------------------
constexpr int PT_STRIDE = 20;                       // LDS row stride of the [sample*10 + c][t] transpose tile
constexpr int PT_FLOATS = 4 * F * PT_STRIDE;        // per wavefront

__device__ __forceinline__ void acc_to_rows(const f32x16& acc, float (&out)[F]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float v0 = acc[r], v1 = acc[4 + r], v2 = acc[8 + r], v3 = acc[12 + r];
        transpose_rows4(v0, v1, v2, v3);
        out[r] = v0;
        out[4 + r] = v1;
        if (8 + r < F) out[8 + r] = v2;
    }
}

// X0[c]: row-mapped statistics, zero in padded lanes / padded sample rows.  pt: per-wave LDS tile.
__device__ __forceinline__ void pearson_rows_mfma(const float (&X0)[F], bool rowok, int N, float* pt, int lane, float (&Arow)[F]) {
    const int srow = lane >> 4, cl = lane & 15;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < F; ++c) pt[(srow * F + c) * PT_STRIDE + cl] = X0[c];
    __builtin_amdgcn_wave_barrier();
    // lane (sample, channel cl) reads its channel's 16 patch values
    const float4* r4 = reinterpret_cast<const float4*>(pt + (srow * F + (cl < F ? cl : 0)) * PT_STRIDE);
    const float4 q0 = r4[0], q1 = r4[1], q2 = r4[2], q3 = r4[3];
    float CT[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) sum += CT[k];                 // patches >= N are zero
    const float mean = sum * (1.0f / (float)N);
    const bool chan_ok = cl < F;
#pragma unroll
    for (int k = 0; k < 16; ++k) CT[k] = (chan_ok && k < N) ? CT[k] - mean : 0.f;
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x1f32(CT[k], CT[k], acc, 0, 0, 0);
    acc_to_rows(acc, Arow);                                     // Arow[c] lane c' = G[c][c']
    float diag = 0.f;
#pragma unroll
    for (int c = 0; c < F; ++c) diag = (cl == c) ? Arow[c] : diag;
    const float nl = sqrtf(diag);                               // |C_c'| in lane c'
    float nb[F];
    nb[0] = dpp<DPP_ROW_NEWBCAST + 0>(nl); nb[1] = dpp<DPP_ROW_NEWBCAST + 1>(nl); nb[2] = dpp<DPP_ROW_NEWBCAST + 2>(nl);
    nb[3] = dpp<DPP_ROW_NEWBCAST + 3>(nl); nb[4] = dpp<DPP_ROW_NEWBCAST + 4>(nl); nb[5] = dpp<DPP_ROW_NEWBCAST + 5>(nl);
    nb[6] = dpp<DPP_ROW_NEWBCAST + 6>(nl); nb[7] = dpp<DPP_ROW_NEWBCAST + 7>(nl); nb[8] = dpp<DPP_ROW_NEWBCAST + 8>(nl);
    nb[9] = dpp<DPP_ROW_NEWBCAST + 9>(nl);
#pragma unroll
    for (int c = 0; c < F; ++c) {
        // dot / (|a||b|): v_rcp_f32 (1 ulp); 0 * rcp(0) = 0 * inf = NaN like the reference's 0/0.
        const float v = Arow[c] * __builtin_amdgcn_rcpf(nb[c] * nl);
        Arow[c] = (chan_ok && rowok) ? v : 0.f;                 // padded channels / padded sample rows stay finite
    }
}

// AX[c] = sum_c' A[c][c'] X[c']  (torch.bmm(A, X), Model.py:87) with the lane-distributed adjacency
__device__ __forceinline__ void adj_aggregate_mfma(const float (&Arow)[F], const float (&X)[F], float (&AX)[F]) {
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < F; ++c) acc = __builtin_amdgcn_mfma_f32_16x16x1f32(Arow[c], X[c], acc, 0, 0, 0);
    acc_to_rows(acc, AX);
}

// 32-bit mix shared bit-for-bit with oracle/stgcn_oracle.py::_lowbias32
__device__ __forceinline__ uint32_t lowbias32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x7FEB352Du;
    h ^= h >> 15;
    h *= 0x846CA68Bu;
    h ^= h >> 16;
    return h;
}

// exact i / d for i < 2^20 given magic = ceil(2^32 / d)
__device__ __forceinline__ uint32_t fastdiv(uint32_t i, uint32_t magic) { return __umulhi(i, magic); }

// ---------------------------------------------------------------------------------------------
// Cooperative, coalesced copy of one wavefront's tile (ns samples = ns*N patches of P floats,
// contiguous in HBM) into LDS with patch stride Ppad.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void stage_tile(const float* __restrict__ g, float* stage, int total, int P, int Ppad,
                                           uint32_t magicP, bool vec4, int lane) {
    if (Ppad == P) {
        if (vec4) {
            // CH loads are issued back to back (clamped index, no branch around the load) before the
            // first LDS store: the rolled load->wait->store loop exposed one HBM latency per float4.
            constexpr int CH = 4;
            float4* s4 = reinterpret_cast<float4*>(stage);
            const int n4 = total >> 2;
            for (int base = lane; base < n4; base += CH * 64) {
                float4 r[CH];
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const int i = base + u * 64;
                    // the windows are read once per step: nontemporal loads (F0 alone 64 -> 60 us)
                    const f32x4 nv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g) + (i < n4 ? i : n4 - 1));
                    r[u] = make_float4(nv[0], nv[1], nv[2], nv[3]);
                }
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const int i = base + u * 64;
                    if (i < n4) s4[i] = r[u];
                }
            }
        } else {
            for (int i = lane; i < total; i += 64) stage[i] = g[i];
        }
    } else {
        for (int i = lane; i < total; i += 64) {
            const uint32_t m = fastdiv((uint32_t)i, magicP);
            stage[m * Ppad + (i - m * P)] = g[i];
        }
    }
}

}  // namespace rulgnn
