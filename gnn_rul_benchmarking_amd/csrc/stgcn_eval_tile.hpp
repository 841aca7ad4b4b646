// The exact-fp32 (VALU / DPP / f32-MFMA) eval forward of ONE wavefront tile, shared by
//   * stgcn_forward.hip     -- the row-mapped fused kernel (every num_patch <= 64), and
//   * stgcn_forward_mx.hip  -- the matrix-core kernel's in-launch fallback for samples whose f16-split
//                              arithmetic left the representable range or that carry NaN / Inf.
// Reference: ST_GCN_model.forward under model.eval() -- models/ST_GCN/Model.py:208-222.
#pragma once
#include "stgcn_device.hpp"

namespace rulgnn {

// LDS-resident weights of the exact path: what varies per lane, zero padded to the row width.  K = MPNN order (Model.py:74-90; 1
// in every wiring of the reference): K theta matrices per layer, ONE bias row per layer (the sum of the K biases).
template <int RW>
struct EvalWeightsLds {
    float* wlds;    // [L K + 1][RW][WS] theta rows per layer and order, then fc1 rows
    float* bnf;     // [L][2][2][F]  folded BatchNorm scale / shift
    float* vecs;    // [L+2][RW]     theta bias per layer (summed over the orders), fc1 bias, fc2 weight
    static __host__ __device__ constexpr int floats(int L, int K = 1) { return (L * K + 1) * RW * wstride<RW>() + L * 4 * F + (L + 2) * RW; }
    __device__ __forceinline__ void bind(float* base, int L, int K = 1) {
        wlds = base;
        bnf = wlds + (L * K + 1) * RW * wstride<RW>();
        vecs = bnf + L * 4 * F;
    }
};

// Fill the LDS weights from the flat parameter / BatchNorm buffers with `nthreads` cooperating threads
// (a workgroup followed by __syncthreads(), or a single wavefront followed by a wave barrier).
template <int RW>
__device__ __forceinline__ void eval_weights_fill(const EvalWeightsLds<RW>& w, const float* __restrict__ prm,
                                                  const float* __restrict__ bn, int N, int L, int tid, int nthreads, int K = 1) {
    constexpr int WS = wstride<RW>();
    const int LS = layer_stride(N, K);
    for (int i = tid; i < (L * K + 1) * RW * RW; i += nthreads) {
        const int m = i / (RW * RW), j = (i / RW) % RW, k = i % RW;
        const float* src = m < L * K ? prm + (m / K) * LS + off_theta_w(N, m % K) : prm + off_fc1_w(N, L, K);
        w.wlds[(m * RW + j) * WS + k] = (j < N && k < N) ? src[j * N + k] : 0.f;
    }
    for (int i = tid; i < (L + 2) * RW; i += nthreads) {
        const int m = i / RW, j = i % RW;
        float v = 0.f;
        if (j < N) {
            if (m < L) {
                for (int kk = 0; kk < K; ++kk) v += prm[m * LS + off_theta_b(N, kk) + j];
            } else {
                v = m == L ? prm[off_fc1_b(N, L, K) + j] : prm[off_fc2_w(N, L, K) + j];
            }
        }
        w.vecs[i] = v;
    }
    for (int i = tid; i < L * 2 * F; i += nthreads) {
        const int l = i / (2 * F), blk = (i / F) % 2, c = i % F;
        const float mean = bn[((l * 2 + blk) * 2 + 0) * F + c];
        const float var = bn[((l * 2 + blk) * 2 + 1) * F + c];
        const float g = prm[l * LS + off_bn_g(N, blk, K) + c], b = prm[l * LS + off_bn_b(N, blk, K) + c];
        const float sc = g / sqrtf(var + BN_EPS);
        w.bnf[((l * 2 + blk) * 2 + 0) * F + c] = sc;
        w.bnf[((l * 2 + blk) * 2 + 1) * F + c] = b - mean * sc;
    }
}

// One tile (64 / RW samples) whose windows are already in `stage` (patch stride Ppad).  Returns the lane's copy of
// the prediction of its sample row (every lane of a row holds it).  `stage` is overwritten (Pearson transpose tile).
template <int RW>
__device__ __forceinline__ float eval_tile_valu(float* stage, int ns, int N, int P, int Ppad, int L, const EvalWeightsLds<RW>& w,
                                                const float* __restrict__ prm, int lane, int K = 1) {
    constexpr int WS = wstride<RW>();
    const int LS = layer_stride(N, K);
    const int srow = lane / RW, t = lane % RW;
    const bool valid = (srow < ns) && (t < N);
    float X[F];
#pragma unroll
    for (int c = 0; c < F; ++c) X[c] = 0.f;
    if (valid) patch_statistics(stage + (srow * N + t) * Ppad, P, X);

    constexpr int NA = RW == 16 ? F : NPAIR;      // RW 16: lane-distributed adjacency rows (MFMA path)
    float A[NA];
    if constexpr (RW == 16) {
        pearson_rows_mfma(X, srow < ns, N, stage, lane, A);
    } else {
        pearson_adjacency<RW>(X, valid, N, A);
    }

    for (int l = 0; l < L; ++l) {
        const float* lp = prm + l * LS;
        const float* bl = w.bnf + l * 4 * F;
        float AX[F], H[F], z[F], o0[F];
        if constexpr (RW == 16) {
            adj_aggregate_mfma(A, X, AX);
        } else {
            adj_aggregate(A, X, AX);
        }
        const float tb = w.vecs[l * RW + t];
#pragma unroll
        for (int c = 0; c < F; ++c) H[c] = tb;
        Row<RW>::project10(H, AX, w.wlds + (l * K * RW + t) * WS, N);   // theta(A.X), Model.py:87
        for (int kk = 1; kk < K; ++kk) {                                 // order kk + 1: theta_kk(A^(kk+1) X), Model.py:82-88 -- A (A^kk X)
            float AXk[F];
            if constexpr (RW == 16) adj_aggregate_mfma(A, AX, AXk); else adj_aggregate(A, AX, AXk);
#pragma unroll
            for (int c = 0; c < F; ++c) AX[c] = AXk[c];
            Row<RW>::project10(H, AX, w.wlds + ((l * K + kk) * RW + t) * WS, N);
        }
#pragma unroll
        for (int c = 0; c < F; ++c) H[c] = leaky(H[c]);
        causal_conv<RW, 1>(H, lp + off_conv_w(N, 0, K), t, z);          // conv_block1, Model.py:134-146
#pragma unroll
        for (int c = 0; c < F; ++c) o0[c] = relu(relu(fmaf(z[c], bl[c], bl[F + c])) + H[c]);
        causal_conv<RW, 2>(o0, lp + off_conv_w(N, 1, K), t, z);         // conv_block2 (dilation 2), Model.py:148-160
#pragma unroll
        for (int c = 0; c < F; ++c) {
            const float o1 = relu(relu(fmaf(z[c], bl[2 * F + c], bl[3 * F + c])) + o0[c]);
            X[c] = valid ? o1 + X[c] : 0.f;                              // Dropout is identity in eval; out += res
        }
    }
    // AdaptiveMaxPool1d over the ten channels (NaN-propagating like torch), Model.py:218-219
    float pooled = X[0];
#pragma unroll
    for (int c = 1; c < F; ++c) pooled = (X[c] > pooled || X[c] != X[c]) ? X[c] : pooled;
    pooled = valid ? pooled : 0.f;
    float y1 = w.vecs[L * RW + t];
    Row<RW>::project1(y1, pooled, w.wlds + (L * K * RW + t) * WS, N);   // fc1, Model.py:220
    y1 = relu(y1);
    return Row<RW>::allsum(y1 * w.vecs[(L + 1) * RW + t]) + prm[off_fc2_b(N, L, K)];   // fc2, Model.py:221
}

}  // namespace rulgnn
