// Constant operands and small helpers of the matrix-core training chains (stgcn_train_mx.hip: num_patch <= 15, four samples per
// wavefront; stgcn_train_mxw.hip: 16 <= num_patch <= 47, one sample per wavefront in column tiles): the split-operand forms of theta,
// of the convolutions (forward and transposed) and the kernels' argument block.
#pragma once
#include <type_traits>

#include "stgcn_mx.hpp"
#include "stgcn_train_layout.hpp"

namespace rulgnn {
namespace {

constexpr int MXT_ZERO_FLOATS = 192;     // zeroed LDS words padded lanes read instead of a tile (largest use: 3 * 55 + 1)
constexpr int MXT_SCRATCH_FLOATS = 128;  // head: d pool / arg-max exchange between the row mapping and the D layout
constexpr int MXT_SHIFT_FLOATS = 264;    // shift tile: [64 lanes + zero slot] x (hi pair | lo pair) = 1040 bytes
constexpr int MXT_WAVES = 4;              // wavefronts per workgroup: they share the BatchNorm table and reduce their sums / gradient rows in LDS
constexpr int MXT_RED_FLOATS = 448;      // gradient row image of a phase: at most 15 x 15 + 15 + 200 floats
constexpr int MXT_BNC = BN_TABLE_ROWS;   // per-BatchNorm constants: mean, istd, gamma, beta, gamma istd, mean(dy), mean(dy xhat)

enum { PH_F = 0, PH_TOP = 1, PH_G = 2 };

struct Op2 { u32x4 h, l; };              // a D-layout tensor as the ({hi | hi}, {lo | lo}) operand pair against a {hi | lo} partner
struct Pk { u32x2 hi, lo; };             // its packed halves: slots 4 g .. 4 g + 3 of this lane's column

__device__ __forceinline__ Pk pack3(float a, float b, float c, float d) {
    const Split2 p01 = split2(a, b), p23 = split2(c, d);
    return Pk{u32x2{p01.hi, p23.hi}, u32x2{p01.lo, p23.lo}};
}
__device__ __forceinline__ u32x4 cat(const u32x2& a, const u32x2& b) { return u32x4{a.x, a.y, b.x, b.y}; }
__device__ __forceinline__ f32x4 mfma16z(const u32x4& a, const u32x4& b) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    return mfma16(a, b, zero);
}
__device__ __forceinline__ bool finite_f(float v) { return __builtin_fabsf(v) <= 3.0e38f; }

// shift tile access for both directions: column t - d (forward taps) or t + d (transposed convolution); `rd` = the lane to read or 64
__device__ __forceinline__ Shifted shift_read(u32x2* tile, int rd, int rd_lo, int lane, const Pk& p) {
    return shift_columns(tile, rd, rd_lo, lane, p.hi, p.lo);
}

struct ConvOp { u32x4 hi, lo; };

// A operand of a forward convolution: row m = col <-> output channel; k-slots [0..3] = tap at t, [4..7] = tap at t - d, each x input
// slot 4 g + r; `scale` multiplies the weights of this lane's output channel, `pre` undoes a factor carried by the data (V = 4 o0),
// `shift` rides in slot 3 of lane group 0 against the constant 1 of the data operand.  In two steps: the raw taps are loaded in front of
// the BatchNorm-table hand-over of the prologue (they do not depend on it), scaled and split behind it.
struct ConvRaw { float wc[4], wd[4]; };
__device__ __forceinline__ ConvRaw conv_fwd_raw(const float* cw, int g, int col) {
    const int co = slot_chan(col), coc = co >= 0 ? co : 0;
    ConvRaw w;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ci = slot_chan(4 * g + r);
        const bool ok = co >= 0 && ci >= 0;
        const float2 taps2 = *reinterpret_cast<const float2*>(cw + (coc * F + (ci >= 0 ? ci : 0)) * 2);
        w.wc[r] = ok ? taps2.y : 0.f;
        w.wd[r] = ok ? taps2.x : 0.f;
    }
    return w;
}
__device__ __forceinline__ ConvOp conv_fwd_operand(const ConvRaw& w, float scale, float shift, float pre, int g, int col) {
    const int co = slot_chan(col);
    float wc[4], wd[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        wc[r] = w.wc[r] * (scale * pre);
        wd[r] = w.wd[r] * (scale * pre);
    }
    if (g == 0) wc[3] = co >= 0 ? shift : 0.f;
    const Split2 c01 = split2(wc[0], wc[1]), c23 = split2(wc[2], wc[3]), d01 = split2(wd[0], wd[1]), d23 = split2(wd[2], wd[3]);
    return ConvOp{u32x4{c01.hi, c23.hi, d01.hi, d23.hi}, u32x4{c01.lo, c23.lo, d01.lo, d23.lo}};
}
// A operand of the TRANSPOSED convolution: row m = col <-> input channel ci; k-slots [0..3] = w[co][ci][tap at t] against d z of
// column t, [4..7] = w[co][ci][tap at t - d] against d z of column t + d, co = slot 4 g + r.
// (loads and conversion apart: a prologue issues every load before it converts anything -- one memory round trip)
__device__ __forceinline__ ConvRaw conv_bwd_raw(const float* cw, int g, int col) {
    const int ci = slot_chan(col), cic = ci >= 0 ? ci : 0;
    ConvRaw w;                               // wc: tap at t, wd: tap at t - d
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = slot_chan(4 * g + r);
        const bool ok = co >= 0 && ci >= 0;
        const float2 taps2 = *reinterpret_cast<const float2*>(cw + ((co >= 0 ? co : 0) * F + cic) * 2);
        w.wc[r] = ok ? taps2.y : 0.f;
        w.wd[r] = ok ? taps2.x : 0.f;
    }
    return w;
}
__device__ __forceinline__ ConvOp conv_bwd_pack(const ConvRaw& w) {
    const Split2 c01 = split2(w.wc[0], w.wc[1]), c23 = split2(w.wc[2], w.wc[3]), d01 = split2(w.wd[0], w.wd[1]), d23 = split2(w.wd[2], w.wd[3]);
    return ConvOp{u32x4{c01.hi, c23.hi, d01.hi, d23.hi}, u32x4{c01.lo, c23.lo, d01.lo, d23.lo}};
}
__device__ __forceinline__ ConvOp conv_bwd_operand(const float* cw, int g, int col) { return conv_bwd_pack(conv_bwd_raw(cw, g, col)); }

struct ThetaOp { u32x4 hi, lo; };
struct ThetaRaw { float w[4]; };
__device__ __forceinline__ ThetaOp theta_pack(const ThetaRaw& t) {
    const Split2 p01 = split2(t.w[0], t.w[1]), p23 = split2(t.w[2], t.w[3]);
    return ThetaOp{u32x4{p01.hi, p23.hi, p01.hi, p23.hi}, u32x4{p01.lo, p23.lo, p01.lo, p23.lo}};
}
// theta^T as B operand of Hp = T x theta^T + b: column j = col, k-slot 4 g + r <-> patch k, k = 15 <-> bias; (1 + a)/2 folded in
__device__ __forceinline__ ThetaRaw theta_t_raw(const float* lp, int N, int g, int col) {
    ThetaRaw t;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = 4 * g + r;
        const bool ok = col < N && (k < N || k == 15);
        const int idx = k == 15 ? off_theta_b(N) + col : off_theta_w(N) + col * N + k;
        const float v = lp[ok ? idx : 0];
        t.w[r] = ok ? v * (0.5f * (1.f + LEAKY)) : 0.f;
    }
    return t;
}
__device__ __forceinline__ ThetaOp theta_t_operand(const float* lp, int N, int g, int col) { return theta_pack(theta_t_raw(lp, N, g, col)); }
// theta as B operand of d X = U x theta: column k' = col, k-slot 4 g + r <-> row j of theta
__device__ __forceinline__ ThetaRaw theta_n_raw(const float* lp, int N, int g, int col) {
    ThetaRaw t;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int j = 4 * g + r;
        const bool ok = col < N && j < N;
        const float v = lp[ok ? off_theta_w(N) + j * N + col : 0];
        t.w[r] = ok ? v : 0.f;
    }
    return t;
}
__device__ __forceinline__ ThetaOp theta_n_operand(const float* lp, int N, int g, int col) { return theta_pack(theta_n_raw(lp, N, g, col)); }

// Everything one layer's forward needs as constants: theta^T, the two convolutions, the affine part of its BatchNorms per D register
struct LayerK {
    ThetaOp th;
    ConvOp w[2];
    float gam[2][3], bet[2][3];
};

// `mode[blk]`: 0 = convolution not needed, 1 = raw weights (its BatchNorm statistics are what this phase computes), 2 = x-hat fold
// (weights x istd, shift -mean istd: the product IS x-hat, y = gamma x-hat + beta one fma behind it)
struct LayerRaw {
    ThetaRaw th;
    ConvRaw w[2];
};
__device__ __forceinline__ void layer_raw(LayerRaw& k, const float* prm, int l, int N, int g, int col, int mode0, int mode1) {
    const float* lp = prm + l * layer_stride(N);
    k.th = theta_t_raw(lp, N, g, col);
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        if ((blk == 0 ? mode0 : mode1) != 0) k.w[blk] = conv_fwd_raw(lp + off_conv_w(N, blk), g, col);
        else k.w[blk] = ConvRaw{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    }
}
__device__ __forceinline__ void layer_constants(LayerK& k, const LayerRaw& raw, const float* bnc, int l, int g, int col, int mode0, int mode1) {
    k.th = theta_pack(raw.th);
    const int co = slot_chan(col), coc = co >= 0 ? co : 0;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const int mode = blk == 0 ? mode0 : mode1;
        const float* q = bnc + (2 * l + blk) * MXT_BNC * F;
        const float istd = mode == 2 ? q[1 * F + coc] : 1.f;
        const float shift = mode == 2 ? -q[0 * F + coc] * istd : 0.f;
        if (mode != 0) k.w[blk] = conv_fwd_operand(raw.w[blk], istd, shift, blk == 0 ? 1.f : 0.25f, g, col);
        else k.w[blk] = ConvOp{u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int c = slot_chan(4 * g + r);
            k.gam[blk][r] = (mode == 2 && c >= 0) ? q[2 * F + c] : 0.f;
            k.bet[blk][r] = (mode == 2 && c >= 0) ? q[3 * F + c] : 0.f;
        }
    }
}

}  // namespace

struct MxTrainK {
    const float* prm;
    const float* y;
    float* pred;
    double* cells;
    float* gpart;
    float* xrec[MX_MAX_LAYERS];   // X_l tiles: [ntiles][10][4 N]
    float* qrec[MX_MAX_LAYERS];   // l >= 1: x-hat of BatchNorm 2l-1 where the gradient passes (ReLU gate and dropout), else +inf
    uint32_t* mrec[MX_MAX_LAYERS];   // dropout masks of layer l, one word per lane and tile (bit 3 s + r = keep of sample s, register r):
                                     // hashed once, by the phase that first applies them (F_{2l+2} / TOP), read by G_{2l+1}
    float* arec;                  // adjacency tiles: [ntiles][4][55]
    float* sb;                    // d(x0 + H): [ntiles][10][4 N]
    float* dx;                    // d X_l: [ntiles][10][4 N]
    float* dtop;                  // d X_L: [ntiles][2][4 N] (value | arg-max channel)
    int64_t B, ntiles, global_batch, sample_offset;
    int N, pcount;
    float dropout_p, drop_scale;
    uint32_t drop_thr;
    float gscale, inv_gscale;
    int do_backward;
};

}  // namespace rulgnn
