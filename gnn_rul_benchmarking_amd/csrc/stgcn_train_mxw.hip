// The matrix-core training chain for 16 <= num_patch <= 47 -- PHM2012's 40 patches of 64 points is the reference's own ST_GCN wiring
// (configs/hparams.py:223,238).  Reference path replaced: ST_GCN.update up to optimizer.step(), algorithms/algorithms.py:481-488; the
// layer is models/ST_GCN/Model.py:74-90 (MPNN_mk), :134-170 (TemporalConvNet), :187-195 (SG_TCN).
//
// Same chain, same recomputation, same arithmetic as stgcn_train_mx.hip (read its header first); what changes is the layout, which is the
// wide eval kernel's (stgcn_forward_mx.hip, stgcn_forward_mxw_kernel): ONE sample per wavefront iteration, its patch axis in NT = 2 | 3
// column tiles of 16 -- a [10, N] tensor is NT x 3 registers, and the column tiles of the sample are the independent MFMA chains that the
// four samples are in the narrow chain.  Consequences:
//   * theta is N x N: Hp = T x theta^T contracts over NT k-tiles into NT column tiles (2 NT^2 products); the backward's d X = U x theta
//     likewise; d theta is NT^2 accumulator tiles.  The theta operands -- (hi | lo) halves per (k-tile, column tile) -- live in LDS,
//     built once per workgroup, shared by its four wavefronts;
//   * the causal taps and the transposed convolution's taps cross tile boundaries: the shift tile is [row group][column 0 .. 16 NT);
//   * the transposed arrangement of the weight gradient holds columns 16 ct + 4 g .. + 3 in lane group g of tile ct: its shift by the
//     dilation takes the pair of the row above, or of row group 3 of the tile before;
//   * head: pooled vector and d y1 go through LDS (broadcast reads) against fc1 / fc1^T tables in LDS; d fc1.w = d y1 (x) pooled is NT^2
//     rank-1 updates on the fp32 matrix cores;
//   * records are per SAMPLE: X_l [10][N] (padded to 16 bytes), 55 adjacency pairs (+ 1 pad), d X_L [2][N], one word of dropout bits per lane.
// A tile is a sample, so there are no partial tiles.  Prepare, the reduction cells, the f16 range guard and finalize are the fp32 chain's.
#include <cstdlib>
#include <type_traits>

#include "stgcn_host.hpp"
#include "stgcn_mx.hpp"
#include "stgcn_train_layout.hpp"
#include "stgcn_train_mx.hpp"
#include "stgcn_train_mx_ops.hpp"

namespace rulgnn {

namespace {
constexpr int MXW_ASTRIDE = 56;                  // adjacency record: 55 (hi | lo) pairs + 1 pad word (16-byte DMA pieces)
__host__ __device__ constexpr int mxw_xstride(int N) { return (10 * N + 3) & ~3; }
__host__ __device__ constexpr int mxw_dstride(int N) { return (2 * N + 3) & ~3; }
}  // namespace

// =====================================================================================================================
// F_0: windows -> X_0 record, adjacency pairs, sum z1, sum z1^2 of BatchNorm 0 (front end of the wide eval kernel + half a layer)
// =====================================================================================================================
template <int NT, int NFIX, int PFIX>
__global__ __launch_bounds__(64 * MXT_WAVES, MX_WAVES_PER_SIMD) void stgcn_train_f0_mxw_kernel(const float* __restrict__ gx, MxTrainK a, int P_, int buf_floats, int cells_stride, HeadScalars hs) {
    if (hs.sc != nullptr && blockIdx.x == 0) head_scalars(hs, threadIdx.x);                   // a step without its prepare launch
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    constexpr int W = 16 * NT, PT = W + 4, TG = 4 * NT;
    const int N = NFIX ? NFIX : a.N, P = PFIX ? PFIX : P_;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, col = lane & 15;
    const int NP = N * P, XS = mxw_xstride(N);
    constexpr int L = 1;                                                  // (cell layout offsets of the forward pair 0 do not depend on L)
    u32x4* const theta_lds = reinterpret_cast<u32x4*>(smem_all);          // [NT][NT][64]
    double* const pairbuf = reinterpret_cast<double*>(smem_all + NT * NT * 64 * 4);       // [MXT_WAVES][2 F]
    constexpr int REGION = (16 * PT > 2 * (4 * W + 1) * 2 ? 16 * PT : 2 * (4 * W + 1) * 2) + 64;
    float* const win = smem_all + NT * NT * 64 * 4 + 2 * MXT_WAVES * 2 * F + wave * (buf_floats + ((REGION + 3) & ~3));
    float* const cur = win + buf_floats;
    const int64_t stride = (int64_t)gridDim.x * MXT_WAVES;
    int64_t smp = (int64_t)blockIdx.x * MXT_WAVES + wave;
    if (smp < a.B) dma_tile(gx + smp * NP, win, NP * 4, lane);

    // theta^T of layer 0 (leaky's (1 + a)/2 folded in, bias in k-slot 15 of the last k-tile) and the RAW conv_block1 weights
    for (int idx = wave; idx < NT * NT; idx += MXT_WAVES) {
        const int ct = idx / NT, jt = idx % NT, j = 16 * jt + col;
        float w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 16 * ct + 4 * g + r;
            const bool bias = ct == NT - 1 && 4 * g + r == 15;
            const bool ok = j < N && (k < N || bias);
            const float v = a.prm[ok ? (bias ? off_theta_b(N) + j : off_theta_w(N) + j * N + k) : 0];
            w[r] = ok ? v * (0.5f * (1.f + LEAKY)) : 0.f;
        }
        const Split2 p01 = split2(w[0], w[1]), p23 = split2(w[2], w[3]);
        theta_lds[idx * 64 + lane] = u32x4{p01.hi, p23.hi, p01.lo, p23.lo};
    }
    const ConvOp w1 = conv_fwd_operand(conv_fwd_raw(a.prm + off_conv_w(N, 0), g, col), 1.f, 0.f, 1.f, g, col);
    const unsigned t_bias = g == 3 ? 0x3C00u << 16 : 0u;
    constexpr int SH_ZERO = 4 * W, SH_LO = 4 * W + 1;
    int sh_wr[NT], sh_rd1[NT], sh_rd1_lo[NT];
    float colm[NT];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
        const int t = 16 * jt + col;
        colm[jt] = t < N ? 1.f : 0.f;
        sh_wr[jt] = g * W + t;
        sh_rd1[jt] = t >= 1 ? g * W + t - 1 : SH_ZERO;
        sh_rd1_lo[jt] = sh_rd1[jt] + SH_LO;
        asm volatile("" : "+v"(sh_rd1_lo[jt]));
    }
    for (int e = lane; e < 2 * PT; e += 64) cur[13 * PT + e] = 0.f;
    __syncthreads();
    const int jc = lane < N ? lane : 0;
    const int ca_col = slot_chan(col);
    float sa[3] = {0.f, 0.f, 0.f}, sb[3] = {0.f, 0.f, 0.f};

    for (; smp < a.B; smp += stride) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        float X0[F];
#pragma unroll
        for (int c = 0; c < F; ++c) X0[c] = 0.f;
        auto request_next = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const int64_t nx = smp + stride;
            if (nx < a.B) {
                if constexpr (NFIX != 0 && PFIX != 0) dma_tile_fixed<4 * NFIX * PFIX>(gx + nx * NP, win, lane);
                else dma_tile(gx + nx * NP, win, NP * 4, lane);
            }
        };
        if constexpr (PFIX == 64) {
            float v[64];
            const float4* p4 = reinterpret_cast<const float4*>(win + jc * 64);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float4 q = p4[(k + lane) & 15];
                v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
            }
            request_next();
            if (lane < N) patch_statistics_lean<64>(v, X0);
        } else {
            if (lane < N) patch_statistics(win + lane * P, P, X0);
            request_next();
        }
        // the statistics as the later phases read them: [10][N], lane = patch
        if (lane < N) {
            float* px = a.xrec[0] + smp * XS + lane;
#pragma unroll
            for (int c = 0; c < F; ++c) __builtin_nontemporal_store(X0[c], px + c * N);
        }
        // Pearson adjacency (Model.py:53-71) and the statistics in the D layout
        __builtin_amdgcn_wave_barrier();
        if (lane < W) {
#pragma unroll
            for (int c = 0; c < F; ++c) cur[chan_slot(c) * PT + lane] = X0[c];
        }
        __builtin_amdgcn_wave_barrier();
        f32x4 gram = {0.f, 0.f, 0.f, 0.f};
        {
            const bool slot_ok = slot_chan(col) >= 0;
            const float4* r4 = reinterpret_cast<const float4*>(cur + (slot_ok ? col : 0) * PT + TG * g);
            float CT[TG];
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                const float4 v = r4[q];
                CT[4 * q] = v.x; CT[4 * q + 1] = v.y; CT[4 * q + 2] = v.z; CT[4 * q + 3] = v.w;
            }
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < TG; ++k) sum += CT[k];
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float mean = sum * (1.0f / (float)N);
            float ss = 0.f;
#pragma unroll
            for (int k = 0; k < TG; ++k) {
                CT[k] = (TG * g + k < N) ? CT[k] - mean : 0.f;
                ss = fmaf(CT[k], CT[k], ss);
            }
            ss += __shfl_xor(ss, 16, 64);
            ss += __shfl_xor(ss, 32, 64);
            const float rn = slot_ok ? __builtin_amdgcn_rsqf(ss) : 0.f;
#pragma unroll
            for (int k = 0; k < TG; ++k) {
                const float y = CT[k] * rn;
                gram = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, gram, 0, 0, 0);
            }
        }
        const Split2 g01 = split2(gram[0], gram[1]), g23 = split2(gram[2], gram[3]);
        const u32x4 adjB = u32x4{g01.hi, g23.hi, g01.lo, g23.lo};
        if (ca_col >= 0) {                           // the 55 unique entries as (hi | lo << 16) pairs
            float* pa = a.arec + smp * MXW_ASTRIDE;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int c = slot_chan(4 * g + r);
                const unsigned ph = r < 2 ? adjB[0] : adjB[1], pl = r < 2 ? adjB[2] : adjB[3];
                const unsigned pair = (r & 1) ? ((ph >> 16) | (pl & 0xFFFF0000u)) : ((ph & 0xFFFFu) | (pl << 16));
                if (c >= 0 && c <= ca_col) pa[sym(c, ca_col)] = __builtin_bit_cast(float, pair);
            }
        }
        float X[NT][3];
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int r = 0; r < 3; ++r) X[ct][r] = cur[(4 * g + r) * PT + 16 * ct + col];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        u32x2* const sh_tile = reinterpret_cast<u32x2*>(cur);
        if (lane < 2) sh_tile[SH_ZERO + SH_LO * lane] = u32x2{0u, 0u};

        // half of layer 0: T, Hp, H, raw z1
        f32x4 T[NT], Hp[NT], z[NT];
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            const Split2 p01 = split2(X[ct][0], X[ct][1]), p2 = split2(X[ct][2], 0.f);
            const u32x4 ah = {p01.hi, p2.hi, p01.hi, p2.hi}, al = {p01.lo, p2.lo, p01.lo, p2.lo};
            T[ct] = mfma16(ah, adjB, zero);
            T[ct] = mfma16(al, adjB, T[ct]);
        }
        __builtin_amdgcn_sched_barrier(0);
        u32x4 tah[NT], tal[NT];
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            const Split2 p01 = split2(T[ct][0], T[ct][1]), p23 = split2(T[ct][2], T[ct][3]);
            const unsigned h23 = ct == NT - 1 ? p23.hi | t_bias : p23.hi;
            tah[ct] = u32x4{p01.hi, h23, p01.hi, h23};
            tal[ct] = u32x4{p01.lo, p23.lo, p01.lo, p23.lo};
        }
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) Hp[jt] = zero;
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                const u32x4 th = theta_lds[(ct * NT + jt) * 64 + lane];
                Hp[jt] = mfma16(tah[ct], th, Hp[jt]);
                Hp[jt] = mfma16(tal[ct], th, Hp[jt]);
            }
        __builtin_amdgcn_sched_barrier(0);
        u32x2 ch[NT], cl[NT];
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            keep_until_here(Hp[jt][3]);
            float H[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) H[r] = fmaf((1.f - LEAKY) / (1.f + LEAKY), __builtin_fabsf(Hp[jt][r]), Hp[jt][r]);
            const Split2 p01 = split2(H[0], H[1]), p2 = split2(H[2], 0.f);
            ch[jt] = u32x2{p01.hi, p2.hi};
            cl[jt] = u32x2{p01.lo, p2.lo};
            sh_tile[sh_wr[jt]] = ch[jt];
            sh_tile[SH_LO + sh_wr[jt]] = cl[jt];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            const u32x2 ph = sh_tile[sh_rd1[jt]], pl = sh_tile[sh_rd1_lo[jt]];
            const u32x4 bh = {ch[jt].x, ch[jt].y, ph.x, ph.y}, bl = {cl[jt].x, cl[jt].y, pl.x, pl.y};
            z[jt] = mfma16(w1.hi, bh, zero);
            z[jt] = mfma16(w1.hi, bl, z[jt]);
            z[jt] = mfma16(w1.lo, bh, z[jt]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            keep_until_here(z[jt][3]);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float zv = z[jt][r] * colm[jt];
                sa[r] += zv;
                sb[r] = fmaf(zv, zv, sb[r]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        for (int e = lane; e < 2 * PT; e += 64) cur[13 * PT + e] = 0.f;         // the shift tile overwrote the conversion tile's padding rows
    }
    // epilogue: the wavefronts' sums combined, one atomic per channel and workgroup
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double da = (double)sa[r], db = (double)sb[r];
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) {
            da += __shfl_xor(da, off, 16);
            db += __shfl_xor(db, off, 16);
        }
        const int c = slot_chan(4 * g + r);
        if (col == 0 && c >= 0) {
            pairbuf[wave * 2 * F + c] = da;
            pairbuf[wave * 2 * F + F + c] = db;
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * F) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < MXT_WAVES; ++w) v += pairbuf[w * 2 * F + threadIdx.x];
        atomicAdd(a.cells + (int64_t)(blockIdx.x % CELL_REPLICAS) * cells_stride + cell_fwd(L) + threadIdx.x, v);
    }
}

// =====================================================================================================================
// the phase kernel (everything but F_0)
// =====================================================================================================================
// NFIX: num_patch as a compile-time constant (40 = PHM2012's wiring; 0 = generic): record strides, LDS offsets and the byte counts of the
// LDS-DMA requests (dma_tile_fixed: one M0 write per 4 KB instead of one per piece) fold
template <int L, int KIND, int IDX, int NT, int NFIX>
__global__ __launch_bounds__(64 * MXT_WAVES, MX_WAVES_PER_SIMD) void stgcn_train_mxw_kernel(MxTrainK a) {
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    constexpr int W = 16 * NT;
    const int N = NFIX ? NFIX : a.N;
    const int LS = layer_stride(N);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, col = lane & 15;
    constexpr int NBN = 2 * L;
    constexpr int CS = cell_stride(L);
    constexpr int LY = KIND == PH_TOP ? L - 1 : IDX / 2;
    constexpr int BLK = KIND == PH_TOP ? 1 : IDX % 2;
    constexpr bool WITH_PREV = KIND == PH_F && BLK == 0 && LY >= 1;
    constexpr bool BWD_PREV = KIND == PH_G && BLK == 0 && LY >= 1;
    constexpr int LIN = WITH_PREV ? LY - 1 : LY;
    constexpr bool GRAD_IN = KIND == PH_G && (BLK == 1 || LY >= 1);
    constexpr bool GRAD_TOP = GRAD_IN && LY == L - 1;
    constexpr bool NEED_SB = KIND == PH_G && BLK == 0;
    // G_{2l} carries the most state (the theta-gradient tiles): its records are read from LDS where they are used instead of being held in
    // registers across the sample, the next sample's records are requested once the region is free, and d X_l is stored without the delay
    constexpr bool LATE = KIND == PH_G && BLK == 0;
    constexpr int NTH = 1 + (WITH_PREV ? 1 : 0) + (BWD_PREV ? 1 : 0);          // theta operand tables: theta^T(LY) | theta^T(LY-1) | theta(LY)
    static_assert(!(KIND == PH_F && IDX == 0), "F_0 is stgcn_train_f0_mxw_kernel");
    const int XS = mxw_xstride(N), DS = mxw_dstride(N);

    // ---- LDS: workgroup [theta tables | fc1 tables (TOP) | BatchNorm constants | row image | pair partials], then one region per wavefront
    constexpr int SH_BNC = (NBN * MXT_BNC * F + 3) & ~3;
    constexpr int TH_FLOATS = NTH * NT * NT * 64 * 4;
    constexpr int FCP = W + 4;                                  // row pitch of the fc1 tables: 16-byte row reads of 16 lanes fall on distinct banks
    constexpr int FC_FLOATS = KIND == PH_TOP ? 2 * W * FCP : 0;
    constexpr int RED_FLOATS = (W * W + W + CONVW + 3) & ~3;
    u32x4* const theta_lds = reinterpret_cast<u32x4*>(smem_all);
    float* const fc1N = smem_all + TH_FLOATS;                  // fc1.w[j][k] at j FCP + k: lane j reads its row (y1 = fc1 x pooled)
    float* const fc1T = fc1N + W * FCP;                        // fc1.w[j][k] at k FCP + j: lane k reads its column (d pooled = fc1^T x d y1)
    constexpr int COPS_FLOATS = LATE ? 31 * 64 : 0;           // G_{2l}: per-lane constant operands kept out of the register file
    float* const bnc = smem_all + TH_FLOATS + FC_FLOATS;
    double* const pairbuf = reinterpret_cast<double*>(bnc + SH_BNC);
    float* const cops = bnc + SH_BNC + 2 * MXT_WAVES * (2 * F + 2);
    float* const red = cops + COPS_FLOATS;                     // the epilogue's row image lies over the wavefronts' regions (all done by then)
    // shift tile: [hi | lo][row group][2 zero entries | column 0 .. W): a tap at t - d (d <= 2) of column 0 / 1 reads the zeros in front of its
    // row, a tap at t + d past the last column the zeros in front of the next row (two more behind the last): every tap is base + constant
    constexpr int WP = W + 2, SH_LO = 4 * WP + 2;
    constexpr int SHIFT_FLOATS = ((2 * SH_LO * 2) + 3) & ~3;
    const int off_zero = 0;                                    // 64 zero words
    const int off_scr = 64;                                    // 4 W floats: pooled | d y1 | d pool | arg-max
    const int off_sh = off_scr + 4 * W;
    const int off_X = off_sh + SHIFT_FLOATS;
    const int off_A = off_X + XS;
    const int off_SB = off_A + MXW_ASTRIDE;
    const int off_DX = off_SB + (NEED_SB ? XS : 0);
    const int off_Q = off_DX + (GRAD_IN ? XS : 0);
    const int wave_floats = off_Q + (BWD_PREV ? XS : 0);
    float* const smem = red + wave * wave_floats;
    static_assert(RED_FLOATS <= MXT_WAVES * (64 + 4 * W + SHIFT_FLOATS + 160 + MXW_ASTRIDE), "row image fits the wavefronts' regions");
    u32x2* const sh_tile = reinterpret_cast<u32x2*>(smem + off_sh);

    int64_t smp = (int64_t)blockIdx.x * MXT_WAVES + wave;
    const int64_t stride = (int64_t)gridDim.x * MXT_WAVES;

    auto dma = [&](const float* src, int off, int floats) {
        if constexpr (NFIX != 0) {
            if (floats == mxw_xstride(NFIX)) { dma_tile_fixed<4 * mxw_xstride(NFIX)>(src, smem + off, lane); return; }
            if (floats == mxw_dstride(NFIX)) { dma_tile_fixed<4 * mxw_dstride(NFIX)>(src, smem + off, lane); return; }
        }
        if (floats == MXW_ASTRIDE) { dma_tile_fixed<4 * MXW_ASTRIDE>(src, smem + off, lane); return; }
        dma_tile(src, smem + off, floats * 4, lane);
    };
    auto req_XA = [&](int64_t s) {
        dma(a.xrec[LIN] + s * XS, off_X, XS);
        dma(a.arec + s * MXW_ASTRIDE, off_A, MXW_ASTRIDE);
    };
    auto req_SB = [&](int64_t s) { if constexpr (NEED_SB) dma(a.sb + s * XS, off_SB, XS); };
    auto req_DX = [&](int64_t s) {
        if constexpr (GRAD_TOP) dma(a.dtop + s * DS, off_DX, DS);
        else if constexpr (GRAD_IN) dma(a.dx + s * XS, off_DX, XS);
    };
    auto req_Q = [&](int64_t s) { if constexpr (BWD_PREV) dma(a.qrec[LY] + s * XS, off_Q, XS); };
    if (smp < a.B) { req_XA(smp); req_SB(smp); req_DX(smp); req_Q(smp); }

    // ---- prologue ----------------------------------------------------------------------------------------------------------
    for (int i = lane; i < 64; i += 64) smem[off_zero + i] = 0.f;
    for (int i = lane; i < 2 * SH_LO; i += 64) sh_tile[i] = u32x2{0u, 0u};
    // theta tables.  Table 0 (and 1): theta^T of layer LY (LY - 1) as B operands of Hp = T x theta^T + b: tile (ct, jt), column j = 16 jt + col,
    // k-slot 4 g + r <-> patch 16 ct + 4 g + r, k-slot 15 of the last k-tile = the bias, (1 + a)/2 folded in.  Last table (G_{2l}, l >= 1):
    // theta itself as B operands of d X = U x theta: tile (jt, kt), column k = 16 kt + col, k-slot <-> row j = 16 jt + 4 g + r.
    // (all loads of the prologue are issued before the first conversion: the tables cost one memory round trip, not one per tile)
    constexpr int TT = NTH * NT * NT, TIT = (TT + MXT_WAVES - 1) / MXT_WAVES;
    float tw[TIT][4];
    bool tok[TIT][4];
#pragma unroll
    for (int it = 0; it < TIT; ++it) {
        const int idx = (wave + MXT_WAVES * it < TT) ? wave + MXT_WAVES * it : TT - 1;        // (the surplus slots of the last round load tile TT - 1 again)
        const int tab = idx / (NT * NT), ct = (idx / NT) % NT, jt = idx % NT;
        const bool plain = BWD_PREV && tab == NTH - 1;
        const int layer = plain ? LY : (tab == 0 ? LY : LY - 1);
        const float* lp = a.prm + layer * LS;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int off;
            if (plain) {
                const int j = 16 * ct + 4 * g + r, k = 16 * jt + col;
                tok[it][r] = j < N && k < N;
                off = off_theta_w(N) + j * N + k;
            } else {
                const int j = 16 * jt + col, k = 16 * ct + 4 * g + r;
                const bool bias = ct == NT - 1 && 4 * g + r == 15;
                tok[it][r] = j < N && (k < N || bias);
                off = bias ? off_theta_b(N) + j : off_theta_w(N) + j * N + k;
            }
            tw[it][r] = lp[tok[it][r] ? off : 0] * (plain ? 1.f : 0.5f * (1.f + LEAKY));
        }
    }
    constexpr int FIT = KIND == PH_TOP ? (W * W) / (64 * MXT_WAVES) : 0;
    static_assert((W * W) % (64 * MXT_WAVES) == 0, "fc1 table elements per thread");
    float fv[FIT > 0 ? FIT : 1];
    if constexpr (KIND == PH_TOP) {
#pragma unroll
        for (int q = 0; q < FIT; ++q) {
            const int i = threadIdx.x + 64 * MXT_WAVES * q, j = i / W, k = i % W;
            const bool ok = j < N && k < N;
            const float v = a.prm[off_fc1_w(N, L) + (ok ? j * N + k : 0)];
            fv[q] = ok ? v : 0.f;
        }
    }
    constexpr int M0_LY = (KIND == PH_F && BLK == 0) ? 1 : 2;
    constexpr int M1_LY = (KIND == PH_F && BLK == 0) ? 0 : ((KIND == PH_F) ? 1 : ((KIND == PH_G && BLK == 0) ? 0 : 2));
    LayerRaw rc, rp;
    layer_raw(rc, a.prm, LY, N, g, col, M0_LY, M1_LY);
    if constexpr (WITH_PREV) layer_raw(rp, a.prm, LY - 1, N, g, col, 2, 2);
    ConvOp wT = ConvOp{u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}};
    u32x4 ident = u32x4{0u, 0u, 0u, 0u};
    ConvRaw wT_raw = ConvRaw{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if constexpr (KIND == PH_G) {
        wT_raw = conv_bwd_raw(a.prm + LY * LS + off_conv_w(N, BLK), g, col);
        unsigned w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = (4 * g + r == col) ? 0x3C00u : 0u;
        const unsigned p01 = w[0] | (w[1] << 16), p23 = w[2] | (w[3] << 16);
        ident = u32x4{p01, p23, p01, p23};
    }
    {
        // (the BatchNorm pairs' loads go out behind the table loads, their fp64 arithmetic runs under those)
        constexpr int FW0 = WITH_PREV ? 2 * LY - 2 : 2 * LY;
        constexpr int NFW = (KIND == PH_F && BLK == 1) || (KIND == PH_G && BLK == 0) ? 1 : 2;
        if (wave < NFW) bn_pair_to_lds(a.cells, a.prm, bnc, L, N, true, FW0 + wave, lane);
        if (KIND == PH_G && wave == NFW) bn_pair_to_lds(a.cells, a.prm, bnc, L, N, false, IDX, lane);
    }
    if constexpr (KIND == PH_G) wT = conv_bwd_pack(wT_raw);
#pragma unroll
    for (int it = 0; it < TIT; ++it) {
        const int idx = wave + MXT_WAVES * it;
        float w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = tok[it][r] ? tw[it][r] : 0.f;
        const Split2 p01 = split2(w[0], w[1]), p23 = split2(w[2], w[3]);
        if (idx < TT) theta_lds[idx * 64 + lane] = u32x4{p01.hi, p23.hi, p01.lo, p23.lo};
    }
    if constexpr (KIND == PH_TOP) {
#pragma unroll
        for (int q = 0; q < FIT; ++q) {
            const int i = threadIdx.x + 64 * MXT_WAVES * q, j = i / W, k = i % W;
            fc1N[j * FCP + k] = fv[q];
            fc1T[k * FCP + j] = fv[q];
        }
    }
    __syncthreads();
    LayerK kc, kp;
    layer_constants(kc, rc, bnc, LY, g, col, M0_LY, M1_LY);
    if constexpr (WITH_PREV) layer_constants(kp, rp, bnc, LY - 1, g, col, 2, 2);
    float bA[3] = {0.f, 0.f, 0.f}, bk1[3] = {0.f, 0.f, 0.f}, bk2[3] = {0.f, 0.f, 0.f};
    if constexpr (KIND == PH_G) {
        const float* q = bnc + IDX * MXT_BNC * F;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int c = slot_chan(4 * g + r);
            bA[r] = c >= 0 ? q[4 * F + c] : 0.f;
            bk1[r] = c >= 0 ? q[5 * F + c] * a.gscale : 0.f;
            bk2[r] = c >= 0 ? q[6 * F + c] * a.gscale : 0.f;
        }
    }
    if constexpr (LATE) {
        // [0..3] x 64 x 16 bytes: conv_block1 forward operand (hi, lo), its transposed operand (hi, lo); then 9 x 64 floats: the BatchNorm
        // backward constants per register row
        if (wave == 0) {
            u32x4* c4 = reinterpret_cast<u32x4*>(cops);
            c4[lane] = kc.w[0].hi; c4[64 + lane] = kc.w[0].lo; c4[128 + lane] = wT.hi; c4[192 + lane] = wT.lo;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                cops[(16 + r) * 64 + lane] = bA[r];
                cops[(19 + r) * 64 + lane] = bk1[r];
                cops[(22 + r) * 64 + lane] = bk2[r];
                cops[(25 + r) * 64 + lane] = kc.gam[0][r];
                cops[(28 + r) * 64 + lane] = kc.bet[0][r];
            }
        }
        __syncthreads();
    }
    auto conv_fwd0 = [&]() -> ConvOp {
        if constexpr (LATE) { const u32x4* c4 = reinterpret_cast<const u32x4*>(cops); return ConvOp{c4[lane], c4[64 + lane]}; }
        else return kc.w[0];
    };
    auto conv_bwd = [&]() -> ConvOp {
        if constexpr (LATE) { const u32x4* c4 = reinterpret_cast<const u32x4*>(cops); return ConvOp{c4[128 + lane], c4[192 + lane]}; }
        else return wT;
    };
    float fc1b = 0.f, fc2w = 0.f, fc2b = 0.f;
    if constexpr (KIND == PH_TOP) {
        const int jc = lane < N ? lane : 0;
        const float b1 = a.prm[off_fc1_b(N, L) + jc], w2 = a.prm[off_fc2_w(N, L) + jc];
        fc1b = lane < N ? b1 : 0.f;
        fc2w = lane < N ? w2 : 0.f;
        fc2b = a.prm[off_fc2_b(N, L)];
    }

    // ---- per-lane addressing ---------------------------------------------------------------------------------------------------
    int chan[3], aoff[3];
    {
        const int cc = slot_chan(col);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            chan[r] = slot_chan(4 * g + r);
            aoff[r] = (chan[r] >= 0 && cc >= 0) ? sym(chan[r], cc) : -1;
        }
    }
    float colm[NT];
    int xoff[NT][3];
    const int shb = g * WP + 2 + col;
    uint32_t dro[NT][3];
    constexpr int DB = BLK == 0 ? 1 : 2;                       // dilation of this phase's backward convolution
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
        const int t = 16 * ct + col;
        colm[ct] = t < N ? 1.f : 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            xoff[ct][r] = (chan[r] >= 0 && t < N) ? chan[r] * N + t : -1;
            dro[ct][r] = (uint32_t)((chan[r] >= 0 ? chan[r] : 0) * N + t);
        }
    }
    auto ld_rec = [&](int base, float (&v)[NT][3]) {
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int r = 0; r < 3; ++r) v[ct][r] = smem[xoff[ct][r] >= 0 ? base + xoff[ct][r] : off_zero];
    };
    auto st_rec = [&](float* dst, const float (&v)[NT][3]) {
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int r = 0; r < 3; ++r)
                if (xoff[ct][r] >= 0) dst[xoff[ct][r]] = v[ct][r];
    };
    const unsigned t_bias = g == 3 ? 0x3C00u << 16 : 0u;
    uint32_t dkey[L];
#pragma unroll
    for (int l = 0; l < L; ++l) dkey[l] = step_scratch(a.cells, L)->drop_key[l];
    const bool use_drop = a.dropout_p > 0.f;

    // ---- persistent accumulators ---------------------------------------------------------------------------------------------------
    float s_a[3] = {0.f, 0.f, 0.f}, s_b[3] = {0.f, 0.f, 0.f};
    f32x4 acc_w0 = {0.f, 0.f, 0.f, 0.f}, acc_w1 = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc_th[NT][NT];                                                   // theta / fc1 weight gradient: tile (j tile, k tile)
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc_th[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float acc_b[NT], acc_w2 = 0.f, acc_b1 = 0.f, acc_b2 = 0.f, acc_loss = 0.f;
#pragma unroll
    for (int i = 0; i < NT; ++i) acc_b[i] = 0.f;
    const float inv_gb = 1.0f / (float)a.global_batch;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    // ---- building blocks -----------------------------------------------------------------------------------------------------------
    auto stage_T = [&](const float (&X)[NT][3], const u32x4& adjB, f32x4 (&T)[NT], Op2 (&xo)[NT]) {
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            const Split2 p01 = split2(X[ct][0], X[ct][1]), p2 = split2(X[ct][2], 0.f);
            xo[ct] = Op2{u32x4{p01.hi, p2.hi, p01.hi, p2.hi}, u32x4{p01.lo, p2.lo, p01.lo, p2.lo}};
            T[ct] = mfma16z(xo[ct].h, adjB);
            T[ct] = mfma16(xo[ct].l, adjB, T[ct]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // out[jt] = sum over k-tiles of in[ct] x table[ct][jt]; `bias`: the constant 1 of the bias k-slot rides in the last k-tile
    auto stage_theta = [&](const f32x4 (&In)[NT], int tab, bool bias, f32x4 (&Out)[NT]) {
        u32x4 ah[NT], al[NT];
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            const Split2 p01 = split2(In[ct][0], In[ct][1]), p23 = split2(In[ct][2], In[ct][3]);
            const unsigned h23 = (bias && ct == NT - 1) ? p23.hi | t_bias : p23.hi;
            ah[ct] = u32x4{p01.hi, h23, p01.hi, h23};
            al[ct] = u32x4{p01.lo, p23.lo, p01.lo, p23.lo};
        }
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) Out[jt] = zero;
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                const u32x4 th = theta_lds[((tab * NT + ct) * NT + jt) * 64 + lane];
                Out[jt] = mfma16(ah[ct], th, Out[jt]);
                Out[jt] = mfma16(al[ct], th, Out[jt]);
            }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto leaky_of = [&](const f32x4 (&Hp)[NT], float (&H)[NT][3]) {
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            keep_until_here(Hp[jt][3]);
#pragma unroll
            for (int r = 0; r < 3; ++r) H[jt][r] = fmaf((1.f - LEAKY) / (1.f + LEAKY), __builtin_fabsf(Hp[jt][r]), Hp[jt][r]);
        }
    };
    // z = W x [D ; D shifted] over the whole patch axis (the shift crosses tile boundaries); `rd` picks the direction / distance
    auto stage_conv = [&](const float (&D)[NT][3], float partner, const ConvOp& w, auto tap, f32x4 (&z)[NT], Pk (&keep)[NT]) {
        constexpr int TAP = decltype(tap)::value;                  // -1 | -2: the causal taps; +1 | +2: the transposed convolution's
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            keep[jt] = pack3(D[jt][0], D[jt][1], D[jt][2], partner);
            sh_tile[shb + 16 * jt] = keep[jt].hi;
            sh_tile[shb + 16 * jt + SH_LO] = keep[jt].lo;
        }
        __builtin_amdgcn_wave_barrier();
        u32x4 bh[NT], bl[NT];
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            const u32x2 ph = sh_tile[shb + 16 * jt + TAP], pl = sh_tile[shb + 16 * jt + TAP + SH_LO];
            bh[jt] = cat(keep[jt].hi, ph);
            bl[jt] = cat(keep[jt].lo, pl);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            z[jt] = mfma16z(w.hi, bh[jt]);
            z[jt] = mfma16(w.hi, bl[jt], z[jt]);
            z[jt] = mfma16(w.lo, bh[jt], z[jt]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // one layer in full: X <- dropout(relu(BN(z2)) + o0) + X
    auto layer_full = [&](float (&X)[NT][3], const u32x4& adjB, const LayerK& k, int tab, uint32_t key, uint32_t sbase, float (*xh1_out)[NT][3],
                          float (*y2_out)[NT][3], float (*q_out)[NT][3], uint32_t* mbits_out) {
        f32x4 T[NT], Hp[NT], z[NT];
        float H[NT][3], V[NT][3];
        Op2 xo[NT];
        Pk pk[NT];
        stage_T(X, adjB, T, xo);
        stage_theta(T, tab, true, Hp);
        leaky_of(Hp, H);
        stage_conv(H, 1.0f, k.w[0], std::integral_constant<int, -1>{}, z, pk);
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            keep_until_here(z[jt][3]);
#pragma unroll
            for (int r = 0; r < 3; ++r) V[jt][r] = relu2(fmaf(2.f, H[jt][r], relu2(fmaf(k.gam[0][r], z[jt][r], k.bet[0][r]))));
        }
        stage_conv(V, 1.0f, k.w[1], std::integral_constant<int, -2>{}, z, pk);
        uint32_t mbits = 0u;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            keep_until_here(z[jt][3]);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float xh = z[jt][r];
                const float y2 = fmaf(k.gam[1][r], xh, k.bet[1][r]);
                if (xh1_out) (*xh1_out)[jt][r] = xh;
                if (y2_out) (*y2_out)[jt][r] = y2;
                float o1 = fmaf(0.5f, relu2(y2), 0.25f * V[jt][r]);
                bool keep = true;
                if (use_drop) {
                    const uint32_t h = lowbias32((sbase + dro[jt][r]) ^ key);
                    keep = h >= a.drop_thr;
                    o1 = keep ? o1 * a.drop_scale : 0.f;
                    mbits |= keep ? 1u << (3 * jt + r) : 0u;
                }
                if (q_out) (*q_out)[jt][r] = (keep && y2 > 0.f) ? xh : INFINITY;
                X[jt][r] = fmaf(colm[jt], o1, X[jt][r]);
            }
        }
        *mbits_out = mbits;
    };

    bool pend = false;
    int64_t pend_smp = 0;
    float pend_v[NT][3], pend_q[NT][3];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 3; ++r) pend_v[i][r] = pend_q[i][r] = 0.f;
    float pend_top0 = 0.f, pend_top1 = 0.f, pend_pred = 0.f;
    uint32_t pend_m = 0u, mask_next = 0u;
    constexpr bool MASK_IN = KIND == PH_G && BLK == 1;
    if constexpr (MASK_IN) {
        if (use_drop && smp < a.B) mask_next = a.mrec[LY][smp * 64 + lane];
    }

    for (; smp < a.B; smp += stride) {
        const int64_t nx = smp + stride;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (pend) {
            if constexpr (WITH_PREV) {
                st_rec(a.xrec[LY] + pend_smp * XS, pend_v);
                st_rec(a.qrec[LY] + pend_smp * XS, pend_q);
                if (use_drop) a.mrec[LY - 1][pend_smp * 64 + lane] = pend_m;
            }
            if constexpr (KIND == PH_G && BLK == 1) st_rec(a.sb + pend_smp * XS, pend_v);
            if constexpr (KIND == PH_TOP) {
                if (lane < N && a.do_backward) {
                    float* p = a.dtop + pend_smp * DS + lane;
                    p[0] = pend_top0;
                    p[N] = pend_top1;
                }
                if (lane == 0) a.pred[pend_smp] = pend_pred;
                if (use_drop && a.do_backward) a.mrec[LY][pend_smp * 64 + lane] = pend_m;
            }
        }
        // ---- inputs --------------------------------------------------------------------------------------------------------------
        float X[NT][3];
        ld_rec(off_X, X);
        u32x4 adjB;
        {
            unsigned q[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) q[r] = __builtin_bit_cast(unsigned, smem[aoff[r] >= 0 ? off_A + aoff[r] : off_zero]);
            adjB = u32x4{__builtin_amdgcn_perm(q[1], q[0], 0x05040100u), q[2] & 0xFFFFu, __builtin_amdgcn_perm(q[1], q[0], 0x07060302u), q[2] >> 16};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (!LATE && nx < a.B) req_XA(nx);
        const uint32_t sbase = (uint32_t)((a.sample_offset + smp) * F) * (uint32_t)N;

        if constexpr (KIND == PH_F) {
            if constexpr (WITH_PREV) {
                layer_full(X, adjB, kp, 1, dkey[LY - 1], sbase, nullptr, nullptr, &pend_q, &pend_m);
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int r = 0; r < 3; ++r) pend_v[i][r] = X[i][r];
            }
            f32x4 T[NT], Hp[NT], z[NT];
            float H[NT][3];
            Op2 xo[NT];
            Pk pk[NT];
            stage_T(X, adjB, T, xo);
            stage_theta(T, 0, true, Hp);
            leaky_of(Hp, H);
            if constexpr (BLK == 0) {
                stage_conv(H, 0.f, kc.w[0], std::integral_constant<int, -1>{}, z, pk);
            } else {
                stage_conv(H, 1.0f, kc.w[0], std::integral_constant<int, -1>{}, z, pk);
                float V[NT][3];
#pragma unroll
                for (int jt = 0; jt < NT; ++jt) {
                    keep_until_here(z[jt][3]);
#pragma unroll
                    for (int r = 0; r < 3; ++r) V[jt][r] = relu2(fmaf(2.f, H[jt][r], relu2(fmaf(kc.gam[0][r], z[jt][r], kc.bet[0][r]))));
                }
                stage_conv(V, 0.f, kc.w[1], std::integral_constant<int, -2>{}, z, pk);
            }
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                keep_until_here(z[jt][3]);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const float zv = z[jt][r] * colm[jt];
                    s_a[r] += zv;
                    s_b[r] = fmaf(zv, zv, s_b[r]);
                }
            }
            if constexpr (WITH_PREV) { pend = true; pend_smp = smp; }
            continue;
        }

        if constexpr (KIND == PH_TOP) {
            float xh1[NT][3], y2[NT][3];
            layer_full(X, adjB, kc, 0, dkey[LY], sbase, &xh1, &y2, nullptr, &pend_m);
            // max over the ten channels with its arg-max; after the register / row transpose lane = patch (16 ct + col)
            float pm[4], pa[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                if (ct < NT) {
                    const float v0 = X[ct][0], v1 = g == 3 ? X[ct][0] : X[ct][1], v2 = g == 3 ? X[ct][0] : X[ct][2];
                    float m = v0;
                    int ar = 0;
                    if (v1 > m) { m = v1; ar = 1; }
                    if (v2 > m) { m = v2; ar = 2; }
                    pm[ct] = fmaf(v0 + v1 + v2, 0.f, m);
                    pa[ct] = __builtin_bit_cast(float, 3 * g + ar);
                } else {
                    pm[ct] = 0.f;
                    pa[ct] = 0.f;
                }
            }
            transpose_rows4(pm[0], pm[1], pm[2], pm[3]);              // in: register = column tile, row = group; out: register = group, row = column tile
            transpose_rows4(pa[0], pa[1], pa[2], pa[3]);
            float pooled = pm[0];
            int arg = __builtin_bit_cast(int, pa[0]);
#pragma unroll
            for (int q = 1; q < 4; ++q) {
                const bool take = pm[q] > pooled;
                pooled = take ? pm[q] : pooled;
                arg = take ? __builtin_bit_cast(int, pa[q]) : arg;
            }
            pooled = fmaf((pm[0] + pm[1]) + (pm[2] + pm[3]), 0.f, pooled);
            const bool valid = lane < N;
            pooled = valid ? pooled : 0.f;
            float* scr = smem + off_scr;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (lane < W) scr[lane] = pooled;
            __builtin_amdgcn_wave_barrier();
            float y1 = fc1b;
            const int jl = lane < W ? lane : 0;
            {
                const float4* row = reinterpret_cast<const float4*>(fc1N + jl * FCP);
                const float4* vec = reinterpret_cast<const float4*>(scr);
#pragma unroll
                for (int k4 = 0; k4 < W / 4; ++k4) {
                    const float4 wv = row[k4], pv = vec[k4];
                    y1 = fmaf(wv.x, pv.x, y1); y1 = fmaf(wv.y, pv.y, y1); y1 = fmaf(wv.z, pv.z, y1); y1 = fmaf(wv.w, pv.w, y1);
                }
            }
            y1 = valid ? relu(y1) : 0.f;
            const float pred = Row<64>::allsum(y1 * fc2w) + fc2b;
            const float diff = pred - a.y[smp];
            const float dpred = 2.f * diff * inv_gb * a.gscale;        // x S
            if (lane == 0) acc_loss = fmaf(diff, diff, acc_loss);
            pend = true; pend_smp = smp; pend_pred = pred;
            if (!a.do_backward) continue;
            const float dy1 = (y1 > 0.f) ? dpred * fc2w : 0.f;         // lane j
            if (lane < W) scr[W + lane] = dy1;
            __builtin_amdgcn_wave_barrier();
            float dpool = 0.f;
            {
                const float4* row = reinterpret_cast<const float4*>(fc1T + jl * FCP);
                const float4* vec = reinterpret_cast<const float4*>(scr + W);
#pragma unroll
                for (int j4 = 0; j4 < W / 4; ++j4) {
                    const float4 wv = row[j4], dv = vec[j4];
                    dpool = fmaf(wv.x, dv.x, dpool); dpool = fmaf(wv.y, dv.y, dpool); dpool = fmaf(wv.z, dv.z, dpool); dpool = fmaf(wv.w, dv.w, dpool);
                }
            }
            dpool = valid ? dpool : 0.f;
            acc_w2 = fmaf(dpred, y1, acc_w2);
            acc_b2 += lane == 0 ? dpred : 0.f;
            acc_b1 += dy1;
            // d fc1.w[j][k] += d y1[j] pooled[k]: rank-1 updates per (j tile, k tile) on the fp32 matrix cores (only k-slice 0 is populated)
            {
                float av[NT], bv[NT];
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    const float d = scr[W + 16 * i + col], p = scr[16 * i + col];
                    av[i] = g == 0 ? d : 0.f;
                    bv[i] = g == 0 ? p : 0.f;
                }
#pragma unroll
                for (int jt = 0; jt < NT; ++jt)
#pragma unroll
                    for (int kt = 0; kt < NT; ++kt) acc_th[jt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[jt], bv[kt], acc_th[jt][kt], 0, 0, 0);
            }
            pend_top0 = dpool;
            pend_top1 = __builtin_bit_cast(float, arg);
            // the sums of the last BatchNorm want d X_L in the D layout
            if (lane < W) {
                scr[2 * W + lane] = dpool;
                scr[3 * W + lane] = __builtin_bit_cast(float, arg);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
                const float dv = scr[2 * W + 16 * ct + col];
                const int da = __builtin_bit_cast(int, scr[3 * W + 16 * ct + col]);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    float gq = (chan[r] == da) ? dv : 0.f;
                    if (use_drop) gq = (pend_m >> (3 * ct + r)) & 1u ? gq * a.drop_scale : 0.f;
                    const float dy = y2[ct][r] > 0.f ? gq : 0.f;
                    s_a[r] += dy;
                    s_b[r] = fmaf(dy, xh1[ct][r], s_b[r]);
                }
            }
            __builtin_amdgcn_wave_barrier();
            continue;
        }

        if constexpr (KIND == PH_G) {
            float gin[NT][3];
            if constexpr (GRAD_TOP && !LATE) {
#pragma unroll
                for (int ct = 0; ct < NT; ++ct) {
                    const int t = 16 * ct + col;
                    const float dv = smem[t < N ? off_DX + t : off_zero];
                    const int da = __builtin_bit_cast(int, smem[t < N ? off_DX + N + t : off_zero]);
#pragma unroll
                    for (int r = 0; r < 3; ++r) gin[ct][r] = (chan[r] == da) ? dv : 0.f;
                }
            } else if constexpr (GRAD_IN && !LATE) {
                ld_rec(off_DX, gin);
            }
            if constexpr (!LATE) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                if (nx < a.B) req_DX(nx);
            }
            uint32_t mask_cur = 0u;
            if constexpr (MASK_IN) {
                mask_cur = mask_next;
                if (use_drop && nx < a.B) mask_next = a.mrec[LY][nx * 64 + lane];
            }

            // layer LY forward again
            f32x4 T[NT], Hp[NT], z[NT], AXd[NT];
            float H[NT][3];
            Op2 xo[NT];
            stage_T(X, adjB, T, xo);
            stage_theta(T, 0, true, Hp);
            leaky_of(Hp, H);
            uint32_t hp_pos = 0u;
#pragma unroll
            for (int jt = 0; jt < NT; ++jt)
#pragma unroll
                for (int r = 0; r < 3; ++r) hp_pos |= Hp[jt][r] > 0.f ? 1u << (3 * jt + r) : 0u;
            Pk hk[NT];
            stage_conv(H, 1.0f, conv_fwd0(), std::integral_constant<int, -1>{}, z, hk);
            float xh0[NT][3], y1[NT][3];
            float gam0[3], bet0[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                gam0[r] = LATE ? cops[(25 + r) * 64 + lane] : kc.gam[0][r];
                bet0[r] = LATE ? cops[(28 + r) * 64 + lane] : kc.bet[0][r];
            }
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                keep_until_here(z[jt][3]);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    xh0[jt][r] = z[jt][r];
                    y1[jt][r] = fmaf(gam0[r], z[jt][r], bet0[r]);
                }
            }
            float dz[NT][3], gsum[NT][3], V[NT][3];
            if constexpr (BLK == 1) {
#pragma unroll
                for (int jt = 0; jt < NT; ++jt)
#pragma unroll
                    for (int r = 0; r < 3; ++r) V[jt][r] = relu2(fmaf(2.f, H[jt][r], relu2(y1[jt][r])));
                stage_conv(V, 1.0f, kc.w[1], std::integral_constant<int, -2>{}, z, hk);             // hk <- V
#pragma unroll
                for (int jt = 0; jt < NT; ++jt) {
                    keep_until_here(z[jt][3]);
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const float xh = z[jt][r];
                        const float y2 = fmaf(kc.gam[1][r], xh, kc.bet[1][r]);
                        float gq = gin[jt][r];
                        if (use_drop) gq = (mask_cur >> (3 * jt + r)) & 1u ? gq * a.drop_scale : 0.f;
                        const bool x1pos = y2 > 0.f;
                        gsum[jt][r] = (x1pos || V[jt][r] > 0.f) ? gq : 0.f;
                        const float dy = x1pos ? gq : 0.f;
                        dz[jt][r] = bA[r] * (fmaf(-xh, bk2[r], dy) - bk1[r]) * colm[jt];
                    }
                }
            } else {
                float cA[3], c1[3], c2[3];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    cA[r] = cops[(16 + r) * 64 + lane];
                    c1[r] = cops[(19 + r) * 64 + lane];
                    c2[r] = cops[(22 + r) * 64 + lane];
                }
#pragma unroll
                for (int jt = 0; jt < NT; ++jt)
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const float sbv = smem[xoff[jt][r] >= 0 ? off_SB + xoff[jt][r] : off_zero];
                        const float dy = y1[jt][r] > 0.f ? sbv : 0.f;
                        dz[jt][r] = cA[r] * (fmaf(-xh0[jt][r], c2[r], dy) - c1[r]) * colm[jt];
                    }
            }
            // d(input of the convolution) = W^T-conv(d z); weight gradient from the transposed tiles
            f32x4 dI[NT];
            {
                Pk dzp[NT];
                stage_conv(dz, 0.f, conv_bwd(), std::integral_constant<int, DB>{}, dI, dzp);
                // per column tile: both tiles transposed (one identity product each), the tap at t - d built in the transposed arrangement --
                // lane group g of tile ct holds columns 16 ct + 4 g .. + 3 as two packed pairs; the pair above comes from lane - 16, or from
                // row group 3 of the tile before -- and the weight-gradient products; the hi . lo cross terms go two tiles per instruction
                const u32x2 z2 = u32x2{0u, 0u};
                Pk qp = Pk{z2, z2}, up = Pk{z2, z2}, usp = Pk{z2, z2};
                unsigned prev_hy = 0u, prev_ly = 0u;
#pragma unroll
                for (int ct = 0; ct < NT; ++ct) {
                    const f32x4 dzT = mfma16z(cat(dzp[ct].hi, dzp[ct].lo), ident);
                    const f32x4 hT = mfma16z(cat(hk[ct].hi, hk[ct].lo), ident);
                    const Pk q = pack3(dzT[0], dzT[1], dzT[2], dzT[3]);
                    const Pk u = pack3(hT[0], hT[1], hT[2], hT[3]);
                    unsigned ah = (unsigned)__builtin_amdgcn_ds_bpermute((lane - 16) << 2, (int)u.hi.y);
                    unsigned al = (unsigned)__builtin_amdgcn_ds_bpermute((lane - 16) << 2, (int)u.lo.y);
                    if (ct > 0) {
                        const unsigned bh2 = (unsigned)__builtin_amdgcn_ds_bpermute((lane + 48) << 2, (int)prev_hy);
                        const unsigned bl2 = (unsigned)__builtin_amdgcn_ds_bpermute((lane + 48) << 2, (int)prev_ly);
                        ah = g == 0 ? bh2 : ah;
                        al = g == 0 ? bl2 : al;
                    } else {
                        ah = g == 0 ? 0u : ah;
                        al = g == 0 ? 0u : al;
                    }
                    Pk us;
                    if constexpr (BLK == 1) {
                        us = Pk{u32x2{ah, u.hi.x}, u32x2{al, u.lo.x}};
                    } else {
                        us = Pk{u32x2{__builtin_amdgcn_perm(u.hi.x, ah, 0x05040302u), __builtin_amdgcn_perm(u.hi.y, u.hi.x, 0x05040302u)},
                                u32x2{__builtin_amdgcn_perm(u.lo.x, al, 0x05040302u), __builtin_amdgcn_perm(u.lo.y, u.lo.x, 0x05040302u)}};
                    }
                    acc_w0 = mfma16(cat(q.hi, q.lo), cat(u.hi, u.hi), acc_w0);
                    acc_w1 = mfma16(cat(q.hi, q.lo), cat(us.hi, us.hi), acc_w1);
                    if (ct % 2 == 1) {
                        acc_w0 = mfma16(cat(qp.hi, q.hi), cat(up.lo, u.lo), acc_w0);
                        acc_w1 = mfma16(cat(qp.hi, q.hi), cat(usp.lo, us.lo), acc_w1);
                    } else if (ct == NT - 1) {
                        acc_w0 = mfma16(cat(q.hi, z2), cat(u.lo, z2), acc_w0);
                        acc_w1 = mfma16(cat(q.hi, z2), cat(us.lo, z2), acc_w1);
                    } else {
                        qp = q; up = u; usp = us;
                    }
                    prev_hy = u.hi.y;
                    prev_ly = u.lo.y;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if constexpr (BLK == 1) {
#pragma unroll
                for (int jt = 0; jt < NT; ++jt) {
                    keep_until_here(dI[jt][3]);
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        float gq = dI[jt][r] + gsum[jt][r];
                        gq = V[jt][r] > 0.f ? gq * colm[jt] : 0.f;
                        pend_v[jt][r] = gq;
                        const float dy = y1[jt][r] > 0.f ? gq : 0.f;
                        s_a[r] += dy;
                        s_b[r] = fmaf(dy, xh0[jt][r], s_b[r]);
                    }
                }
                pend = true; pend_smp = smp;
            } else {
                Pk dk[NT];
#pragma unroll
                for (int jt = 0; jt < NT; ++jt) {
                    keep_until_here(dI[jt][3]);
                    float dHp[3];
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const float gq = dI[jt][r] + smem[xoff[jt][r] >= 0 ? off_SB + xoff[jt][r] : off_zero];
                        dHp[r] = ((hp_pos >> (3 * jt + r)) & 1u ? gq : gq * LEAKY) * colm[jt];
                        acc_b[jt] += dHp[r];
                    }
                    dk[jt] = pack3(dHp[0], dHp[1], dHp[2], 0.f);
                }
                {
                    // A x X (the other operand of the theta gradient) from the X record, which is still in LDS
                    Pk ax[NT];
                    const u32x2 z2 = u32x2{0u, 0u};
                    {
                        float X2[NT][3];
                        ld_rec(off_X, X2);
#pragma unroll
                        for (int ct = 0; ct < NT; ++ct) {
                            const Split2 p01 = split2(X2[ct][0], X2[ct][1]), p2 = split2(X2[ct][2], 0.f);
                            AXd[ct] = mfma16z(adjB, u32x4{p01.hi, p2.hi, p01.hi, p2.hi});
                            AXd[ct] = mfma16(adjB, u32x4{p01.lo, p2.lo, p01.lo, p2.lo}, AXd[ct]);
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_wave_barrier();
                        if (nx < a.B) { req_XA(nx); req_SB(nx); }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kt = 0; kt < NT; ++kt) {
                        keep_until_here(AXd[kt][3]);
                        ax[kt] = pack3(AXd[kt][0], AXd[kt][1], AXd[kt][2], 0.f);
                    }
#pragma unroll
                    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
                        for (int kt = 0; kt < NT; ++kt) {
                            acc_th[jt][kt] = mfma16(cat(dk[jt].hi, dk[jt].lo), cat(ax[kt].hi, ax[kt].hi), acc_th[jt][kt]);
                            acc_th[jt][kt] = mfma16(cat(dk[jt].hi, z2), cat(ax[kt].lo, z2), acc_th[jt][kt]);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (LY >= 1) {
                    f32x4 U[NT], dXl[NT];
#pragma unroll
                    for (int jt = 0; jt < NT; ++jt) {
                        U[jt] = mfma16z(cat(dk[jt].hi, dk[jt].hi), adjB);
                        U[jt] = mfma16(cat(dk[jt].lo, dk[jt].lo), adjB, U[jt]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    stage_theta(U, NTH - 1, false, dXl);
                    float Q[NT][3], dXo[NT][3];
                    if constexpr (GRAD_TOP) {
#pragma unroll
                        for (int ct = 0; ct < NT; ++ct) {
                            const int t = 16 * ct + col;
                            const float dv = smem[t < N ? off_DX + t : off_zero];
                            const int da = __builtin_bit_cast(int, smem[t < N ? off_DX + N + t : off_zero]);
#pragma unroll
                            for (int r = 0; r < 3; ++r) gin[ct][r] = (chan[r] == da) ? dv : 0.f;
                        }
                    } else {
                        ld_rec(off_DX, gin);
                    }
                    ld_rec(off_Q, Q);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                    if (nx < a.B) { req_DX(nx); req_Q(nx); }
#pragma unroll
                    for (int kt = 0; kt < NT; ++kt) {
                        keep_until_here(dXl[kt][3]);
#pragma unroll
                        for (int r = 0; r < 3; ++r) {
                            const float dX = (dXl[kt][r] + gin[kt][r]) * colm[kt];
                            dXo[kt][r] = dX;
                            const bool open = Q[kt][r] < INFINITY;
                            const float dy = open ? dX * a.drop_scale : 0.f;
                            s_a[r] += dy;
                            s_b[r] = fmaf(dy, open ? Q[kt][r] : 0.f, s_b[r]);
                        }
                    }
                    st_rec(a.dx + smp * XS, dXo);
                }
            }
        }
    }

    // ---- the last sample's outputs ---------------------------------------------------------------------------------------------------
    if (pend) {
        if constexpr (WITH_PREV) {
            st_rec(a.xrec[LY] + pend_smp * XS, pend_v);
            st_rec(a.qrec[LY] + pend_smp * XS, pend_q);
            if (use_drop) a.mrec[LY - 1][pend_smp * 64 + lane] = pend_m;
        }
        if constexpr (KIND == PH_G && BLK == 1) st_rec(a.sb + pend_smp * XS, pend_v);
        if constexpr (KIND == PH_TOP) {
            if (lane < N && a.do_backward) {
                float* p = a.dtop + pend_smp * DS + lane;
                p[0] = pend_top0;
                p[N] = pend_top1;
            }
            if (lane == 0) a.pred[pend_smp] = pend_pred;
            if (use_drop && a.do_backward) a.mrec[LY][pend_smp * 64 + lane] = pend_m;
        }
    }

    // ---- epilogue ---------------------------------------------------------------------------------------------------------------------
    StepScratch* const sc = step_scratch(a.cells, L);
    bool bad = false;
    const bool has_pair = (KIND == PH_F) || (KIND == PH_TOP && a.do_backward) || (KIND == PH_G && IDX > 0);
    constexpr int PBW = 2 * F + 2;
    if (has_pair) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            double da = (double)s_a[r], db = (double)s_b[r];
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) {
                da += __shfl_xor(da, off, 16);
                db += __shfl_xor(db, off, 16);
            }
            if (col == 0 && chan[r] >= 0) {
                pairbuf[wave * PBW + chan[r]] = da;
                pairbuf[wave * PBW + F + chan[r]] = db;
            }
        }
    }
    if constexpr (KIND == PH_TOP) {
        float vl = acc_loss;
        vl += __shfl_xor(vl, 16, 64);
        vl += __shfl_xor(vl, 32, 64);
        if (lane == 0) pairbuf[wave * PBW + 2 * F] = (double)vl;
    }
    __syncthreads();
    if (has_pair && threadIdx.x < 2 * F) {
        double* cell = a.cells + (int64_t)(blockIdx.x % CELL_REPLICAS) * CS;
        if (KIND == PH_F) cell += cell_fwd(L) + IDX * 2 * F;
        else if (KIND == PH_TOP) cell += cell_bwd(L) + (NBN - 1) * 2 * F;
        else cell += cell_bwd(L) + (IDX - 1) * 2 * F;
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < MXT_WAVES; ++w) v += pairbuf[w * PBW + threadIdx.x];
        atomicAdd(cell + threadIdx.x, KIND == PH_F ? v : v * (double)a.inv_gscale);
    }
    if (KIND == PH_TOP && threadIdx.x == 2 * F) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < MXT_WAVES; ++w) v += pairbuf[w * PBW + 2 * F];
        atomicAdd(a.cells + (int64_t)(blockIdx.x % CELL_REPLICAS) * CS + cell_loss(L), v);
        bad |= !(v <= 1.0e300 && v >= 0.0);
    }
    if constexpr (KIND == PH_F) return;
    if (KIND == PH_TOP && !a.do_backward) {
        if (__any(bad) && lane == 0) atomicOr(&sc->pad[0], 1u);
        return;
    }

    // ---- the workgroup's row of partial gradients (LDS image of the phase's contiguous parameter range) --------------------------------
    const float us = a.inv_gscale;
    const int NN = N * N;
    // two images (wavefronts 0 | 1 store, then 2 | 3 add), summed on the way out: two rounds, a fixed order of additions
    static_assert(2 * RED_FLOATS <= MXT_WAVES * (64 + 4 * W + SHIFT_FLOATS + 160 + MXW_ASTRIDE), "two row images fit the wavefronts' regions");
    for (int w = 0; w < 2; ++w) {
        if ((wave >> 1) == w) {
            float* const img = red + (wave & 1) * RED_FLOATS;
            auto put = [&](int idx, float v) { img[idx] = (w == 0) ? v : img[idx] + v; };
            // [N][N] matrix (theta or fc1): tile (jt, kt) in the MFMA D layout, row j = 16 jt + 4 g + r, column k = 16 kt + col
            if (KIND == PH_TOP || BLK == 0) {
#pragma unroll
                for (int jt = 0; jt < NT; ++jt)
#pragma unroll
                    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int j = 16 * jt + 4 * g + r, k = 16 * kt + col;
                            if (j < N && k < N) put(j * N + k, acc_th[jt][kt][r]);
                        }
            }
            if constexpr (KIND == PH_TOP) {
                if (lane < N) { put(NN + lane, acc_b1); put(NN + N + lane, acc_w2); }      // lane = j
                float v2 = acc_b2;
                if (lane == 0) put(NN + 2 * N, v2);
            }
            if constexpr (KIND == PH_G) {
                const int ci = slot_chan(col);
                const int cbase = BLK == 0 ? NN + N : 0;
                const float ws = BLK == 1 ? 0.25f : 1.f;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int co = chan[r];
                    if (co >= 0 && ci >= 0) {
                        put(cbase + (co * F + ci) * 2 + 1, acc_w0[r] * ws);
                        put(cbase + (co * F + ci) * 2 + 0, acc_w1[r] * ws);
                    }
                }
                if constexpr (BLK == 0) {
#pragma unroll
                    for (int jt = 0; jt < NT; ++jt) {
                        float vb = acc_b[jt];
                        vb += __shfl_xor(vb, 16, 64);
                        vb += __shfl_xor(vb, 32, 64);
                        const int j = 16 * jt + col;
                        if (g == 0 && j < N) put(NN + j, vb);
                    }
                }
            }
        }
        __syncthreads();
    }
    int rbase = 0, rlen = 0;
    if constexpr (KIND == PH_TOP) { rbase = off_fc1_w(N, L); rlen = NN + 2 * N + 1; }
    else if constexpr (BLK == 0) { rbase = LY * LS + off_theta_w(N); rlen = NN + N + CONVW; }
    else { rbase = LY * LS + off_conv_w(N, 1); rlen = CONVW; }
    float* row = a.gpart + (size_t)blockIdx.x * a.pcount + rbase;
    for (int i = threadIdx.x; i < rlen; i += 64 * MXT_WAVES) {
        const float v = (red[i] + red[RED_FLOATS + i]) * us;
        row[i] = v;
        bad |= !finite_f(v);
    }
    if (__any(bad) && lane == 0) atomicOr(&sc->pad[0], 1u);
}

// =====================================================================================================================
// host side
// =====================================================================================================================
static size_t mxtw_lds_bytes(int L, int kind, int idx, int N, int NT) {
    const int W = 16 * NT;
    const int blk = kind == PH_TOP ? 1 : idx % 2, ly = kind == PH_TOP ? L - 1 : idx / 2;
    const bool with_prev = kind == PH_F && blk == 0 && ly >= 1, bwd_prev = kind == PH_G && blk == 0 && ly >= 1;
    const bool need_sb = kind == PH_G && blk == 0, grad_in = kind == PH_G && (blk == 1 || ly >= 1);
    const int nth = 1 + (with_prev ? 1 : 0) + (bwd_prev ? 1 : 0);
    const int XS = mxw_xstride(N);
    const bool late = kind == PH_G && blk == 0;
    const size_t shared = (size_t)nth * NT * NT * 64 * 4 + (kind == PH_TOP ? 2 * W * (W + 4) : 0) + ((2 * L * MXT_BNC * F + 3) & ~3) + 2 * MXT_WAVES * (2 * F + 2) +
                          (late ? 31 * 64 : 0);
    const size_t red = (W * W + W + CONVW + 3) & ~3;
    const size_t wave = 64 + 4 * W + (((2 * (4 * (W + 2) + 2) * 2) + 3) & ~3) + XS + MXW_ASTRIDE + (need_sb ? XS : 0) + (grad_in ? XS : 0) + (bwd_prev ? XS : 0);
    return (shared + (MXT_WAVES * wave > red ? MXT_WAVES * wave : red)) * sizeof(float);
}

// Two workgroups per CU: the backward phases hold > 128 registers; a third and fourth workgroup of the forward phases (~110 registers)
// were measured (F_1 / F_3 at 40 x 64, batch 16384: 23.7 / 23.1 / 23.7 us with 2 / 3 / 4) -- the issue port is the limit, not latency
template <typename K>
static int mxtw_grid(K kern, size_t lds, int64_t B, int max_grid, int* grid_out) {
    constexpr int cap = MX_WAVES_PER_SIMD;
    if (lds > 160 * 1024) return RULGNN_EUNSUPPORTED;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return RULGNN_EHIP;
    int dev = 0, cus = 256, per_cu = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 64 * MXT_WAVES, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    if (per_cu > cap) per_cu = cap;
    int64_t grid = (int64_t)cus * per_cu;
    const int64_t want = (B + MXT_WAVES - 1) / MXT_WAVES;
    if (grid > want) grid = want;
    if (grid > max_grid) grid = max_grid;
    if (grid_out) *grid_out = (int)grid;
    return RULGNN_OK;
}

template <int L, int KIND, int IDX, int NT>
static int mxtw_launch(const MxTrainK& k, hipStream_t stream, int max_grid, int* grid_out) {
    const size_t lds = mxtw_lds_bytes(L, KIND, IDX, k.N, NT);
    auto go = [&](auto kern) -> int {
        int grid = 0;
        const int rc = mxtw_grid(kern, lds, k.B, max_grid, &grid);
        if (rc != RULGNN_OK) return rc;
        if (grid_out) *grid_out = grid;
        (void)hipGetLastError();
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * MXT_WAVES), lds, stream, k);
        return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
    };
    if constexpr (NT == 3) {
        if (k.N == 40) return go(&stgcn_train_mxw_kernel<L, KIND, IDX, NT, 40>);          // PHM2012's wiring
    }
    return go(&stgcn_train_mxw_kernel<L, KIND, IDX, NT, 0>);
}

template <int L, int NT, int I>
struct MxtwPhase {
    static int run(int kind, int idx, const MxTrainK& k, hipStream_t st, int mg, int* go) {
        if (idx == I) {
            if (kind == PH_F) {
                if constexpr (I >= 1) return mxtw_launch<L, PH_F, I, NT>(k, st, mg, go);
                else return RULGNN_EINVAL;
            }
            if (kind == PH_G) return mxtw_launch<L, PH_G, I, NT>(k, st, mg, go);
        }
        if constexpr (I > 0) return MxtwPhase<L, NT, I - 1>::run(kind, idx, k, st, mg, go);
        return RULGNN_EINVAL;
    }
};

bool stgcn_train_mxw_shape_ok(const rulgnn_stgcn_shape* s, const float* x) {
    const int N = s->num_patch, P = s->patch_size, L = s->num_layers;
    if (N < 16 || N > 47 || L < 1 || L > 2 || s->mpnn_k != 1) return false;
    if (((int64_t)N * P) % 4 != 0 || (reinterpret_cast<uintptr_t>(x) & 15) != 0) return false;
    if ((size_t)N * P * 4 > 40 * 1024) return false;                    // the window buffer of a wavefront
    return true;
}

static MxTrainK mxtw_args(const MxTrainArgs& m) {
    MxTrainK k;
    k.prm = m.prm; k.y = m.y; k.pred = m.pred; k.cells = m.cells; k.gpart = m.gpart;
    for (int l = 0; l < MX_MAX_LAYERS; ++l) { k.xrec[l] = m.xrec[l]; k.qrec[l] = m.qrec[l]; k.mrec[l] = m.mrec[l]; }
    k.arec = m.arec; k.sb = m.sb; k.dx = m.dx; k.dtop = m.dtop;
    k.B = m.B; k.ntiles = m.B; k.global_batch = m.global_batch; k.sample_offset = m.sample_offset;
    k.N = m.N; k.pcount = m.pcount;
    k.dropout_p = m.dropout_p; k.drop_scale = m.drop_scale; k.drop_thr = m.drop_thr;
    k.gscale = stgcn_train_mx_grad_scale(m.global_batch);
    k.inv_gscale = 1.0f / k.gscale;
    k.do_backward = m.do_backward;
    return k;
}

int stgcn_train_mxw_f0(const MxTrainArgs& m, const float* x, int P, hipStream_t stream, const HeadScalars* head) {
    const MxTrainK k = mxtw_args(m);
    HeadScalars hs{};
    if (head) hs = *head;
    if (m.B == 0) return RULGNN_OK;
    const int NT = m.N <= 31 ? 2 : 3, W = 16 * NT, PT = W + 4;
    const int buf_floats = (m.N * P + 3) & ~3;
    const int region = ((16 * PT > 2 * (4 * W + 1) * 2 ? 16 * PT : 2 * (4 * W + 1) * 2) + 64 + 3) & ~3;
    const size_t lds = ((size_t)NT * NT * 64 * 4 + 2 * MXT_WAVES * 2 * F + (size_t)MXT_WAVES * (buf_floats + region)) * sizeof(float);
    auto go = [&](auto kern) -> int {
        int grid = 0;
        const int rc = mxtw_grid(kern, lds, m.B, 1 << 30, &grid);
        if (rc != RULGNN_OK) return rc;
        (void)hipGetLastError();
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * MXT_WAVES), lds, stream, x, k, P, buf_floats, cell_stride(m.L), hs);
        return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
    };
    if (m.N == 40 && P == 64) return go(&stgcn_train_f0_mxw_kernel<3, 40, 64>);
    if (NT == 2) return go(&stgcn_train_f0_mxw_kernel<2, 0, 0>);
    return go(&stgcn_train_f0_mxw_kernel<3, 0, 0>);
}

int stgcn_train_mxw_phase(const MxTrainArgs& m, int kind, int idx, hipStream_t stream, int max_grid, int* grid_out) {
    const MxTrainK k = mxtw_args(m);
    if (m.B == 0) { if (grid_out) *grid_out = 0; return RULGNN_OK; }
    const int NT = m.N <= 31 ? 2 : 3;
    if (kind == PH_TOP) {
        if (m.L == 1) return NT == 2 ? mxtw_launch<1, PH_TOP, 0, 2>(k, stream, max_grid, grid_out) : mxtw_launch<1, PH_TOP, 0, 3>(k, stream, max_grid, grid_out);
        if (m.L == 2) return NT == 2 ? mxtw_launch<2, PH_TOP, 0, 2>(k, stream, max_grid, grid_out) : mxtw_launch<2, PH_TOP, 0, 3>(k, stream, max_grid, grid_out);
        return RULGNN_EUNSUPPORTED;
    }
    if (m.L == 1) return NT == 2 ? MxtwPhase<1, 2, 1>::run(kind, idx, k, stream, max_grid, grid_out) : MxtwPhase<1, 3, 1>::run(kind, idx, k, stream, max_grid, grid_out);
    if (m.L == 2) return NT == 2 ? MxtwPhase<2, 2, 3>::run(kind, idx, k, stream, max_grid, grid_out) : MxtwPhase<2, 3, 3>::run(kind, idx, k, stream, max_grid, grid_out);
    return RULGNN_EUNSUPPORTED;
}

}  // namespace rulgnn
