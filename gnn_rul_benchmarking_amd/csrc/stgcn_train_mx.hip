// Train-mode ST_GCN forward + backward on the f16 matrix cores of gfx950 for num_patch <= 15 (the C-MAPSS shapes): the phase chain of
// stgcn_train.hip with every contraction on v_mfma_f32_16x16x32_f16 (2-way split operands, fp32 accumulation: stgcn_mx.hpp) and --
// the point of this file -- with RECOMPUTATION instead of saved activations.
//
// Reference path replaced: ST_GCN.update up to optimizer.step() -- algorithms/algorithms.py:481-488 (model(X) under model.train(),
// MSE, loss.backward()); the layer is models/ST_GCN/Model.py:74-90 (MPNN_mk), :134-170 (TemporalConvNet), :187-195 (SG_TCN).
//
// Why.  The row-mapped chain hands eight [10, N] activation tensors per sample from phase to phase (H, z1, o0, z2 of every layer):
// 26.0 KB of HBM traffic per sample against 1.7 KB of input, 4.2 TB/s of real traffic -- the step was bound by its own saved tensors.
// A layer forward on the matrix cores costs ~10 MFMAs per sample, so here every phase re-derives what it needs from the layer INPUT:
//
//     F_0 (stgcn_forward_mx.hip)  windows -> X_0, adjacency (55 floats), sum z1, sum z1^2 of BatchNorm 0
//     F_{2l+1}                    X_l, A -> layer l up to conv_block2 -> sums of BatchNorm 2l+1                     nothing written
//     F_{2l}, l >= 1              X_{l-1}, A -> layer l-1 in full -> X_l (written) -> layer l up to conv_block1 -> sums of BatchNorm 2l
//     TOP                         X_{L-1}, A -> layer L-1, head, loss, head backward -> d X_L as (value, arg-max channel), sums of BN 2L-1
//     G_{2l+1}                    X_l, A, d X_{l+1} -> layer l again, BatchNorm 2l+1 / conv_block2 backward -> d(x0 + H) (written)
//     G_{2l}                      X_l, A, d(x0 + H) -> BatchNorm 2l / conv_block1 / theta backward; l >= 1: d X_l (written) and, with
//                                 layer l-1 recomputed from X_{l-1}, the sums of BatchNorm 2l-1
//
// Per sample at 14 x 30, L = 2: 13.5 KB instead of 26.0 KB.  Between phases only layer inputs and two gradient tensors cross HBM.
//
// Layout.  The "D layout" of stgcn_forward_mx.hip: one 16x16 tile per sample, column = patch t = lane & 15, row = channel slot
// 4 (lane >> 4) + r; a [10, N] tensor is three registers and four samples are in flight per wavefront.  The backward uses the same
// chains with transposed constant operands:
//     d o0 = W2^T-conv(d z2), d H = W1^T-conv(d z1)    A operand [ci][tap, co], data = [d z | d z of column t + d]         3 MFMAs
//     d X  = A . (d Hp . theta) = ((A d Hp)^T)^T-chain: U = d Hp^T x Adj (2 MFMAs), d X = U x theta (2 MFMAs)
//     d theta[j][k] = sum_c d Hp[c][j] (A X)[c][k]      contraction over channel slots = the packed D registers themselves
//     d W[co][ci, tap] = sum_t d z[co][t] h[ci][t - d tap]: contraction over COLUMNS -- both operands transposed by one MFMA
//                                                      against an identity operand ({hi | lo} halves of K: exact to 22 bits)
// Weight-gradient accumulators live in MFMA accumulator registers across the persistent tile loop; every wavefront writes one row of
// partial gradients, summed in a fixed order by stgcn_train_finalize_kernel (shared with the fp32 chain, as are prepare and the cells).
//
// f16 range.  Gradients are carried multiplied by S = 2^(ceil(log2 global_batch) + 3) (d pred = 2 diff / B would sit in f16's
// subnormals) and unscaled exactly when rows / sums leave the wavefront.  Activations are not rescaled: statistics of inputs scaled to
// O(1) (every dataset the reference wires) stay far inside the f16 range.  A value that leaves it ends as Inf / NaN in a BatchNorm
// sum, the loss or a gradient row; every TOP / G wavefront checks what it writes and raises the step's status word
// (StepScratch::pad[0]), on which finalize leaves parameters and optimizer state untouched and reports a NaN loss -- the caller
// repeats the step on the exact fp32 chain (RULGNN_STEP_CHAIN; stgcn.py does).
//
// Memory.  Inputs arrive by LDS-DMA one tile ahead (each record is requested again as soon as its last LDS read has retired), outputs
// leave one tile late, right behind the s_waitcnt vmcnt(0) that also counts stores (stgcn_forward_mx.hip, round 3).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "stgcn_host.hpp"
#include "stgcn_mx.hpp"
#include "stgcn_train_layout.hpp"
#include "stgcn_train_mx.hpp"
#include "stgcn_train_mx_ops.hpp"

namespace rulgnn {

// =====================================================================================================================
// the phase kernel (everything but F_0)
// =====================================================================================================================
// The body of a phase.  PERSIST: the phase runs inside the single small-batch launch (stgcn_train_mx_persist_kernel below): the sums its
// BatchNorm constants come from were completed by the other workgroups of the SAME launch -- it waits for their arrivals (`target` on the
// step's counter) right in front of the cell reads, i.e. behind its first tile's requests and the BatchNorm-independent half of the
// prologue, and reads the cells with agent-scope atomic loads.  Everything else a phase reads was written by the wavefront itself (tile
// t belongs to the same wavefront in every phase of a launch) or by an earlier launch.
#ifdef MXP_TRACE
__device__ unsigned mxp_ts[128];
__device__ int mxp_n;
__device__ unsigned mxp_id[128];
#define MXP_MARKI(ID) do { if (PERSIST && blockIdx.x == 0 && threadIdx.x == 0 && mxp_n < 128) { mxp_id[mxp_n] = ID; mxp_ts[mxp_n++] = (unsigned)wall_clock64(); } } while (0)
#define MXP_MARK() MXP_MARKI(0)
#else
#define MXP_MARK() do {} while (0)
#define MXP_MARKI(ID) do {} while (0)
#endif
template <int L, int KIND, int IDX, int NFIX, bool PERSIST>
__device__ __forceinline__ void mxt_phase_body(const MxTrainK& a, float* smem_all, unsigned target) {
    MXP_MARKI(100 + KIND * 10 + IDX);                                              // phase entry
    const int N = NFIX ? NFIX : a.N;
    const int LS = layer_stride(N);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, col = lane & 15;
    constexpr int NBN = 2 * L;
    constexpr int CS = cell_stride(L);
    constexpr int LY = KIND == PH_TOP ? L - 1 : IDX / 2;
    constexpr int BLK = KIND == PH_TOP ? 1 : IDX % 2;
    constexpr bool WITH_PREV = KIND == PH_F && BLK == 0 && LY >= 1;     // F_{2l}: layer l-1 in full first (its input is this phase's input)
    constexpr bool BWD_PREV = KIND == PH_G && BLK == 0 && LY >= 1;      // G_{2l}: the sums of BatchNorm 2l-1 (its gated x-hat comes from F_{2l})
    constexpr int LIN = WITH_PREV ? LY - 1 : LY;                        // the layer whose input record is the main input
    constexpr bool GRAD_IN = KIND == PH_G && (BLK == 1 || LY >= 1);     // a gradient tensor enters: d X_{l+1}
    constexpr bool GRAD_TOP = GRAD_IN && LY == L - 1;                   // ... in TOP's (value, arg-max) form
    constexpr bool NEED_SB = KIND == PH_G && BLK == 0;
    static_assert(!(KIND == PH_F && IDX == 0), "F_0 is stgcn_train_f0_mx_kernel");

    // ---- LDS carve (floats) ------------------------------------------------------------------------------------------------
    const int XF = 40 * N;                                   // one [10][4 N] tile
    constexpr int AF = 220;                                  // the adjacency tile
    // workgroup: [BatchNorm table | gradient row image | pair partials]; then one region per wavefront (the wavefronts only meet in the
    // prologue and the epilogue)
    constexpr int SH_BNC = (NBN * MXT_BNC * F + 3) & ~3;
    float* const bnc = smem_all;                                                 // [NBN][BN_TABLE_ROWS][F], the rows this phase uses
    float* const red = smem_all + SH_BNC;                                       // [MXT_RED_FLOATS]
    double* const pairbuf = reinterpret_cast<double*>(red + MXT_RED_FLOATS);    // [MXT_WAVES][2 F + 1]
    const int off_zero = 0;
    const int off_scr = off_zero + MXT_ZERO_FLOATS;
    const int off_sh = off_scr + MXT_SCRATCH_FLOATS;
    const int off_X = off_sh + MXT_SHIFT_FLOATS;
    const int off_A = off_X + XF;
    const int off_SB = off_A + AF;                           // G_{2l}
    const int off_DX = off_SB + (NEED_SB ? XF : 0);          // gradient in (full tile or TOP's two rows)
    const int off_XP = off_DX + (GRAD_IN ? XF : 0);          // G_{2l}, l >= 1: the gated x-hat of BatchNorm 2l-1
    const int wave_floats = off_XP + (BWD_PREV ? XF : 0);
    float* const smem = smem_all + SH_BNC + MXT_RED_FLOATS + 2 * MXT_WAVES * (2 * F + 2) + wave * wave_floats;
    u32x2* const sh_tile = reinterpret_cast<u32x2*>(smem + off_sh);

    int64_t tile = (int64_t)blockIdx.x * MXT_WAVES + wave;
    const int64_t tstride = (int64_t)gridDim.x * MXT_WAVES;

    // ---- requests --------------------------------------------------------------------------------------------------------
    auto dma = [&](const float* src, int off, int bytes) {
        if constexpr (NFIX != 0) {
            if (bytes == 160 * NFIX) { dma_tile_fixed<160 * NFIX>(src, smem + off, lane); return; }
            if (bytes == 32 * NFIX) { dma_tile_fixed<32 * NFIX>(src, smem + off, lane); return; }
        }
        if (bytes == 4 * AF) { dma_tile_fixed<4 * AF>(src, smem + off, lane); return; }
        dma_tile(src, smem + off, bytes, lane);
    };
    auto req_XA = [&](int64_t t) {
        dma(a.xrec[LIN] + t * XF, off_X, 4 * XF);
        dma(a.arec + t * AF, off_A, 4 * AF);
    };
    auto req_SB = [&](int64_t t) { if constexpr (NEED_SB) dma(a.sb + t * XF, off_SB, 4 * XF); };
    auto req_DX = [&](int64_t t) {
        if constexpr (GRAD_TOP) dma(a.dtop + t * (8 * N), off_DX, 32 * N);
        else if constexpr (GRAD_IN) dma(a.dx + t * XF, off_DX, 4 * XF);
    };
    auto req_XP = [&](int64_t t) { if constexpr (BWD_PREV) dma(a.qrec[LY] + t * XF, off_XP, 4 * XF); };
    if (tile < a.ntiles) { req_XA(tile); req_SB(tile); req_DX(tile); req_XP(tile); }

    // ---- prologue.  First everything that does not depend on a BatchNorm (parameter loads, theta / transposed-convolution / identity
    // operands); then the BatchNorm constants from the reduction cells (one pair per wavefront, stgcn_train_layout.hpp); then the operands
    // that fold a BatchNorm in ----------------------------------------------------------------------------------------------------
    for (int i = lane; i < MXT_ZERO_FLOATS; i += 64) smem[off_zero + i] = 0.f;
    if (lane < 2) sh_tile[64 + 65 * lane] = u32x2{0u, 0u};
    // which convolutions of the main layer / the previous layer this phase runs, and how
    constexpr int M0_LY = (KIND == PH_F && BLK == 0) ? 1 : 2;                          // conv_block1 of layer LY
    constexpr int M1_LY = (KIND == PH_F && BLK == 0) ? 0 : ((KIND == PH_F) ? 1 : ((KIND == PH_G && BLK == 0) ? 0 : 2));   // conv_block2
    MXP_MARKI(1);
    LayerRaw rc, rp;
    layer_raw(rc, a.prm, LY, N, g, col, M0_LY, M1_LY);
    if constexpr (WITH_PREV) layer_raw(rp, a.prm, LY - 1, N, g, col, 2, 2);
    ConvOp wT = ConvOp{u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}};
    ThetaOp thN = ThetaOp{u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}};
    u32x4 ident = u32x4{0u, 0u, 0u, 0u};
    ConvRaw wT_raw = ConvRaw{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};          // (converted behind the BatchNorm cells' loads: one round trip)
    ThetaRaw thN_raw = ThetaRaw{{0.f, 0.f, 0.f, 0.f}};
    if constexpr (KIND == PH_G) {
        wT_raw = conv_bwd_raw(a.prm + LY * LS + off_conv_w(N, BLK), g, col);
        if constexpr (BLK == 0 && LY >= 1) thN_raw = theta_n_raw(a.prm + LY * LS, N, g, col);
        unsigned w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = (4 * g + r == col) ? 0x3C00u : 0u;      // f16 1.0
        const unsigned p01 = w[0] | (w[1] << 16), p23 = w[2] | (w[3] << 16);
        ident = u32x4{p01, p23, p01, p23};
    }
    {
        // the reduction pairs this phase's BatchNorm constants come from, one per wavefront: the forward pairs of the layers it runs
        // (F_{2l+1}: 2l; F_{2l}, l >= 1: 2l-2, 2l-1; TOP, G_{2l+1}: 2l, 2l+1; G_{2l}: 2l) and the backward pair of BatchNorm IDX (G)
        constexpr int FW0 = WITH_PREV ? 2 * LY - 2 : 2 * LY;                       // first forward pair
        constexpr int NFW = (KIND == PH_F && BLK == 1) || (KIND == PH_G && BLK == 0) ? 1 : 2;
        static_assert(NFW + (KIND == PH_G ? 1 : 0) <= MXT_WAVES, "one reduction pair per wavefront");
        MXP_MARKI(2);                                                              // independent prologue done
        if constexpr (PERSIST) {
            if (threadIdx.x == 0) {
                unsigned* const ctr = step_barrier(a.cells, L);
                // (a workgroup that never arrives -- a grid that is not co-resident -- must not hang the GPU: after ~2^22 polls, seconds,
                // the step is rejected like a guard trip)
                unsigned spins = 0;
                while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 22)) { atomicOr(&step_scratch(a.cells, L)->pad[0], 4u); break; }
                }
            }
            __syncthreads();
        }
        constexpr bool COH = PERSIST;
        MXP_MARKI(3);                                                              // barrier passed
        if (wave < NFW) bn_pair_to_lds<COH>(a.cells, a.prm, bnc, L, N, true, FW0 + wave, lane);
        if (KIND == PH_G && wave == NFW) bn_pair_to_lds<COH>(a.cells, a.prm, bnc, L, N, false, IDX, lane);
        if constexpr (KIND == PH_G) {
            wT = conv_bwd_pack(wT_raw);
            if constexpr (BLK == 0 && LY >= 1) thN = theta_pack(thN_raw);
        }
        __syncthreads();
    }
    MXP_MARKI(4);                                                                  // BatchNorm table in LDS
    LayerK kc;                                   // layer LY
    layer_constants(kc, rc, bnc, LY, g, col, M0_LY, M1_LY);
    LayerK kp;                                   // layer LY - 1 (F_{2l}, l >= 1)
    if constexpr (WITH_PREV) layer_constants(kp, rp, bnc, LY - 1, g, col, 2, 2);
    float bA[3] = {0.f, 0.f, 0.f}, bk1[3] = {0.f, 0.f, 0.f}, bk2[3] = {0.f, 0.f, 0.f};     // BatchNorm IDX backward: gamma istd, S mean(dy), S mean(dy xhat)
    if constexpr (KIND == PH_G) {
        const float* q = bnc + IDX * MXT_BNC * F;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int c = slot_chan(4 * g + r);
            bA[r] = c >= 0 ? q[4 * F + c] : 0.f;
            bk1[r] = c >= 0 ? q[5 * F + c] * a.gscale : 0.f;      // carried x S like the gradients
            bk2[r] = c >= 0 ? q[6 * F + c] * a.gscale : 0.f;
        }
    }
    // head (TOP), row mapping: lane (sample row, t) holds row t and column t of fc1
    float fc1w[16], fc1wT[16];
    float fc1b = 0.f, fc2w = 0.f, fc2b = 0.f;
    if constexpr (KIND == PH_TOP) {
        const int colc = col < N ? col : 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int kc2 = k < N ? k : 0;
            const float v = a.prm[off_fc1_w(N, L) + colc * N + kc2], vt = a.prm[off_fc1_w(N, L) + kc2 * N + colc];
            fc1w[k] = (col < N && k < N) ? v : 0.f;
            fc1wT[k] = (col < N && k < N) ? vt : 0.f;
        }
        const float b1 = a.prm[off_fc1_b(N, L) + colc], w2 = a.prm[off_fc2_w(N, L) + colc];
        fc1b = col < N ? b1 : 0.f;
        fc2w = col < N ? w2 : 0.f;
        fc2b = a.prm[off_fc2_b(N, L)];
    }

    // ---- per-lane addressing ---------------------------------------------------------------------------------------------------
    const bool col_ok = col < N;
    const float colm = col_ok ? 1.f : 0.f;
    const int pitch = 4 * N;
    int xoff[3], aoff[3], chan[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int c = slot_chan(4 * g + r);
        chan[r] = c;
        xoff[r] = (c >= 0 && col_ok) ? c * pitch + col : -1;      // inside a [10][4 N] tile; -1: padding -> the zero words
    }
    {
        const int cc = slot_chan(col);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int c = slot_chan(4 * g + r);
            aoff[r] = (c >= 0 && cc >= 0) ? sym(c, cc) : -1;
        }
    }
    auto ld_tile = [&](int base, float (&v)[4][3]) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int r = 0; r < 3; ++r) v[s][r] = smem[(xoff[r] >= 0 ? base + xoff[r] : off_zero) + s * N];
    };
    auto st_tile = [&](float* dst, const float (&v)[4][3], int ns) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
            if (s < ns) {
#pragma unroll
                for (int r = 0; r < 3; ++r)
                    if (xoff[r] >= 0) dst[xoff[r] + s * N] = v[s][r];
            }
    };
    const int sh_rd1 = col >= 1 ? lane - 1 : 64, sh_rd2 = col >= 2 ? lane - 2 : 64;     // forward taps: column t - d
    int sh_rd1_lo = sh_rd1 + 65, sh_rd2_lo = sh_rd2 + 65;
    const int sh_bk = col + (BLK == 0 ? 1 : 2) < 16 ? lane + (BLK == 0 ? 1 : 2) : 64;   // transposed convolution of this phase: column t + d
    int sh_bk_lo = sh_bk + 65;
    asm volatile("" : "+v"(sh_rd1_lo), "+v"(sh_rd2_lo), "+v"(sh_bk_lo));
    const unsigned t_bias = g == 3 ? 0x3C00u << 16 : 0u;
    uint32_t dkey[L];
#pragma unroll
    for (int l = 0; l < L; ++l) dkey[l] = step_scratch(a.cells, L)->drop_key[l];
    uint32_t dro[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) dro[r] = (uint32_t)((chan[r] >= 0 ? chan[r] : 0) * N + col);
    const bool use_drop = a.dropout_p > 0.f;

    // ---- persistent accumulators ---------------------------------------------------------------------------------------------------
    float s_a[3] = {0.f, 0.f, 0.f}, s_b[3] = {0.f, 0.f, 0.f};      // BatchNorm reduction pair of this lane's channels
    f32x4 acc_w0 = {0.f, 0.f, 0.f, 0.f}, acc_w1 = {0.f, 0.f, 0.f, 0.f};      // conv weight gradient: tap at t / tap at t - d
    f32x4 acc_th = {0.f, 0.f, 0.f, 0.f};                                   // theta / fc1 weight gradient
    float acc_b = 0.f, acc_w2 = 0.f, acc_b2 = 0.f, acc_loss = 0.f;
    const float inv_gb = 1.0f / (float)a.global_batch;

    // ---- building blocks of a tile ------------------------------------------------------------------------------------------------------
    // T = (A X)^T, Hp' = (1 + a)/2 (theta (A X) + b), H = leaky(Hp)
    // (generic in the number of samples in flight: four in the forward phases, two per half tile in the backward phases, whose working set
    // does not fit 256 registers at four)
    auto stage_T = [&](const auto& X, const auto& adjB, auto& T, auto& xo) {
        constexpr int NS = std::extent_v<std::remove_reference_t<decltype(T)>>;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const Split2 p01 = split2(X[s][0], X[s][1]), p2 = split2(X[s][2], 0.f);
            const Op2 o = {u32x4{p01.hi, p2.hi, p01.hi, p2.hi}, u32x4{p01.lo, p2.lo, p01.lo, p2.lo}};
            xo[s] = o;
            T[s] = mfma16z(o.h, adjB[s]);
            T[s] = mfma16(o.l, adjB[s], T[s]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto stage_Hp = [&](const auto& T, const ThetaOp& th, auto& Hp) {
        constexpr int NS = std::extent_v<std::remove_reference_t<decltype(Hp)>>;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const Split2 p01 = split2(T[s][0], T[s][1]), p23 = split2(T[s][2], T[s][3]);
            const u32x4 ta = {p01.hi, p23.hi | t_bias, p01.lo, p23.lo};
            Hp[s] = mfma16z(ta, th.hi);
            Hp[s] = mfma16(ta, th.lo, Hp[s]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // z = W x [D ; D of column t - d] (+ shift x partner); keeps the packed operand halves of D and of the shifted column
    auto stage_conv = [&](const auto& D, float partner, const ConvOp& w, int rd, int rd_lo, auto& z, auto& keep, auto& keep_sh) {
        constexpr int NS = std::extent_v<std::remove_reference_t<decltype(z)>>;
        u32x4 bh[NS], bl[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const Pk p = pack3(D[s][0], D[s][1], D[s][2], partner);
            const Shifted prev = shift_read(sh_tile, rd, rd_lo, lane, p);
            bh[s] = cat(p.hi, prev.hi);
            bl[s] = cat(p.lo, prev.lo);
            keep[s] = p;
            keep_sh[s] = prev;
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            z[s] = mfma16z(w.hi, bh[s]);
            z[s] = mfma16(w.hi, bl[s], z[s]);
            z[s] = mfma16(w.lo, bh[s], z[s]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto leaky_of = [&](const auto& Hp, auto& H) {
        constexpr int NS = std::extent_v<std::remove_reference_t<decltype(H)>>;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            keep_until_here(Hp[s][3]);
#pragma unroll
            for (int r = 0; r < 3; ++r) H[s][r] = fmaf((1.f - LEAKY) / (1.f + LEAKY), __builtin_fabsf(Hp[s][r]), Hp[s][r]);
        }
    };
    // One layer in full (both BatchNorms known): X <- dropout(relu(BN(z2)) + o0) + X.  `sbase[s]`: dropout counter of (sample s, channel 0, patch 0).
    auto layer_full = [&](float (&X)[4][3], const u32x4 (&adjB)[4], const LayerK& k, uint32_t key, const uint32_t (&sbase)[4],
                          float (*xh1_out)[4][3], float (*y2_out)[4][3], float (*q_out)[4][3], uint32_t* mbits_out) {
        f32x4 T[4], Hp[4], z[4];
        float H[4][3], V[4][3];
        Op2 xo[4];
        Pk pk[4];
        Shifted ps[4];
        stage_T(X, adjB, T, xo);
        stage_Hp(T, k.th, Hp);
        leaky_of(Hp, H);
        stage_conv(H, 1.0f, k.w[0], sh_rd1, sh_rd1_lo, z, pk, ps);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            keep_until_here(z[s][3]);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float y1 = fmaf(k.gam[0][r], z[s][r], k.bet[0][r]);
                V[s][r] = relu2(fmaf(2.f, H[s][r], relu2(y1)));                  // 4 o0
            }
        }
        stage_conv(V, 1.0f, k.w[1], sh_rd2, sh_rd2_lo, z, pk, ps);
        uint32_t mbits = 0u;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            keep_until_here(z[s][3]);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float xh = z[s][r];
                const float y2 = fmaf(k.gam[1][r], xh, k.bet[1][r]);
                if (xh1_out) (*xh1_out)[s][r] = xh;
                if (y2_out) (*y2_out)[s][r] = y2;
                float o1 = fmaf(0.5f, relu2(y2), 0.25f * V[s][r]);               // relu(BN(z2)) + o0, both >= 0
                bool keep = true;
                if (use_drop) {
                    const uint32_t h = lowbias32((sbase[s] + dro[r]) ^ key);
                    keep = h >= a.drop_thr;
                    o1 = keep ? o1 * a.drop_scale : 0.f;
                    mbits |= keep ? 1u << (3 * s + r) : 0u;
                }
                // what the backward sums of this BatchNorm need: x-hat where the gradient passes the ReLU and the dropout, else +inf
                if (q_out) (*q_out)[s][r] = (keep && y2 > 0.f) ? xh : INFINITY;
                X[s][r] = fmaf(colm, o1, X[s][r]);
            }
        }
        *mbits_out = mbits;
    };

    bool pend = false;
    int64_t pend_tile = 0;
    int pend_ns = 0;
    float pend_v[4][3], pend_q[4][3];
    float pend_top0 = 0.f, pend_top1 = 0.f, pend_pred = 0.f;
    uint32_t pend_m = 0u;                       // dropout mask bits of the tile (F_{2l}, TOP: written; G_{2l+1}: this tile's, read one tile ahead)
    uint32_t mask_next = 0u;
    constexpr bool MASK_IN = KIND == PH_G && BLK == 1;
    if constexpr (MASK_IN) {
        if (use_drop && tile < a.ntiles) mask_next = a.mrec[LY][tile * 64 + lane];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int r = 0; r < 3; ++r) pend_v[s][r] = pend_q[s][r] = 0.f;

    MXP_MARKI(5);                                                                  // constants ready
    for (; tile < a.ntiles; tile += tstride) {
        const int64_t s0 = tile * 4;
        const int ns_tile = (int)((a.B - s0) < 4 ? (a.B - s0) : 4);
        // Two instantiations of the tile body: every tile but (at most) the last one is FULL -- there `ns` is the constant 4 and the
        // per-sample guards of the partial tile (selects on every gradient element, scalar branches around the sums) compile away.
        auto tile_body = [&](auto FULL_) {
        constexpr bool FULL = decltype(FULL_)::value;
        const int ns = FULL ? 4 : ns_tile;
        const int64_t nt = tile + tstride;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // ---- what the previous tile leaves: stored HERE, behind the wait that also counts stores ------------------------------------
        if (pend) {
            if constexpr (WITH_PREV) {
                st_tile(a.xrec[LY] + pend_tile * XF, pend_v, pend_ns);
                st_tile(a.qrec[LY] + pend_tile * XF, pend_q, pend_ns);
                if (use_drop) a.mrec[LY - 1][pend_tile * 64 + lane] = pend_m;
            }
            if constexpr (KIND == PH_G && BLK == 1) st_tile(a.sb + pend_tile * XF, pend_v, pend_ns);
            if constexpr (BWD_PREV) st_tile(a.dx + pend_tile * XF, pend_v, pend_ns);
            if constexpr (KIND == PH_TOP) {
                if (g < pend_ns && col_ok && a.do_backward) {
                    float* p = a.dtop + pend_tile * (8 * N) + g * N + col;
                    p[0] = pend_top0;
                    p[pitch] = pend_top1;
                }
                if (g < pend_ns && col == 0) a.pred[pend_tile * 4 + g] = pend_pred;
                if (use_drop && a.do_backward) a.mrec[LY][pend_tile * 64 + lane] = pend_m;
            }
        }
        // ---- inputs ------------------------------------------------------------------------------------------------------------------
        float X[4][3];
        u32x4 adjB[4];
        ld_tile(off_X, X);
        {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                // F_0 left every entry as its (hi | lo << 16) f16 pair: the operand is four byte permutes, no split (slot 3 of every
                // lane group is padding: always zero)
                unsigned q[3];
#pragma unroll
                for (int r = 0; r < 3; ++r) q[r] = __builtin_bit_cast(unsigned, smem[(aoff[r] >= 0 ? off_A + aoff[r] : off_zero) + s * 55]);
                adjB[s] = u32x4{__builtin_amdgcn_perm(q[1], q[0], 0x05040100u), q[2] & 0xFFFFu, __builtin_amdgcn_perm(q[1], q[0], 0x07060302u), q[2] >> 16};
            }
            if (ns < 4) {                       // samples beyond the batch (last tile): whatever the workspace holds there must not reach an accumulator
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    if (s >= ns) { adjB[s] = u32x4{0u, 0u, 0u, 0u}; X[s][0] = X[s][1] = X[s][2] = 0.f; }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (nt < a.ntiles) req_XA(nt);
        uint32_t sbase[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) sbase[s] = (uint32_t)((a.sample_offset + s0 + s) * F) * (uint32_t)N;
        const bool any_row = true;
        (void)any_row;

        if constexpr (KIND == PH_F) {
            // ===== forward statistics phases ==============================================================================================
            if constexpr (WITH_PREV) {
                layer_full(X, adjB, kp, dkey[LY - 1], sbase, nullptr, nullptr, &pend_q, &pend_m);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int r = 0; r < 3; ++r) pend_v[s][r] = X[s][r];
            }
            f32x4 T[4], Hp[4], z[4];
            float H[4][3];
            Op2 xo[4];
            Pk pk[4];
            Shifted ps[4];
            stage_T(X, adjB, T, xo);
            stage_Hp(T, kc.th, Hp);
            leaky_of(Hp, H);
            if constexpr (BLK == 0) {
                stage_conv(H, 0.f, kc.w[0], sh_rd1, sh_rd1_lo, z, pk, ps);          // raw z1
            } else {
                stage_conv(H, 1.0f, kc.w[0], sh_rd1, sh_rd1_lo, z, pk, ps);         // x-hat of BatchNorm 2l
                float V[4][3];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    keep_until_here(z[s][3]);
#pragma unroll
                    for (int r = 0; r < 3; ++r) V[s][r] = relu2(fmaf(2.f, H[s][r], relu2(fmaf(kc.gam[0][r], z[s][r], kc.bet[0][r]))));
                }
                stage_conv(V, 0.f, kc.w[1], sh_rd2, sh_rd2_lo, z, pk, ps);          // raw z2
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                keep_until_here(z[s][3]);
                if (s < ns) {
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const float zv = z[s][r] * colm;
                        s_a[r] += zv;
                        s_b[r] = fmaf(zv, zv, s_b[r]);
                    }
                }
            }
            if constexpr (WITH_PREV) { pend = true; pend_tile = tile; pend_ns = ns; }
            return;
        }

        if constexpr (KIND == PH_TOP) {
            // ===== last layer, head, loss, head backward ===================================================================================
            float xh1[4][3], y2[4][3];
            // (this row's label, requested before the last layer's forward: read where it is used -- `if (rowok) ... a.y[s0 + g]`, a load
            // under a condition -- it was a memory round trip per tile in the middle of the kernel)
            const float ylab = a.y[s0 + (g < ns ? g : ns - 1)];
            layer_full(X, adjB, kc, dkey[LY], sbase, &xh1, &y2, nullptr, &pend_m);
            // max over the ten channels with its arg-max: per lane over its (up to three) channels, then across the four lane groups
            float pm[4], pa[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float v0 = X[s][0], v1 = g == 3 ? X[s][0] : X[s][1], v2 = g == 3 ? X[s][0] : X[s][2];
                float m = v0;
                int ar = 0;
                if (v1 > m) { m = v1; ar = 1; }
                if (v2 > m) { m = v2; ar = 2; }
                pm[s] = fmaf(v0 + v1 + v2, 0.f, m);                      // NaN / Inf stay
                pa[s] = __builtin_bit_cast(float, 3 * g + ar);            // channel of slot 4 g + ar
            }
            transpose_rows4(pm[0], pm[1], pm[2], pm[3]);                  // out: register = lane group, row = sample
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
            const bool rowok = g < ns, valid = rowok && col_ok;
            pooled = valid ? pooled : 0.f;
            float y1 = fc1b;
            fmac1_rowbcast<0>(y1, pooled, fc1w[0]);   fmac1_rowbcast<1>(y1, pooled, fc1w[1]);   fmac1_rowbcast<2>(y1, pooled, fc1w[2]);
            fmac1_rowbcast<3>(y1, pooled, fc1w[3]);   fmac1_rowbcast<4>(y1, pooled, fc1w[4]);   fmac1_rowbcast<5>(y1, pooled, fc1w[5]);
            fmac1_rowbcast<6>(y1, pooled, fc1w[6]);   fmac1_rowbcast<7>(y1, pooled, fc1w[7]);   fmac1_rowbcast<8>(y1, pooled, fc1w[8]);
            fmac1_rowbcast<9>(y1, pooled, fc1w[9]);   fmac1_rowbcast<10>(y1, pooled, fc1w[10]); fmac1_rowbcast<11>(y1, pooled, fc1w[11]);
            fmac1_rowbcast<12>(y1, pooled, fc1w[12]); fmac1_rowbcast<13>(y1, pooled, fc1w[13]); fmac1_rowbcast<14>(y1, pooled, fc1w[14]);
            fmac1_rowbcast<15>(y1, pooled, fc1w[15]);
            y1 = relu(y1);
            const float pred = Row<16>::allsum(y1 * fc2w) + fc2b;
            float dpred = 0.f;                                             // x S
            if (rowok) {
                const float diff = pred - ylab;
                dpred = 2.f * diff * inv_gb * a.gscale;
                if (col == 0) acc_loss = fmaf(diff, diff, acc_loss);
            }
            pend = true; pend_tile = tile; pend_ns = ns; pend_pred = pred;
            if (!a.do_backward) return;
            const float dy1 = (y1 > 0.f) ? dpred * fc2w : 0.f;            // d(fc1 pre-activation), lane j
            float dpool = 0.f;
            fmac1_rowbcast<0>(dpool, dy1, fc1wT[0]);   fmac1_rowbcast<1>(dpool, dy1, fc1wT[1]);   fmac1_rowbcast<2>(dpool, dy1, fc1wT[2]);
            fmac1_rowbcast<3>(dpool, dy1, fc1wT[3]);   fmac1_rowbcast<4>(dpool, dy1, fc1wT[4]);   fmac1_rowbcast<5>(dpool, dy1, fc1wT[5]);
            fmac1_rowbcast<6>(dpool, dy1, fc1wT[6]);   fmac1_rowbcast<7>(dpool, dy1, fc1wT[7]);   fmac1_rowbcast<8>(dpool, dy1, fc1wT[8]);
            fmac1_rowbcast<9>(dpool, dy1, fc1wT[9]);   fmac1_rowbcast<10>(dpool, dy1, fc1wT[10]); fmac1_rowbcast<11>(dpool, dy1, fc1wT[11]);
            fmac1_rowbcast<12>(dpool, dy1, fc1wT[12]); fmac1_rowbcast<13>(dpool, dy1, fc1wT[13]); fmac1_rowbcast<14>(dpool, dy1, fc1wT[14]);
            fmac1_rowbcast<15>(dpool, dy1, fc1wT[15]);
            dpool = valid ? dpool : 0.f;
            acc_w2 = fmaf(dpred, y1, acc_w2);
            acc_b2 += (col == 0) ? dpred : 0.f;
            acc_b += dy1;
            acc_th = __builtin_amdgcn_mfma_f32_16x16x4f32(dy1, pooled, acc_th, 0, 0, 0);      // d fc1.w[j][t]: k = the four samples
            pend_top0 = dpool;
            pend_top1 = __builtin_bit_cast(float, arg);
            // the sums of the last BatchNorm want d X_L in the D layout: through the scratch words
            float* scr = smem + off_scr;
            scr[lane] = dpool;
            scr[64 + lane] = __builtin_bit_cast(float, arg);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float dv = scr[16 * s + col];
                const int da = __builtin_bit_cast(int, scr[64 + 16 * s + col]);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    float gq = (chan[r] == da) ? dv : 0.f;
                    if (use_drop) gq = (pend_m >> (3 * s + r)) & 1u ? gq * a.drop_scale : 0.f;       // the mask layer_full hashed
                    const float dy = y2[s][r] > 0.f ? gq : 0.f;
                    s_a[r] += dy;
                    s_b[r] = fmaf(dy, xh1[s][r], s_b[r]);
                }
            }
            __builtin_amdgcn_wave_barrier();
            return;
        }

        if constexpr (KIND == PH_G) {
            // ===== backward phases =========================================================================================================
            // the gradient entering the layer: d X_{l+1}, D layout, x S
            float gin[4][3];
            if constexpr (GRAD_TOP) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float dv = smem[(col_ok ? off_DX + col : off_zero) + s * N];
                    const int da = __builtin_bit_cast(int, smem[(col_ok ? off_DX + pitch + col : off_zero) + s * N]);
#pragma unroll
                    for (int r = 0; r < 3; ++r) gin[s][r] = (chan[r] == da && s < ns) ? dv : 0.f;
                }
            } else if constexpr (GRAD_IN) {
                ld_tile(off_DX, gin);
                if (ns < 4) {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        if (s >= ns) gin[s][0] = gin[s][1] = gin[s][2] = 0.f;
                }
            }
            float SB[4][3];
            if constexpr (NEED_SB) {
                ld_tile(off_SB, SB);
                if (ns < 4) {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        if (s >= ns) SB[s][0] = SB[s][1] = SB[s][2] = 0.f;
                }
            }
            float Q[4][3];
            if constexpr (BWD_PREV) {
                ld_tile(off_XP, Q);
                if (ns < 4) {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        if (s >= ns) Q[s][0] = Q[s][1] = Q[s][2] = INFINITY;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (nt < a.ntiles) { req_SB(nt); req_DX(nt); req_XP(nt); }

            uint32_t mask_cur = 0u;
            if constexpr (MASK_IN) {
                mask_cur = mask_next;
                if (use_drop && nt < a.ntiles) mask_next = a.mrec[LY][nt * 64 + lane];
            }
            // ---- two samples at a time (the working set of four does not fit the register file) ------------------------------------------------
            auto g_half = [&](auto HS_) {
                constexpr int HS = decltype(HS_)::value;
                constexpr int NS = 2;
                float Xh[NS][3];
                u32x4 ad[NS];
#pragma unroll
                for (int e = 0; e < NS; ++e) {
                    ad[e] = adjB[HS + e];
#pragma unroll
                    for (int r = 0; r < 3; ++r) Xh[e][r] = X[HS + e][r];
                }
                // layer LY forward again, as far as this phase needs it
                f32x4 T[NS], Hp[NS], z[NS], AXd[NS];
                float H[NS][3];
                Op2 xo[NS];
                stage_T(Xh, ad, T, xo);
                if constexpr (BLK == 0) {
#pragma unroll
                    for (int e = 0; e < NS; ++e) {
                        AXd[e] = mfma16z(ad[e], xo[e].h);                                    // (A X) in the D layout: rows c, columns k
                        AXd[e] = mfma16(ad[e], xo[e].l, AXd[e]);
                    }
                }
                stage_Hp(T, kc.th, Hp);
                leaky_of(Hp, H);
                Pk hk[NS];
                Shifted hks[NS];
                stage_conv(H, 1.0f, kc.w[0], sh_rd1, sh_rd1_lo, z, hk, hks);                  // x-hat of BatchNorm 2l
                float xh0[NS][3], y1[NS][3];
#pragma unroll
                for (int e = 0; e < NS; ++e) {
                    keep_until_here(z[e][3]);
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        xh0[e][r] = z[e][r];
                        y1[e][r] = fmaf(kc.gam[0][r], z[e][r], kc.bet[0][r]);
                    }
                }
                // the operand the weight gradient transposes: conv_block1: H; conv_block2: V = 4 o0
                Pk dk[NS];                              // packed d z (BLK 1) -- later the packed d Hp (BLK 0)
                float dz[NS][3], gsum[NS][3], V[NS][3];
                if constexpr (BLK == 1) {
                    // ---- G_{2l+1}: BatchNorm 2l+1 backward, conv_block2 gradient, d(x0 + H) ---------------------------------------------------
#pragma unroll
                    for (int e = 0; e < NS; ++e)
#pragma unroll
                        for (int r = 0; r < 3; ++r) V[e][r] = relu2(fmaf(2.f, H[e][r], relu2(y1[e][r])));
                    stage_conv(V, 1.0f, kc.w[1], sh_rd2, sh_rd2_lo, z, hk, hks);              // x-hat of BatchNorm 2l+1; hk, hks <- V
#pragma unroll
                    for (int e = 0; e < NS; ++e) {
                        keep_until_here(z[e][3]);
#pragma unroll
                        for (int r = 0; r < 3; ++r) {
                            const float xh = z[e][r];
                            const float y2 = fmaf(kc.gam[1][r], xh, kc.bet[1][r]);
                            float gq = gin[HS + e][r];
                            if (use_drop) gq = (mask_cur >> (3 * (HS + e) + r)) & 1u ? gq * a.drop_scale : 0.f;   // hashed by F_{2l+2} / TOP
                            const bool x1pos = y2 > 0.f;
                            gsum[e][r] = (x1pos || V[e][r] > 0.f) ? gq : 0.f;                // d(x1 + o0): o1 = relu(x1 + o0) > 0
                            const float dy = x1pos ? gq : 0.f;
                            const float v = bA[r] * (fmaf(-xh, bk2[r], dy) - bk1[r]);
                            dz[e][r] = (HS + e < ns) ? v * colm : 0.f;
                        }
                    }
                } else {
                    // ---- G_{2l}: BatchNorm 2l backward, conv_block1 + theta gradients, d X_l -------------------------------------------------
#pragma unroll
                    for (int e = 0; e < NS; ++e)
#pragma unroll
                        for (int r = 0; r < 3; ++r) {
                            const float dy = y1[e][r] > 0.f ? SB[HS + e][r] : 0.f;
                            const float v = bA[r] * (fmaf(-xh0[e][r], bk2[r], dy) - bk1[r]);
                            dz[e][r] = (HS + e < ns) ? v * colm : 0.f;
                        }
                }
                // d(input of the convolution) = W^T-conv(d z); weight gradient from the transposed tiles: acc += dzT x [hT | hsT]
                // (contraction over the columns t = 4 g + r of the transposed tiles)
                f32x4 dI[NS];
                {
                    u32x4 bh[NS], bl[NS];
                    f32x4 dzT[NS], hT[NS];
#pragma unroll
                    for (int e = 0; e < NS; ++e) {
                        const Pk p = pack3(dz[e][0], dz[e][1], dz[e][2], 0.f);
                        const Shifted nx = shift_read(sh_tile, sh_bk, sh_bk_lo, lane, p);
                        bh[e] = cat(p.hi, nx.hi);
                        bl[e] = cat(p.lo, nx.lo);
                        dzT[e] = mfma16z(cat(p.hi, p.lo), ident);
                        hT[e] = mfma16z(cat(hk[e].hi, hk[e].lo), ident);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int e = 0; e < NS; ++e) {
                        dI[e] = mfma16z(wT.hi, bh[e]);
                        dI[e] = mfma16(wT.hi, bl[e], dI[e]);
                        dI[e] = mfma16(wT.lo, bh[e], dI[e]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    Pk q[NS], u[NS], us[NS];
#pragma unroll
                    for (int e = 0; e < NS; ++e) {
                        q[e] = pack3(dzT[e][0], dzT[e][1], dzT[e][2], dzT[e][3]);
                        u[e] = pack3(hT[e][0], hT[e][1], hT[e][2], hT[e][3]);
                        // The operand of the tap at t - d is the SAME transposed tile shifted by d columns.  In this arrangement lane
                        // (g, slot) holds columns 4 g .. 4 g + 3 as two packed pairs, so the shift is a move between registers plus the
                        // last pair of the row above (lane - 16; nothing above row 0 = the causal zeros): no third transposing MFMA, no
                        // third split.  d = 2 (conv_block2): pairs move whole; d = 1 (conv_block1): halves recombine (two byte permutes).
                        const unsigned uh = __builtin_amdgcn_ds_bpermute((lane - 16) << 2, (int)u[e].hi.y), ul = __builtin_amdgcn_ds_bpermute((lane - 16) << 2, (int)u[e].lo.y);
                        const unsigned ah = g == 0 ? 0u : uh, al = g == 0 ? 0u : ul;
                        if constexpr (BLK == 1) {
                            us[e] = Pk{u32x2{ah, u[e].hi.x}, u32x2{al, u[e].lo.x}};
                        } else {
                            us[e] = Pk{u32x2{__builtin_amdgcn_perm(u[e].hi.x, ah, 0x05040302u), __builtin_amdgcn_perm(u[e].hi.y, u[e].hi.x, 0x05040302u)},
                                       u32x2{__builtin_amdgcn_perm(u[e].lo.x, al, 0x05040302u), __builtin_amdgcn_perm(u[e].lo.y, u[e].lo.x, 0x05040302u)}};
                        }
                        acc_w0 = mfma16(cat(q[e].hi, q[e].lo), cat(u[e].hi, u[e].hi), acc_w0);
                        acc_w1 = mfma16(cat(q[e].hi, q[e].lo), cat(us[e].hi, us[e].hi), acc_w1);
                    }
                    acc_w0 = mfma16(cat(q[0].hi, q[1].hi), cat(u[0].lo, u[1].lo), acc_w0);
                    acc_w1 = mfma16(cat(q[0].hi, q[1].hi), cat(us[0].lo, us[1].lo), acc_w1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (BLK == 1) {
                    // d(x0 + H) and the sums of BatchNorm 2l
#pragma unroll
                    for (int e = 0; e < NS; ++e) {
                        keep_until_here(dI[e][3]);
#pragma unroll
                        for (int r = 0; r < 3; ++r) {
                            float gq = dI[e][r] + gsum[e][r];
                            gq = (V[e][r] > 0.f && HS + e < ns) ? gq * colm : 0.f;
                            pend_v[HS + e][r] = gq;
                            const float dy = y1[e][r] > 0.f ? gq : 0.f;
                            s_a[r] += dy;
                            s_b[r] = fmaf(dy, xh0[e][r], s_b[r]);
                        }
                    }
                } else {
                    // d Hp = leaky'(Hp) (d H + d(x0 + H)); theta gradient: contraction over the channel slots
#pragma unroll
                    for (int e = 0; e < NS; ++e) {
                        keep_until_here(dI[e][3]);
                        float dHp[3];
#pragma unroll
                        for (int r = 0; r < 3; ++r) {
                            const float gq = dI[e][r] + SB[HS + e][r];
                            const float v = Hp[e][r] > 0.f ? gq : gq * LEAKY;
                            dHp[r] = (HS + e < ns) ? v * colm : 0.f;
                            acc_b += dHp[r];
                        }
                        dk[e] = pack3(dHp[0], dHp[1], dHp[2], 0.f);
                    }
                    {
                        Pk ax[NS];
#pragma unroll
                        for (int e = 0; e < NS; ++e) {
                            keep_until_here(AXd[e][3]);
                            ax[e] = pack3(AXd[e][0], AXd[e][1], AXd[e][2], 0.f);
                            acc_th = mfma16(cat(dk[e].hi, dk[e].lo), cat(ax[e].hi, ax[e].hi), acc_th);
                        }
                        acc_th = mfma16(cat(dk[0].hi, dk[1].hi), cat(ax[0].lo, ax[1].lo), acc_th);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (LY >= 1) {
                        // d X_l = A (d Hp theta) + d X_{l+1}: U = (A d Hp)^T, then U x theta
                        f32x4 U[NS], dXl[NS];
#pragma unroll
                        for (int e = 0; e < NS; ++e) {
                            U[e] = mfma16z(cat(dk[e].hi, dk[e].hi), ad[e]);
                            U[e] = mfma16(cat(dk[e].lo, dk[e].lo), ad[e], U[e]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int e = 0; e < NS; ++e) {
                            const Split2 p01 = split2(U[e][0], U[e][1]), p23 = split2(U[e][2], U[e][3]);
                            const u32x4 ua = {p01.hi, p23.hi, p01.lo, p23.lo};
                            dXl[e] = mfma16z(ua, thN.hi);
                            dXl[e] = mfma16(ua, thN.lo, dXl[e]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        // the sums of BatchNorm 2l-1: F_{2l} left its x-hat where the gradient passes (ReLU gate, dropout), +inf elsewhere
#pragma unroll
                        for (int e = 0; e < NS; ++e) {
                            keep_until_here(dXl[e][3]);
#pragma unroll
                            for (int r = 0; r < 3; ++r) {
                                const float dX = (HS + e < ns) ? (dXl[e][r] + gin[HS + e][r]) * colm : 0.f;
                                pend_v[HS + e][r] = dX;
                                const bool open = Q[HS + e][r] < INFINITY;
                                const float dy = open ? dX * a.drop_scale : 0.f;
                                s_a[r] += dy;
                                s_b[r] = fmaf(dy, open ? Q[HS + e][r] : 0.f, s_b[r]);
                            }
                        }
                    }
                }
            };
            g_half(std::integral_constant<int, 0>{});
            __builtin_amdgcn_sched_barrier(0);
            g_half(std::integral_constant<int, 2>{});
            if constexpr (BLK == 1 || LY >= 1) { pend = true; pend_tile = tile; pend_ns = ns; }
        }
        };
        if (ns_tile == 4) tile_body(std::true_type{});
        else tile_body(std::false_type{});
    }

    MXP_MARKI(6);                                                                  // tile loop done
    // ---- the last tile's outputs ---------------------------------------------------------------------------------------------------------
    if (pend) {
        if constexpr (WITH_PREV) {
            st_tile(a.xrec[LY] + pend_tile * XF, pend_v, pend_ns);
            st_tile(a.qrec[LY] + pend_tile * XF, pend_q, pend_ns);
            if (use_drop) a.mrec[LY - 1][pend_tile * 64 + lane] = pend_m;
        }
        if constexpr (KIND == PH_G && BLK == 1) st_tile(a.sb + pend_tile * XF, pend_v, pend_ns);
        if constexpr (BWD_PREV) st_tile(a.dx + pend_tile * XF, pend_v, pend_ns);
        if constexpr (KIND == PH_TOP) {
            if (g < pend_ns && col_ok && a.do_backward) {
                float* p = a.dtop + pend_tile * (8 * N) + g * N + col;
                p[0] = pend_top0;
                p[pitch] = pend_top1;
            }
            if (g < pend_ns && col == 0) a.pred[pend_tile * 4 + g] = pend_pred;
            if (use_drop && a.do_backward) a.mrec[LY][pend_tile * 64 + lane] = pend_m;
        }
    }

    MXP_MARKI(7);                                                                  // last tile's stores issued
    // ---- epilogue: the four wavefronts meet here.  BatchNorm pair: fp64 from the 16-lane reduction on, the wavefronts' sums combined in
    // a fixed order, one atomic per channel and workgroup ---------------------------------------------------------------------------------
    StepScratch* const sc = step_scratch(a.cells, L);
    bool bad = false;
    const bool has_pair = (KIND == PH_F) || (KIND == PH_TOP && a.do_backward) || (KIND == PH_G && IDX > 0);
    constexpr int PBW = 2 * F + 2;                       // doubles per wavefront in pairbuf: the pair, the loss
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
    MXP_MARKI(8);                                                                  // pair sums in LDS
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

    MXP_MARKI(9);                                                                  // atomics issued
    // ---- the workgroup's row of partial gradients: the wavefronts add their accumulators into an LDS image of the phase's contiguous
    // parameter range in a fixed order, then the image leaves in coalesced stores -------------------------------------------------------------
    const float us = a.inv_gscale;
    const int NN = N * N;
    int rbase = 0, rlen = 0;
    // (two images -- the second over the wavefronts' regions, dead by now: wavefronts 0 | 1 store, then 2 | 3 add: two rounds, fixed order)
    float* const red2 = smem_all + SH_BNC + MXT_RED_FLOATS + 2 * MXT_WAVES * (2 * F + 2);
    for (int w = 0; w < 2; ++w) {
        if ((wave >> 1) == w) {
            float* const img = (wave & 1) ? red2 : red;
            auto put = [&](int idx, float v) { img[idx] = (w == 0) ? v : img[idx] + v; };
            if constexpr (KIND == PH_TOP) {
                // [fc1.w | fc1.b | fc2.w | fc2.b]; fc1.w in the fp32 MFMA's D layout: row j = 4 g + r, column k = col
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = 4 * g + r;
                    if (j < N && col_ok) put(j * N + col, acc_th[r]);
                }
                float vb = acc_b, vw = acc_w2, v2 = acc_b2;
                vb += __shfl_xor(vb, 16, 64); vb += __shfl_xor(vb, 32, 64);
                vw += __shfl_xor(vw, 16, 64); vw += __shfl_xor(vw, 32, 64);
                v2 += __shfl_xor(v2, 16, 64); v2 += __shfl_xor(v2, 32, 64);
                if (g == 0 && col_ok) { put(NN + col, vb); put(NN + N + col, vw); }
                if (lane == 0) put(NN + 2 * N, v2);
            }
            if constexpr (KIND == PH_G) {
                const int ci = slot_chan(col);
                const int cbase = BLK == 0 ? NN + N : 0;                  // G_{2l}: [theta.w | theta.b | conv_block1.w]; G_{2l+1}: [conv_block2.w]
                const float ws = BLK == 1 ? 0.25f : 1.f;                  // conv_block2's data operand was V = 4 o0
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
                    for (int r = 0; r < 4; ++r) {
                        const int j = 4 * g + r;
                        if (j < N && col_ok) put(j * N + col, acc_th[r]);
                    }
                    float vb = acc_b;
                    vb += __shfl_xor(vb, 16, 64); vb += __shfl_xor(vb, 32, 64);
                    if (g == 0 && col_ok) put(NN + col, vb);
                }
            }
        }
        __syncthreads();
    }
    MXP_MARKI(10);                                                                 // gradient image complete
    if constexpr (KIND == PH_TOP) { rbase = off_fc1_w(N, L); rlen = NN + 2 * N + 1; }
    else if constexpr (BLK == 0) { rbase = LY * LS + off_theta_w(N); rlen = NN + N + CONVW; }
    else { rbase = LY * LS + off_conv_w(N, 1); rlen = CONVW; }
    float* row = a.gpart + (size_t)blockIdx.x * a.pcount + rbase;
    for (int i = threadIdx.x; i < rlen; i += 64 * MXT_WAVES) {
        const float v = (red[i] + red2[i]) * us;
        row[i] = v;
        bad |= !finite_f(v);
    }
    if (__any(bad) && lane == 0) atomicOr(&sc->pad[0], 1u);
}

template <int L, int KIND, int IDX, int NFIX>
__global__ __launch_bounds__(64 * MXT_WAVES, MX_WAVES_PER_SIMD) void stgcn_train_mx_kernel(MxTrainK a) {
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    mxt_phase_body<L, KIND, IDX, NFIX, false>(a, smem_all, 0u);
}

// =====================================================================================================================
// Small batches: F_1 .. G_0 as ONE launch (RULGNN_STEP_MX_PERSIST; an experiment kept as an explicit, tested option -- it does NOT pay:
// 85 us per step at batch 100 against 77 us for the ten launches, profiles/r06_notes.md section 4).  At the reference protocol's batch
// (100: configs/hparams.py:16-27) a phase is one tile per wavefront on seven workgroups and the ten-launch chain is ten dispatches + ten
// prologues at their latency floor.  Here every workgroup walks the phases itself; between two phases stands an arrival counter instead
// of a kernel boundary:
//   * arrive: the workgroup's cell atomics (agent scope) and its own record stores are drained (s_waitcnt vmcnt(0)), the wavefronts meet,
//     one agent-scope atomic add;
//   * wait: in the NEXT phase's prologue, right in front of the cell reads (mxt_phase_body<PERSIST>): the first tile's LDS-DMA requests
//     and the parameter loads / operand conversions that need no BatchNorm are in flight by then.
// No fence on either side: the only data that crosses a workgroup inside the launch are the fp64 cells -- atomics on the producing and
// the consuming side (MI355X_MICROARCH.md, inter-workgroup visibility: "8-B agent atomics both sides"); tiles stay on the wavefront that
// owns them (same grid, same tile -> wavefront map in every phase), the gradient rows and the status word are read by the finalize
// LAUNCH.  F_0 stays its own launch (other translation unit; it also carries the head-of-step scalars).
// The grid must be co-resident: the host launches it only when every workgroup has a CU of its own.
// =====================================================================================================================
// (inlined: as calls -- 80 callee-saved registers through scratch per phase -- the step measured 94 us instead of 85)
template <int L, int KIND, int IDX, int NFIX>
__device__ __forceinline__ void mxt_persist_phase(const MxTrainK& a, float* smem_all, unsigned& arrived_phases) {
    mxt_phase_body<L, KIND, IDX, NFIX, true>(a, smem_all, arrived_phases * gridDim.x);
    constexpr bool PERSIST = true;
    MXP_MARKI(11);                                                                 // body returned
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    MXP_MARKI(12);                                                                 // drained
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(step_barrier(a.cells, L), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ++arrived_phases;
}
template <int L, int NFIX, int I>
struct MxtPersistChain {
    static __device__ __forceinline__ void forward(const MxTrainK& a, float* smem_all, unsigned& n) {
        if constexpr (I > 1) MxtPersistChain<L, NFIX, I - 1>::forward(a, smem_all, n);
        mxt_persist_phase<L, PH_F, I, NFIX>(a, smem_all, n);
    }
    static __device__ __forceinline__ void backward(const MxTrainK& a, float* smem_all, unsigned& n) {
        mxt_persist_phase<L, PH_G, I, NFIX>(a, smem_all, n);
        if constexpr (I > 0) MxtPersistChain<L, NFIX, I - 1>::backward(a, smem_all, n);
    }
};
template <int L, int NFIX>
__global__ __launch_bounds__(64 * MXT_WAVES, MX_WAVES_PER_SIMD) void stgcn_train_mx_persist_kernel(MxTrainK a) {
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    unsigned n = 0;                                   // phases this workgroup has arrived at; F_1 reads what F_0's LAUNCH left: target 0
    MxtPersistChain<L, NFIX, 2 * L - 1>::forward(a, smem_all, n);
    mxt_persist_phase<L, PH_TOP, 0, NFIX>(a, smem_all, n);
    MxtPersistChain<L, NFIX, 2 * L - 1>::backward(a, smem_all, n);
#ifdef MXP_TRACE
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (int i = 0; i < mxp_n; ++i) {
            if (mxp_id[i] >= 100) printf("\nphase %u:", mxp_id[i]);
            else printf(" [%u] +%u", mxp_id[i], mxp_ts[i] - mxp_ts[i - 1]);
        }
        printf("\n");
        mxp_n = 0;
    }
#endif
}

// =====================================================================================================================
// host side
// =====================================================================================================================
static size_t mxt_lds_bytes(int L, int kind, int idx, int N) {
    const int blk = kind == PH_TOP ? 1 : idx % 2, ly = kind == PH_TOP ? L - 1 : idx / 2;
    const bool need_sb = kind == PH_G && blk == 0, grad_in = kind == PH_G && (blk == 1 || ly >= 1), bwd_prev = kind == PH_G && blk == 0 && ly >= 1;
    const int XF = 40 * N;
    const size_t wave = (size_t)MXT_ZERO_FLOATS + MXT_SCRATCH_FLOATS + MXT_SHIFT_FLOATS + XF + 220 + (need_sb ? XF : 0) + (grad_in ? XF : 0) +
                        (bwd_prev ? XF : 0);
    const size_t shared = (size_t)((2 * L * MXT_BNC * F + 3) & ~3) + MXT_RED_FLOATS + 2 * MXT_WAVES * (2 * F + 2);
    return (shared + MXT_WAVES * wave) * sizeof(float);
}

template <int L, int KIND, int IDX, int NFIX>
static int mxt_launch(const MxTrainK& k, hipStream_t stream, int max_grid, int* grid_out) {
    auto kern = &stgcn_train_mx_kernel<L, KIND, IDX, NFIX>;
    const size_t lds = mxt_lds_bytes(L, KIND, IDX, k.N);
    if (lds > 80 * 1024) return RULGNN_EUNSUPPORTED;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return RULGNN_EHIP;
    int dev = 0, cus = 256, per_cu = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 64 * MXT_WAVES, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    if (per_cu > MX_WAVES_PER_SIMD) per_cu = MX_WAVES_PER_SIMD;           // a workgroup is one wavefront per SIMD
    if (const char* e = getenv("RULGNN_MXT_BLOCKS_PER_CU")) { const int v = atoi(e); if (v > 0) per_cu = v; }   // tuning aid
    int64_t grid = (int64_t)cus * per_cu;
    const int64_t want = (k.ntiles + MXT_WAVES - 1) / MXT_WAVES;
    if (grid > want) grid = want;
    if (grid > max_grid) grid = max_grid;
    if (grid_out) *grid_out = (int)grid;
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * MXT_WAVES), lds, stream, k);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

template <int L, int KIND, int IDX>
static int mxt_launch_n(const MxTrainK& k, hipStream_t stream, int max_grid, int* grid_out) {
    if (k.N == 14) return mxt_launch<L, KIND, IDX, 14>(k, stream, max_grid, grid_out);
    return mxt_launch<L, KIND, IDX, 0>(k, stream, max_grid, grid_out);
}

template <int L, int I>
struct MxtPhase {
    static int run(int kind, int idx, const MxTrainK& k, hipStream_t st, int mg, int* go) {
        if (idx == I) {
            if (kind == PH_F) {
                if constexpr (I >= 1) return mxt_launch_n<L, PH_F, I>(k, st, mg, go);
                else return RULGNN_EINVAL;
            }
            if (kind == PH_G) return mxt_launch_n<L, PH_G, I>(k, st, mg, go);
        }
        if constexpr (I > 0) return MxtPhase<L, I - 1>::run(kind, idx, k, st, mg, go);
        return RULGNN_EINVAL;
    }
};

bool stgcn_train_mx_shape_ok(const rulgnn_stgcn_shape* s, const float* x) {
    const int N = s->num_patch, P = s->patch_size, L = s->num_layers;
    if (N < 2 || N > 15 || L < 1 || L > MX_MAX_LAYERS || s->mpnn_k != 1) return false;
    if (((int64_t)N * P) % 4 != 0 || (reinterpret_cast<uintptr_t>(x) & 15) != 0) return false;     // F_0's 16-byte DMA pieces
    return true;
}

float stgcn_train_mx_grad_scale(int64_t global_batch) {
    int e = 0;
    while (((int64_t)1 << e) < global_batch && e < 40) ++e;
    return ldexpf(1.0f, e + 3);
}

static MxTrainK mxt_kernel_args(const MxTrainArgs& m) {
    MxTrainK k;
    k.prm = m.prm; k.y = m.y; k.pred = m.pred; k.cells = m.cells; k.gpart = m.gpart;
    for (int l = 0; l < MX_MAX_LAYERS; ++l) { k.xrec[l] = m.xrec[l]; k.qrec[l] = m.qrec[l]; k.mrec[l] = m.mrec[l]; }
    k.arec = m.arec; k.sb = m.sb; k.dx = m.dx; k.dtop = m.dtop;
    k.B = m.B; k.ntiles = (m.B + 3) / 4; k.global_batch = m.global_batch; k.sample_offset = m.sample_offset;
    k.N = m.N; k.pcount = m.pcount;
    k.dropout_p = m.dropout_p; k.drop_scale = m.drop_scale; k.drop_thr = m.drop_thr;
    k.gscale = stgcn_train_mx_grad_scale(m.global_batch);
    k.inv_gscale = 1.0f / k.gscale;
    k.do_backward = m.do_backward;
    return k;
}

// F_1 .. G_0 of a whole step as one launch (stgcn_train_mx_persist_kernel): two layers, a batch small enough that every workgroup of
// the phases' common grid (one tile per wavefront) has a CU of its own -- RULGNN_EUNSUPPORTED otherwise, and the caller runs the phases
// as launches.  *grid_out = the grid: every phase's partial gradient rows (finalize).
template <int NFIX>
static int mxt_persist_launch(const MxTrainK& k, hipStream_t stream, int64_t grid) {
    constexpr int L = 2;
    auto kern = &stgcn_train_mx_persist_kernel<L, NFIX>;
    size_t lds = 0;
    for (int i = 1; i < 2 * L; ++i) lds = std::max(lds, mxt_lds_bytes(L, PH_F, i, k.N));
    lds = std::max(lds, mxt_lds_bytes(L, PH_TOP, 0, k.N));
    for (int i = 0; i < 2 * L; ++i) lds = std::max(lds, mxt_lds_bytes(L, PH_G, i, k.N));
    if (lds > 80 * 1024) return RULGNN_EUNSUPPORTED;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return RULGNN_EHIP;
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * MXT_WAVES), lds, stream, k);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}
int stgcn_train_mx_persistent_grid(int64_t batch, int num_layers, int max_grid) {
    if (num_layers != 2 || batch <= 0) return 0;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const int64_t grid = ((batch + 3) / 4 + MXT_WAVES - 1) / MXT_WAVES;
    // one workgroup per CU: co-resident whatever else runs.  (Measured with a 512-workgroup grid at the headline batch: 0.44 ms against
    // 0.345 for the launches, 0.23 against 0.14 at 16 384 -- one register allocation for eight bodies spills in the tile loops.)
    return grid <= cus && grid <= max_grid ? (int)grid : 0;
}
int stgcn_train_mx_persistent(const MxTrainArgs& m, hipStream_t stream, int max_grid, int* grid_out) {
    const int grid = m.do_backward ? stgcn_train_mx_persistent_grid(m.B, m.L, max_grid) : 0;
    if (grid == 0) return RULGNN_EUNSUPPORTED;
    const MxTrainK k = mxt_kernel_args(m);
    if (grid_out) *grid_out = grid;
    return k.N == 14 ? mxt_persist_launch<14>(k, stream, grid) : mxt_persist_launch<0>(k, stream, grid);
}

int stgcn_train_mx_phase(const MxTrainArgs& m, int kind, int idx, hipStream_t stream, int max_grid, int* grid_out) {
    const MxTrainK k = mxt_kernel_args(m);
    if (m.B == 0) { if (grid_out) *grid_out = 0; return RULGNN_OK; }
    const int L = m.L;
    if (kind == PH_TOP) {
        switch (L) {
            case 1: return mxt_launch_n<1, PH_TOP, 0>(k, stream, max_grid, grid_out);
            case 2: return mxt_launch_n<2, PH_TOP, 0>(k, stream, max_grid, grid_out);
            case 3: return mxt_launch_n<3, PH_TOP, 0>(k, stream, max_grid, grid_out);
            default: return RULGNN_EUNSUPPORTED;
        }
    }
    switch (L) {
        case 1: return MxtPhase<1, 1>::run(kind, idx, k, stream, max_grid, grid_out);
        case 2: return MxtPhase<2, 3>::run(kind, idx, k, stream, max_grid, grid_out);
        case 3: return MxtPhase<3, 5>::run(kind, idx, k, stream, max_grid, grid_out);
        default: return RULGNN_EUNSUPPORTED;
    }
}

}  // namespace rulgnn
