// Train-mode ST_GCN forward + backward for gfx950 (row width 16: num_patch <= 16).
//
// Reference path replaced: ST_GCN.update up to optimizer.step() -- algorithms/algorithms.py:481-488
// (model(X) under model.train(), MSE, loss.backward()).
//
// Train-mode BatchNorm couples all samples of the batch: each of the 2L BatchNorm layers needs a
// batch-wide mean/var in the forward and two batch-wide sums in the backward before anything
// behind it can proceed.  The step is therefore a chain of 4L+1 phase kernels separated by
// grid-wide reductions (fp64 atomics into "cells"):
//
//     F_0 .. F_{2L-1}   forward up to the input of BatchNorm i, accumulate sum(z), sum(z^2)
//     TOP               full forward -> pred, loss; backward of fc2/fc1/max-pool down to dy of
//                       BatchNorm 2L-1, accumulate sum(dy), sum(dy*xhat)
//     G_{2L-1} .. G_0   BatchNorm-i backward -> dz_i, weight gradients of conv_i (and theta of the
//                       layer when i is even), data gradient down to dy_{i-1}, accumulate its sums
//
// Per sample the phases hand each other 640-byte [10][N] tensors through HBM: the input-only statistics X0 and Pearson
// adjacency (written by F_0), and per layer the tensors listed at SavedSlot.  The first versions recomputed activations from
// the cache in every phase; the kernels turned out VALU-issue-bound with HBM two thirds idle, so each phase now reads what an
// earlier phase already had in registers instead of redoing a theta projection and up to two convolutions.
//
// Matrix-core use: the contraction of the weight gradients runs over (sample, patch) or
// (sample, channel), i.e. over lanes/registers of the row mapping:
//   d theta[j][k] = sum_{s,c} dHpre[s,c,j] * AX[s,c,k]  -> v_mfma_f32_16x16x4_f32 fed DIRECTLY from
//       the row-mapped registers (lane>>4 = sample = k-slice, lane&15 = patch = matrix row/col);
//   d convW[co][ci,tap] = sum_{s,t} dz[co][s,t] * h[ci][s,t-d*tap'] -> same MFMA after an in-wave
//       LDS transpose ([channel][lane] tile, ds_read_b128 fragments).
// Accumulators stay in the MFMA accumulator registers across the whole persistent tile loop.
#include "stgcn_device.hpp"
#include "stgcn_host.hpp"
#include "stgcn_train_layout.hpp"
#include "stgcn_train_mx.hpp"

namespace rulgnn {


// Row widths of the training kernels: 16 (num_patch <= 16: four samples per wavefront, DPP + MFMA tricks
// that rely on 16-lane rows) and 64 (num_patch <= 64: one sample per wavefront, generic cross-lane
// primitives; correctness path for the reference-wired PHM2012 40x64 shape).
constexpr int TT_STRIDE = 68;               // LDS row stride of the [channel][lane] transpose tile
constexpr int TT_ROWS = 30;
constexpr int CONVT_FLOATS = F * 2 * F;     // one transposed conv weight block in LDS (multiple of 4 floats)
constexpr int BNC = 7;                      // per-BatchNorm constants: mean, istd, scale, shift, gamma*istd, k1, k2

enum PhaseKind { PH_F = 0, PH_TOP = 1, PH_G = 2 };
// Wavefronts per SIMD the phase kernels are compiled for (launch bounds; measured per phase at batch 65536: tighter bounds spill)
constexpr int PHASE_WAVES_F1 = 5, PHASE_WAVES_F2 = 4, PHASE_WAVES_TOP = 3, PHASE_WAVES_G1 = 3, PHASE_WAVES_G0 = 3;

// Activations carried from phase to phase through HBM, one [ntiles][F][64] lane-major tensor per slot (ONE base pointer in the
// kernel arguments: the phase kernels are SGPR-bound).  Every tensor is written once per step by the phase that first has it
// and read by the phases that would otherwise recompute it (theta projection + two convolutions per layer):
//   X(l)   input of layer l >= 1                                       F_{2l}   -> F_{2l} (next use), TOP, G_{2l}
//   H(l)   leaky(theta(A X_l))                                          F_{2l}   -> F_{2l+1}, G_{2l}
//   Z1(l)  conv_block1 output (BatchNorm 2l input)                      F_{2l}   -> F_{2l+1}, G_{2l+1}, G_{2l}
//   O0(l)  relu(relu(BN(z1)) + H): conv_block2 input                    F_{2l+1} -> F_{2l+2} / TOP, G_{2l+1}
//   Z2(l)  conv_block2 output (BatchNorm 2l+1 input)                    F_{2l+1} -> F_{2l+2} / TOP, G_{2l+1}, G_{2l+2} (x-hat and
//          the ReLU gate of BatchNorm 2l+1 for its backward sums: round 2 carried them in a tensor of their own)
template <int L>
struct SavedSlot {
    static constexpr int X(int l) { return l - 1; }
    static constexpr int H(int l) { return (L - 1) + l; }
    static constexpr int Z1(int l) { return (L - 1) + L + l; }
    static constexpr int O0(int l) { return (L - 1) + 2 * L + l; }
    static constexpr int Z2(int l) { return (L - 1) + 3 * L + l; }
};
static inline int saved_slots(int L) { return 5 * L - 1; }

struct TrainK {
    // workspace regions
    float* cacheX;        // [ntiles][F][64]
    float* cacheA;        // [ntiles][F][64]  lane-distributed adjacency rows
    double* cells;        // [CELL_REPLICAS][cell_stride(L)] reduction cells (see CellLayout), step scratch behind them
    float* gpart;         // [grid][param_count] per-block partial gradients
    float* saved;         // [5L-1][ntiles][F][64]  activations carried between phases, see SavedSlot
    float* rbuf;          // [ntiles][F][64]  d X_{l+1}: gradient entering layer l's backward (TOP / G_{2l+2} -> G_{2l+1}, G_{2l})
    float* sbuf;          // [ntiles][F][64]  d(x0 + H) of layer l (G_{2l+1} -> G_{2l})
    // outputs
    float* pred;
    // sizes
    int64_t B, ntiles, global_batch, sample_offset;
    int N, P, Ppad, vec4, stage_floats;
    uint32_t magicP;
    int do_backward;      // TOP: 0 = forward only
    int wave_area_floats;
    int write_pred;
    int has_dpred;        // 1: gy is d(loss)/d(pred); 0: gy is y (MSE); 2: neither (forward only)
    float dropout_p, drop_scale;
    uint32_t drop_thr;
    int pcount;
    int K;                // MPNN order (Model.py:74-90): theta matrices per layer; the layout offsets take it
};

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {      // total over 64 lanes, valid in every lane
    v = Row<16>::allsum(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

template <int RW, int D, typename WP>
__device__ __forceinline__ void causal_conv_T(const float (&dz)[F], WP w, int t, float (&dh)[F]) {
    // transpose of causal_conv: dh[ci][t] = sum_co w[co][ci][1] dz[co][t] + w[co][ci][0] dz[co][t + D]
    float dzs[F];
#pragma unroll
    for (int c = 0; c < F; ++c) dzs[c] = Row<RW>::template shl<D>(dz[c], t);
#pragma unroll
    for (int ci = 0; ci < F; ++ci) {
        float acc = 0.f;
#pragma unroll
        for (int co = 0; co < F; ++co) {
            acc = fmaf(w[(co * F + ci) * 2 + 1], dz[co], acc);
            acc = fmaf(w[(co * F + ci) * 2 + 0], dzs[co], acc);
        }
        dh[ci] = acc;
    }
}

template <int RW, int D>
__device__ __forceinline__ void causal_conv_T_lds(const float (&dz)[F], const float* wT, int t, float (&dh)[F]) {
    float dzs[F];
#pragma unroll
    for (int c = 0; c < F; ++c) dzs[c] = Row<RW>::template shl<D>(dz[c], t);
    conv_rows_lds(dzs, dz, wT, dh);
}

// In-wave LDS transpose + MFMA: acc0/acc1 += dz (10 x 64 lanes) . [h | hs]^T (64 lanes x 20).
// T: this wavefront's [TT_ROWS][TT_STRIDE] tile.  Row r of the tile holds one channel for all 64
// lanes; the A/B fragments of v_mfma_f32_16x16x4_f32 are ds_read_b128 of 4 consecutive lanes
// (the k index is permuted identically for A and B, which a contraction does not see).
__device__ __forceinline__ void conv_wgrad_mfma(float* T, const float (&dz)[F], const float (&h)[F], const float (&hs)[F],
                                                int lane, f32x4& acc0, f32x4& acc1) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < F; ++c) {
        T[c * TT_STRIDE + lane] = dz[c];
        T[(F + c) * TT_STRIDE + lane] = h[c];
        T[(2 * F + c) * TT_STRIDE + lane] = hs[c];
    }
    __builtin_amdgcn_wave_barrier();
    const int i = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
        const float4 a = *reinterpret_cast<const float4*>(&T[i * TT_STRIDE + 16 * kq + 4 * jp]);
        const float4 b0 = *reinterpret_cast<const float4*>(&T[(F + i) * TT_STRIDE + 16 * kq + 4 * jp]);
        const float4 b1 = *reinterpret_cast<const float4*>(&T[(i < 4 ? 2 * F + 6 + i : 3 * F - 1) * TT_STRIDE + 16 * kq + 4 * jp]);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b1.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b0.y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b0.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b1.z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b0.w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b1.w, acc1, 0, 0, 0);
    }
}

// Generic (any row width) weight gradient of a [N][N] matrix: D[j][k] += sum_{sample, c} P[c][sample, j] * Q[c][sample, k]
// by the same in-wave LDS transpose + v_mfma_f32_16x16x4_f32, tiled (RW/16)^2.  Used by the RW = 64 kernels
// for theta (NC = 10) and fc1 (NC = 1); the RW = 16 kernels feed the MFMA straight from registers instead.
template <int RW, int NC>
__device__ __forceinline__ void outer_grad_mfma(float* T, const float (&P)[NC], const float (&Q)[NC], int lane,
                                                f32x4 (&acc)[(RW / 16) * (RW / 16)]) {
    constexpr int NT = RW / 16, KS = (NC + 3) / 4, ROWS = KS * 4, SPW = 64 / RW;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < ROWS; ++c) {
        T[c * TT_STRIDE + lane] = c < NC ? P[c < NC ? c : 0] : 0.f;
        T[(ROWS + c) * TT_STRIDE + lane] = c < NC ? Q[c < NC ? c : 0] : 0.f;
    }
    __builtin_amdgcn_wave_barrier();
    const int i = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int s = 0; s < SPW; ++s) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            float av[NT], bv[NT];
#pragma unroll
            for (int m = 0; m < NT; ++m) {
                av[m] = T[(4 * ks + kq) * TT_STRIDE + s * RW + 16 * m + i];
                bv[m] = T[(ROWS + 4 * ks + kq) * TT_STRIDE + s * RW + 16 * m + i];
            }
#pragma unroll
            for (int m = 0; m < NT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m * NT + n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[n], acc[m * NT + n], 0, 0, 0);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// the phase kernel
// ------------------------------------------------------------------------------------------------
// IDX: BatchNorm index (0 .. 2L-1) for F and G kernels; unused for TOP.
// Wavefronts per SIMD the register allocation aims for (second __launch_bounds__ argument).  The phases that only stream
// saved activations through a convolution or the head want more wavefronts in flight; the gradient phases need the registers.
constexpr int phase_min_waves(int RW, int KIND, int IDX) {
    if (RW != 16) return 1;
    if (KIND == PH_F) return IDX % 2 == 1 ? PHASE_WAVES_F1 : (IDX == 0 ? 4 : PHASE_WAVES_F2);
    if (KIND == PH_TOP) return PHASE_WAVES_TOP;
    return IDX % 2 == 1 ? PHASE_WAVES_G1 : PHASE_WAVES_G0;
}

// NFIX: num_patch known at compile time (14 = C-MAPSS, the headline shape; 0 = read it from the arguments).  With a constant
// pitch the 10 element addresses of a saved tensor become immediate offsets of one base: ~60 64-bit address computations per
// tile and the scalar registers that carried them disappear.
// The body is a device function so that the same code runs as one kernel per phase (the chain, below) and as the stages of the
// single cooperative launch that small batches use (stgcn_train_coop_kernel).
// KORD: MPNN order of the phases that contain the theta projection (F_{2l}) or its backward (G_{2l}) -- compile-time there, because the
// weight-gradient accumulators of every order live in registers; every other phase is instantiated with KORD = 1 and takes the order
// from a.K for the parameter offsets only.
template <int RW, int L, int KIND, int IDX, int NFIX = 0, int PFIX = 0, int KORD = 1>
__device__ __forceinline__ void train_phase_body(const float* __restrict__ gx, const float* __restrict__ prm,
                                                 const float* __restrict__ gy,   // y or dpred (TOP only)
                                                 const TrainK& a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TRW = RW, TSPW = 64 / RW, TWS = wstride<RW>();
    using R16 = Row<RW>;                       // (name kept from the first, 16-lane-only version)
    constexpr int NTH = RW == 16 ? 1 : (RW / 16) * (RW / 16);     // theta-gradient MFMA tiles (generic path)
    constexpr int NBN = 2 * L;
    // BatchNorm layers whose forward statistics this kernel needs: F_i applies BN 0..i-1.
    constexpr int NFWD = KIND == PH_F ? IDX : NBN;
    static_assert(KORD == 1 || ((KIND == PH_F || KIND == PH_G) && IDX % 2 == 0), "only the theta phases are specialised on the order");
    static_assert(KORD == 1 || NFIX == 0, "the compile-time shapes are order 1");
    const int K = NFIX ? 1 : (KORD > 1 ? KORD : a.K);
    const int N = NFIX ? NFIX : a.N, LS = layer_stride(N, K);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int srow = lane / RW, t = lane % RW;
    // dropout keys, read ONCE from the step scratch (kept out of the tile loops: the compiler will not hoist a load through a
    // plain pointer)
    uint32_t dkey[L];
#pragma unroll
    for (int l = 0; l < L; ++l) dkey[l] = step_scratch(a.cells, L)->drop_key[l];

    // Which layer does this kernel work in, and where does its forward start?
    //   F_{2l}, l >= 1 : first finishes layer l-1 (its BatchNorm statistics are complete now) and stores X_l
    //                    and, per element, "x-hat of BatchNorm 2l-1 if the gradient passes both ReLUs, else +inf"
    //                    (psave) -- all that G_{2l} needs from layer l-1 to form that BatchNorm's backward sums
    constexpr int LY = KIND == PH_TOP ? L - 1 : IDX / 2;
    constexpr int BLK = KIND == PH_TOP ? 1 : IDX % 2;
    constexpr bool WITH_PREV = (KIND == PH_F) && (BLK == 0) && (LY >= 1);
    constexpr int LSTART = WITH_PREV ? LY - 1 : LY;
    using SV = SavedSlot<L>;
    // where the LDS weights pay (measured per phase, batch 65536): G_{2l} (86 -> 64 us, 46 -> 40 us: the scalar path spilled),
    // F_{2l+1} (29 -> 28 us); G_{2l+1} and F_{2l} are 1-2 us better on scalar-operand weights
    constexpr bool CONV_FROM_LDS = (KIND == PH_G && IDX % 2 == 0) || (KIND == PH_F && IDX % 2 == 1);
    // what this phase reads besides the saved activations: the layer input (F_{2l}: aggregation, G_{2l}: theta gradient,
    // TOP: last residual) and the adjacency (F_{2l}, G_{2l})
    constexpr bool NEED_A = KIND != PH_TOP && BLK == 0;
    constexpr bool NEED_X = NEED_A || KIND == PH_TOP;

    // ---- LDS carve --------------------------------------------------------------------------------
    // three zero-padded [RW][TWS] weight slots: theta of layer LY | theta of layer LY-1 (F_{2l}) or fc1 (TOP) |
    // the transposed matrix the backward needs (theta^T for G_{2l}, fc1^T for TOP)
    // Order KORD > 1 (F_{2l} / G_{2l} only): KORD slots in all -- theta_kk for F_{2l}, theta_kk^T for G_{2l}.
    constexpr int CS = cell_stride(L);
    constexpr int NCUR = KORD == 1 ? 1 : (KIND == PH_F ? KORD : 0), NAUX = KORD == 1 ? 1 : 0, NTR = KORD == 1 ? 1 : (KIND == PH_G ? KORD : 0);
    double* cellsum = reinterpret_cast<double*>(smem);     // [CS] the reduction cells, replicas summed
    float* w_cur = smem + 2 * CS;
    float* w_aux = w_cur + NCUR * TRW * TWS;
    float* w_tr = w_aux + NAUX * TRW * TWS;
    float* vecs = w_tr + NTR * TRW * TWS;                 // [L+2][RW]        theta bias (summed over the orders), fc1 bias, fc2 weight
    float* bnc = vecs + (L + 2) * TRW;                    // [NBN][BNC][F] (+pad to 4)
    constexpr int RED_TH = RW == 16 ? 4 : 4 * NTH;        // rows of one [N][N] weight-gradient accumulator in the block reduction
    constexpr int RED_K1 = 15 + (RW == 16 ? 0 : 4 * NTH);
    constexpr int RED_K = RED_K1 + (KORD - 1) * RED_TH;   // orders 2.. behind the rows of order 1
    float* red = bnc + ((NBN * BNC * F + 3) & ~3);        // [RED_K][64] block reduction of the gradient accumulators
    float* redp = red + RED_K * 64;                       // [4 waves][24] BatchNorm pair / loss partials
    float* convT = redp + WAVES_PER_BLOCK * 24;           // [F][2F] conv_block1 weights of layer LY as [ci][co][tap] (G_{2l} only)
    float* wave_area = convT + CONVT_FLOATS;              // per-wave: staging (F_0) or transpose tile (TOP / G)
    const int wave_area_floats = a.wave_area_floats;
    float* mywave = wave_area + wave * wave_area_floats;

    // ---- prologue: weights to LDS, BatchNorm constants from the reduction cells ---------------------
    if constexpr (KORD == 1) {
        const float* th_cur = prm + LY * LS + off_theta_w(N);
        const float* aux = KIND == PH_TOP ? prm + off_fc1_w(N, L, K) : prm + (LY >= 1 ? LY - 1 : 0) * LS + off_theta_w(N);
        const float* trs = KIND == PH_TOP ? prm + off_fc1_w(N, L, K) : th_cur;
        for (int i = threadIdx.x; i < TRW * TRW; i += BLOCK) {
            const int j = i / TRW, k = i % TRW;
            const bool in = j < N && k < N;
            w_cur[j * TWS + k] = in ? th_cur[j * N + k] : 0.f;
            w_aux[j * TWS + k] = in ? aux[j * N + k] : 0.f;
            w_tr[k * TWS + j] = in ? trs[j * N + k] : 0.f;
        }
    } else {
        for (int i = threadIdx.x; i < KORD * TRW * TRW; i += BLOCK) {
            const int kk = i / (TRW * TRW), j = (i / TRW) % TRW, k = i % TRW;
            const float v = (j < N && k < N) ? prm[LY * LS + off_theta_w(N, kk) + j * N + k] : 0.f;
            if constexpr (KIND == PH_F) w_cur[(kk * TRW + j) * TWS + k] = v;
            else w_tr[(kk * TRW + k) * TWS + j] = v;
        }
    }
    for (int i = threadIdx.x; i < (L + 2) * TRW; i += BLOCK) {
        const int m = i / TRW, j = i % TRW;
        float v = 0.f;
        if (j < N) {
            if (m < L) {
                for (int kk = 0; kk < K; ++kk) v += prm[m * LS + off_theta_b(N, kk) + j];       // one bias row: the sum over the orders
            } else {
                v = m == L ? prm[off_fc1_b(N, L, K) + j] : prm[off_fc2_w(N, L, K) + j];
            }
        }
        vecs[i] = v;
    }
    if constexpr (CONV_FROM_LDS) {                        // this phase's ONE convolution reads its weights from LDS
        const float* cw = prm + (IDX / 2) * LS + off_conv_w(N, IDX % 2, K);
        for (int i = threadIdx.x; i < F * F * 2; i += BLOCK) {
            const int tap = i & 1, ci = (i >> 1) % F, co = (i >> 1) / F;
            convT[KIND == PH_G ? ci * 2 * F + co * 2 + tap : i] = cw[i];      // backward: [ci][co][tap]; forward: as stored
        }
    }
    for (int i = threadIdx.x; i < CS; i += BLOCK) cellsum[i] = cell_sum(a.cells, L, i);
    __syncthreads();
    const double* cells_fwd = cellsum + cell_fwd(L);
    const double* cells_bwd = cellsum + cell_bwd(L);
    const double cnt = step_scratch(a.cells, L)->bn_count;   // values per channel behind the cells (shard, or global batch under SyncBN)
    for (int i = threadIdx.x; i < NBN * F; i += BLOCK) {
        const int b = i / F, c = i % F;
        const int l = b / 2, blk = b % 2;
        float* o = bnc + b * BNC * F;
        if (b < NFWD) {
            const double s1 = cells_fwd[(b * 2 + 0) * F + c], s2 = cells_fwd[(b * 2 + 1) * F + c];
            const double mean = s1 / cnt;
            double var = s2 / cnt - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            const double istd = 1.0 / sqrt(var + (double)BN_EPS);
            const double g = prm[l * LS + off_bn_g(N, blk, K) + c], be = prm[l * LS + off_bn_b(N, blk, K) + c];
            o[0 * F + c] = (float)mean;
            o[1 * F + c] = (float)istd;
            o[2 * F + c] = (float)(g * istd);
            o[3 * F + c] = (float)(be - mean * g * istd);
            o[4 * F + c] = (float)(g * istd);
            if (KIND == PH_G && b >= IDX) {                // BatchNorm backward constants, known for b >= IDX
                o[5 * F + c] = (float)(cells_bwd[(b * 2 + 0) * F + c] / cnt);
                o[6 * F + c] = (float)(cells_bwd[(b * 2 + 1) * F + c] / cnt);
            } else {
                o[5 * F + c] = 0.f;
                o[6 * F + c] = 0.f;
            }
        }
    }
    __syncthreads();

    // ---- persistent per-wave accumulators ------------------------------------------------------------
    float s_a[F], s_b[F];                 // BatchNorm reduction pair (fwd: z, z^2; bwd: dy, dy*xhat)
#pragma unroll
    for (int c = 0; c < F; ++c) s_a[c] = s_b[c] = 0.f;
    f32x4 acc_c0 = {0.f, 0.f, 0.f, 0.f}, acc_c1 = {0.f, 0.f, 0.f, 0.f};   // conv weight gradient (MFMA)
    f32x4 acc_th = {0.f, 0.f, 0.f, 0.f};                                   // theta / fc1 weight gradient (MFMA, RW 16)
    f32x4 acc_thg[NTH];                                                    // same, tiled, for the generic row width
#pragma unroll
    for (int i = 0; i < NTH; ++i) acc_thg[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int KX = KORD > 1 ? KORD - 1 : 1;                            // theta gradients of the orders 2.. (G_{2l}, KORD > 1)
    f32x4 acc_thk[KX], acc_thgk[KX][NTH];
#pragma unroll
    for (int q = 0; q < KX; ++q) {
        acc_thk[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NTH; ++i) acc_thgk[q][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    float acc_b = 0.f, acc_w2 = 0.f, acc_b2 = 0.f, acc_loss = 0.f;         // theta/fc1 bias, fc2 weight, fc2 bias, loss

    const float fc2_b = prm[off_fc2_b(N, L, K)];
    const int P = PFIX ? PFIX : a.P;                      // compile-time window length: the F_0 statistics passes unroll fully
    const int64_t sampleNP = (int64_t)N * P;
    const float inv_gb = 1.0f / (float)a.global_batch;

    // Saved tensors are packed: of each RW-lane row only the N patch lanes are stored (element (c, sample row, patch t) of a
    // tile at c * pitch + row * N + t), padded lanes read back as zero -- 12.5 % less traffic at N = 14.  The lane-distributed
    // adjacency rows (RW 16) use F of the 16 lanes.
    const int pitch = TSPW * N, loff = srow * N + t;
    const bool lane_ok = t < N;
    const size_t tile_floats = (size_t)F * pitch;
    // rows of samples beyond the batch (last tile) read as zero, whatever an earlier step -- or the matrix-core chain, which shares the
    // workspace and keeps +inf sentinels in it -- left there: 0 x inf would poison the MFMA weight-gradient accumulators
    bool tile_row_ok = true;
    auto load_tile = [&](const float* p, float (&v)[F]) {
#pragma unroll
        for (int c = 0; c < F; ++c) v[c] = 0.f;
        if (lane_ok && tile_row_ok) {
#pragma unroll
            for (int c = 0; c < F; ++c) v[c] = p[c * pitch];
        }
    };
    auto store_tile = [&](float* p, const float (&v)[F]) {
        if (lane_ok) {
#pragma unroll
            for (int c = 0; c < F; ++c) {
                p[c * pitch] = v[c];
            }
        }
    };
    // tensors whose next reader is several phases away (X_l, the x-hat mask): optionally stored past the caches
    auto store_tile_cold = [&](float* p, const float (&v)[F]) {
        if (lane_ok) {
#pragma unroll
            for (int c = 0; c < F; ++c) {
                __builtin_nontemporal_store(v[c], p + c * pitch);
            }
        }
    };
    // d X_L out of TOP is one value per (sample, patch): the max-pool routes the gradient to the arg-max channel.  It travels as
    // (value, channel) in the first two channel rows of rbuf instead of ten rows, nine of them zero.
    auto store_top_grad = [&](float* p, float v, int arg) {
        if (lane_ok) {
            p[0] = v;
            p[pitch] = __builtin_bit_cast(float, arg);
        }
    };
    auto load_top_grad = [&](const float* p, float (&v)[F]) {
        float val = 0.f;
        int arg = -1;
        if (lane_ok && tile_row_ok) {
            val = p[0];
            arg = __builtin_bit_cast(int, p[pitch]);
        }
#pragma unroll
        for (int c = 0; c < F; ++c) v[c] = c == arg ? val : 0.f;
    };
    constexpr int pitch_a = TSPW * F;
    const int loff_a = srow * F + t;
    const bool lane_ok_a = t < F;

    // Successive phases walk the tiles in opposite directions: what a phase wrote (or read) last is what the next one
    // touches first, while it is still in the 256 MB Infinity Cache.
    constexpr int CHAIN_POS = KIND == PH_F ? IDX : (KIND == PH_TOP ? NBN : NBN + 1 + (NBN - 1 - IDX));
    constexpr bool REVERSED = CHAIN_POS % 2 == 1;
    for (int64_t it = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave; it < a.ntiles; it += (int64_t)gridDim.x * WAVES_PER_BLOCK) {
        const int64_t tile = REVERSED ? a.ntiles - 1 - it : it;
        const int64_t s0 = tile * TSPW;
        const int ns = (int)((a.B - s0) < TSPW ? (a.B - s0) : TSPW);
        const bool rowok = srow < ns;
        const bool valid = rowok && (t < N);
        tile_row_ok = rowok;
        constexpr int NA = RW == 16 ? F : NPAIR;   // RW 16: lane-distributed adjacency rows (MFMA); else 55 row-uniform values
        float X[F], A[NA];
        auto slot = [&](int k) { return a.saved + ((size_t)k * a.ntiles + tile) * tile_floats + loff; };

        // ---- inputs: patch statistics + Pearson adjacency (F_0 computes and caches), or the saved X_l ----
        if constexpr (KIND == PH_F && IDX == 0) {
            __builtin_amdgcn_wave_barrier();
            stage_tile(gx + s0 * sampleNP, mywave, ns * (int)sampleNP, P, a.Ppad, a.magicP, a.vec4, lane);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c = 0; c < F; ++c) X[c] = 0.f;
            if (valid) {
                // even compile-time window: the patch is read from LDS once and held in registers for both passes (Ppad is even
                // whenever P is, so the 8-byte reads stay aligned)
                if constexpr (PFIX != 0 && PFIX % 2 == 0 && PFIX <= 64) patch_statistics_regs<PFIX, false>(mywave + (srow * N + t) * a.Ppad, X);
                else patch_statistics(mywave + (srow * N + t) * a.Ppad, P, X);
            }
            if constexpr (RW == 16) {
                pearson_rows_mfma(X, rowok, N, mywave, lane, A);   // padded sample rows are kept finite (zero) inside
                float* ca = a.cacheA + tile * (size_t)(F * pitch_a) + loff_a;
                if (lane_ok_a) {
#pragma unroll
                    for (int c = 0; c < F; ++c) ca[c * pitch_a] = A[c];
                }
            } else {
                pearson_adjacency<RW>(X, valid, N, A);
                float v = 0.f;                              // one sample per wavefront: lane i keeps entry i
#pragma unroll
                for (int i = 0; i < NPAIR; ++i) {
                    A[i] = rowok ? A[i] : 0.f;
                    v = (lane == i) ? A[i] : v;
                }
                a.cacheA[tile * 64 + lane] = v;
            }
            store_tile_cold(a.cacheX + tile * tile_floats + loff, X);
        } else {
            if constexpr (NEED_X) {
                load_tile(LSTART == 0 ? a.cacheX + tile * tile_floats + loff : slot(SV::X(LSTART)), X);
            }
            if constexpr (NEED_A) {
                if constexpr (RW == 16) {
                    const float* ca = a.cacheA + tile * (size_t)(F * pitch_a) + loff_a;
#pragma unroll
                    for (int c = 0; c < F; ++c) A[c] = 0.f;
                    if (lane_ok_a) {
#pragma unroll
                        for (int c = 0; c < F; ++c) A[c] = ca[c * pitch_a];
                    }
                } else {
                    const int v = __builtin_bit_cast(int, a.cacheA[tile * 64 + lane]);
#pragma unroll
                    for (int i = 0; i < NPAIR; ++i) A[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(v, i));
                }
            }
        }
        const uint32_t ctr_base = (uint32_t)((a.sample_offset + s0 + srow) * F) * (uint32_t)N + (uint32_t)t;

        // ---- previous layer in full (F_{2l} with l >= 1) ---------------------------------------------------
        if constexpr (WITH_PREV) {
            float pz2[F], po0[F];
            constexpr int lq = LY - 1;
            const float* b2 = bnc + (2 * lq + 1) * BNC * F;
            load_tile(slot(SV::O0(lq)), po0);
            load_tile(slot(SV::Z2(lq)), pz2);
#pragma unroll
            for (int c = 0; c < F; ++c) {
                const float x1 = relu(fmaf(pz2[c], b2[2 * F + c], b2[3 * F + c]));
                float o1 = relu(x1 + po0[c]);
                if (a.dropout_p > 0.f) {
                    const uint32_t h = lowbias32((ctr_base + (uint32_t)(c * N)) ^ dkey[lq]);
                    o1 = h >= a.drop_thr ? o1 * a.drop_scale : 0.f;
                }
                X[c] = valid ? o1 + X[c] : 0.f;
            }
            store_tile_cold(slot(SV::X(LY)), X);            // X_l is final from here on: stored for the later phases
        }

        // ---- layer LY forward, as far as this phase needs it ---------------------------------------------------
        const float* lp = prm + LY * LS;
        const float* b1 = bnc + (2 * LY) * BNC * F;
        const float* b2 = bnc + (2 * LY + 1) * BNC * F;
        float AX[F], H[F], z1[F];
        if constexpr (NEED_A) { if constexpr (RW == 16) adj_aggregate_mfma(A, X, AX); else adj_aggregate(A, X, AX); }

        if constexpr (KIND == PH_F && BLK == 0) {       // F_{2l}: theta projection, conv_block1, its statistics
            const float tb = vecs[LY * TRW + t];
#pragma unroll
            for (int c = 0; c < F; ++c) H[c] = tb;
            R16::project10(H, AX, w_cur + t * TWS, N);
            if constexpr (KORD > 1) {                   // orders 2..: theta_kk(A^(kk+1) X), Model.py:82-88, as A (A^kk X)
                float AXa[F], AXb[F];
#pragma unroll
                for (int c = 0; c < F; ++c) AXa[c] = AX[c];
#pragma unroll
                for (int kk = 1; kk < KORD; ++kk) {
                    if constexpr (RW == 16) adj_aggregate_mfma(A, AXa, AXb); else adj_aggregate(A, AXa, AXb);
                    R16::project10(H, AXb, w_cur + (kk * TRW + t) * TWS, N);
#pragma unroll
                    for (int c = 0; c < F; ++c) AXa[c] = AXb[c];
                }
            }
#pragma unroll
            for (int c = 0; c < F; ++c) H[c] = leaky(H[c]);
            if constexpr (CONV_FROM_LDS) causal_conv_lds<TRW, 1>(H, convT, t, z1);
            else causal_conv<TRW, 1>(H, conv_weights<true, 5>(lp + off_conv_w(N, 0, K), (int)tile), t, z1);
            store_tile(slot(SV::H(LY)), H);
            store_tile(slot(SV::Z1(LY)), z1);
#pragma unroll
            for (int c = 0; c < F; ++c) {
                const float z = valid ? z1[c] : 0.f;
                s_a[c] += z;
                s_b[c] = fmaf(z, z, s_b[c]);
            }
            continue;
        }
        if constexpr (KIND != PH_TOP) {                  // F_{2l+1}, G_{2l+1}, G_{2l}: conv_block1 output as F_{2l} left it
            load_tile(slot(SV::Z1(LY)), z1);
        }
        if constexpr (KIND != PH_TOP && !(KIND == PH_G && BLK == 1)) {
            load_tile(slot(SV::H(LY)), H);
        }

        if constexpr (KIND == PH_G && BLK == 0) {
            // ---- G_{2l}: BatchNorm 2l backward, conv_block1 + theta gradients, dX_l ------------------------------
            float g0[F], dz[F];
            load_tile(a.sbuf + tile * tile_floats + loff, g0);                    // d(x0 + H), written by G_{2l+1}
#pragma unroll
            for (int c = 0; c < F; ++c) {
                const float x0 = relu(fmaf(z1[c], b1[2 * F + c], b1[3 * F + c]));
                const float dy = (x0 > 0.f && valid) ? g0[c] : 0.f;
                const float xh = (z1[c] - b1[0 * F + c]) * b1[1 * F + c];
                const float v = b1[4 * F + c] * (dy - b1[5 * F + c] - xh * b1[6 * F + c]);
                dz[c] = valid ? v : 0.f;
            }
            {
                float hs[F];
#pragma unroll
                for (int c = 0; c < F; ++c) hs[c] = R16::template shr<1>(H[c], t);
                conv_wgrad_mfma(mywave, dz, H, hs, lane, acc_c0, acc_c1);
            }
            float dH[F];
            if constexpr (CONV_FROM_LDS) causal_conv_T_lds<RW, 1>(dz, convT, t, dH);
            else causal_conv_T<RW, 1>(dz, conv_weights<false, 1>(lp + off_conv_w(N, 0, K), (int)tile), t, dH);
#pragma unroll
            for (int c = 0; c < F; ++c) {
                const float g = dH[c] + g0[c];
                dH[c] = valid ? (H[c] > 0.f ? g : g * LEAKY) : 0.f;               // d(theta pre-activation)
                if constexpr (RW == 16) acc_th = __builtin_amdgcn_mfma_f32_16x16x4f32(dH[c], AX[c], acc_th, 0, 0, 0);
                acc_b += dH[c];
            }
            if constexpr (RW != 16) outer_grad_mfma<RW, F>(mywave, dH, AX, lane, acc_thg);
            if constexpr (KORD > 1) {                   // d theta_kk = d Hpre^T (A^(kk+1) X)
                float AXa[F], AXb[F];
#pragma unroll
                for (int c = 0; c < F; ++c) AXa[c] = AX[c];
#pragma unroll
                for (int kk = 1; kk < KORD; ++kk) {
                    if constexpr (RW == 16) adj_aggregate_mfma(A, AXa, AXb); else adj_aggregate(A, AXa, AXb);
                    if constexpr (RW == 16) {
#pragma unroll
                        for (int c = 0; c < F; ++c) acc_thk[kk - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(dH[c], AXb[c], acc_thk[kk - 1], 0, 0, 0);
                    } else {
                        outer_grad_mfma<RW, F>(mywave, dH, AXb, lane, acc_thgk[kk - 1]);
                    }
#pragma unroll
                    for (int c = 0; c < F; ++c) AXa[c] = AXb[c];
                }
            }
            if constexpr (LY > 0) {
                float dAX[F], dXl[F];
#pragma unroll
                for (int c = 0; c < F; ++c) dAX[c] = 0.f;
                if constexpr (KORD > 1) {
                    // d X = sum_kk A^(kk+1) (d Hpre theta_kk) = A (u_0 + A (u_1 + A u_2 ...)), u_kk = d Hpre theta_kk (A symmetric)
                    R16::project10(dAX, dH, w_tr + ((KORD - 1) * TRW + t) * TWS, N);
#pragma unroll
                    for (int kk = KORD - 2; kk >= 0; --kk) {
                        float nxt[F];
                        if constexpr (RW == 16) adj_aggregate_mfma(A, dAX, nxt); else adj_aggregate(A, dAX, nxt);
                        R16::project10(nxt, dH, w_tr + (kk * TRW + t) * TWS, N);
#pragma unroll
                        for (int c = 0; c < F; ++c) dAX[c] = nxt[c];
                    }
                } else {
                R16::project10(dAX, dH, w_tr + t * TWS, N);                         // dHpre . theta
                }
                if constexpr (RW == 16) adj_aggregate_mfma(A, dAX, dXl); else adj_aggregate(A, dAX, dXl);   // A is symmetric: A^T = A
                float* rb = a.rbuf + tile * tile_floats + loff;
                constexpr int lq = LY - 1;
                float rbv[F], pz2[F];
                if constexpr (LY == L - 1) load_top_grad(rb, rbv); else load_tile(rb, rbv);
                // top of layer l-1: the sums of BatchNorm 2l-1 need its x-hat where the gradient passes.  Both come from the stored
                // conv_block2 output: the gradient passes where x1 = relu(BN(z2)) > 0 (then o1 = relu(x1 + o0) > 0 as well, o0 >= 0)
                load_tile(slot(SV::Z2(lq)), pz2);
                const float* q2 = bnc + (2 * lq + 1) * BNC * F;
#pragma unroll
                for (int c = 0; c < F; ++c) {
                    const float dX = valid ? dXl[c] + rbv[c] : 0.f;               // + residual branch: d X_{l+1}
                    rbv[c] = dX;                                                   // = d X_l, read by G_{2l-1}
                    const bool open = valid && fmaf(pz2[c], q2[2 * F + c], q2[3 * F + c]) > 0.f;
                    const float xh = open ? (pz2[c] - q2[0 * F + c]) * q2[1 * F + c] : INFINITY;
                    float g = dX;
                    if (a.dropout_p > 0.f) {
                        const uint32_t h = lowbias32((ctr_base + (uint32_t)(c * N)) ^ dkey[lq]);
                        g = h >= a.drop_thr ? g * a.drop_scale : 0.f;
                    }
                    const bool pass = xh < INFINITY;
                    const float dy = pass ? g : 0.f;
                    s_a[c] += dy;
                    s_b[c] = fmaf(dy, pass ? xh : 0.f, s_b[c]);
                }
                store_tile(rb, rbv);
            }
            continue;
        }

        float o0[F], z2[F];
        if constexpr (KIND == PH_F && BLK == 1) {       // F_{2l+1}: conv_block2 and its statistics
#pragma unroll
            for (int c = 0; c < F; ++c) o0[c] = relu(relu(fmaf(z1[c], b1[2 * F + c], b1[3 * F + c])) + H[c]);
            if constexpr (CONV_FROM_LDS) causal_conv_lds<TRW, 2>(o0, convT, t, z2);
            else causal_conv<TRW, 2>(o0, conv_weights<true, 6>(lp + off_conv_w(N, 1, K), (int)tile), t, z2);
            store_tile(slot(SV::O0(LY)), o0);
            store_tile(slot(SV::Z2(LY)), z2);
#pragma unroll
            for (int c = 0; c < F; ++c) {
                const float z = valid ? z2[c] : 0.f;
                s_a[c] += z;
                s_b[c] = fmaf(z, z, s_b[c]);
            }
            continue;
        }
        load_tile(slot(SV::O0(LY)), o0);
        load_tile(slot(SV::Z2(LY)), z2);

        if constexpr (KIND == PH_TOP) {
            // ---- layer L-1 output, head forward (Model.py:218-221), head backward --------------------------------
            float x1v[F], o1v[F];
#pragma unroll
            for (int c = 0; c < F; ++c) {
                x1v[c] = relu(fmaf(z2[c], b2[2 * F + c], b2[3 * F + c]));
                o1v[c] = relu(x1v[c] + o0[c]);
                float o1 = o1v[c];
                if (a.dropout_p > 0.f) {
                    const uint32_t h = lowbias32((ctr_base + (uint32_t)(c * N)) ^ dkey[LY]);
                    o1 = h >= a.drop_thr ? o1 * a.drop_scale : 0.f;
                }
                X[c] = valid ? o1 + X[c] : 0.f;
            }
            float pooled = X[0];
            int arg = 0;
#pragma unroll
            for (int c = 1; c < F; ++c) {
                const bool take = (X[c] > pooled) || (X[c] != X[c] && pooled == pooled);
                pooled = take ? X[c] : pooled;
                arg = take ? c : arg;
            }
            pooled = valid ? pooled : 0.f;
            float y1 = vecs[L * TRW + t];
            R16::project1(y1, pooled, w_aux + t * TWS, N);
            y1 = relu(y1);
            const float w2 = vecs[(L + 1) * TRW + t];
            const float pred = R16::allsum(y1 * w2) + fc2_b;
            float dpred = 0.f;
            if (rowok) {
                if (a.has_dpred == 1) {
                    dpred = gy[s0 + srow];
                } else if (a.has_dpred == 0) {
                    const float diff = pred - gy[s0 + srow];
                    dpred = 2.f * diff * inv_gb;
                    if (t == 0) acc_loss = fmaf(diff, diff, acc_loss);
                }
                if (t == 0 && a.write_pred) a.pred[s0 + srow] = pred;
            }
            if (!a.do_backward) continue;
            const float dy1 = (y1 > 0.f) ? dpred * w2 : 0.f;                       // d(fc1 pre-activation), lane j
            float dpool = 0.f;
            R16::project1(dpool, dy1, w_tr + t * TWS, N);                          // sum_j dy1[j] fc1.w[j][t]
            acc_w2 = fmaf(dpred, y1, acc_w2);
            acc_b2 += (t == 0) ? dpred : 0.f;
            acc_b += dy1;
            if constexpr (RW == 16) {
                acc_th = __builtin_amdgcn_mfma_f32_16x16x4f32(dy1, pooled, acc_th, 0, 0, 0);   // d fc1.w[j][t]
            } else {
                const float p1[1] = {dy1}, q1[1] = {pooled};
                outer_grad_mfma<RW, 1>(mywave, p1, q1, lane, acc_thg);
            }
            float rbv[F];
#pragma unroll
            for (int c = 0; c < F; ++c) {
                const float dX = (valid && c == arg) ? dpool : 0.f;                 // d X_L (max-pool routes to the arg-max channel)
                rbv[c] = dX;
                float g = dX;
                if (a.dropout_p > 0.f) {
                    const uint32_t h = lowbias32((ctr_base + (uint32_t)(c * N)) ^ dkey[LY]);
                    g = h >= a.drop_thr ? g * a.drop_scale : 0.f;
                }
                const float dy = (o1v[c] > 0.f && x1v[c] > 0.f && valid) ? g : 0.f;
                const float xh = (z2[c] - b2[0 * F + c]) * b2[1 * F + c];
                s_a[c] += dy;
                s_b[c] = fmaf(dy, xh, s_b[c]);
            }
            store_top_grad(a.rbuf + tile * tile_floats + loff, valid ? dpool : 0.f, arg);
            continue;
        }

        if constexpr (KIND == PH_G && BLK == 1) {
            // ---- G_{2l+1}: BatchNorm 2l+1 backward, conv_block2 gradient, d(x0 + H) -----------------------------------
            float gsum[F], dz[F], rbv[F];
            if constexpr (LY == L - 1) load_top_grad(a.rbuf + tile * tile_floats + loff, rbv);
            else load_tile(a.rbuf + tile * tile_floats + loff, rbv);
#pragma unroll
            for (int c = 0; c < F; ++c) {
                const float x1 = relu(fmaf(z2[c], b2[2 * F + c], b2[3 * F + c]));
                const float o1 = relu(x1 + o0[c]);
                float g = rbv[c];                                                   // d X_{l+1}
                if (a.dropout_p > 0.f) {
                    const uint32_t h = lowbias32((ctr_base + (uint32_t)(c * N)) ^ dkey[LY]);
                    g = h >= a.drop_thr ? g * a.drop_scale : 0.f;
                }
                g = (o1 > 0.f && valid) ? g : 0.f;
                gsum[c] = g;                                                        // d(x1 + o0)
                const float dy = (x1 > 0.f) ? g : 0.f;
                const float xh = (z2[c] - b2[0 * F + c]) * b2[1 * F + c];
                const float v = b2[4 * F + c] * (dy - b2[5 * F + c] - xh * b2[6 * F + c]);
                dz[c] = valid ? v : 0.f;
            }
            {
                float hs[F];
#pragma unroll
                for (int c = 0; c < F; ++c) hs[c] = R16::template shr<2>(o0[c], t);
                conv_wgrad_mfma(mywave, dz, o0, hs, lane, acc_c0, acc_c1);
            }
            float d_o0[F];
            if constexpr (CONV_FROM_LDS) causal_conv_T_lds<RW, 2>(dz, convT, t, d_o0);
            else causal_conv_T<RW, 2>(dz, conv_weights<true, 2>(lp + off_conv_w(N, 1, K), (int)tile), t, d_o0);
            float sbv[F];
#pragma unroll
            for (int c = 0; c < F; ++c) {
                const float x0 = relu(fmaf(z1[c], b1[2 * F + c], b1[3 * F + c]));
                float g = d_o0[c] + gsum[c];
                g = (o0[c] > 0.f && valid) ? g : 0.f;
                sbv[c] = g;                                                         // d(x0 + H), read by G_{2l}
                const float dy = (x0 > 0.f) ? g : 0.f;
                const float xh = (z1[c] - b1[0 * F + c]) * b1[1 * F + c];
                s_a[c] += dy;
                s_b[c] = fmaf(dy, xh, s_b[c]);
            }
            store_tile(a.sbuf + tile * tile_floats + loff, sbv);
            continue;
        }
    }

    // ---- epilogue: block reduction, then ONE fp64 atomic per cell / one partial row per block -------------
    // (1) BatchNorm pair: every F kernel, TOP (if backward) and G_i with i > 0
    const bool has_pair = (KIND == PH_F) || (KIND == PH_TOP && a.do_backward) || (KIND == PH_G && IDX > 0);
    if (has_pair) {
#pragma unroll
        for (int c = 0; c < F; ++c) {
            const float va = wave_sum(s_a[c]), vb = wave_sum(s_b[c]);
            if (lane == 0) {
                redp[wave * 24 + c] = va;
                redp[wave * 24 + F + c] = vb;
            }
        }
    }
    if (KIND == PH_TOP) {
        const float vl = wave_sum(acc_loss);
        if (lane == 0) redp[wave * 24 + 2 * F] = vl;
    }
    __syncthreads();
    if (has_pair && threadIdx.x < 2 * F) {
        double v = 0.0;
        for (int w = 0; w < WAVES_PER_BLOCK; ++w) v += (double)redp[w * 24 + threadIdx.x];
        const int which = threadIdx.x / F, c = threadIdx.x % F;
        double* cell = a.cells + (blockIdx.x % CELL_REPLICAS) * CS;
        if (KIND == PH_F) cell += cell_fwd(L) + (IDX * 2 + which) * F + c;
        else if (KIND == PH_TOP) cell += cell_bwd(L) + ((NBN - 1) * 2 + which) * F + c;
        else cell += cell_bwd(L) + ((IDX - 1) * 2 + which) * F + c;
        atomicAdd(cell, v);
    }
    if (KIND == PH_TOP && threadIdx.x == 2 * F && a.has_dpred == 0) {
        double v = 0.0;
        for (int w = 0; w < WAVES_PER_BLOCK; ++w) v += (double)redp[w * 24 + 2 * F];
        atomicAdd(a.cells + (blockIdx.x % CELL_REPLICAS) * CS + cell_loss(L), v);
    }
    if constexpr (KIND == PH_F) return;
    if (KIND == PH_TOP && !a.do_backward) return;

    // (2) weight-gradient accumulators -> per-block partial row (waves add in a fixed order: deterministic)
    float* row = a.gpart + (size_t)blockIdx.x * a.pcount;
    for (int w = 0; w < WAVES_PER_BLOCK; ++w) {
        if (wave == w) {
            float* r = red + lane;
            const float v[15] = {acc_th[0], acc_th[1], acc_th[2], acc_th[3], acc_c0[0], acc_c0[1], acc_c0[2], acc_c0[3],
                                 acc_c1[0], acc_c1[1], acc_c1[2], acc_c1[3], acc_b, acc_w2, acc_b2};
#pragma unroll
            for (int k = 0; k < 15; ++k) r[k * 64] = (w == 0) ? v[k] : r[k * 64] + v[k];
            if constexpr (RW != 16) {
#pragma unroll
                for (int q = 0; q < NTH; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int k = 15 + q * 4 + e;
                        r[k * 64] = (w == 0) ? acc_thg[q][e] : r[k * 64] + acc_thg[q][e];
                    }
            }
            if constexpr (KORD > 1 && KIND == PH_G) {
#pragma unroll
                for (int kk = 1; kk < KORD; ++kk)
#pragma unroll
                    for (int q = 0; q < (RW == 16 ? 1 : NTH); ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int k = RED_K1 + (kk - 1) * RED_TH + q * 4 + e;
                            const float v2 = RW == 16 ? acc_thk[kk - 1][e] : acc_thgk[kk - 1][q][e];
                            r[k * 64] = (w == 0) ? v2 : r[k * 64] + v2;
                        }
            }
        }
        __syncthreads();
    }
    // [N][N] matrix gradient (theta or fc1): MFMA D layout -> row j = 16m + 4*(lane>>4) + reg, column k = 16n + (lane & 15)
    // (`first`: the accumulator's first row in `red`: 0 (RW 16) / 15 (generic) for order 1 and fc1, RED_K1 + ... for the orders 2..)
    auto write_matrix = [&](float* dst, int first) {
        if constexpr (RW == 16) {
            for (int i = threadIdx.x; i < 4 * 64; i += BLOCK) {
                const int rg = i / 64, ln = i % 64, j = 4 * (ln >> 4) + rg, k = ln & 15;
                if (j < N && k < N) dst[j * N + k] = red[(first + rg) * 64 + ln];
            }
        } else {
            constexpr int NT = RW / 16;
            for (int i = threadIdx.x; i < NTH * 4 * 64; i += BLOCK) {
                const int q = i / 256, rg = (i / 64) % 4, ln = i % 64;
                const int j = 16 * (q / NT) + 4 * (ln >> 4) + rg, k = 16 * (q % NT) + (ln & 15);
                if (j < N && k < N) dst[j * N + k] = red[(first + q * 4 + rg) * 64 + ln];
            }
        }
    };
    constexpr int RED_FIRST = RW == 16 ? 0 : 15;
    if (KIND == PH_TOP) {
        write_matrix(row + off_fc1_w(N, L, K), RED_FIRST);
        if (threadIdx.x < N) {
            const int j = threadIdx.x;
            float vb = 0.f, vw = 0.f;
            for (int s = 0; s < TSPW; ++s) { vb += red[12 * 64 + s * TRW + j]; vw += red[13 * 64 + s * TRW + j]; }
            row[off_fc1_b(N, L, K) + j] = vb;
            row[off_fc2_w(N, L, K) + j] = vw;
        }
        if (threadIdx.x == 0) {
            float v = 0.f;
            for (int s = 0; s < TSPW; ++s) v += red[14 * 64 + s * TRW];
            row[off_fc2_b(N, L, K)] = v;
        }
    }
    if (KIND == PH_G) {
        constexpr int l = IDX / 2, blk = IDX % 2;
        float* lrow = row + l * LS;
        // conv weight [co][ci][tap]: acc0 column j<10 -> (ci=j, tap 1); 10..15 -> (ci=j-10, tap 0); acc1 column j<4 -> (ci=6+j, tap 0)
        for (int i = threadIdx.x; i < 8 * 64; i += BLOCK) {
            const int rg = (i / 64) % 4, which = i / 256, ln = i % 64;
            const int co = 4 * (ln >> 4) + rg, j = ln & 15;
            int ci, tap;
            if (which == 0) { ci = j < F ? j : j - F; tap = j < F ? 1 : 0; }
            else { ci = 6 + j; tap = 0; }
            if (co < F && ci < F && (which == 0 || j < 4))
                lrow[off_conv_w(N, blk, K) + (co * F + ci) * 2 + tap] = red[(4 + which * 4 + rg) * 64 + ln];
        }
        if (blk == 0) {
            write_matrix(lrow + off_theta_w(N), RED_FIRST);
            if constexpr (KORD > 1) {
#pragma unroll
                for (int kk = 1; kk < KORD; ++kk) write_matrix(lrow + off_theta_w(N, kk), RED_K1 + (kk - 1) * RED_TH);
            }
            if (threadIdx.x < N) {
                float vb = 0.f;
                for (int s = 0; s < TSPW; ++s) vb += red[12 * 64 + s * TRW + threadIdx.x];
                for (int kk = 0; kk < KORD; ++kk) lrow[off_theta_b(N, kk) + threadIdx.x] = vb;       // every order's bias sees d Hpre summed
            }
        }
    }
}

template <int RW, int L, int KIND, int IDX, int NFIX = 0, int PFIX = 0, int KORD = 1>
__global__ __launch_bounds__(BLOCK, KORD > 1 ? 1 : phase_min_waves(RW, KIND, IDX)) void stgcn_train_phase_kernel(const float* __restrict__ gx,
                                                                  const float* __restrict__ prm,
                                                                  const float* __restrict__ gy, TrainK a) {
    train_phase_body<RW, L, KIND, IDX, NFIX, PFIX, KORD>(gx, prm, gy, a);
}

// ------------------------------------------------------------------------------------------------
// finalize: per-block partial rows + BatchNorm cells -> flat gradient, loss, batch statistics
// ------------------------------------------------------------------------------------------------
struct FinalizeK {
    const float* gpart;
    double* cells;
    float* grads;
    float* loss;
    float* bn_batch;
    int grid_top;
    int grid_g[16];       // grid of G_i, i = BatchNorm index
    int N, L, pcount;
    int K;                // MPNN order (parameter offsets)
    int64_t B, global_batch;
    int write_grads, write_loss;
    float moment_weight;
    float cell_grad_scale;   // 1; under SyncBN the caller's factor: the cells then hold GLOBAL sums on every rank and the gradient
                             // bucket is summed over the ranks afterwards (1 on one rank and 0 elsewhere, or 1 / world_size)
    // optional fused optimizer (single-GPU step): Adam on the parameter this wavefront just reduced, BatchNorm
    // running statistics from the batch statistics block 0 just finished
    float* params;
    float* exp_avg;
    float* exp_avg_sq;
    float* bn_running;
    float beta1, beta2, eps, weight_decay, bn_momentum;
    int fused_opt;
    int guard;               // matrix-core chain: a raised status word (a value left the f16 range, stgcn_train_mx.hip) leaves parameters,
                             // optimizer state and running statistics untouched and reports a NaN loss
    int clean;               // matrix-core chain: the last workgroup zeroes the cells and the status word and sets the clean token
};

// Gradient rows of the phase kernels' workgroups -> gradient (+ Adam): a workgroup owns FIN_COLS consecutive parameters (lane =
// parameter, so every row read is one coalesced 256-byte line) and its FIN_SLICES wavefronts split the rows; the slices are
// combined in a fixed order (bit-reproducible).  One wavefront per parameter, as before round-1h, read the rows with a
// 6-KB stride: 12 us for 7.8 MB.
constexpr int FIN_COLS = 64, FIN_SLICES = 16;

// One unit = FIN_COLS consecutive parameters.  `part` is a [FIN_SLICES][FIN_COLS] LDS scratch; the calling workgroup's wavefronts
// split the FIN_SLICES row slices among themselves (SPW slices per wavefront: 1 in the finalize kernel with its 16 wavefronts, 4 in
// the cooperative kernel with its 4) -- every slice is summed in the same order by whoever owns it and the slices are combined in a
// fixed order, so both launch forms produce the same bits.
template <int SPW>
__device__ __forceinline__ void finalize_unit(const FinalizeK& f, int unit, float (*part)[FIN_COLS], int lane, int wave) {
    const int N = f.N, L = f.L, K = f.K, LS = layer_stride(N, K);
    const float lr_over_bc1 = step_scratch(f.cells, L)->lr_over_bc1, inv_sqrt_bc2 = step_scratch(f.cells, L)->inv_sqrt_bc2;
    const int p = unit * FIN_COLS + lane;
    const bool valid = p < f.pcount;
    // (the optimizer's operands are requested in front of the row sums: they do not depend on them -- one memory round trip less behind
    // the reduction)
    float p_old = 0.f, m_old = 0.f, v_old = 0.f;
    bool skip_opt = true;
    if (wave == 0 && valid && f.fused_opt) {
        p_old = f.params[p];
        m_old = f.exp_avg[p];
        v_old = f.exp_avg_sq[p];
        skip_opt = f.guard && step_scratch(f.cells, L)->pad[0] != 0u;
    }
    int nblk = 0;
    bool from_cells = false;
    int bn = 0, which = 0, c = 0;
    if (valid) {
        if (p >= L * LS) {
            nblk = f.grid_top;
        } else {
            const int l = p / LS, o = p % LS;
            if (o < off_conv_w(N, 0, K)) nblk = f.grid_g[2 * l];                  // theta w/b
            else {
                const int blk = o >= off_conv_w(N, 1, K) ? 1 : 0;
                const int oo = o - off_conv_w(N, blk, K);
                nblk = f.grid_g[2 * l + blk];
                if (oo >= CONVW) { from_cells = true; bn = 2 * l + blk; which = (oo - CONVW) / F; c = (oo - CONVW) % F; }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < SPW; ++q) {
        const int slice = wave * SPW + q;
        float v = 0.f;
        if (valid && !from_cells) {
            // latency-bound: thirty-two (then sixteen) independent row reads in flight per lane, then the tail -- the 512 rows of the
            // matrix-core chain are ONE round trip per lane; the order of the additions is the row order either way
            const float* col = f.gpart + p;
            const size_t rs = (size_t)f.pcount * FIN_SLICES;
            int b = slice;
            for (; b + 31 * FIN_SLICES < nblk; b += 32 * FIN_SLICES) {
                float r[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) r[u] = col[(size_t)b * f.pcount + u * rs];
#pragma unroll
                for (int u = 0; u < 32; ++u) v += r[u];
            }
            for (; b + 15 * FIN_SLICES < nblk; b += 16 * FIN_SLICES) {
                float r[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) r[u] = col[(size_t)b * f.pcount + u * rs];
#pragma unroll
                for (int u = 0; u < 16; ++u) v += r[u];
            }
            for (; b < nblk; b += FIN_SLICES) v += col[(size_t)b * f.pcount];
        }
        part[slice][lane] = v;
    }
    __syncthreads();
    if (wave == 0 && valid) {
        float v;
        if (from_cells) {
            // d gamma = sum dy*xhat, d beta = sum dy
            v = (float)(cell_sum(f.cells, L, cell_bwd(L) + (bn * 2 + (which == 0 ? 1 : 0)) * F + c) * (double)f.cell_grad_scale);
        } else {
            v = 0.f;
#pragma unroll
            for (int sl = 0; sl < FIN_SLICES; ++sl) v += part[sl][lane];
        }
        f.grads[p] = v;
        if (f.fused_opt && !skip_opt) {     // torch.optim.Adam, same arithmetic as adam_step_kernel
            const float pi = p_old;
            const float gi = fmaf(f.weight_decay, pi, v);
            const float mi = fmaf(f.beta1, m_old, (1.f - f.beta1) * gi);
            const float vi = fmaf(f.beta2, v_old, (1.f - f.beta2) * gi * gi);
            f.exp_avg[p] = mi;
            f.exp_avg_sq[p] = vi;
            f.params[p] = pi - lr_over_bc1 * (mi / (sqrtf(vi) * inv_sqrt_bc2 + f.eps));
        }
    }
    __syncthreads();
}

// loss, batch statistics and (fused optimizer) the BatchNorm running statistics: one workgroup
__device__ __forceinline__ void finalize_stats(const FinalizeK& f, int tid, int nthreads) {
    const int N = f.N, L = f.L;
    (void)N;
    const double cnt = step_scratch(f.cells, L)->bn_count;
    const bool tripped = f.guard && step_scratch(f.cells, L)->pad[0] != 0u;
    if (tid == 0 && f.write_loss) f.loss[0] = tripped ? __builtin_nanf("") : (float)(cell_sum(f.cells, L, cell_loss(L)) / (double)f.global_batch);
    if (tid == 0 && tripped) step_scratch(f.cells, L)->pad[3] += 1u;      // sticky: a rejected step never vanishes (a caller that does not read
                                                                          // the loss every step checks this counter instead)
    for (int i = tid; i < 2 * L * F; i += nthreads) {
        const int b = i / F, c = i % F;
        const double s2 = cell_sum(f.cells, L, cell_fwd(L) + (b * 2 + 1) * F + c);
        const double mean = cell_sum(f.cells, L, cell_fwd(L) + (b * 2 + 0) * F + c) / cnt;
        double var = s2 / cnt - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        if (f.moment_weight > 0.f) {
            f.bn_batch[(b * 2 + 0) * F + c] = (float)(mean * (double)f.moment_weight);
            f.bn_batch[(b * 2 + 1) * F + c] = (float)(s2 / cnt * (double)f.moment_weight);
        } else {
            f.bn_batch[(b * 2 + 0) * F + c] = (float)mean;
            f.bn_batch[(b * 2 + 1) * F + c] = (float)var;
        }
        if (f.fused_opt && f.bn_running && !tripped) {          // nn.BatchNorm1d running statistics (unbiased running variance)
            const float unbias = cnt > 1.0 ? (float)(cnt / (cnt - 1.0)) : 1.f;
            float* rm = f.bn_running + (b * 2 + 0) * F + c;
            float* rv = f.bn_running + (b * 2 + 1) * F + c;
            *rm = (1.f - f.bn_momentum) * *rm + f.bn_momentum * (float)mean;
            *rv = (1.f - f.bn_momentum) * *rv + f.bn_momentum * ((float)var * unbias);
        }
    }
}

__global__ __launch_bounds__(FIN_COLS * FIN_SLICES) void stgcn_train_finalize_kernel(FinalizeK f) {
    __shared__ float part[FIN_SLICES][FIN_COLS];
    if (f.write_grads) finalize_unit<1>(f, blockIdx.x, part, threadIdx.x & 63, threadIdx.x >> 6);
    if (blockIdx.x == 0) finalize_stats(f, threadIdx.x, blockDim.x);
    if (f.clean) {
        // A matrix-core step leaves its workspace ready for a step WITHOUT a prepare launch (RULGNN_TRAIN_WS_CLEAN): the workgroup that
        // finishes last -- every other one has read the cells and the status word by then -- zeroes them and sets the clean token.
        __shared__ int last;
        StepScratch* sc = step_scratch(f.cells, f.L);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) last = atomicAdd(&sc->pad[2], 1u) == gridDim.x - 1 ? 1 : 0;
        __syncthreads();
        if (last) {
            const int stride = cell_stride(f.L);
            for (int i = threadIdx.x; i < stride * CELL_REPLICAS; i += blockDim.x) f.cells[i] = 0.0;
            if (threadIdx.x == 0) { sc->pad[0] = 0u; sc->pad[2] = 0u; sc->pad[1] = WS_CLEAN_TOKEN; *step_barrier(f.cells, f.L) = 0u; }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct WsLayout {
    size_t off_cacheX, off_cacheA, off_cells, off_gpart, off_saved, off_rbuf, off_sbuf, total;
    size_t cells_bytes;
    int max_grid;
};

static int max_resident_blocks() {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    return cus * 8;
}

// training geometry: RW 16 for num_patch <= 16, else one sample per wavefront (RW 64)
static int train_geometry(const rulgnn_stgcn_shape* s, TileGeom* g) {
    const int rc = tile_geometry(s, g);
    if (rc != RULGNN_OK) return rc;
    if (g->RW != 16) {
        g->RW = 64;
        g->SPW = 1;
        const int raw = s->num_patch * g->Ppad;
        g->stage_floats = (raw + 3) & ~3;
    }
    g->ntiles = (s->batch + g->SPW - 1) / g->SPW;
    return RULGNN_OK;
}

static void ws_layout(const rulgnn_stgcn_shape* s, const TileGeom& g, WsLayout* w) {
    const int L = s->num_layers, N = s->num_patch;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t o = 0;
    w->off_cacheX = o; o = al(o + (size_t)g.ntiles * F * 64 * sizeof(float));
    w->off_cacheA = o; o = al(o + (size_t)g.ntiles * (g.RW == 16 ? F : 1) * 64 * sizeof(float));
    w->cells_bytes = sizeof(double) * (size_t)CELL_REPLICAS * cell_stride(L) + sizeof(StepScratch) + 64;   // + grid-barrier counter
    w->off_cells = o; o = al(o + w->cells_bytes);
    w->max_grid = 2048;
    w->off_gpart = o; o = al(o + (size_t)w->max_grid * param_count(N, L, s->mpnn_k) * sizeof(float));
    const size_t tile_bytes = (size_t)g.ntiles * F * 64 * sizeof(float);
    w->off_saved = o; o = al(o + (size_t)saved_slots(L) * tile_bytes);
    w->off_rbuf = o; o = al(o + tile_bytes);
    w->off_sbuf = o; o = al(o + tile_bytes);
    w->total = o;
}

size_t stgcn_train_workspace_bytes(const rulgnn_stgcn_shape* s) {
    TileGeom g;
    if (train_geometry(s, &g) != RULGNN_OK) return 0;
    if (s->num_layers > 3 || (g.RW != 16 && s->num_layers > 2)) return 0;
    WsLayout w;
    ws_layout(s, g, &w);
    return w.total;
}

static int wave_area_for(int kind, int idx, const TileGeom& g) {
    if (kind == PH_F) return idx == 0 ? g.stage_floats : 0;
    if (kind == PH_G) return TT_ROWS * TT_STRIDE;
    return g.RW == 16 ? 0 : TT_ROWS * TT_STRIDE;          // TOP, generic row width: fc1 gradient through the transpose tile
}

// (kord: the order a theta phase is specialised on -- 1 for every other phase; mirrors the LDS carve of train_phase_body)
static size_t train_lds_bytes(int RW, int L, int wave_area, int kord = 1) {
    const int tws = RW + 4, nth = RW == 16 ? 0 : (RW / 16) * (RW / 16);
    const int slots = kord == 1 ? 3 : kord, red_th = RW == 16 ? 4 : 4 * nth;
    const size_t fl = (size_t)2 * cell_stride(L) + (size_t)slots * RW * tws + (size_t)(L + 2) * RW + (size_t)((2 * L * BNC * F + 3) & ~3) +
                      (size_t)(15 + 4 * nth + (kord - 1) * red_th) * 64 + (size_t)WAVES_PER_BLOCK * 24 + CONVT_FLOATS +
                      (size_t)WAVES_PER_BLOCK * wave_area;
    return fl * sizeof(float);
}

template <int RW, int L, int KIND, int IDX, int NFIX, int PFIX = 0, int KORD = 1>
static int launch_phase_n(const TrainK& k_in, const float* x, const float* prm, const float* gy, const TileGeom& g, int max_grid,
                          hipStream_t stream, int* grid_out) {
    auto kern = stgcn_train_phase_kernel<RW, L, KIND, IDX, NFIX, PFIX, KORD>;
    TrainK k = k_in;
    k.wave_area_floats = wave_area_for(KIND, IDX, g);
    const size_t lds = train_lds_bytes(RW, L, k.wave_area_floats, KORD);
    if (lds > 160 * 1024) return RULGNN_EUNSUPPORTED;
    if (lds > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
            return RULGNN_EHIP;
    }
    int grid = persistent_grid(kern, k.ntiles, lds);
    if (grid > max_grid) grid = max_grid;
    if (grid_out) *grid_out = grid;
    (void)hipGetLastError();   // drop any stale error of the caller's earlier HIP calls
    hipLaunchKernelGGL(kern, dim3(grid), dim3(BLOCK), lds, stream, x, prm, gy, k);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

template <int RW, int L, int KIND, int IDX>
static int launch_phase(const TrainK& k, const float* x, const float* prm, const float* gy, const TileGeom& g, int max_grid,
                        hipStream_t stream, int* grid_out) {
    if (k.K > 1) {
        // MPNN order 2, 3: the theta phases (F_{2l}, G_{2l}) specialised on the order, the others as they are (offsets from k.K)
        if constexpr ((KIND == PH_F || KIND == PH_G) && IDX % 2 == 0) {
            if (k.K == 2) return launch_phase_n<RW, L, KIND, IDX, 0, 0, 2>(k, x, prm, gy, g, max_grid, stream, grid_out);
            if (k.K == 3) return launch_phase_n<RW, L, KIND, IDX, 0, 0, 3>(k, x, prm, gy, g, max_grid, stream, grid_out);
            return RULGNN_EUNSUPPORTED;
        }
        return launch_phase_n<RW, L, KIND, IDX, 0>(k, x, prm, gy, g, max_grid, stream, grid_out);
    }
    if constexpr (RW == 16 && KIND == PH_F && IDX == 0) {
        // F_0 on the f16 matrix cores (stgcn_forward_mx.hip: the eval kernel's front end and the first half of layer 0) where its
        // shape rules hold (num_patch <= 15, 16-byte pieces); the row-mapped fp32 phase kernel below otherwise
        rulgnn_stgcn_shape shp;
        shp.batch = k.B; shp.num_patch = k.N; shp.patch_size = k.P; shp.num_layers = L; shp.mpnn_k = 1;
        const size_t tile_floats = (size_t)F * (64 / RW) * k.N;
        const int rc = stgcn_train_f0_mx(&shp, x, prm, k.cacheX, k.cacheA, k.saved + (size_t)SavedSlot<L>::H(0) * k.ntiles * tile_floats,
                                         k.saved + (size_t)SavedSlot<L>::Z1(0) * k.ntiles * tile_floats, k.cells + cell_fwd(L), cell_stride(L),
                                         CELL_REPLICAS, stream);
        if (rc != RULGNN_EUNSUPPORTED) {
            if (grid_out) *grid_out = 0;
            return rc;
        }
    }
    if constexpr (RW == 16 && L == 2) {
        if constexpr (KIND == PH_F && IDX == 0) {          // the only phase that reads the windows
            if (k.N == 14 && k.P == 30) return launch_phase_n<RW, L, KIND, IDX, 14, 30>(k, x, prm, gy, g, max_grid, stream, grid_out);
            if (k.N == 14 && k.P == 50) return launch_phase_n<RW, L, KIND, IDX, 14, 50>(k, x, prm, gy, g, max_grid, stream, grid_out);
        }
        if (k.N == 14) return launch_phase_n<RW, L, KIND, IDX, 14>(k, x, prm, gy, g, max_grid, stream, grid_out);
    }
    return launch_phase_n<RW, L, KIND, IDX, 0>(k, x, prm, gy, g, max_grid, stream, grid_out);
}

enum TrainMode { TM_FORWARD = 0, TM_BACKWARD = 1, TM_FWDBWD = 2 };

// Synchronised BatchNorm ("per-BN all-reduce of [sum x, sum x^2, count] in forward and the matching [sum dy, sum dy xhat]
// in backward"): after every phase that completes a reduction pair its 16 replicas are collapsed into replica 0 (the others
// zeroed, so that the consumers' replica sum is unchanged) and the caller's all-reduce runs on those 2 F contiguous doubles.
struct SyncHook {
    float bn_param_grad_scale;
    rulgnn_allreduce_f64_fn fn;
    void* user;
};

__global__ void stgcn_cells_collapse_kernel(double* cells, int off, int n, int stride) {
    const int i = threadIdx.x;
    if (i >= n) return;
    double v = 0.0;
    for (int r = 0; r < CELL_REPLICAS; ++r) {          // same fixed order as cell_sum()
        v += cells[r * stride + off + i];
        if (r) cells[r * stride + off + i] = 0.0;
    }
    cells[off + i] = v;
}

template <int L>
static int sync_pair(const TrainK& k, int off, const SyncHook* h, hipStream_t st) {
    if (!h) return RULGNN_OK;
    (void)hipGetLastError();
    hipLaunchKernelGGL(stgcn_cells_collapse_kernel, dim3(1), dim3(64), 0, st, k.cells, off, 2 * F, cell_stride(L));
    if (hipGetLastError() != hipSuccess) return RULGNN_EHIP;
    return h->fn(h->user, k.cells + off, 2 * F, st) == 0 ? RULGNN_OK : RULGNN_ECALLBACK;
}

template <int RW, int L, int I>
struct PhaseChain {
    static int forward_stats(const TrainK& k, const float* x, const float* prm, const TileGeom& lds, int mg, hipStream_t st,
                             const SyncHook* h) {
        if constexpr (I > 0) {
            const int rc = PhaseChain<RW, L, I - 1>::forward_stats(k, x, prm, lds, mg, st, h);
            if (rc != RULGNN_OK) return rc;
        }
        const int rc = launch_phase<RW, L, PH_F, I>(k, x, prm, nullptr, lds, mg, st, nullptr);
        if (rc != RULGNN_OK) return rc;
        return sync_pair<L>(k, cell_fwd(L) + I * 2 * F, h, st);          // sum z, sum z^2 of BatchNorm I
    }
    static int backward(const TrainK& k, const float* x, const float* prm, const float* gy, const TileGeom& lds, int mg, hipStream_t st,
                        int* grids, const SyncHook* h) {
        int rc = launch_phase<RW, L, PH_G, I>(k, x, prm, gy, lds, mg, st, &grids[I]);
        if (rc != RULGNN_OK) return rc;
        if constexpr (I > 0) {
            rc = sync_pair<L>(k, cell_bwd(L) + (I - 1) * 2 * F, h, st);  // G_I leaves sum dy, sum dy xhat of BatchNorm I - 1
            if (rc != RULGNN_OK) return rc;
            return PhaseChain<RW, L, I - 1>::backward(k, x, prm, gy, lds, mg, st, grids, h);
        }
        return RULGNN_OK;
    }
};

template <int L>
static int setup_train(const rulgnn_stgcn_shape* s, const rulgnn_stgcn_train_args* a, int mode, TrainK* kp, WsLayout* wp,
                       TileGeom* ldsp) {
    TileGeom& g = *ldsp;
    int rc = train_geometry(s, &g);
    if (rc != RULGNN_OK) return rc;
    if (g.RW != 16 && L > 2) return RULGNN_EUNSUPPORTED;
    WsLayout& w = *wp;
    ws_layout(s, g, &w);
    if (a->workspace_bytes < w.total) return RULGNN_EWORKSPACE;
    char* ws = static_cast<char*>(a->workspace);
    const int N = s->num_patch;
    TrainK& k = *kp;
    k.cacheX = reinterpret_cast<float*>(ws + w.off_cacheX);
    k.cacheA = reinterpret_cast<float*>(ws + w.off_cacheA);
    k.cells = reinterpret_cast<double*>(ws + w.off_cells);
    k.gpart = reinterpret_cast<float*>(ws + w.off_gpart);
    k.saved = reinterpret_cast<float*>(ws + w.off_saved);
    k.rbuf = reinterpret_cast<float*>(ws + w.off_rbuf);
    k.sbuf = reinterpret_cast<float*>(ws + w.off_sbuf);
    k.write_pred = 1;
    k.pred = a->pred;
    k.B = s->batch; k.ntiles = g.ntiles; k.global_batch = a->global_batch; k.sample_offset = a->sample_offset;
    k.N = N; k.P = s->patch_size; k.Ppad = g.Ppad;
    k.vec4 = g.vec4 && ((reinterpret_cast<uintptr_t>(a->x) & 15) == 0);
    k.stage_floats = g.stage_floats; k.magicP = g.magicP;
    k.do_backward = mode != TM_FORWARD;
    k.has_dpred = a->dpred ? 1 : (a->y ? 0 : 2);
    k.dropout_p = a->dropout_p;
    k.drop_scale = a->dropout_p > 0.f ? 1.0f / (1.0f - a->dropout_p) : 1.0f;
    {
        double thr = (double)a->dropout_p * 4294967296.0;
        thr = thr < 0 ? 0 : thr;
        const uint64_t ti = (uint64_t)(thr + 0.5);
        k.drop_thr = ti > 4294967295ull ? 4294967295u : (uint32_t)ti;
    }
    k.K = s->mpnn_k;
    k.pcount = param_count(N, L, k.K);
    k.wave_area_floats = 0;
    return RULGNN_OK;
}

// Head of every step: clears the reduction cells (what a memset did before) and writes the step scratch -- dropout keys of
// (seed, step) and, for the fused optimizer, Adam's bias corrections.  With a device step state the counters are advanced and
// read there (hipGraph replay), else they come from the arguments.
__device__ __forceinline__ void prepare_body(double* cells, int zero_from, int nzero, int stride, StepScratch* sc, StepState* st,
                                             uint64_t seed, uint64_t step, int L, int new_forward, int has_adam, int64_t adam_step,
                                             float lr, float beta1, float beta2, double bn_count) {
    // the scalar pieces on different wavefronts, beside the zeroing (one lane doing keys, two fp64 pow() and the stores in sequence was
    // most of this kernel's 5 us)
    const int tid = threadIdx.x;
    if (tid == 0) {
        sc->bn_count = bn_count;
        if (new_forward) {                             // status word of the matrix-core chain (stgcn_train_mx.hip), its clean token and
            sc->pad[0] = 0u;                           // the finalize kernel's ticket (stgcn_train_layout.hpp)
            sc->pad[1] = 0u;
            sc->pad[2] = 0u;
        }
    }
    if (new_forward && tid >= 64 && tid < 72) {        // one dropout key per lane
        const int l = tid - 64;
        if (st) {
            if (l == 0) step = ++st->dropout_step;
            step = __shfl(step, 0, 64);
        }
        sc->drop_key[l] = l < L ? dropout_layer_key(seed, step, l) : 0u;
    }
    if (has_adam && tid == 128) {
        if (st) adam_step = ++st->adam_step;
        const double bc1 = 1.0 - pow((double)beta1, (double)adam_step);
        const double bc2 = 1.0 - pow((double)beta2, (double)adam_step);
        sc->lr_over_bc1 = (float)((double)lr / bc1);
        sc->inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    }
    for (int i = tid; i < nzero * CELL_REPLICAS; i += blockDim.x) cells[(i / nzero) * stride + zero_from + i % nzero] = 0.0;
}

__global__ void stgcn_prepare_kernel(double* cells, int zero_from, int nzero, int stride, StepScratch* sc, StepState* st, uint64_t seed,
                                     uint64_t step, int L, int new_forward, int has_adam, int64_t adam_step, float lr, float beta1,
                                     float beta2, double bn_count) {
    prepare_body(cells, zero_from, nzero, stride, sc, st, seed, step, L, new_forward, has_adam, adam_step, lr, beta1, beta2, bn_count);
    if (threadIdx.x == 0) *reinterpret_cast<unsigned*>(reinterpret_cast<char*>(sc) + sizeof(StepScratch)) = 0u;      // step_barrier()
}

// ------------------------------------------------------------------------------------------------
// Small batches: the whole step as ONE launch (RULGNN_STEP_COOP; an experiment kept as an explicit, tested option -- it does NOT pay).
// At the reference protocol's batch size (100; configs/hparams.py:223) the chain is eleven dependent kernels of ~9 us each.  When
// every tile of the batch has its own resident wavefront the same phase bodies run back to back inside one kernel, the BatchNorm
// reductions behind device-side grid barriers instead of kernel boundaries: prepare | F_0 .. F_{2L-1} | TOP | G_{2L-1} .. G_0 |
// finalize (+Adam, running statistics).  Same grid in every stage as the chain uses for such a batch (one tile per wavefront), same
// per-workgroup partial rows, same finalize arithmetic: bit-identical to the chain (tests/test_coop_step_gpu.py).  Measured: 97 us
// against the chain's 89 us at batch 100 -- the ~9 us of a phase are its prologue and one single-wavefront pass, not its launch.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float* coop_smem() {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    return smem;
}

struct CoopK {
    unsigned* barrier;          // one zeroed 32-bit counter in the workspace
    int wa_f0, wa_g, wa_top;    // per-wavefront LDS area of F_0 / the G phases / TOP (wave_area_for)
    StepState* st;              // optional device step state
    uint64_t seed, step;
    int has_adam;
    int64_t adam_step;
    float lr, beta1, beta2;
    double bn_count;
    FinalizeK fin;
};

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned& epoch) {
    __syncthreads();
    ++epoch;
    if (threadIdx.x == 0) {
        __threadfence();
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = epoch * gridDim.x;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
        __threadfence();
    }
    __syncthreads();
}

template <int RW, int L, int NFIX, int PFIX, int I>
struct CoopChain {
    static __device__ __forceinline__ void forward(const float* gx, const float* prm, const TrainK& a, const CoopK& c, unsigned& epoch) {
        if constexpr (I > 0) CoopChain<RW, L, NFIX, PFIX, I - 1>::forward(gx, prm, a, c, epoch);
        TrainK b = a;
        b.wave_area_floats = I == 0 ? c.wa_f0 : 0;
        train_phase_body<RW, L, PH_F, I, NFIX, (I == 0 ? PFIX : 0)>(gx, prm, nullptr, b);
        grid_barrier(c.barrier, epoch);
    }
    static __device__ __forceinline__ void backward(const float* gx, const float* prm, const float* gy, const TrainK& a, const CoopK& c,
                                                    unsigned& epoch) {
        TrainK b = a;
        b.wave_area_floats = c.wa_g;
        train_phase_body<RW, L, PH_G, I, NFIX, 0>(gx, prm, gy, b);
        grid_barrier(c.barrier, epoch);
        if constexpr (I > 0) CoopChain<RW, L, NFIX, PFIX, I - 1>::backward(gx, prm, gy, a, c, epoch);
    }
};

template <int RW, int L, int NFIX, int PFIX>
__global__ __launch_bounds__(BLOCK) void stgcn_train_coop_kernel(const float* __restrict__ gx, const float* __restrict__ prm,
                                                                 const float* __restrict__ gy, TrainK a, CoopK c) {
    unsigned epoch = 0;
    if (blockIdx.x == 0)
        prepare_body(a.cells, 0, cell_stride(L), cell_stride(L), step_scratch(a.cells, L), c.st, c.seed, c.step, L, 1, c.has_adam,
                     c.adam_step, c.lr, c.beta1, c.beta2, c.bn_count);
    grid_barrier(c.barrier, epoch);
    CoopChain<RW, L, NFIX, PFIX, 2 * L - 1>::forward(gx, prm, a, c, epoch);
    {
        TrainK b = a;
        b.wave_area_floats = c.wa_top;
        train_phase_body<RW, L, PH_TOP, 0, NFIX, 0>(gx, prm, gy, b);
        grid_barrier(c.barrier, epoch);
    }
    CoopChain<RW, L, NFIX, PFIX, 2 * L - 1>::backward(gx, prm, gy, a, c, epoch);
    // finalize: the parameter units over the workgroups, then the statistics on workgroup 0
    float(*part)[FIN_COLS] = reinterpret_cast<float(*)[FIN_COLS]>(coop_smem());
    const int units = (c.fin.pcount + FIN_COLS - 1) / FIN_COLS;
    for (int u = blockIdx.x; u < units; u += gridDim.x) finalize_unit<FIN_SLICES / WAVES_PER_BLOCK>(c.fin, u, part, threadIdx.x & 63, threadIdx.x >> 6);
    if (blockIdx.x == 0) finalize_stats(c.fin, threadIdx.x, blockDim.x);
}

// Host side of the cooperative step: applicable when every tile gets its own co-resident wavefront.
template <int RW, int L, int NFIX, int PFIX>
static int launch_coop(const rulgnn_stgcn_shape* s, const rulgnn_stgcn_train_args* a, hipStream_t stream, const TrainK& k,
                       const WsLayout& w, const TileGeom& g, const rulgnn_adam_args* opt, double bn_count) {
    auto kern = stgcn_train_coop_kernel<RW, L, NFIX, PFIX>;
    CoopK c{};
    c.wa_f0 = wave_area_for(PH_F, 0, g);
    c.wa_g = wave_area_for(PH_G, 0, g);
    c.wa_top = wave_area_for(PH_TOP, 0, g);
    int wa = c.wa_f0 > c.wa_g ? c.wa_f0 : c.wa_g;
    wa = wa > c.wa_top ? wa : c.wa_top;
    size_t lds = train_lds_bytes(RW, L, wa);
    if (lds < sizeof(float) * FIN_SLICES * FIN_COLS) lds = sizeof(float) * FIN_SLICES * FIN_COLS;
    if (lds > 160 * 1024) return RULGNN_EUNSUPPORTED;
    const int64_t grid = (k.ntiles + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    if (grid > w.max_grid) return RULGNN_EUNSUPPORTED;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return RULGNN_EHIP;
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return RULGNN_EHIP;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, BLOCK, lds) != hipSuccess || per_cu < 1) return RULGNN_EUNSUPPORTED;
    // every workgroup must be resident for the grid barriers; one workgroup per CU keeps them evenly spread as well
    if (grid > (int64_t)cus) return RULGNN_EUNSUPPORTED;
    const bool fused_adam = opt != nullptr;
    void* adam_state = fused_adam ? opt->step_state : nullptr;
    if (adam_state && a->step_state && adam_state != a->step_state) return RULGNN_EINVAL;
    StepScratch* sc = step_scratch(k.cells, L);
    c.barrier = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(sc) + sizeof(StepScratch));
    c.st = a->step_state ? static_cast<StepState*>(a->step_state) : nullptr;      // as the chain's prepare kernel
    c.seed = a->seed; c.step = a->step;
    c.has_adam = fused_adam ? 1 : 0;
    c.adam_step = fused_adam ? opt->step : 0;
    c.lr = fused_adam ? opt->lr : 0.f; c.beta1 = fused_adam ? opt->beta1 : 0.f; c.beta2 = fused_adam ? opt->beta2 : 0.f;
    c.bn_count = bn_count;
    FinalizeK& f = c.fin;
    f.gpart = k.gpart; f.cells = k.cells;
    f.grads = a->grads; f.loss = a->loss; f.bn_batch = a->bn_batch;
    f.grid_top = (int)grid;
    for (int i = 0; i < 16; ++i) f.grid_g[i] = (int)grid;
    f.N = k.N; f.L = L; f.K = 1; f.pcount = k.pcount; f.B = s->batch; f.global_batch = a->global_batch;
    f.moment_weight = a->bn_moment_weight;
    f.cell_grad_scale = 1.0f;
    f.guard = 0;
    f.fused_opt = fused_adam ? 1 : 0;
    f.params = fused_adam ? opt->params : nullptr; f.exp_avg = fused_adam ? opt->exp_avg : nullptr;
    f.exp_avg_sq = fused_adam ? opt->exp_avg_sq : nullptr; f.bn_running = fused_adam ? opt->bn_stats : nullptr;
    f.beta1 = fused_adam ? opt->beta1 : 0.f; f.beta2 = fused_adam ? opt->beta2 : 0.f; f.eps = fused_adam ? opt->eps : 0.f;
    f.weight_decay = fused_adam ? opt->weight_decay : 0.f; f.bn_momentum = fused_adam ? opt->bn_momentum : 0.f;
    f.write_grads = 1;
    f.write_loss = (k.has_dpred == 0) && a->loss;
    const float* gy = a->dpred ? a->dpred : a->y;
    if (hipMemsetAsync(c.barrier, 0, sizeof(unsigned), stream) != hipSuccess) return RULGNN_EHIP;
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(BLOCK), lds, stream, a->x, a->params, gy, k, c);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

template <int RW, int L>
static int run_train_coop(const rulgnn_stgcn_shape* s, const rulgnn_stgcn_train_args* a, hipStream_t stream, const TrainK& k,
                          const WsLayout& w, const TileGeom& g, const rulgnn_adam_args* opt, double bn_count) {
    if constexpr (RW == 16 && L == 2) {
        if (k.N == 14 && k.P == 30) return launch_coop<RW, L, 14, 30>(s, a, stream, k, w, g, opt, bn_count);
        if (k.N == 14 && k.P == 50) return launch_coop<RW, L, 14, 50>(s, a, stream, k, w, g, opt, bn_count);
    }
    return launch_coop<RW, L, 0, 0>(s, a, stream, k, w, g, opt, bn_count);
}

// ---- the matrix-core chain's view of the workspace (stgcn_train_mx.hip): X_0 tiles in cacheX, packed adjacency tiles in cacheA, X_l in
// the saved slots X(l), TOP's sparse gradient in slot H(0), d(x0 + H) in sbuf, d X_l in rbuf ---------------------------------------------
// The wide chain (stgcn_train_mxw.hip, 16 <= num_patch <= 47) keeps per-SAMPLE records in the same slots (a slot is ntiles x 640 floats in
// either geometry: >= 160 floats per sample, which holds its largest record, 10 x 47 floats, whenever the geometry is one sample per tile).
template <int L>
static MxTrainArgs mx_args(const rulgnn_stgcn_shape* s, const rulgnn_stgcn_train_args* a, const TrainK& k, bool wide = false) {
    MxTrainArgs m;
    const size_t tile_floats = wide ? (size_t)F * 64 : (size_t)F * 4 * k.N;
    m.prm = a->params; m.y = a->y; m.pred = a->pred; m.cells = k.cells; m.gpart = k.gpart;
    m.xrec[0] = k.cacheX;
    m.qrec[0] = nullptr;
    for (int l = 0; l < 3; ++l)
        m.mrec[l] = l < L ? reinterpret_cast<uint32_t*>(k.saved + (size_t)SavedSlot<L>::O0(l) * k.ntiles * tile_floats) : nullptr;
    for (int l = 1; l < 3; ++l) {
        m.xrec[l] = l < L ? k.saved + (size_t)SavedSlot<L>::X(l) * k.ntiles * tile_floats : nullptr;
        m.qrec[l] = l < L ? k.saved + (size_t)SavedSlot<L>::Z2(l - 1) * k.ntiles * tile_floats : nullptr;
    }
    m.arec = k.cacheA;
    m.sb = k.sbuf; m.dx = k.rbuf;
    m.dtop = k.saved + (size_t)SavedSlot<L>::H(0) * k.ntiles * tile_floats;
    m.B = s->batch; m.global_batch = a->global_batch; m.sample_offset = a->sample_offset;
    m.N = k.N; m.L = L; m.pcount = k.pcount;
    m.dropout_p = k.dropout_p; m.drop_scale = k.drop_scale; m.drop_thr = k.drop_thr;
    m.do_backward = k.do_backward;
    return m;
}
// phase numbering of rulgnn_stgcn_train_phase_f32: 0 .. 2L-1 = F_i, 2L = TOP, 2L+1+j = G_{2L-1-j}
template <int L>
static int mx_phase(const rulgnn_stgcn_shape* s, const rulgnn_stgcn_train_args* a, const TrainK& k, const MxTrainArgs& m, int ph,
                    hipStream_t stream, int max_grid, int* grid_out, bool wide = false, const HeadScalars* head = nullptr) {
    if (wide) {
        if (ph == 0) return stgcn_train_mxw_f0(m, a->x, s->patch_size, stream, head);
        if (ph < 2 * L) return stgcn_train_mxw_phase(m, PH_F, ph, stream, max_grid, grid_out);
        if (ph == 2 * L) return stgcn_train_mxw_phase(m, PH_TOP, 0, stream, max_grid, grid_out);
        return stgcn_train_mxw_phase(m, PH_G, 4 * L - ph, stream, max_grid, grid_out);
    }
    if (ph == 0)
        return stgcn_train_f0_mx_packed(s, a->x, a->params, m.xrec[0], m.arec, k.cells + cell_fwd(L), cell_stride(L), CELL_REPLICAS, stream, head);
    if (ph < 2 * L) return stgcn_train_mx_phase(m, PH_F, ph, stream, max_grid, grid_out);
    if (ph == 2 * L) return stgcn_train_mx_phase(m, PH_TOP, 0, stream, max_grid, grid_out);
    return stgcn_train_mx_phase(m, PH_G, 4 * L - ph, stream, max_grid, grid_out);
}

template <int RW, int L>
static int run_train_rw(const rulgnn_stgcn_shape* s, const rulgnn_stgcn_train_args* a, int mode, hipStream_t stream,
                        TrainK& k, WsLayout& w, TileGeom& lds, const rulgnn_adam_args* opt, const SyncHook* hook, int path) {
    int rc = RULGNN_OK;
    const double bn_count = (double)(hook ? a->global_batch : s->batch) * (double)s->num_patch;
    // RULGNN_STEP_AUTO is the chain: measured on MI355X the cooperative launch is SLOWER at every batch it applies to (batch 100:
    // 97 vs 89 us, batch 4096: 275 vs 104 us) -- a phase costs its prologue plus one single-wavefront pass (~2000 instructions at one
    // issue per ~5 cycles), not its launch, and eleven grid barriers cost more than the ten launches they replace
    if (path == RULGNN_STEP_COOP) {
        if (mode != TM_FWDBWD || hook || k.K != 1) return RULGNN_EUNSUPPORTED;
        return run_train_coop<RW, L>(s, a, stream, k, w, lds, opt, bn_count);
    }
    const float* gy = a->dpred ? a->dpred : a->y;
    // The matrix-core chain (stgcn_train_mx.hip): same prepare / cells / finalize, phases that recompute instead of reading saved
    // activations.  Whole MSE steps only (the autograd split and upstream gradients of unknown magnitude stay on the fp32 phases).
    const bool mx_step = path != RULGNN_STEP_CHAIN && mode == TM_FWDBWD && k.has_dpred == 0;
    const int mx_kind = mx_step ? stgcn_train_mx_kind(s, a->x) : 0;                         // the ONE predicate rulgnn_stgcn_train_step_resolve uses
    const bool use_mxw = mx_kind == 2;                                                      // 16 <= num_patch <= 47: the wide chain
    const bool use_mx = mx_kind != 0;
    if ((path == RULGNN_STEP_MX || path == RULGNN_STEP_MX_PERSIST) && !use_mx) return RULGNN_EUNSUPPORTED;

    StepScratch* sc = step_scratch(k.cells, L);
    const bool fused_adam = opt && mode == TM_FWDBWD;
    void* adam_state = fused_adam ? opt->step_state : nullptr;
    if (adam_state && a->step_state && adam_state != a->step_state) return RULGNN_EINVAL;
    StepState* st = static_cast<StepState*>(a->step_state ? a->step_state : adam_state);
    (void)hipGetLastError();
    // RULGNN_TRAIN_WS_CLEAN: the caller vouches that the previous matrix-core step on this workspace was the last thing to touch it -- its
    // finalize kernel left the cells zero, F_0's workgroup 0 writes the head-of-step scalars (and checks the clean token)
    const bool skip_prepare = use_mx && (a->flags & RULGNN_TRAIN_WS_CLEAN) != 0 && !a->step_state && !adam_state && s->batch > 0;
    HeadScalars head{};
    if (skip_prepare) {
        head.sc = sc; head.seed = a->seed; head.step = a->step; head.L = L; head.has_adam = fused_adam ? 1 : 0;
        head.adam_step = fused_adam ? opt->step : 0;
        head.lr = fused_adam ? opt->lr : 0.f; head.beta1 = fused_adam ? opt->beta1 : 0.f; head.beta2 = fused_adam ? opt->beta2 : 0.f;
        head.bn_count = bn_count;
    }
    if (skip_prepare) {
    } else if (mode == TM_FORWARD || mode == TM_FWDBWD) {
        // a new forward: all cells, fresh dropout keys (a backward-only call below reuses the keys of its forward)
        hipLaunchKernelGGL(stgcn_prepare_kernel, dim3(1), dim3(1024), 0, stream, k.cells, 0, cell_stride(L), cell_stride(L), sc,
                           a->step_state ? st : nullptr, a->seed, a->step, L, 1, fused_adam ? 1 : 0, fused_adam ? opt->step : 0,
                           fused_adam ? opt->lr : 0.f, fused_adam ? opt->beta1 : 0.f, fused_adam ? opt->beta2 : 0.f, bn_count);
        if (hipGetLastError() != hipSuccess) return RULGNN_EHIP;
        if (!use_mx) {
            rc = PhaseChain<RW, L, 2 * L - 1>::forward_stats(k, a->x, a->params, lds, w.max_grid, stream, hook);
            if (rc != RULGNN_OK) return rc;
        }
    } else {
        // backward after a separate forward: forward cells are valid, clear the backward ones + loss
        hipLaunchKernelGGL(stgcn_prepare_kernel, dim3(1), dim3(1024), 0, stream, k.cells, cell_bwd(L), cell_stride(L) - cell_bwd(L),
                           cell_stride(L), sc, (StepState*)nullptr,
                           a->seed, a->step, L, 0, 0, (int64_t)0, 0.f, 0.f, 0.f, bn_count);
        if (hipGetLastError() != hipSuccess) return RULGNN_EHIP;
    }
    int grid_top = 0;
    int grids[16] = {0};
    if (use_mx) {
        const MxTrainArgs m = mx_args<L>(s, a, k, use_mxw);
        // RULGNN_STEP_MX_PERSIST, small batches (every workgroup of the phases' grid on a CU of its own): F_1 .. G_0 as ONE launch
        // behind F_0 -- the BatchNorm reductions behind arrival counters instead of kernel boundaries (stgcn_train_mx.hip; measured
        // SLOWER than the launches, hence explicit only).  Not under synchronised BatchNorm: the collectives sit between the phases.
        const bool persist = path == RULGNN_STEP_MX_PERSIST;
        if (persist && (use_mxw || hook || stgcn_train_mx_persistent_grid(s->batch, L, w.max_grid) == 0)) return RULGNN_EUNSUPPORTED;
        for (int ph = 0; ph <= (persist ? 0 : 4 * L); ++ph) {
            int grid = 0;
            rc = mx_phase<L>(s, a, k, m, ph, stream, w.max_grid, &grid, use_mxw, ph == 0 && skip_prepare ? &head : nullptr);
            if (rc != RULGNN_OK) return rc;
            // the reduction pair a phase completes (all-reduced here under synchronised BatchNorm; the later phases read the cells):
            // F_i -> forward pair i, TOP -> backward pair 2L-1, G_i -> backward pair i-1
            if (ph < 2 * L) rc = sync_pair<L>(k, cell_fwd(L) + ph * 2 * F, hook, stream);
            else if (ph == 2 * L) { grid_top = grid; rc = sync_pair<L>(k, cell_bwd(L) + (2 * L - 1) * 2 * F, hook, stream); }
            else {
                const int i = 4 * L - ph;
                grids[i] = grid;
                if (i > 0) rc = sync_pair<L>(k, cell_bwd(L) + (i - 1) * 2 * F, hook, stream);
            }
            if (rc != RULGNN_OK) return rc;
        }
        if (persist) {
            int grid = 0;
            rc = stgcn_train_mx_persistent(m, stream, w.max_grid, &grid);
            if (rc != RULGNN_OK) return rc;
            grid_top = grid;
            for (int i = 0; i < 2 * L; ++i) grids[i] = grid;
        }
    } else {
    rc = launch_phase<RW, L, PH_TOP, 0>(k, a->x, a->params, gy, lds, w.max_grid, stream, &grid_top);
    if (rc != RULGNN_OK) return rc;
    }
    if (mode != TM_FORWARD && !use_mx) {
        rc = sync_pair<L>(k, cell_bwd(L) + (2 * L - 1) * 2 * F, hook, stream);      // TOP leaves the pair of the last BatchNorm
        if (rc != RULGNN_OK) return rc;
        rc = PhaseChain<RW, L, 2 * L - 1>::backward(k, a->x, a->params, gy, lds, w.max_grid, stream, grids, hook);
        if (rc != RULGNN_OK) return rc;
    }
    FinalizeK f{};
    f.gpart = k.gpart; f.cells = k.cells;
    f.grads = a->grads; f.loss = a->loss; f.bn_batch = a->bn_batch;
    f.grid_top = grid_top;
    for (int i = 0; i < 16; ++i) f.grid_g[i] = grids[i];
    f.N = k.N; f.L = L; f.K = k.K; f.pcount = k.pcount; f.B = s->batch; f.global_batch = a->global_batch;
    f.moment_weight = a->bn_moment_weight;
    f.cell_grad_scale = hook ? hook->bn_param_grad_scale : 1.0f;
    f.guard = use_mx ? 1 : 0;
    f.clean = use_mx && mode == TM_FWDBWD ? 1 : 0;
    f.fused_opt = 0;
    f.params = nullptr; f.exp_avg = nullptr; f.exp_avg_sq = nullptr; f.bn_running = nullptr;
    f.beta1 = f.beta2 = f.eps = f.weight_decay = f.bn_momentum = 0.f;
    if (fused_adam) {                      // the bias corrections are in the step scratch (stgcn_prepare_kernel)
        f.fused_opt = 1;
        f.params = opt->params; f.exp_avg = opt->exp_avg; f.exp_avg_sq = opt->exp_avg_sq; f.bn_running = opt->bn_stats;
        f.beta1 = opt->beta1; f.beta2 = opt->beta2; f.eps = opt->eps; f.weight_decay = opt->weight_decay;
        f.bn_momentum = opt->bn_momentum;
    }
    f.write_grads = mode != TM_FORWARD;
    f.write_loss = (k.has_dpred == 0) && a->loss;
    const int fgrid = f.write_grads ? (k.pcount + FIN_COLS - 1) / FIN_COLS : 1;
    (void)hipGetLastError();   // drop any stale error of the caller's earlier HIP calls
    hipLaunchKernelGGL(stgcn_train_finalize_kernel, dim3(fgrid), dim3(FIN_COLS * FIN_SLICES), 0, stream, f);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

template <int L>
static int run_train(const rulgnn_stgcn_shape* s, const rulgnn_stgcn_train_args* a, int mode, hipStream_t stream,
                     const rulgnn_adam_args* opt, const SyncHook* hook, int path) {
    TrainK k;
    WsLayout w;
    TileGeom lds;
    const int rc = setup_train<L>(s, a, mode, &k, &w, &lds);
    if (rc != RULGNN_OK) return rc;
    if (lds.RW == 16) return run_train_rw<16, L>(s, a, mode, stream, k, w, lds, opt, hook, path);
    if constexpr (L <= 2) return run_train_rw<64, L>(s, a, mode, stream, k, w, lds, opt, hook, path);
    return RULGNN_EUNSUPPORTED;
}

// One phase kernel alone (profiling / roofline timing): the reduction cells are NOT cleared, so
// repeated launches keep accumulating into them -- durations are valid, results are not.
template <int RW, int L, int PHASE>
struct SinglePhase {
    static int run(int phase, const TrainK& k, const float* x, const float* prm, const float* gy, const TileGeom& lds, int mg,
                   hipStream_t st) {
        if (phase == PHASE) {
            if constexpr (PHASE < 2 * L) return launch_phase<RW, L, PH_F, PHASE>(k, x, prm, gy, lds, mg, st, nullptr);
            else if constexpr (PHASE == 2 * L) return launch_phase<RW, L, PH_TOP, 0>(k, x, prm, gy, lds, mg, st, nullptr);
            else return launch_phase<RW, L, PH_G, 4 * L - PHASE>(k, x, prm, gy, lds, mg, st, nullptr);
        }
        if constexpr (PHASE > 0) return SinglePhase<RW, L, PHASE - 1>::run(phase, k, x, prm, gy, lds, mg, st);
        return RULGNN_EINVAL;
    }
};

template <int L>
static int run_phase(const rulgnn_stgcn_shape* s, const rulgnn_stgcn_train_args* a, int phase, hipStream_t stream, int path) {
    TrainK k;
    WsLayout w;
    TileGeom lds;
    const int rc = setup_train<L>(s, a, TM_FWDBWD, &k, &w, &lds);
    if (rc != RULGNN_OK) return rc;
    if (phase == -1) {
        // the step's prepare kernel alone: clears the reduction cells (same dropout step) so that a harness timing the phases one by one
        // runs them on valid BatchNorm statistics -- cells that keep accumulating from launch to launch drive the statistics out of range
        (void)hipGetLastError();
        hipLaunchKernelGGL(stgcn_prepare_kernel, dim3(1), dim3(1024), 0, stream, k.cells, 0, cell_stride(L), cell_stride(L), step_scratch(k.cells, L),
                           (StepState*)nullptr, a->seed, a->step, L, 1, 0, (int64_t)0, 0.f, 0.f, 0.f, (double)s->batch * (double)s->num_patch);
        return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
    }
    if (phase < 0 || phase > 4 * L) return RULGNN_EINVAL;
    const float* gy = a->dpred ? a->dpred : a->y;
    if (path != RULGNN_STEP_CHAIN && k.has_dpred == 0 && L <= 2 && stgcn_train_mxw_shape_ok(s, a->x)) {
        const MxTrainArgs m = mx_args<L>(s, a, k, true);
        return mx_phase<L>(s, a, k, m, phase, stream, w.max_grid, nullptr, true);
    }
    if (lds.RW == 16 && path != RULGNN_STEP_CHAIN && k.has_dpred == 0 && stgcn_train_mx_shape_ok(s, a->x)) {
        const MxTrainArgs m = mx_args<L>(s, a, k);
        return mx_phase<L>(s, a, k, m, phase, stream, w.max_grid, nullptr);
    }
    if (path == RULGNN_STEP_MX) return RULGNN_EUNSUPPORTED;
    if (lds.RW == 16) return SinglePhase<16, L, 4 * L>::run(phase, k, a->x, a->params, gy, lds, w.max_grid, stream);
    if constexpr (L <= 2) return SinglePhase<64, L, 4 * L>::run(phase, k, a->x, a->params, gy, lds, w.max_grid, stream);
    return RULGNN_EUNSUPPORTED;
}

int stgcn_train_phase(const rulgnn_stgcn_shape* s, const rulgnn_stgcn_train_args* a, int phase, hipStream_t stream, int path) {
    switch (s->num_layers) {
        case 1: return run_phase<1>(s, a, phase, stream, path);
        case 2: return run_phase<2>(s, a, phase, stream, path);
        case 3: return run_phase<3>(s, a, phase, stream, path);
        default: return RULGNN_EUNSUPPORTED;
    }
}

static int dispatch_train(const rulgnn_stgcn_shape* s, const rulgnn_stgcn_train_args* a, int mode, hipStream_t stream,
                          const rulgnn_adam_args* opt = nullptr, const SyncHook* hook = nullptr, int path = RULGNN_STEP_AUTO) {
    switch (s->num_layers) {
        case 1: return run_train<1>(s, a, mode, stream, opt, hook, path);
        case 2: return run_train<2>(s, a, mode, stream, opt, hook, path);
        case 3: return run_train<3>(s, a, mode, stream, opt, hook, path);
        default: return RULGNN_EUNSUPPORTED;
    }
}

int stgcn_train_step(const rulgnn_stgcn_shape* s, const rulgnn_stgcn_train_args* a, const rulgnn_adam_args* opt,
                     hipStream_t st, int path) {
    return dispatch_train(s, a, TM_FWDBWD, st, opt, nullptr, path);
}

int stgcn_train_forward(const rulgnn_stgcn_shape* s, const rulgnn_stgcn_train_args* a, hipStream_t st) {
    return dispatch_train(s, a, TM_FORWARD, st);
}
int stgcn_train_backward(const rulgnn_stgcn_shape* s, const rulgnn_stgcn_train_args* a, hipStream_t st) {
    return dispatch_train(s, a, TM_BACKWARD, st);
}
int stgcn_train_fwdbwd(const rulgnn_stgcn_shape* s, const rulgnn_stgcn_train_args* a, hipStream_t st) {
    // The split entry (gradients now, the caller's own optimizer call later) stays on the fp32 phases: the matrix-core chain reports an
    // f16 range violation as a NaN loss + untouched state, which only a caller that knows the guard protocol handles
    // (rulgnn_stgcn_train_step_path_f32 with an explicit path is that caller's entry).
    return dispatch_train(s, a, TM_FWDBWD, st, nullptr, nullptr, RULGNN_STEP_CHAIN);
}
int stgcn_train_fwdbwd_syncbn(const rulgnn_stgcn_shape* s, const rulgnn_stgcn_train_args* a, float bn_param_grad_scale,
                              rulgnn_allreduce_f64_fn allreduce, void* user, hipStream_t st, int path) {
    const SyncHook hook{bn_param_grad_scale, allreduce, user};
    return dispatch_train(s, a, TM_FWDBWD, st, nullptr, &hook, path);
}

// Which matrix-core chain a whole MSE step of this shape runs on: 0 none (fp32 phases), 1 the 4-sample-tile chain (num_patch <= 15,
// stgcn_train_mx.hip), 2 the wide chain (16 <= num_patch <= 47, stgcn_train_mxw.hip).  Shared by run_train_rw and
// rulgnn_stgcn_train_step_resolve so that the two dispatch rules cannot drift apart.
int stgcn_train_mx_kind(const rulgnn_stgcn_shape* s, const float* x) {
    TileGeom g;
    if (train_geometry(s, &g) != RULGNN_OK) return 0;
    if (s->num_layers <= 2 && stgcn_train_mxw_shape_ok(s, x)) return 2;
    if (g.RW == 16 && stgcn_train_mx_shape_ok(s, x)) return 1;
    return 0;
}

// Byte offset of the sticky guard counter (StepScratch::pad[3]) inside a training workspace of this shape; -1 where the phase chain
// does not apply (the tiled path).
int64_t stgcn_train_guard_counter_offset(const rulgnn_stgcn_shape* s) {
    TileGeom g;
    if (train_geometry(s, &g) != RULGNN_OK) return -1;
    if (s->num_layers > 3 || (g.RW != 16 && s->num_layers > 2)) return -1;
    WsLayout w;
    ws_layout(s, g, &w);
    return (int64_t)(w.off_cells + sizeof(double) * (size_t)CELL_REPLICAS * cell_stride(s->num_layers) + offsetof(StepScratch, pad) + 3 * sizeof(uint32_t));
}

}  // namespace rulgnn
