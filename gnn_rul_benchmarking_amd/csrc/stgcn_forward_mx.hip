// Matrix-core eval forward of ST_GCN for num_patch <= 15 (the C-MAPSS shapes) on gfx950.
// Reference path replaced: ST_GCN_model.forward under model.eval() -- models/ST_GCN/Model.py:208-222.
//
// Why a second kernel.  The row-mapped kernel (stgcn_forward.hip) is VALU-issue bound: per 4-sample tile it spends
// ~820 cycles per convolution on scalar-operand FMAs (half rate on gfx950) and ~670 per theta projection on DPP FMAs
// (half rate).  Here every contraction of a layer runs on the f16 matrix cores (v_mfma_f32_16x16x32_f16, 16.6 cycles for
// 16 k-MACs x 256 outputs) with fp32 accumulation and 2-way SPLIT operands: a = hi + lo, hi = f16(a), lo = f16(a - hi), and
// a.b ~ hi.hi + hi.lo + lo.hi (+ lo.lo where it is free), measured error 5e-8 of sum|a.b| (tools/probe_mx.hip) -- fp32-class,
// far inside the 1e-4 gate.  (fp32-input MFMAs run at the VALU FMA rate AND block the VALU while they run, measured in
// tools/probe_overlap.hip; bf16 splits are 50x less accurate for the same cost.)
//
// Data layout ("D layout").  One 16x16 MFMA result tile per SAMPLE: column n = lane & 15 is the patch t, row m = 4 (lane >> 4) + r
// is a channel SLOT, r = accumulator register.  The ten statistic channels sit in slots 0,1,2, 4,5,6, 8,9,10, 12
// (chan_slot), so register r = 3 is never used and a [10, N] activation tensor of a sample is THREE registers.  An MFMA
// operand is 8 f16 per lane with k = 8 (lane >> 4) + i: the packed pairs (r0,r1), (r2,-) of a D-layout tensor ARE the
// k-slots 4 (lane >> 4) + r of an operand, so a layer chains through the matrix cores with no data movement at all:
//   T  = X^T-as-A-operand x Adj            (A.X, Model.py:87; rows t, columns c: the transposed tile)       2 MFMAs
//   Hp = T-as-A-operand x theta^T + b      (theta(A.X); rows c, columns j: D layout again)                  2 MFMAs
//   z1 = W1 x [H ; H shifted by 1]         (conv_block1 with BatchNorm folded in; the causal tap is the PACKED operand
//                                           registers of column t - 1, fetched through an LDS tile on the LDS port;
//                                           lanes t < d read a zero slot = the causal padding)                3 MFMAs
//   z2 = W2 x [o0 ; o0 shifted by 2]       (conv_block2, dilation 2)                                        3 MFMAs
// Biases ride in spare k-slots against a constant 1 (theta: k = 15, so num_patch <= 15; conv: slot 3 of lane group 0).
// ReLU is |x| + x = 2 relu(x) (one full-rate op, NaN-preserving where v_max is neither); the powers of two are folded
// into the next weights; LeakyReLU is one fma, (1 + a)/2 folded into theta.  Four samples are in flight per wavefront so that
// dependent MFMAs never wait.
//
// What the SIMD's issue port costs (tools/probe_mixed.hip, round 5; profiles/r05_mfma_valu_overlap.md): an f16 16x16x32 MFMA is 16.3
// cycles with either C form (round 3's "10 with an inline-zero C" came from a probe the compiler had CSE'd); against full-rate fp32 VALU
// work it holds the port for all of them, also against the OTHER wavefront of the SIMD; beside the multi-cycle forms (v_cvt_pk_f16_f32,
// DPP, v_max/min, packed f16: 4 cycles on the lanes, less on the port) about half of it hides.  Plain fp32 ops 2.4.  So the kernel is
// shaped by instruction count, by spreading the MFMAs among the conversions, and by keeping a wavefront's own stalls (LDS round trips,
// LDS-DMA issue, dependent chains) from coinciding with the other wavefront's.
//
// Around the layers: the patch statistics (Model.py:7-52) and the Pearson Gram matrix (Model.py:53-71, f32 4-block MFMAs,
// exact) run in the row mapping of the exact kernel and are converted through the wavefront's LDS tile; the head
// (channel max-pool, fc1, fc2) returns to the row mapping with one 4x4 register/row transpose.
//
// Memory: windows arrive by LDS-DMA (global_load_lds_dwordx4, no VGPRs) into ONE 6.7-KB buffer per wavefront: the patches go to
// registers at the top of a tile and the next tile is requested right behind them, a whole iteration ahead of its use; M0 is
// written once per 4 KB (the immediate offset advances both addresses): an LDS-DMA piece with its own M0 round trip keeps the
// issuing wavefront busy for ~190 cycles, back to back ~85.  4 bytes per sample leave, one tile late (s_waitcnt vmcnt also counts
// stores).  One wavefront per workgroup (wavefronts never synchronise), 12.9 KB of LDS each at 14x30.
//
// Safety net.  f16 overflows at 65504.  Every pointwise step here preserves NaN / Inf, so a sample whose arithmetic
// left the f16 range (or whose input statistics are NaN: constant patch, Model.py:41-52) ends up non-finite; such
// samples are recomputed by the exact fp32 tile routine (stgcn_eval_tile.hpp) inside the same launch, which also
// reproduces the reference's NaN placement.  Inputs scaled to [0, 1] (every dataset the reference wires) never take it.
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "stgcn_eval_tile.hpp"
#include "stgcn_host.hpp"
#include "stgcn_mx.hpp"
#include "stgcn_train_layout.hpp"
#include "stgcn_train_mx.hpp"

namespace rulgnn {
struct MxArgs {
    int64_t B;
    int64_t ntiles;
    int N, P, L;
    int buf_floats;        // the window buffer of a wavefront (the conversion region follows it)
    float* taps;           // debug: raw register dumps of tile 0 (TAPS builds only)
};

// Development aid (variant builds with -DMX_STAGE_CLOCKS only; tools/mx_stage_clocks.py): s_memtime stamps at the stage boundaries of
// a tile, consumed at the END of the iteration (an SMEM result needs lgkmcnt(0): consumed in place it would serialise the LDS round
// trips the stages overlap), summed per wavefront and added to a device array.  The product build compiles every call to nothing.
#ifdef MX_STAGE_CLOCKS
constexpr int MX_CLK_STAMPS = 20;
constexpr int MX_CLK_SLOTS = 4096;          // one row per workgroup (= wavefront): plain stores, no same-address atomics at the end of the kernel
                                            // (20 480 wavefronts adding into 20 addresses cost ~150 us per launch: a first version did)
__device__ unsigned long long g_mx_stage_clocks[MX_CLK_SLOTS][MX_CLK_STAMPS + 2];
struct StageClk {
    unsigned long long t[MX_CLK_STAMPS];
    template <int I> __device__ __forceinline__ void stamp() {
        if (MX_STAGE_CLOCKS == 2 && I != 0 && I != 18) return;      // light form: only the ends of the iteration
        if (MX_STAGE_CLOCKS == 3) return;                           // lifetime form: no stamp inside the loop at all (an outstanding SMEM
                                                                    // result turns every LDS wait of the iteration into lgkmcnt(0))
        __builtin_amdgcn_sched_barrier(0);
        t[I] = __builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);
    }
};
#else
struct StageClk {
    template <int I> __device__ __forceinline__ void stamp() {}
};
#endif

// Per-wavefront constant MFMA operands of one layer (built in the prologue from the flat parameter buffer).
struct LayerOps {
    u32x4 theta_hi, theta_lo;   // B operands of Hp: lane (kg, j): theta[j][4 kg + r] as (hi | hi) and (lo | lo) against the data's
                                //   (hi | lo): all four products in two MFMAs; k = 15 carries the bias
    u32x4 w_hi[2];         // A operands of the convolutions: lane (kg, slot(co)): [tap t | tap t-d] x slot 4 kg + r, BatchNorm
    u32x4 w_lo[2];         //   scale and the 2^k of the ReLU form folded in; slot 3 of lane group 0 carries the BatchNorm shift
};

template <bool TAPS>
__device__ __forceinline__ void tap(float* taps, bool on, int slot, int lane, float v) {
    if constexpr (TAPS) {
        if (on) taps[slot * 64 + lane] = v;
    }
}

// ---- front end of a tile (shared by the eval kernel and the training phase F_0) ------------------------------------------------------
// Patch statistics (row mapping), the request for the next tile's windows, the Pearson adjacency as split B operands, and the statistics
// in the D layout.  `gram` keeps the fp32 adjacency (gram[4 b + r] in lane (g, col) = Adj_b[slot 4 g + r][slot col]).
template <int NFIX, int PFIX, bool TAPS>
__device__ __forceinline__ void mx_front_end(const float* __restrict__ gx, const MxArgs& a, float* tileA, float* cur, int64_t tile, int64_t tstride,
                                             int ns, int N, int P, int lane, bool tapon, float (&X0)[F], f32x16& gram, u32x4 (&adjB)[4],
                                             float (&X)[4][3], StageClk& ck) {
    const int g = lane >> 4, col = lane & 15;
    const int tileNP = N * P;
    // ---- patch statistics, row mapping: lane (sample row g, patch col) ------------------------------------------
    const bool valid = (g < ns) && (col < N);
#pragma unroll
    for (int c = 0; c < F; ++c) X0[c] = 0.f;
    // The next tile is requested as soon as this one has left the LDS: ONE window buffer per wavefront (12.9 KB with the conversion
    // tile at 14x30 -> twelve wavefronts per CU, three per SIMD), and the copy has the rest of the iteration to land.
    auto request_next = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const int64_t nt = tile + tstride;
        if (nt < a.ntiles) {
            const int64_t n0 = nt * 4;
            const int nns = (int)((a.B - n0) < 4 ? (a.B - n0) : 4);
            if constexpr (NFIX != 0 && PFIX != 0) {
                if (nns == 4) dma_tile_fixed<16 * NFIX * PFIX>(gx + n0 * tileNP, tileA, lane);
                else dma_tile(gx + n0 * tileNP, tileA, nns * tileNP * 4, lane);
            } else {
                dma_tile(gx + n0 * tileNP, tileA, nns * tileNP * 4, lane);
            }
        }
    };
    if constexpr (PFIX != 0 && PFIX % 2 == 0 && PFIX <= 64) {
        // every lane loads (padding lanes: a patch that exists; their statistics are never computed): the copy below must run
        // with all 64 lanes enabled -- a lane transfers ITS 16 bytes -- so nothing in front of it may tempt the compiler into
        // one divergent region around the loads, the copy and the arithmetic
        float v[PFIX];
        patch_load<PFIX>(tileA + ((g < ns ? g : 0) * N + (col < N ? col : 0)) * P, v);
        request_next();
        if (valid) patch_statistics_lean<PFIX>(v, X0);

    } else {
        if (valid) patch_statistics(tileA + (g * N + col) * P, P, X0);
        request_next();
    }
#pragma unroll
    for (int c = 0; c < F; ++c) tap<TAPS>(a.taps, tapon, c, lane, X0[c]);

    ck.template stamp<2>();
    // ---- Pearson adjacency: rows = channel slots, one 16x16 block per sample (f32 4-block MFMA, exact) ---------
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < F; ++c) cur[(g * 16 + chan_slot(c)) * PT_STRIDE + col] = X0[c];
    __builtin_amdgcn_wave_barrier();
    gram = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    {
        // lane (sample row g, slot col) reads its channel's patch series; normalised BEFORE the Gram product
        // (dot / (|a| |b|) == (a / |a|) . (b / |b|)); a constant series gives 0 * Inf = NaN like the reference's 0 / 0
        // Lanes whose slot is padding read slot 0's series and scale it by 0 (one select instead of thirty).
        const bool slot_ok = slot_chan(col) >= 0 && g < ns;
        const float4* r4 = reinterpret_cast<const float4*>(cur + (g * 16 + (slot_chan(col) >= 0 ? col : 0)) * PT_STRIDE);
        const float4 q0 = r4[0], q1 = r4[1], q2 = r4[2], q3 = r4[3];
        float CT[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
        float sum = 0.f;                           // columns t >= N were written as zeros
#pragma unroll
        for (int k = 0; k < 16; ++k) sum += CT[k];
        const float mean = sum * (1.0f / (float)N);
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (NFIX ? k < NFIX : true) {
                CT[k] = (NFIX || k < N) ? CT[k] - mean : 0.f;
                ss = fmaf(CT[k], CT[k], ss);
            }
        }
        const float rn = slot_ok ? __builtin_amdgcn_rsqf(ss) : 0.f;      // ss == 0 -> Inf: 0 * Inf = NaN like the reference
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (k < N) {
                const float y = CT[k] * rn;
                gram = __builtin_amdgcn_mfma_f32_16x16x1f32(y, y, gram, 0, 0, 0);
            }
        }
    }
    ck.template stamp<3>();
    // gram[4 b + r] in lane (g, col) = Adj_b[slot 4 g + r][slot col]: the D layout of sample b's adjacency
#pragma unroll
    for (int b = 0; b < 4; ++b) {
#pragma unroll
        for (int r = 0; r < 4; ++r) tap<TAPS>(a.taps, tapon, 10 + 4 * b + r, lane, gram[4 * b + r]);
        const Split2 p01 = split2(gram[4 * b + 0], gram[4 * b + 1]), p23 = split2(gram[4 * b + 2], gram[4 * b + 3]);
        adjB[b] = u32x4{p01.hi, p23.hi, p01.lo, p23.lo};
    }

    // ---- statistics in the D layout: the SAME tile read by (slot row, column) -- rows 13, 14 of a sample's block are the padding
    // rows lane group 3 reads as its registers 1, 2: zeroed once by the caller (mx_zero_padding_rows), never written here
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            X[s][r] = cur[(16 * s + 4 * g + r) * PT_STRIDE + col];
            tap<TAPS>(a.taps, tapon, 26 + 3 * s + r, lane, X[s][r]);
        }
    }
}
__device__ __forceinline__ void mx_zero_padding_rows(float* cur, int lane) {
    for (int e = lane; e < 4 * 2 * PT_STRIDE; e += 64) cur[((e / (2 * PT_STRIDE)) * 16 + 13) * PT_STRIDE + e % (2 * PT_STRIDE)] = 0.f;
}

template <int LFIX, int NFIX, int PFIX, bool TAPS>
__global__ __launch_bounds__(64, MX_WAVES_PER_SIMD) void stgcn_forward_mx_kernel(const float* __restrict__ gx, const float* __restrict__ prm,
                                                                 const float* __restrict__ bn, float* __restrict__ out, MxArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int L = LFIX;
    const int N = NFIX ? NFIX : a.N, P = PFIX ? PFIX : a.P;
    const int LS = layer_stride(N);
    const int lane = threadIdx.x;
    const int g = lane >> 4, col = lane & 15;            // D layout: row group / column; row mapping: sample row / patch
    const int tileNP = N * P;                            // floats per sample

    int64_t tile = blockIdx.x;
    if (tile >= a.ntiles) return;
#ifdef MX_STAGE_CLOCKS
    const unsigned long long clk_entry = __builtin_readcyclecounter();
    const unsigned long long wall_entry = wall_clock64();          // the constant 100 MHz counter: shader ticks / wall ticks = the clock
#endif
    {   // first tile on its way before the weights are touched
        const int64_t s0 = tile * 4;
        const int ns = (int)((a.B - s0) < 4 ? (a.B - s0) : 4);
        dma_tile(gx + s0 * tileNP, smem, ns * tileNP * 4, lane);
    }

    // ---- prologue: constant operands -------------------------------------------------------------------------------
    // Every load is unconditional (index clamped to 0 where the slot is padding, value selected afterwards): ~80 loads per lane go
    // out back to back and the wavefront pays ONE memory latency; as `cond ? load : 0` they compiled to a branch per load.
    LayerOps ops[L];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const float* lp = prm + l * LS;
        {   // theta^T as B operand: column j = col, k-slot 4 g + r <-> patch k; k = 15 <-> bias (its A side is the constant 1)
            float w[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 4 * g + r;
                const bool ok = col < N && (k < N || k == 15);
                const int idx = k == 15 ? off_theta_b(N) + col : off_theta_w(N) + col * N + k;
                const float v = lp[ok ? idx : 0];
                w[r] = ok ? v * (0.5f * (1.f + LEAKY)) : 0.f;      // Hp arrives as (1 + a)/2 Hp: leaky() is then ONE fma
            }
            const Split2 p01 = split2(w[0], w[1]), p23 = split2(w[2], w[3]);
            ops[l].theta_hi = u32x4{p01.hi, p23.hi, p01.hi, p23.hi};
            ops[l].theta_lo = u32x4{p01.lo, p23.lo, p01.lo, p23.lo};
        }
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            // A operand: row m = col <-> output channel slot_chan(col); k-slots [0..3] = tap at t, [4..7] = tap at t - d, each x input
            // slot 4 g + r.  Input of conv_block2 arrives as 4 o0 (two ReLUs in 2 relu form): fold 1/4.
            const int co = slot_chan(col), coc = co >= 0 ? co : 0;
            const float mean = bn[((l * 2 + blk) * 2 + 0) * F + coc], var = bn[((l * 2 + blk) * 2 + 1) * F + coc];
            const float gam = lp[off_bn_g(N, blk) + coc], bet = lp[off_bn_b(N, blk) + coc];
            const float sc = gam / sqrtf(var + BN_EPS);
            const float shift = bet - mean * sc;
            const float wsc = (blk == 0 ? 1.f : 0.25f) * sc;
            float wc[4], wd[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = slot_chan(4 * g + r);
                const bool ok = co >= 0 && ci >= 0;
                const float2 taps2 = *reinterpret_cast<const float2*>(lp + off_conv_w(N, blk) + (coc * F + (ci >= 0 ? ci : 0)) * 2);
                wc[r] = ok ? taps2.y * wsc : 0.f;     // tap at t
                wd[r] = ok ? taps2.x * wsc : 0.f;     // tap at t - d
            }
            // x the constant 1 that rides in slot 3 of the data operand.  (A select, not `if (g == 0)`: mean and beta only feed `shift`,
            // and under the `if` the compiler sank their two loads into the branch behind a wait of their own -- four extra dependent
            // memory round trips in front of the first tile.)
            wc[3] = g == 0 ? (co >= 0 ? shift : 0.f) : wc[3];
            const Split2 c01 = split2(wc[0], wc[1]), c23 = split2(wc[2], wc[3]), d01 = split2(wd[0], wd[1]), d23 = split2(wd[2], wd[3]);
            ops[l].w_hi[blk] = u32x4{c01.hi, c23.hi, d01.hi, d23.hi};
            ops[l].w_lo[blk] = u32x4{c01.lo, c23.lo, d01.lo, d23.lo};
        }
    }
    // head, row mapping: lane (sample row, t) holds row t of fc1
    float fc1w[16];
    const int colc = col < N ? col : 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const float v = prm[off_fc1_w(N, L) + colc * N + (k < N ? k : 0)];
        fc1w[k] = (col < N && k < N) ? v : 0.f;
    }
    const float fc1b_raw = prm[off_fc1_b(N, L) + colc], fc2w_raw = prm[off_fc2_w(N, L) + colc];
    const float fc1b = col < N ? fc1b_raw : 0.f;
    const float fc2w_half = col < N ? 0.5f * fc2w_raw : 0.f;                         // fc1's ReLU arrives as 2 relu
    const float fc2b = prm[off_fc2_b(N, L)];
    // row t = 15 of T is 1 (the k = 15 slot of theta^T is the bias): OR-ed into the hi operand of T (column 15 of X is zero, so
    // the matrix cores leave an exact 0 there): no separate bias add, every MFMA chain starts from C = 0
    const unsigned t_bias = g == 3 ? 0x3C00u << 16 : 0u;
    const float half_ok = col < N ? 0.5f : 0.f, quarter_ok = col < N ? 0.25f : 0.f;  // residual scales; padded columns stay 0

    const int sh_rd1 = col >= 1 ? lane - 1 : 64, sh_rd2 = col >= 2 ? lane - 2 : 64;     // shift tile: column t - d, or the zero slot
    int sh_rd1_lo = sh_rd1 + 65, sh_rd2_lo = sh_rd2 + 65;
    asm volatile("" : "+v"(sh_rd1_lo), "+v"(sh_rd2_lo));

    mx_zero_padding_rows(smem + a.buf_floats, lane);
    bool any_bad = false, pend_mine = false;
    int64_t pend_idx = 0;
    float pend_pred = 0.f;
#ifdef MX_STAGE_CLOCKS
    unsigned clk_acc[MX_CLK_STAMPS] = {};
    unsigned clk_tiles = 0;
#endif
    for (int it = 0; tile < a.ntiles; ++it, tile += gridDim.x) {
        StageClk ck;
        ck.template stamp<0>();
        float* const tileA = smem;                       // this tile's windows (LDS-DMA target); free again once the patches are in registers
        float* const cur = smem + a.buf_floats;          // layout-conversion tile + shift tile (plain arithmetic on the __shared__ base
                                                         // keeps these LDS, not flat, accesses)
        const int64_t s0 = tile * 4;
        const int ns = (int)((a.B - s0) < 4 ? (a.B - s0) : 4);
        const bool tapon = TAPS && a.taps != nullptr && tile == 0;
        // this tile's windows have landed; every LDS access of the previous tile has retired
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // The previous tile's predictions leave HERE, not at the end of their own iteration: the wait above also counts stores, and a
        // store issued just in front of it kept the wavefront parked for the write acknowledgement (~1600 cycles per tile, s_memtime).
        if (pend_mine) out[pend_idx] = pend_pred;
        ck.template stamp<1>();

        u32x2* sh_tile = reinterpret_cast<u32x2*>(cur + MX_MIN_BUF_BYTES / 4);      // behind the layout-conversion tile

        const bool valid = (g < ns) && (col < N);
        float X0[F], X[4][3];
        f32x16 gram;
        u32x4 adjB[4];
        mx_front_end<NFIX, PFIX, TAPS>(gx, a, tileA, cur, tile, gridDim.x, ns, N, P, lane, tapon, X0, gram, adjB, X, ck);
        ck.template stamp<4>();
        if (lane < 2) sh_tile[64 + 65 * lane] = u32x2{0u, 0u};      // the padding slot of the shift tile (hi and lo halves)

        // The layers are the dense part of a tile; statistics, Pearson and the head are chains of LDS round trips and dependent
        // instructions.  The wavefront that is in a sparse stage goes first whenever it has something to issue; the other one fills the
        // gaps from its dense stage (s_setprio 1 / 0: 665 -> 648 us at 1M samples).
        __builtin_amdgcn_s_setprio(0);
        // ---- the layers, four samples in flight ---------------------------------------------------------------------
        // Scheduling barriers between the five stages of a layer: inside a stage the samples follow one another (operands of
        // sample s, then its MFMA chain), so the results of sample 0 are three samples old when the next stage reads them.
        // Left alone the scheduler groups the whole layer by sample and pays an s_nop 7 after every MFMA chain.
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const int tb = 38 + MX_TAPS_PER_LAYER * l;
            f32x4 T[4], Hp[4], z[4];
            __builtin_amdgcn_sched_barrier(0);
            // T = (A.X)^T : rows t, columns c
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const Split2 p01 = split2(X[s][0], X[s][1]), p2 = split2(X[s][2], 0.f);
                const u32x4 ah = {p01.hi, p2.hi, p01.hi, p2.hi}, al = {p01.lo, p2.lo, p01.lo, p2.lo};
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                T[s] = mfma16(ah, adjB[s], zero);
                T[s] = mfma16(al, adjB[s], T[s]);
            }
            if (l == 0) ck.template stamp<5>(); else ck.template stamp<11>();
            __builtin_amdgcn_sched_barrier(0);
            // Hp = theta(A.X) + b : rows c, columns j
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int r = 0; r < 4; ++r) tap<TAPS>(a.taps, tapon, tb + 4 * s + r, lane, T[s][r]);
                const Split2 p01 = split2(T[s][0], T[s][1]), p23 = split2(T[s][2], T[s][3]);
                const u32x4 ta = {p01.hi, p23.hi | t_bias, p01.lo, p23.lo};
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                Hp[s] = mfma16(ta, ops[l].theta_hi, zero);
                Hp[s] = mfma16(ta, ops[l].theta_lo, Hp[s]);
            }
            float H[4][3], V[4][3];
            u32x4 bh[4], bl[4];
            if (l == 0) ck.template stamp<6>(); else ck.template stamp<12>();
            __builtin_amdgcn_sched_barrier(0);
            // conv_block1 on H = leaky(Hp)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int r = 0; r < 4; ++r) tap<TAPS>(a.taps, tapon, tb + 16 + 4 * s + r, lane, Hp[s][r] * (2.f / (1.f + LEAKY)));
                keep_until_here(Hp[s][3]);
                // Hp carries the factor (1 + a)/2 (folded into theta): leaky(x) = x' + (1 - a)/(1 + a) |x'|, one full-rate fma; NaN stays NaN
#pragma unroll
                for (int r = 0; r < 3; ++r) H[s][r] = fmaf((1.f - LEAKY) / (1.f + LEAKY), __builtin_fabsf(Hp[s][r]), Hp[s][r]);
                // slot 3 = 1: the BatchNorm shift's partner rides through the split (its lo part is 0)
                const Split2 p01 = split2(H[s][0], H[s][1]), p2 = split2(H[s][2], 1.0f);
                const Shifted prev = shift_columns(sh_tile, sh_rd1, sh_rd1_lo, lane, u32x2{p01.hi, p2.hi}, u32x2{p01.lo, p2.lo});   // column t - 1
                bh[s] = u32x4{p01.hi, p2.hi, prev.hi.x, prev.hi.y};
                bl[s] = u32x4{p01.lo, p2.lo, prev.lo.x, prev.lo.y};
                __builtin_amdgcn_sched_barrier(0);          // keep each sample's LDS round trip behind ITS arithmetic, not behind all four
            }
            // the shifted columns of all four samples are in flight through the LDS before the first product needs one
            if (l == 0) ck.template stamp<7>(); else ck.template stamp<13>();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                z[s] = mfma16(ops[l].w_hi[0], bh[s], zero);
                z[s] = mfma16(ops[l].w_hi[0], bl[s], z[s]);
                z[s] = mfma16(ops[l].w_lo[0], bh[s], z[s]);
            }
            if (l == 0) ck.template stamp<8>(); else ck.template stamp<14>();
            __builtin_amdgcn_sched_barrier(0);
            // o0 = relu(relu(z1) + H), carried as V = 4 o0;  conv_block2 (dilation 2)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int r = 0; r < 4; ++r) tap<TAPS>(a.taps, tapon, tb + 32 + 4 * s + r, lane, z[s][r]);
                keep_until_here(z[s][3]);
#pragma unroll
                for (int r = 0; r < 3; ++r) V[s][r] = relu2(fmaf(2.f, H[s][r], relu2(z[s][r])));
#pragma unroll
                for (int r = 0; r < 3; ++r) tap<TAPS>(a.taps, tapon, tb + 48 + 3 * s + r, lane, V[s][r]);
                const Split2 p01 = split2(V[s][0], V[s][1]), p2 = split2(V[s][2], 1.0f);
                const Shifted prev = shift_columns(sh_tile, sh_rd2, sh_rd2_lo, lane, u32x2{p01.hi, p2.hi}, u32x2{p01.lo, p2.lo});   // column t - 2
                bh[s] = u32x4{p01.hi, p2.hi, prev.hi.x, prev.hi.y};
                bl[s] = u32x4{p01.lo, p2.lo, prev.lo.x, prev.lo.y};
                __builtin_amdgcn_sched_barrier(0);          // keep each sample's LDS round trip behind ITS arithmetic, not behind all four
            }
            if (l == 0) ck.template stamp<9>(); else ck.template stamp<15>();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                z[s] = mfma16(ops[l].w_hi[1], bh[s], zero);
                z[s] = mfma16(ops[l].w_hi[1], bl[s], z[s]);
                z[s] = mfma16(ops[l].w_lo[1], bh[s], z[s]);
            }
            if (l == 0) ck.template stamp<10>(); else ck.template stamp<16>();
            __builtin_amdgcn_sched_barrier(0);
            // o1 = relu(z2) + o0 (both >= 0: the outer ReLU is the identity); out = dropout_eval(o1) + X
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int r = 0; r < 4; ++r) tap<TAPS>(a.taps, tapon, tb + 60 + 4 * s + r, lane, z[s][r]);
                keep_until_here(z[s][3]);
#pragma unroll
                for (int r = 0; r < 3; ++r) X[s][r] = fmaf(half_ok, relu2(z[s][r]), fmaf(quarter_ok, V[s][r], X[s][r]));
#pragma unroll
                for (int r = 0; r < 3; ++r) tap<TAPS>(a.taps, tapon, tb + 76 + 3 * s + r, lane, X[s][r]);
            }
        }

        ck.template stamp<17>();
        __builtin_amdgcn_s_setprio(1);
        // ---- head: max over the ten channels (Model.py:218-219), fc1, fc2 -------------------------------------------
        float pm[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            // lane group 3 holds one channel (slot 12) in r = 0; its r = 1, 2 are padding rows
            const float x1 = g == 3 ? X[s][0] : X[s][1], x2 = g == 3 ? X[s][0] : X[s][2];
            const float m = vmax3(X[s][0], x1, x2);
            pm[s] = fmaf(X[s][0] + x1 + x2, 0.f, m);                 // v_max drops NaN: put it (and Inf) back
        }
        transpose_rows4(pm[0], pm[1], pm[2], pm[3]);                  // in: register = sample, row = group; out: register = group, row = sample
        float pooled = vmax(vmax3(pm[0], pm[1], pm[2]), pm[3]);
        pooled = fmaf((pm[0] + pm[1]) + (pm[2] + pm[3]), 0.f, pooled);
        pooled = valid ? pooled : 0.f;
        tap<TAPS>(a.taps, tapon, 38 + MX_TAPS_PER_LAYER * MX_MAX_LAYERS, lane, pooled);
        float y1 = fc1b;
        fmac1_rowbcast<0>(y1, pooled, fc1w[0]);   fmac1_rowbcast<1>(y1, pooled, fc1w[1]);   fmac1_rowbcast<2>(y1, pooled, fc1w[2]);
        fmac1_rowbcast<3>(y1, pooled, fc1w[3]);   fmac1_rowbcast<4>(y1, pooled, fc1w[4]);   fmac1_rowbcast<5>(y1, pooled, fc1w[5]);
        fmac1_rowbcast<6>(y1, pooled, fc1w[6]);   fmac1_rowbcast<7>(y1, pooled, fc1w[7]);   fmac1_rowbcast<8>(y1, pooled, fc1w[8]);
        fmac1_rowbcast<9>(y1, pooled, fc1w[9]);   fmac1_rowbcast<10>(y1, pooled, fc1w[10]); fmac1_rowbcast<11>(y1, pooled, fc1w[11]);
        fmac1_rowbcast<12>(y1, pooled, fc1w[12]); fmac1_rowbcast<13>(y1, pooled, fc1w[13]); fmac1_rowbcast<14>(y1, pooled, fc1w[14]);
        fmac1_rowbcast<15>(y1, pooled, fc1w[15]);
        const float pred = Row<16>::allsum(relu2(y1) * fc2w_half) + fc2b;
        tap<TAPS>(a.taps, tapon, 38 + MX_TAPS_PER_LAYER * MX_MAX_LAYERS + 1, lane, pred);
        const bool mine = col == 0 && g < ns;
        pend_mine = mine; pend_idx = s0 + g; pend_pred = pred;
        any_bad |= __any(mine && !(__builtin_fabsf(pred) <= 3.0e38f)) != 0;
#ifdef MX_STAGE_CLOCKS
        ck.template stamp<18>();
        if constexpr (L == 2 && MX_STAGE_CLOCKS != 3) {
            if (MX_STAGE_CLOCKS == 2) {
                clk_acc[0] += (unsigned)(ck.t[18] - ck.t[0]);
            } else {
#pragma unroll
                for (int i = 0; i < 18; ++i) {
                    clk_acc[i] += (unsigned)(ck.t[i + 1] - ck.t[i]);
                    asm volatile("" : "+v"(clk_acc[i]));
                }
            }
            ++clk_tiles;
        }
#endif
    }
#ifdef MX_STAGE_CLOCKS
    if (lane == 0 && blockIdx.x < MX_CLK_SLOTS) {
        // [0..17] stage sums, [18] wavefronts, [19] entry-to-here ticks, [20] tile passes; accumulated over the launches since the last reset
        unsigned long long* row = g_mx_stage_clocks[blockIdx.x];
        for (int i = 0; i < 18; ++i) row[i] += (unsigned long long)clk_acc[i];
        row[18] += 1ull;
        row[19] += __builtin_readcyclecounter() - clk_entry;
        row[MX_CLK_STAMPS] += (unsigned long long)clk_tiles;
        row[MX_CLK_STAMPS + 1] += wall_clock64() - wall_entry;
    }
#endif

    if (pend_mine) out[pend_idx] = pend_pred;
    // ---- safety net: recompute non-finite samples with the exact fp32 tile routine ----------------------------------
    if (any_bad) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // Cold path: hide the parameter pointers' uniformity so that its ~800 conv weights arrive through per-lane loads instead
        // of scalar loads -- the scalar form makes the whole kernel spill SGPRs (106 live at once here), which the hot loop
        // would pay for in v_readlane traffic.
        const float* prm_cold = prm;
        const float* bn_cold = bn;
        asm volatile("" : "+v"(prm_cold), "+v"(bn_cold));
        EvalWeightsLds<16> w;
        w.bind(smem + a.buf_floats, L);
        eval_weights_fill<16>(w, prm_cold, bn_cold, N, L, lane, 64);
        __builtin_amdgcn_wave_barrier();
        for (tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
            const int64_t s0 = tile * 4;
            const int ns = (int)((a.B - s0) < 4 ? (a.B - s0) : 4);
            const bool mine = col == 0 && g < ns;
            const float have = mine ? __builtin_nontemporal_load(out + s0 + g) : 0.f;
            if (!__any(!(__builtin_fabsf(have) <= 3.0e38f))) continue;
            __builtin_amdgcn_wave_barrier();
            stage_tile(gx + s0 * tileNP, smem, ns * tileNP, P, P, 0u, true, lane);
            __builtin_amdgcn_wave_barrier();
            const float pred = eval_tile_valu<16>(smem, ns, N, P, P, L, w, prm_cold, lane);
            // only the samples that need it: the others keep the matrix-core result, bit for bit independent of their tile mates
            if (mine && !(__builtin_fabsf(have) <= 3.0e38f)) out[s0 + g] = pred;
        }
    }
}

// =====================================================================================================================
// Training phase F_0 on the same arithmetic (stgcn_train.hip: "forward up to the input of BatchNorm 0").
// Reference path: ST_GCN_model.forward under model.train() up to conv_block1 of layer 0 -- models/ST_GCN/Model.py:208-216,
// :74-90 (MPNN_mk), :134-146 (conv_block1).  Writes what the later phases read (the packed [10][4 N] tiles of stgcn_train.hip):
// the statistics X0, the Pearson adjacency, H = leaky(theta(A X0)), z1 = conv_block1(H), and adds sum z1, sum z1^2 per channel
// to the BatchNorm-0 reduction cells (fp64 atomics, replica blockIdx % 16).
//
// No safety net is needed here: up to z1 the layer is positively homogeneous in (X0, bias), so every sample is scaled by a power of
// two 2^-k that brings its largest statistic below 128 (the bias partner of theta becomes 2^-k, an exact f16 down to k = 24) and
// H, z1 are scaled back exactly: statistics up to ~1e9 stay inside the f16 range, NaN (constant patch) stays NaN.
constexpr int MXF0_WAVES = 4;
struct MxF0Out {
    float* cacheX;         // [ntiles][10][4 N]
    float* cacheA;         // [ntiles][10][40]
    float* H0;             // [ntiles][10][4 N]   SavedSlot::H(0)
    float* Z1;             // [ntiles][10][4 N]   SavedSlot::Z1(0)
    double* cells;         // replica 0 of the BatchNorm-0 forward pair: [sum z (10) | sum z^2 (10)]
    int cell_stride;       // doubles from one replica to the next
    int replicas;
};

// PACKED (the matrix-core chain of stgcn_train_mx.hip): the adjacency leaves as its 55 unique entries per sample ([tile][4][55], 220
// bytes instead of the 400-byte lane layout of the row-mapped phases) and H, z1 are not written at all -- the later phases recompute.
template <int NFIX, int PFIX, bool PACKED>
__global__ __launch_bounds__(64 * MXF0_WAVES, MX_WAVES_PER_SIMD) void stgcn_train_f0_mx_kernel(const float* __restrict__ gx, const float* __restrict__ prm,
                                                                                             MxArgs a, MxF0Out o, HeadScalars hs) {
    if (PACKED && hs.sc != nullptr && blockIdx.x == 0) head_scalars(hs, threadIdx.x);       // a step without its prepare launch
    // Four wavefronts per workgroup, each on its own tiles with its own LDS region; they meet once, in the epilogue, where their BatchNorm
    // sums are combined in LDS: one fp64 atomic per channel and WORKGROUP (2048 single-wavefront workgroups adding into 16 replicas of
    // the same two lines cost ~4 us of serialised atomics at the end of the kernel).
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    const int N = NFIX ? NFIX : a.N, P = PFIX ? PFIX : a.P;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, col = lane & 15;
    const int tileNP = N * P;
    double* const pairbuf = reinterpret_cast<double*>(smem_all);                       // [MXF0_WAVES][2 F]
    float* const smem = smem_all + 2 * MXF0_WAVES * 2 * F + wave * (a.buf_floats + (MX_MIN_BUF_BYTES + MX_SHIFT_TILE_BYTES) / 4);

    int64_t tile = (int64_t)blockIdx.x * MXF0_WAVES + wave;
    const int64_t tstride = (int64_t)gridDim.x * MXF0_WAVES;
    if (tile < a.ntiles) {
        const int64_t s0 = tile * 4;
        const int ns = (int)((a.B - s0) < 4 ? (a.B - s0) : 4);
        dma_tile(gx + s0 * tileNP, smem, ns * tileNP * 4, lane);
    }

    // ---- prologue: theta^T (leaky's (1 + a)/2 folded in) and the RAW conv_block1 weights of layer 0 as split operands ----------------
    u32x4 theta_hi, theta_lo, w_hi, w_lo;
    {
        float w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 4 * g + r;
            const bool ok = col < N && (k < N || k == 15);
            const int idx = k == 15 ? off_theta_b(N) + col : off_theta_w(N) + col * N + k;
            const float v = prm[ok ? idx : 0];
            w[r] = ok ? v * (0.5f * (1.f + LEAKY)) : 0.f;
        }
        const Split2 p01 = split2(w[0], w[1]), p23 = split2(w[2], w[3]);
        theta_hi = u32x4{p01.hi, p23.hi, p01.hi, p23.hi};
        theta_lo = u32x4{p01.lo, p23.lo, p01.lo, p23.lo};
        const int co = slot_chan(col), coc = co >= 0 ? co : 0;
        float wc[4], wd[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = slot_chan(4 * g + r);
            const bool ok = co >= 0 && ci >= 0;
            const float2 taps2 = *reinterpret_cast<const float2*>(prm + off_conv_w(N, 0) + (coc * F + (ci >= 0 ? ci : 0)) * 2);
            wc[r] = ok ? taps2.y : 0.f;
            wd[r] = ok ? taps2.x : 0.f;
        }
        const Split2 c01 = split2(wc[0], wc[1]), c23 = split2(wc[2], wc[3]), d01 = split2(wd[0], wd[1]), d23 = split2(wd[2], wd[3]);
        w_hi = u32x4{c01.hi, c23.hi, d01.hi, d23.hi};
        w_lo = u32x4{c01.lo, c23.lo, d01.lo, d23.lo};
    }
    const int sh_rd1 = col >= 1 ? lane - 1 : 64;
    int sh_rd1_lo = sh_rd1 + 65;
    asm volatile("" : "+v"(sh_rd1_lo));

    // where this lane's values go in a packed [10][4 N] tile: row mapping (sample g, patch col), and D layout (channel of slot 4 g + r)
    const int pitch = 4 * N, tile_floats = F * pitch;
    const bool col_ok = col < N;
    int d_off[3];
    bool d_ok[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int c = slot_chan(4 * g + r);
        d_ok[r] = c >= 0 && col_ok;
        d_off[r] = (c >= 0 ? c : 0) * pitch + col;
    }
    const int ca_col = slot_chan(col);                    // adjacency: gram[4 b + r] of lane (g, col) = A_b[slot 4 g + r][slot col]
    float sa[3] = {0.f, 0.f, 0.f}, sb[3] = {0.f, 0.f, 0.f};
    mx_zero_padding_rows(smem + a.buf_floats, lane);

    for (int it = 0; tile < a.ntiles; ++it, tile += tstride) {
        float* const tileA = smem;
        float* const cur = smem + a.buf_floats;
        const int64_t s0 = tile * 4;
        const int ns = (int)((a.B - s0) < 4 ? (a.B - s0) : 4);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        u32x2* sh_tile = reinterpret_cast<u32x2*>(cur + MX_MIN_BUF_BYTES / 4);

        float X0[F], X[4][3];
        f32x16 gram;
        u32x4 adjB[4];
        StageClk ck;
        mx_front_end<NFIX, PFIX, false>(gx, a, tileA, cur, tile, tstride, ns, N, P, lane, false, X0, gram, adjB, X, ck);
        if (lane < 2) sh_tile[64 + 65 * lane] = u32x2{0u, 0u};

        // ---- what the later phases read of the inputs: statistics (row mapping) and adjacency -----------------------------
        if (g < ns && col_ok) {
            float* px = o.cacheX + tile * (int64_t)tile_floats + g * N + col;
#pragma unroll
            for (int c = 0; c < F; ++c) __builtin_nontemporal_store(X0[c], px + c * pitch);
        }
        if constexpr (PACKED) {
            if (ca_col >= 0) {
                float* pa = o.cacheA + tile * (int64_t)(4 * NPAIR);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const int c = slot_chan(4 * g + r);
                        // the entry as its (hi | lo << 16) f16 pair, as the front end split it: the later phases permute bytes instead of splitting
                        const unsigned ph = r < 2 ? adjB[b][0] : adjB[b][1], pl = r < 2 ? adjB[b][2] : adjB[b][3];
                        const unsigned pair = (r & 1) ? ((ph >> 16) | (pl & 0xFFFF0000u)) : ((ph & 0xFFFFu) | (pl << 16));
                        if (c >= 0 && c <= ca_col) pa[b * NPAIR + sym(c, ca_col)] = __builtin_bit_cast(float, pair);
                    }
                }
            }
        } else if (ca_col >= 0) {
            float* pa = o.cacheA + tile * (int64_t)(F * 40) + ca_col;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int c = slot_chan(4 * g + r);
                    if (c >= 0) pa[c * 40 + b * 10] = gram[4 * b + r];
                }
            }
        }

        // ---- per-sample power-of-two scale: the largest |statistic| of the sample below 128 ------------------------------------
        float m = 0.f;
#pragma unroll
        for (int c = 0; c < F; ++c) m = fmaxf(m, __builtin_fabsf(X0[c]));
        m = fmaxf(m, dpp<DPP_QUAD_XOR1>(m));
        m = fmaxf(m, dpp<DPP_QUAD_XOR2>(m));
        m = fmaxf(m, dpp<DPP_ROW_HALF_MIRROR>(m));
        m = fmaxf(m, dpp<DPP_ROW_MIRROR>(m));
        int kexp = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xFFu) - (127 + 6);
        kexp = kexp < 0 ? 0 : (kexp > 24 ? 24 : kexp);
        float sc[4], isc[4];
        unsigned t_bias[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = __builtin_amdgcn_readlane(kexp, 16 * s);
            sc[s] = __builtin_bit_cast(float, (unsigned)(127 - k) << 23);
            isc[s] = __builtin_bit_cast(float, (unsigned)(127 + k) << 23);
            t_bias[s] = g == 3 ? (pk_f16(0.f, sc[s]) & 0xFFFF0000u) : 0u;      // the bias partner of theta: 2^-k instead of 1
        }

        __builtin_amdgcn_s_setprio(0);
        f32x4 T[4], Hp[4], z[4];
        float H[4][3];
        u32x4 bh[4], bl[4];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const Split2 p01 = split2(X[s][0] * sc[s], X[s][1] * sc[s]), p2 = split2(X[s][2] * sc[s], 0.f);
            const u32x4 ah = {p01.hi, p2.hi, p01.hi, p2.hi}, al = {p01.lo, p2.lo, p01.lo, p2.lo};
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            T[s] = mfma16(ah, adjB[s], zero);
            T[s] = mfma16(al, adjB[s], T[s]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const Split2 p01 = split2(T[s][0], T[s][1]), p23 = split2(T[s][2], T[s][3]);
            const u32x4 ta = {p01.hi, p23.hi | t_bias[s], p01.lo, p23.lo};
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            Hp[s] = mfma16(ta, theta_hi, zero);
            Hp[s] = mfma16(ta, theta_lo, Hp[s]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            keep_until_here(Hp[s][3]);
#pragma unroll
            for (int r = 0; r < 3; ++r) H[s][r] = fmaf((1.f - LEAKY) / (1.f + LEAKY), __builtin_fabsf(Hp[s][r]), Hp[s][r]);
            const Split2 p01 = split2(H[s][0], H[s][1]), p2 = split2(H[s][2], 0.f);
            const Shifted prev = shift_columns(sh_tile, sh_rd1, sh_rd1_lo, lane, u32x2{p01.hi, p2.hi}, u32x2{p01.lo, p2.lo});
            bh[s] = u32x4{p01.hi, p2.hi, prev.hi.x, prev.hi.y};
            bl[s] = u32x4{p01.lo, p2.lo, prev.lo.x, prev.lo.y};
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            z[s] = mfma16(w_hi, bh[s], zero);
            z[s] = mfma16(w_hi, bl[s], z[s]);
            z[s] = mfma16(w_lo, bh[s], z[s]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- H and z1 back to their true scale, out to the phases behind, z1 into the BatchNorm sums ----------------------------
        {
            float* ph = o.H0 + tile * (int64_t)tile_floats;
            float* pz = o.Z1 + tile * (int64_t)tile_floats;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                keep_until_here(z[s][3]);
                if (s < ns) {
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        if (d_ok[r]) {
                            const float hv = H[s][r] * isc[s], zv = z[s][r] * isc[s];
                            if constexpr (!PACKED) {
                                ph[d_off[r] + s * N] = hv;
                                pz[d_off[r] + s * N] = zv;
                            }
                            sa[r] += zv;
                            sb[r] = fmaf(zv, zv, sb[r]);
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_s_setprio(1);
    }

    // ---- epilogue: sum z1, sum z1^2 (fp64 from the 16-lane reduction on), the wavefronts' shares combined in a fixed order, one atomic
    // per channel and workgroup ---------------------------------------------------------------------------------------------------------
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
        for (int w = 0; w < MXF0_WAVES; ++w) v += pairbuf[w * 2 * F + threadIdx.x];
        atomicAdd(o.cells + (int64_t)(blockIdx.x % o.replicas) * o.cell_stride + threadIdx.x, v);
    }
}

// =====================================================================================================================
// The same arithmetic for 16 <= num_patch <= 47 -- PHM2012's 40 patches of 64 points is the reference's own ST_GCN wiring
// (configs/hparams.py:238); the row-mapped kernel runs it at 0.05 of the HBM roofline.  ONE sample per wavefront iteration, its patch
// axis in NT = 2 | 3 column tiles of 16: a [10, N] tensor is NT x 3 registers, the tiles of a sample are the independent MFMA chains
// that the four samples are in the narrow kernel.  What changes against it:
//  * theta is N x N: the projection contracts over NT k-tiles into NT column tiles (2 NT^2 products per layer); its operands
//    -- NT^2 per layer, (hi | lo) halves, each against the data's (hi | hi) and (lo | lo) -- live in LDS, shared by the EIGHT
//    wavefronts of a workgroup (one workgroup per CU: 18 KB of operands + 8 x 13.6 KB at 40 x 64);
//  * the causal taps cross tile boundaries: the shift tile is indexed [row group][column 0 .. 16 NT);
//  * the patch (64 points = 256 bytes) is read from the linearly copied window in a per-lane ROTATED chunk order -- lane p reads its
//    16-byte chunks (k + p) mod 16, k = 0..15: every statistic is a symmetric function of the patch, and a 256-byte lane stride
//    would put all lanes on the same four banks;
//  * Pearson on v_mfma_f32_16x16x4_f32 (one sample: A = B = the normalised series, lane (g, slot) holding patches 4 NT g .. + 4 NT),
//    whose result IS the adjacency's D layout; the statistics reach the D layout from the same LDS tile (one write, two reads);
//  * head: the pooled patch vector goes through LDS (broadcast reads) against fc1 rows in registers;
//  * non-finite predictions (f16 range, NaN statistics) are recomputed by a second, scanning launch of the exact kernel
//    (stgcn_forward.hip: stgcn_forward_fixup) instead of inside this one.
constexpr int MXW_WAVES = 8;
template <int NT> struct MxwGeom {
    static constexpr int W = 16 * NT, PT = W + 4, TG = 4 * NT;
    static constexpr int conv_bytes = 16 * PT * 4 + 256;                   // [16 slots][PT] + 64 floats of head scratch
    static constexpr int shift_bytes = 2 * (4 * W + 1) * 8;                // hi plane and lo plane: [4 row groups][W] + the zero slot
    static constexpr int region_bytes = ((conv_bytes > shift_bytes ? conv_bytes : shift_bytes) + 15) & ~15;
    static constexpr int theta_bytes(int L) { return L * NT * NT * 64 * 16; }
};

template <int LFIX, int NT, int NFIX, int PFIX>
__global__ __launch_bounds__(64 * MXW_WAVES, 2) void stgcn_forward_mxw_kernel(const float* __restrict__ gx, const float* __restrict__ prm,
                                                                             const float* __restrict__ bn, float* __restrict__ out, MxArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef MxwGeom<NT> G;
    constexpr int L = LFIX, W = G::W, PT = G::PT, TG = G::TG;
    const int N = NFIX ? NFIX : a.N, P = PFIX ? PFIX : a.P;
    const int LS = layer_stride(N);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, col = lane & 15;
    const int NP = N * P;
    u32x4* const theta_lds = reinterpret_cast<u32x4*>(smem);
    float* const win = smem + G::theta_bytes(L) / 4 + wave * (a.buf_floats + G::region_bytes / 4);
    float* const cur = win + a.buf_floats;
    const int64_t stride = (int64_t)gridDim.x * MXW_WAVES;
    int64_t smp = (int64_t)blockIdx.x * MXW_WAVES + wave;

    if (smp < a.B) dma_tile(gx + smp * NP, win, NP * 4, lane);          // first window on its way before the weights are touched

    // ---- prologue ------------------------------------------------------------------------------------------------------
    // theta^T as B operands, tile (ct, jt): column j = 16 jt + col, k-slot 4 g + r <-> patch 16 ct + 4 g + r; k-slot 15 of the last
    // k-tile carries the bias (its A side is the constant 1: num_patch <= 16 NT - 1)
    for (int idx = wave; idx < L * NT * NT; idx += MXW_WAVES) {
        const int l = idx / (NT * NT), ct = (idx / NT) % NT, jt = idx % NT;
        const float* lp = prm + l * LS;
        const int j = 16 * jt + col;
        float w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 16 * ct + 4 * g + r;
            const bool bias = ct == NT - 1 && 4 * g + r == 15;
            const bool ok = j < N && (k < N || bias);
            const float v = lp[ok ? (bias ? off_theta_b(N) + j : off_theta_w(N) + j * N + k) : 0];
            w[r] = ok ? v * (0.5f * (1.f + LEAKY)) : 0.f;
        }
        const Split2 p01 = split2(w[0], w[1]), p23 = split2(w[2], w[3]);
        theta_lds[idx * 64 + lane] = u32x4{p01.hi, p23.hi, p01.lo, p23.lo};      // (hi | lo) against the data's (hi | hi) and (lo | lo)
    }
    u32x4 w_hi[L][2], w_lo[L][2];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const float* lp = prm + l * LS;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {                              // see the narrow kernel: BatchNorm and the ReLU form's 2^k folded in
            const int co = slot_chan(col), coc = co >= 0 ? co : 0;
            const float mean = bn[((l * 2 + blk) * 2 + 0) * F + coc], var = bn[((l * 2 + blk) * 2 + 1) * F + coc];
            const float gam = lp[off_bn_g(N, blk) + coc], bet = lp[off_bn_b(N, blk) + coc];
            const float sc = gam / sqrtf(var + BN_EPS);
            const float shift = bet - mean * sc;
            const float wsc = (blk == 0 ? 1.f : 0.25f) * sc;
            float wc[4], wd[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = slot_chan(4 * g + r);
                const bool ok = co >= 0 && ci >= 0;
                const float2 taps2 = *reinterpret_cast<const float2*>(lp + off_conv_w(N, blk) + (coc * F + (ci >= 0 ? ci : 0)) * 2);
                wc[r] = ok ? taps2.y * wsc : 0.f;
                wd[r] = ok ? taps2.x * wsc : 0.f;
            }
            if (g == 0) wc[3] = co >= 0 ? shift : 0.f;
            const Split2 c01 = split2(wc[0], wc[1]), c23 = split2(wc[2], wc[3]), d01 = split2(wd[0], wd[1]), d23 = split2(wd[2], wd[3]);
            w_hi[l][blk] = u32x4{c01.hi, c23.hi, d01.hi, d23.hi};
            w_lo[l][blk] = u32x4{c01.lo, c23.lo, d01.lo, d23.lo};
        }
    }
    // head, row mapping: lane j holds row j of fc1
    float fc1w[W];
    const int jc = lane < N ? lane : 0;
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const float v = prm[off_fc1_w(N, L) + jc * N + (k < N ? k : 0)];
        fc1w[k] = (lane < N && k < N) ? v : 0.f;
    }
    const float fc1b_raw = prm[off_fc1_b(N, L) + jc], fc2w_raw = prm[off_fc2_w(N, L) + jc];
    const float fc1b = lane < N ? fc1b_raw : 0.f;
    const float fc2w_half = lane < N ? 0.5f * fc2w_raw : 0.f;
    const float fc2b = prm[off_fc2_b(N, L)];
    const unsigned t_bias = g == 3 ? 0x3C00u << 16 : 0u;
    float half_ok[NT], quarter_ok[NT];
    int sh_wr[NT], sh_rd1[NT], sh_rd2[NT], sh_rd1_lo[NT], sh_rd2_lo[NT];
    constexpr int SH_ZERO = 4 * W, SH_LO = 4 * W + 1;                    // index of the zero slot; offset of the lo plane
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
        const int t = 16 * jt + col;
        half_ok[jt] = t < N ? 0.5f : 0.f;
        quarter_ok[jt] = t < N ? 0.25f : 0.f;
        sh_wr[jt] = g * W + t;
        sh_rd1[jt] = t >= 1 ? g * W + t - 1 : SH_ZERO;
        sh_rd2[jt] = t >= 2 ? g * W + t - 2 : SH_ZERO;
        // through opaque registers: seeing the constant distance the compiler fuses the hi and the lo read into one ds_read2_b64, whose four
        // consecutive result registers then have to be moved apart into the two MFMA operands they belong to
        sh_rd1_lo[jt] = sh_rd1[jt] + SH_LO;
        sh_rd2_lo[jt] = sh_rd2[jt] + SH_LO;
        asm volatile("" : "+v"(sh_rd1_lo[jt]), "+v"(sh_rd2_lo[jt]));
    }
    // rows 13, 14 of the conversion tile are the padding rows lane group 3 reads as its registers 1, 2: zero, once
    for (int e = lane; e < 2 * PT; e += 64) cur[13 * PT + e] = 0.f;
    __syncthreads();

    bool pend_mine = false;
    int64_t pend_idx = 0;
    float pend_pred = 0.f;
    for (; smp < a.B; smp += stride) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (pend_mine) out[pend_idx] = pend_pred;

        // ---- patch statistics, row mapping: lane = patch ---------------------------------------------------------------
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

        // ---- Pearson adjacency (Model.py:53-71) and the statistics in the D layout --------------------------------------
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
        u32x4 adjB;
        {
            const Split2 p01 = split2(gram[0], gram[1]), p23 = split2(gram[2], gram[3]);
            adjB = u32x4{p01.hi, p23.hi, p01.lo, p23.lo};
        }
        float X[NT][3];
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
#pragma unroll
            for (int r = 0; r < 3; ++r) X[ct][r] = cur[(4 * g + r) * PT + 16 * ct + col];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        u32x2* const sh_tile = reinterpret_cast<u32x2*>(cur);             // the shift tile reuses the conversion tile's bytes
        if (lane < 2) sh_tile[SH_ZERO + SH_LO * lane] = u32x2{0u, 0u};

        // ---- the layers (the dense stage: the wavefront in a sparse stage goes first, see the narrow kernel) -------------------
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int l = 0; l < L; ++l) {
            f32x4 T[NT], Hp[NT], z[NT];
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            __builtin_amdgcn_sched_barrier(0);
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
            for (int ct = 0; ct < NT; ++ct) {
#pragma unroll
                for (int jt = 0; jt < NT; ++jt) {
                    const u32x4 th = theta_lds[((l * NT + ct) * NT + jt) * 64 + lane];
                    Hp[jt] = mfma16(tah[ct], th, Hp[jt]);
                    Hp[jt] = mfma16(tal[ct], th, Hp[jt]);
                }
            }
            float H[NT][3], V[NT][3];
            u32x2 ch[NT], cl[NT];
            u32x4 bh[NT], bl[NT];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                keep_until_here(Hp[jt][3]);
#pragma unroll
                for (int r = 0; r < 3; ++r) H[jt][r] = fmaf((1.f - LEAKY) / (1.f + LEAKY), __builtin_fabsf(Hp[jt][r]), Hp[jt][r]);
                const Split2 p01 = split2(H[jt][0], H[jt][1]), p2 = split2(H[jt][2], 1.0f);
                ch[jt] = u32x2{p01.hi, p2.hi};
                cl[jt] = u32x2{p01.lo, p2.lo};
                sh_tile[sh_wr[jt]] = ch[jt];
                sh_tile[SH_LO + sh_wr[jt]] = cl[jt];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                const u32x2 ph = sh_tile[sh_rd1[jt]], pl = sh_tile[sh_rd1_lo[jt]];
                bh[jt] = u32x4{ch[jt].x, ch[jt].y, ph.x, ph.y};
                bl[jt] = u32x4{cl[jt].x, cl[jt].y, pl.x, pl.y};
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                z[jt] = mfma16(w_hi[l][0], bh[jt], zero);
                z[jt] = mfma16(w_hi[l][0], bl[jt], z[jt]);
                z[jt] = mfma16(w_lo[l][0], bh[jt], z[jt]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                keep_until_here(z[jt][3]);
#pragma unroll
                for (int r = 0; r < 3; ++r) V[jt][r] = relu2(fmaf(2.f, H[jt][r], relu2(z[jt][r])));
                const Split2 p01 = split2(V[jt][0], V[jt][1]), p2 = split2(V[jt][2], 1.0f);
                ch[jt] = u32x2{p01.hi, p2.hi};
                cl[jt] = u32x2{p01.lo, p2.lo};
                sh_tile[sh_wr[jt]] = ch[jt];
                sh_tile[SH_LO + sh_wr[jt]] = cl[jt];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                const u32x2 ph = sh_tile[sh_rd2[jt]], pl = sh_tile[sh_rd2_lo[jt]];
                bh[jt] = u32x4{ch[jt].x, ch[jt].y, ph.x, ph.y};
                bl[jt] = u32x4{cl[jt].x, cl[jt].y, pl.x, pl.y};
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                z[jt] = mfma16(w_hi[l][1], bh[jt], zero);
                z[jt] = mfma16(w_hi[l][1], bl[jt], z[jt]);
                z[jt] = mfma16(w_lo[l][1], bh[jt], z[jt]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                keep_until_here(z[jt][3]);
#pragma unroll
                for (int r = 0; r < 3; ++r) X[jt][r] = fmaf(half_ok[jt], relu2(z[jt][r]), fmaf(quarter_ok[jt], V[jt][r], X[jt][r]));
            }
        }

        __builtin_amdgcn_s_setprio(1);
        // ---- head: max over the ten channels (Model.py:218-219), fc1, fc2 ---------------------------------------------------
        float pm[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            if (ct < NT) {
                const float x1 = g == 3 ? X[ct][0] : X[ct][1], x2 = g == 3 ? X[ct][0] : X[ct][2];
                const float m = vmax3(X[ct][0], x1, x2);
                pm[ct] = fmaf(X[ct][0] + x1 + x2, 0.f, m);               // v_max drops NaN: put it (and Inf) back
            } else {
                pm[ct] = 0.f;
            }
        }
        transpose_rows4(pm[0], pm[1], pm[2], pm[3]);                      // in: register = column tile, row = group; out: the reverse
        float pooled = vmax(vmax3(pm[0], pm[1], pm[2]), pm[3]);
        pooled = fmaf((pm[0] + pm[1]) + (pm[2] + pm[3]), 0.f, pooled);
        pooled = lane < N ? pooled : 0.f;                                 // lane (row ct, col) = patch 16 ct + col = lane
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        float* const scratch = cur + 16 * PT;
        scratch[lane] = pooled;
        __builtin_amdgcn_wave_barrier();
        float y1 = fc1b, y1b = 0.f;
#pragma unroll
        for (int q = 0; q < W / 4; ++q) {
            const float4 pv = *reinterpret_cast<const float4*>(scratch + 4 * q);
            y1 = fmaf(fc1w[4 * q], pv.x, y1);
            y1b = fmaf(fc1w[4 * q + 1], pv.y, y1b);
            y1 = fmaf(fc1w[4 * q + 2], pv.z, y1);
            y1b = fmaf(fc1w[4 * q + 3], pv.w, y1b);
        }
        y1 += y1b;
        const float pred = Row<64>::allsum(relu2(y1) * fc2w_half) + fc2b;
        pend_mine = lane == 0; pend_idx = smp; pend_pred = pred;
        // the conversion tile's padding rows were overwritten by the shift tile: restore them for the next sample
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        for (int e = lane; e < 2 * PT; e += 64) cur[13 * PT + e] = 0.f;
    }
    if (pend_mine) out[pend_idx] = pend_pred;
}

// ---- host side ----------------------------------------------------------------------------------------------------------
static bool mx_shape_ok(const rulgnn_stgcn_shape* s, const float* x) {
    const int N = s->num_patch, P = s->patch_size, L = s->num_layers;
    if (N < 2 || N > 15 || L < 1 || L > MX_MAX_LAYERS || s->mpnn_k != 1) return false;
    if (((int64_t)N * P) % 4 != 0 || (reinterpret_cast<uintptr_t>(x) & 15) != 0) return false;     // 16-byte DMA pieces
    return true;
}

// LDS of a wavefront: [window buffer: 4 samples | conversion tile + shift tile].  The safety net stages its tile (and its transpose
// scratch, PT_FLOATS) in the first region and keeps its weights in the second.
constexpr int MX_CONV_BYTES = MX_MIN_BUF_BYTES + MX_SHIFT_TILE_BYTES;
static_assert(EvalWeightsLds<16>::floats(MX_MAX_LAYERS) * 4 <= MX_CONV_BYTES, "the safety net's weights live in the conversion region");
static int mx_buf_floats(const rulgnn_stgcn_shape* s) {
    int bytes = 4 * s->num_patch * s->patch_size * 4;
    if (bytes < PT_FLOATS * 4) bytes = PT_FLOATS * 4;
    return ((bytes + 15) & ~15) / 4;
}

template <int L, int NFIX, int PFIX, bool TAPS>
static int mx_launch(const rulgnn_stgcn_shape* s, const float* x, const float* prm, const float* bn, float* out, hipStream_t stream,
                     float* taps) {
    MxArgs a;
    a.B = s->batch; a.ntiles = (s->batch + 3) / 4; a.N = s->num_patch; a.P = s->patch_size; a.L = s->num_layers;
    a.buf_floats = mx_buf_floats(s);
    a.taps = taps;
    const size_t lds = (size_t)a.buf_floats * sizeof(float) + MX_CONV_BYTES;
    if (lds > 64 * 1024) return RULGNN_EUNSUPPORTED;
    auto kern = &stgcn_forward_mx_kernel<L, NFIX, PFIX, TAPS>;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return RULGNN_EHIP;
    int dev = 0, cus = 256, per_cu = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 64, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    // Wavefronts per CU: as many as fit, in whole multiples of the four SIMDs (an uneven spread costs more than the extra
    // wavefront brings: at 1M samples 8 per CU 806 us, 9: 981, 10: 912, 11: 850 -- round 2's kernel).  Three per SIMD at 14x30.
    if (per_cu > MX_BLOCKS_PER_CU) per_cu = MX_BLOCKS_PER_CU;
    if (per_cu > 4) per_cu -= per_cu % 4;
    if (const char* e = getenv("RULGNN_MX_BLOCKS_PER_CU")) { const int v = atoi(e); if (v > 0) per_cu = v; }   // tuning aid
    int64_t grid = (int64_t)cus * per_cu;
    if (grid > a.ntiles) grid = a.ntiles;
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), lds, stream, x, prm, bn, out, a);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

template <int L, bool TAPS>
static int mx_dispatch_shape(const rulgnn_stgcn_shape* s, const float* x, const float* prm, const float* bn, float* out,
                             hipStream_t stream, float* taps) {
    if constexpr (!TAPS) {
        if (s->num_patch == 14 && s->patch_size == 30) return mx_launch<L, 14, 30, false>(s, x, prm, bn, out, stream, taps);
        if (s->num_patch == 14 && s->patch_size == 50) return mx_launch<L, 14, 50, false>(s, x, prm, bn, out, stream, taps);
    }
    return mx_launch<L, 0, 0, TAPS>(s, x, prm, bn, out, stream, taps);
}

int stgcn_forward_eval_mx(const rulgnn_stgcn_shape* s, const float* x, const float* prm, const float* bn, float* out,
                          hipStream_t stream, float* taps) {
    const int rc = validate_shape(s);
    if (rc != RULGNN_OK) return rc;
    if (!mx_shape_ok(s, x)) return RULGNN_EUNSUPPORTED;
    if (s->batch == 0) return RULGNN_OK;
    if (taps) {
        switch (s->num_layers) {
            case 1: return mx_dispatch_shape<1, true>(s, x, prm, bn, out, stream, taps);
            case 2: return mx_dispatch_shape<2, true>(s, x, prm, bn, out, stream, taps);
            default: return mx_dispatch_shape<3, true>(s, x, prm, bn, out, stream, taps);
        }
    }
    switch (s->num_layers) {
        case 1: return mx_dispatch_shape<1, false>(s, x, prm, bn, out, stream, taps);
        case 2: return mx_dispatch_shape<2, false>(s, x, prm, bn, out, stream, taps);
        default: return mx_dispatch_shape<3, false>(s, x, prm, bn, out, stream, taps);
    }
}

// Training phase F_0 (see stgcn_train_f0_mx_kernel): same shape rules as the eval kernel; RULGNN_EUNSUPPORTED -> the caller runs the
// row-mapped fp32 phase kernel.
template <bool PACKED>
static int train_f0_mx_launch(const rulgnn_stgcn_shape* s, const float* x, const float* prm, float* cacheX, float* cacheA, float* H0, float* Z1,
                              double* cells_bn0, int cell_stride_doubles, int replicas, hipStream_t stream, const HeadScalars* head = nullptr) {
    HeadScalars hs{};
    if (head) hs = *head;
    if (!mx_shape_ok(s, x)) return RULGNN_EUNSUPPORTED;
    if (s->batch == 0) return RULGNN_OK;
    MxArgs a;
    a.B = s->batch; a.ntiles = (s->batch + 3) / 4; a.N = s->num_patch; a.P = s->patch_size; a.L = s->num_layers;
    a.buf_floats = mx_buf_floats(s);
    a.taps = nullptr;
    MxF0Out o;
    o.cacheX = cacheX; o.cacheA = cacheA; o.H0 = H0; o.Z1 = Z1; o.cells = cells_bn0; o.cell_stride = cell_stride_doubles; o.replicas = replicas;
    const size_t lds = MXF0_WAVES * ((size_t)a.buf_floats * sizeof(float) + MX_CONV_BYTES) + sizeof(double) * MXF0_WAVES * 2 * F;
    if (lds > 80 * 1024) return RULGNN_EUNSUPPORTED;
    auto launch = [&](auto kern) -> int {
        if (lds > 48 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return RULGNN_EHIP;
        int dev = 0, cus = 256, per_cu = 0;
        if (hipGetDevice(&dev) == hipSuccess) {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        }
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 64 * MXF0_WAVES, lds) != hipSuccess || per_cu < 1) per_cu = 1;
        if (per_cu > MX_WAVES_PER_SIMD) per_cu = MX_WAVES_PER_SIMD;         // a workgroup is one wavefront per SIMD
        int64_t grid = (int64_t)cus * per_cu;
        const int64_t want = (a.ntiles + MXF0_WAVES - 1) / MXF0_WAVES;
        if (grid > want) grid = want;
        (void)hipGetLastError();
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * MXF0_WAVES), lds, stream, x, prm, a, o, hs);
        return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
    };
    if (s->num_patch == 14 && s->patch_size == 30) return launch(&stgcn_train_f0_mx_kernel<14, 30, PACKED>);
    if (s->num_patch == 14 && s->patch_size == 50) return launch(&stgcn_train_f0_mx_kernel<14, 50, PACKED>);
    return launch(&stgcn_train_f0_mx_kernel<0, 0, PACKED>);
}

int stgcn_train_f0_mx(const rulgnn_stgcn_shape* s, const float* x, const float* prm, float* cacheX, float* cacheA, float* H0, float* Z1,
                      double* cells_bn0, int cell_stride_doubles, int replicas, hipStream_t stream) {
    return train_f0_mx_launch<false>(s, x, prm, cacheX, cacheA, H0, Z1, cells_bn0, cell_stride_doubles, replicas, stream);
}
int stgcn_train_f0_mx_packed(const rulgnn_stgcn_shape* s, const float* x, const float* prm, float* xrec0, float* arec, double* cells_bn0,
                             int cell_stride_doubles, int replicas, hipStream_t stream, const HeadScalars* head) {
    return train_f0_mx_launch<true>(s, x, prm, xrec0, arec, nullptr, nullptr, cells_bn0, cell_stride_doubles, replicas, stream, head);
}

// ---- wide shapes ---------------------------------------------------------------------------------------------------------
template <int L, int NT, int NFIX, int PFIX>
static int mxw_launch(const rulgnn_stgcn_shape* s, const float* x, const float* prm, const float* bn, float* out, hipStream_t stream) {
    typedef MxwGeom<NT> G;
    MxArgs a;
    a.B = s->batch; a.ntiles = s->batch; a.N = s->num_patch; a.P = s->patch_size; a.L = s->num_layers;
    a.buf_floats = (s->num_patch * s->patch_size + 3) & ~3;
    a.taps = nullptr;
    const size_t lds = (size_t)G::theta_bytes(L) + (size_t)MXW_WAVES * ((size_t)a.buf_floats * 4 + G::region_bytes);
    if (lds > 160 * 1024) return RULGNN_EUNSUPPORTED;
    auto kern = &stgcn_forward_mxw_kernel<L, NT, NFIX, PFIX>;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return RULGNN_EHIP;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    int64_t grid = (s->batch + MXW_WAVES - 1) / MXW_WAVES;
    if (grid > cus) grid = cus;
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * MXW_WAVES), lds, stream, x, prm, bn, out, a);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

template <int L>
static int mxw_dispatch(const rulgnn_stgcn_shape* s, const float* x, const float* prm, const float* bn, float* out, hipStream_t stream) {
    if (s->num_patch == 40 && s->patch_size == 64) return mxw_launch<L, 3, 40, 64>(s, x, prm, bn, out, stream);
    if (s->num_patch <= 31) return mxw_launch<L, 2, 0, 0>(s, x, prm, bn, out, stream);
    return mxw_launch<L, 3, 0, 0>(s, x, prm, bn, out, stream);
}

// 16 <= num_patch <= 47.  The caller follows a successful launch with stgcn_forward_fixup (non-finite predictions -> exact kernel).
int stgcn_forward_eval_mxw(const rulgnn_stgcn_shape* s, const float* x, const float* prm, const float* bn, float* out, hipStream_t stream) {
    const int rc = validate_shape(s);
    if (rc != RULGNN_OK) return rc;
    const int N = s->num_patch, P = s->patch_size, L = s->num_layers;
    if (N < 16 || N > 47 || L < 1 || L > MX_MAX_LAYERS || s->mpnn_k != 1) return RULGNN_EUNSUPPORTED;
    if (((int64_t)N * P) % 4 != 0 || (reinterpret_cast<uintptr_t>(x) & 15) != 0) return RULGNN_EUNSUPPORTED;
    if (s->batch == 0) return RULGNN_OK;
    switch (L) {
        case 1: return mxw_dispatch<1>(s, x, prm, bn, out, stream);
        case 2: return mxw_dispatch<2>(s, x, prm, bn, out, stream);
        default: return mxw_dispatch<3>(s, x, prm, bn, out, stream);
    }
}

int stgcn_forward_mx_tap_floats() { return MX_TAP_SLOTS * 64; }

}  // namespace rulgnn

#ifdef MX_STAGE_CLOCKS
// variant builds only (tools/mx_stage_clocks.py): the stage clock sums of the launches since the last reset
extern "C" __attribute__((visibility("default"))) int rulgnn_debug_mx_stage_clocks(unsigned long long* out, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    constexpr int W = rulgnn::MX_CLK_STAMPS + 2;
    static unsigned long long host[rulgnn::MX_CLK_SLOTS][W];
    if (out) {
        if (hipMemcpyFromSymbol(host, HIP_SYMBOL(rulgnn::g_mx_stage_clocks), sizeof(host)) != hipSuccess) return -2;
        for (int i = 0; i < W; ++i) {
            out[i] = 0;
            for (int r = 0; r < rulgnn::MX_CLK_SLOTS; ++r) out[i] += host[r][i];
        }
    }
    if (reset) {
        void* dptr = nullptr;
        if (hipGetSymbolAddress(&dptr, HIP_SYMBOL(rulgnn::g_mx_stage_clocks)) != hipSuccess) return -3;
        if (hipMemset(dptr, 0, sizeof(host)) != hipSuccess) return -3;
    }
    return 0;
}
#endif
