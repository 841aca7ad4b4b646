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
//   z1 = W1 x [H ; H shifted by 1]         (conv_block1 with BatchNorm folded in; the causal tap is a DPP row_shr of
//                                           the PACKED operand registers, zero fill = the causal padding)    3 MFMAs
//   z2 = W2 x [o0 ; o0 shifted by 2]       (conv_block2, dilation 2)                                        3 MFMAs
// Biases ride in spare k-slots against a constant 1 (theta: k = 15, so num_patch <= 15; conv: slot 3 of lane group 0).
// ReLU is |x| + x = 2 relu(x) (one full-rate op, NaN-preserving where v_max is neither); the powers of two are folded
// into the next weights.  Four samples are in flight per wavefront so that dependent MFMAs never wait.
//
// Around the layers: the patch statistics (Model.py:7-52) and the Pearson Gram matrix (Model.py:53-71, f32 4-block MFMAs,
// exact) run in the row mapping of the exact kernel and are converted through the wavefront's LDS tile; the head
// (channel max-pool, fc1, fc2) returns to the row mapping with one 4x4 register/row transpose.
//
// Memory: windows arrive by LDS-DMA (global_load_lds_dwordx4, no VGPRs), double buffered one tile ahead; 4 bytes per
// sample leave.  One wavefront per workgroup (wavefronts never synchronise), 2 x 6.7 KB of LDS each at 14x30.
//
// Safety net.  f16 overflows at 65504.  Every pointwise step here preserves NaN / Inf, so a sample whose arithmetic
// left the f16 range (or whose input statistics are NaN: constant patch, Model.py:41-52) ends up non-finite; such
// samples are recomputed by the exact fp32 tile routine (stgcn_eval_tile.hpp) inside the same launch, which also
// reproduces the reference's NaN placement.  Inputs scaled to [0, 1] (every dataset the reference wires) never take it.
#include <cstdlib>

#include "stgcn_eval_tile.hpp"
#include "stgcn_host.hpp"

namespace rulgnn {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int MX_MAX_LAYERS = 3;
constexpr int MX_BLOCKS_PER_CU = 8;
constexpr int MX_MIN_BUF_BYTES = 5120;        // [64][20] floats: layout-conversion tile (both uses)
constexpr int MX_TAPS_PER_LAYER = 88;
constexpr int MX_TAP_SLOTS = 38 + MX_TAPS_PER_LAYER * MX_MAX_LAYERS + 2;

// channel <-> row slot of the 16-row tile (see the header comment)
__host__ __device__ constexpr int chan_slot(int c) { return c + c / 3; }
__host__ __device__ constexpr int slot_chan(int m) { return (m & 3) == 3 ? -1 : (m == 12 ? 9 : (m > 12 ? -1 : m - (m >> 2))); }
static_assert(chan_slot(9) == 12 && slot_chan(12) == 9 && slot_chan(10) == 8 && slot_chan(11) == -1 && slot_chan(4) == 3, "slot map");

// ---- f16 split ------------------------------------------------------------------------------------------------------
// v_cvt_pk_f16_f32: two fp32 -> one register of two f16, round to nearest (|x| >= 65520 -> Inf, which is what trips the safety net).
__device__ __forceinline__ unsigned pk_f16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
// a = hi + lo with hi = the top 11 significant bits of a (mask: exactly representable in f16, so its conversion is exact) and
// lo = f16(a - hi), the difference being exact in fp32: 22 significant bits in all.  Per PAIR of values: two v_and, two
// subtractions, two packing conversions (the form with hi = f16(a) rounded needs two v_cvt_f32_f16 on top).
struct Split2 { unsigned hi, lo; };
__device__ __forceinline__ Split2 split2(float a, float b) {
    const float ha = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a) & 0xFFFFE000u);
    const float hb = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, b) & 0xFFFFE000u);
    Split2 s;
    s.hi = pk_f16(ha, hb);
    s.lo = pk_f16(a - ha, b - hb);
    return s;
}

__device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <int D>
__device__ __forceinline__ unsigned shr_packed(unsigned v) {     // operand registers of column t - D; zero for t < D
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_SHR + D, 0xf, 0xf, true);
}

// Row 3 of every lane group is padding, so element 3 of a D-layout accumulator is dead on arrival -- and the register allocator
// hands it out as a scratch register while the MFMA that writes it is still in flight: a write-after-write hazard the
// compiler pads with s_nop 7.  Naming the element at the point where its siblings are consumed keeps it reserved until then.
__device__ __forceinline__ void keep_until_here(float v) { asm volatile("" ::"v"(v)); }

__device__ __forceinline__ float relu2(float v) { return __builtin_fabsf(v) + v; }     // 2 relu(v); NaN / +Inf preserving

// ---- LDS-DMA --------------------------------------------------------------------------------------------------------
// Copies `bytes` (multiple of 16) from global memory to the wavefront's LDS buffer; every instruction moves 1 KB
// (lane i: 16 bytes to dst + 16 i).  Completion is counted in vmcnt by the hardware, not by the compiler: the caller waits.
// Inline asm on purpose: the builtin form is counted by the compiler, which then puts s_waitcnt vmcnt(0) in front of EVERY later LDS
// read (it cannot tell the two buffers apart) -- the prefetch would be waited for at once.  M0 carries the LDS base and is
// compiler-reserved: saved and restored inside the statement.
__device__ __forceinline__ void dma_tile(const float* __restrict__ g, float* lds_dst, int bytes, int lane) {
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_dst);    // LDS byte address, wave-uniform
    for (int off = 0; off < bytes; off += 1024) {
        if (off + lane * 16 < bytes) {
            const float* src = g + (off >> 2) + lane * 4;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\t"
                         "s_mov_b32 m0, %2\n\t"
                         "s_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, off\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(src), "s"(base + (unsigned)off)
                         : "memory");
        }
    }
}

struct MxArgs {
    int64_t B;
    int64_t ntiles;
    int N, P, L;
    int buf_floats;        // one of the two LDS buffers of a wavefront
    float* taps;           // debug: raw register dumps of tile 0 (TAPS builds only)
};

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

template <int LFIX, int NFIX, int PFIX, bool TAPS>
__global__ __launch_bounds__(64, 2) void stgcn_forward_mx_kernel(const float* __restrict__ gx, const float* __restrict__ prm,
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
                w[r] = ok ? v : 0.f;
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
            if (g == 0) wc[3] = co >= 0 ? shift : 0.f;    // x the constant 1 that rides in slot 3 of the data operand
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
    // T's accumulator starts with row t = 15 at 1: the k = 15 slot of theta^T is the bias
    const f32x4 t_init = {0.f, 0.f, 0.f, g == 3 ? 1.f : 0.f};
    const float half_ok = col < N ? 0.5f : 0.f, quarter_ok = col < N ? 0.25f : 0.f;  // residual scales; padded columns stay 0
    const unsigned one_hi = 0x3C00u << 16;                                            // f16 1.0 in the upper half

    bool any_bad = false;
    for (int it = 0; tile < a.ntiles; ++it, tile += gridDim.x) {
        float* cur = smem + (it & 1) * a.buf_floats;     // plain arithmetic on the __shared__ base keeps these LDS (not flat) accesses
        const int64_t s0 = tile * 4;
        const int ns = (int)((a.B - s0) < 4 ? (a.B - s0) : 4);
        const bool tapon = TAPS && a.taps != nullptr && tile == 0;
        // this tile's windows have landed; every LDS access of the previous tile has retired
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        {
            const int64_t nt = tile + gridDim.x;
            if (nt < a.ntiles) {
                const int64_t n0 = nt * 4;
                const int nns = (int)((a.B - n0) < 4 ? (a.B - n0) : 4);
                dma_tile(gx + n0 * tileNP, smem + ((it + 1) & 1) * a.buf_floats, nns * tileNP * 4, lane);
            }
        }

        // ---- patch statistics, row mapping: lane (sample row g, patch col) ------------------------------------------
        const bool valid = (g < ns) && (col < N);
        float X0[F];
#pragma unroll
        for (int c = 0; c < F; ++c) X0[c] = 0.f;
        if (valid) {
            if constexpr (PFIX != 0 && PFIX % 2 == 0 && PFIX <= 64) patch_statistics_regs<PFIX, true>(cur + (g * N + col) * P, X0);
            else patch_statistics(cur + (g * N + col) * P, P, X0);
        }
#pragma unroll
        for (int c = 0; c < F; ++c) tap<TAPS>(a.taps, tapon, c, lane, X0[c]);

        // ---- Pearson adjacency: rows = channel slots, one 16x16 block per sample (f32 4-block MFMA, exact) ---------
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < F; ++c) cur[(g * 16 + chan_slot(c)) * PT_STRIDE + col] = X0[c];
        __builtin_amdgcn_wave_barrier();
        f32x16 gram = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
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
        // gram[4 b + r] in lane (g, col) = Adj_b[slot 4 g + r][slot col]: the D layout of sample b's adjacency
        u32x4 adjB[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
#pragma unroll
            for (int r = 0; r < 4; ++r) tap<TAPS>(a.taps, tapon, 10 + 4 * b + r, lane, gram[4 * b + r]);
            const Split2 p01 = split2(gram[4 * b + 0], gram[4 * b + 1]), p23 = split2(gram[4 * b + 2], gram[4 * b + 3]);
            adjB[b] = u32x4{p01.hi, p23.hi, p01.lo, p23.lo};
        }

        // ---- statistics into the D layout: [row-mapped lane][16 slots] through the LDS tile ------------------------
        __builtin_amdgcn_wave_barrier();
        {
            float4* w4 = reinterpret_cast<float4*>(cur + lane * PT_STRIDE);
            w4[0] = make_float4(X0[0], X0[1], X0[2], 0.f);
            w4[1] = make_float4(X0[3], X0[4], X0[5], 0.f);
            w4[2] = make_float4(X0[6], X0[7], X0[8], 0.f);
            w4[3] = make_float4(X0[9], 0.f, 0.f, 0.f);
        }
        __builtin_amdgcn_wave_barrier();
        float X[4][3];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float4 v = *reinterpret_cast<const float4*>(cur + (16 * s + col) * PT_STRIDE + 4 * g);
            X[s][0] = v.x; X[s][1] = v.y; X[s][2] = v.z;
#pragma unroll
            for (int r = 0; r < 3; ++r) tap<TAPS>(a.taps, tapon, 26 + 3 * s + r, lane, X[s][r]);
        }

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
                T[s] = mfma16(ah, adjB[s], t_init);
                T[s] = mfma16(al, adjB[s], T[s]);
            }
            __builtin_amdgcn_sched_barrier(0);
            // Hp = theta(A.X) + b : rows c, columns j
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int r = 0; r < 4; ++r) tap<TAPS>(a.taps, tapon, tb + 4 * s + r, lane, T[s][r]);
                const Split2 p01 = split2(T[s][0], T[s][1]), p23 = split2(T[s][2], T[s][3]);
                const u32x4 ta = {p01.hi, p23.hi, p01.lo, p23.lo};
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                Hp[s] = mfma16(ta, ops[l].theta_hi, zero);
                Hp[s] = mfma16(ta, ops[l].theta_lo, Hp[s]);
            }
            float H[4][3], V[4][3];
            __builtin_amdgcn_sched_barrier(0);
            // conv_block1 on H = leaky(Hp)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int r = 0; r < 4; ++r) tap<TAPS>(a.taps, tapon, tb + 16 + 4 * s + r, lane, Hp[s][r]);
                keep_until_here(Hp[s][3]);
#pragma unroll
                // leaky(x) = (1 + a)/2 x + (1 - a)/2 |x|: two full-rate instructions (v_max is half rate); NaN stays NaN
                for (int r = 0; r < 3; ++r) H[s][r] = fmaf(0.5f * (1.f - LEAKY), __builtin_fabsf(Hp[s][r]), (0.5f * (1.f + LEAKY)) * Hp[s][r]);
                const Split2 p01 = split2(H[s][0], H[s][1]), p2 = split2(H[s][2], 0.f);
                const unsigned h2 = p2.hi | one_hi;                                           // slot 3 = 1: the BatchNorm shift's partner
                const u32x4 bh = {p01.hi, h2, shr_packed<1>(p01.hi), shr_packed<1>(p2.hi)};
                const u32x4 bl = {p01.lo, p2.lo, shr_packed<1>(p01.lo), shr_packed<1>(p2.lo)};
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                z[s] = mfma16(ops[l].w_hi[0], bh, zero);
                z[s] = mfma16(ops[l].w_hi[0], bl, z[s]);
                z[s] = mfma16(ops[l].w_lo[0], bh, z[s]);
            }
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
                const Split2 p01 = split2(V[s][0], V[s][1]), p2 = split2(V[s][2], 0.f);
                const unsigned h2 = p2.hi | one_hi;
                const u32x4 bh = {p01.hi, h2, shr_packed<2>(p01.hi), shr_packed<2>(p2.hi)};
                const u32x4 bl = {p01.lo, p2.lo, shr_packed<2>(p01.lo), shr_packed<2>(p2.lo)};
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                z[s] = mfma16(ops[l].w_hi[1], bh, zero);
                z[s] = mfma16(ops[l].w_hi[1], bl, z[s]);
                z[s] = mfma16(ops[l].w_lo[1], bh, z[s]);
            }
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
        if (mine) out[s0 + g] = pred;
        any_bad |= __any(mine && !(__builtin_fabsf(pred) <= 3.0e38f)) != 0;
    }

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

// ---- host side ----------------------------------------------------------------------------------------------------------
static bool mx_shape_ok(const rulgnn_stgcn_shape* s, const float* x) {
    const int N = s->num_patch, P = s->patch_size, L = s->num_layers;
    if (N < 2 || N > 15 || L < 1 || L > MX_MAX_LAYERS || s->mpnn_k != 1) return false;
    if (((int64_t)N * P) % 4 != 0 || (reinterpret_cast<uintptr_t>(x) & 15) != 0) return false;     // 16-byte DMA pieces
    return true;
}

static int mx_buf_floats(const rulgnn_stgcn_shape* s) {
    int bytes = 4 * s->num_patch * s->patch_size * 4;
    if (bytes < MX_MIN_BUF_BYTES) bytes = MX_MIN_BUF_BYTES;
    const int fb = EvalWeightsLds<16>::floats(s->num_layers) * 4;        // the safety net's weights live in buffer 1
    if (bytes < fb) bytes = fb;
    return ((bytes + 15) & ~15) / 4;
}

template <int L, int NFIX, int PFIX, bool TAPS>
static int mx_launch(const rulgnn_stgcn_shape* s, const float* x, const float* prm, const float* bn, float* out, hipStream_t stream,
                     float* taps) {
    MxArgs a;
    a.B = s->batch; a.ntiles = (s->batch + 3) / 4; a.N = s->num_patch; a.P = s->patch_size; a.L = s->num_layers;
    a.buf_floats = mx_buf_floats(s);
    a.taps = taps;
    const size_t lds = (size_t)2 * a.buf_floats * sizeof(float);
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
    // The kernel is issue-bound, not latency-bound: two wavefronts per SIMD (8 one-wave workgroups per CU) saturate it, and an
    // even spread over the four SIMDs matters more than a third wavefront (measured at 1M samples: 8 per CU 806 us, 9: 981,
    // 10: 912, 11: 850, 12 -- which the occupancy API grants but the LDS granule does not fit -- 1044).
    if (per_cu > MX_BLOCKS_PER_CU) per_cu = MX_BLOCKS_PER_CU;
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

int stgcn_forward_mx_tap_floats() { return MX_TAP_SLOTS * 64; }

}  // namespace rulgnn
