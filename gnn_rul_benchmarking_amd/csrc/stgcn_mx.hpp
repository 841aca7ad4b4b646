// Building blocks of the matrix-core ("D layout") ST_GCN kernels for gfx950, shared by the eval forward (stgcn_forward_mx.hip) and the
// matrix-core training chain (stgcn_train_mx.hip): 2-way split f16 operands, the 16x16x32 f16 MFMA wrapper, the LDS shift tile of the
// causal taps, and the LDS-DMA copies with hand-counted vmcnt.  See the header comment of stgcn_forward_mx.hip for the layout.
#pragma once
#include "stgcn_device.hpp"

namespace rulgnn {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int MX_MAX_LAYERS = 3;
constexpr int MX_WAVES_PER_SIMD = 2;          // a third one (168 VGPRs, 12.9 KB of LDS each: it fits) is 7 % SLOWER: 688 vs 641 us at 1M samples
constexpr int MX_BLOCKS_PER_CU = 4 * MX_WAVES_PER_SIMD;
constexpr int MX_MIN_BUF_BYTES = 5120;        // [64][20] floats: layout-conversion tile (both uses)
constexpr int MX_SHIFT_TILE_BYTES = 2 * 65 * 8;  // hi pairs and lo pairs of [64 lanes + the zero slot], behind the conversion tile
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
// subtractions, two packing conversions (the form with hi = f16(a) rounded needs two v_cvt_f32_f16 on top; with the residuals from
// v_fma_mixlo/mixhi_f16 -- three instructions per pair instead of six -- it passes every test and is NOT faster: 686 vs 684 us at 1M).
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

// The causal tap of a convolution is the operand registers of column t - d: a lane shift inside the 16-lane row.  As four DPP moves
// it costs 17 cycles of the VALU port per sample and convolution (a DPP move issues at half rate); through the wavefront's LDS tile it
// is one ds_write_b128 + one ds_read_b128 on the LDS port, which nothing else here keeps busy.  Lanes t < d read the zero slot (index
// 64) = the causal padding.  LDS operations of one wavefront execute in order, so one tile serves the samples back to back.
// 8-byte pieces on purpose: the hi pair and the lo pair of the shifted column land straight in the upper halves of the two MFMA
// operands they belong to (a 16-byte read would need four register moves to get them there).
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
struct Shifted { u32x2 hi, lo; };
// (rd_lo = rd + 65 arrives through an opaque register: seeing the constant distance the compiler fuses the two reads into one
// ds_read2_b64, whose four consecutive result registers then have to be moved apart.)
__device__ __forceinline__ Shifted shift_columns(u32x2* tile, int rd, int rd_lo, int lane, const u32x2& hi, const u32x2& lo) {
    tile[lane] = hi;
    tile[65 + lane] = lo;
    Shifted r;
    r.hi = tile[rd];
    r.lo = tile[rd_lo];
    return r;
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
// The same copy for a byte count known at compile time: M0 is written once per 4 KB (the instruction's immediate offset advances the
// global AND the LDS address), so the pieces go out back to back instead of one M0 round trip each.
template <int BYTES>
__device__ __forceinline__ void dma_tile_fixed(const float* __restrict__ g, float* lds_dst, int lane) {
    static_assert(BYTES % 16 == 0 && BYTES <= 16384, "16-byte pieces, at most four M0 windows");
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_dst);
    const float* src = g + lane * 4;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0" : "=s"(keep) :: "memory");
#pragma unroll
    for (int win = 0; win < BYTES; win += 4096) {
        const float* s4 = src + win / 4;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(base + (unsigned)win) : "memory");
        if (win + 1024 <= BYTES) asm volatile("global_load_lds_dwordx4 %0, off" :: "v"(s4) : "memory");
        if (win + 2048 <= BYTES) asm volatile("global_load_lds_dwordx4 %0, off offset:1024" :: "v"(s4) : "memory");
        if (win + 3072 <= BYTES) asm volatile("global_load_lds_dwordx4 %0, off offset:2048" :: "v"(s4) : "memory");
        if (win + 4096 <= BYTES) asm volatile("global_load_lds_dwordx4 %0, off offset:3072" :: "v"(s4) : "memory");
        constexpr int full = BYTES / 1024 * 1024;          // the last, partial piece: the lanes below the end
        if (win <= full && full < win + 4096 && full < BYTES) {
            if (lane * 16 < BYTES - full) {
                if (full - win == 0) asm volatile("global_load_lds_dwordx4 %0, off" :: "v"(s4) : "memory");
                if (full - win == 1024) asm volatile("global_load_lds_dwordx4 %0, off offset:1024" :: "v"(s4) : "memory");
                if (full - win == 2048) asm volatile("global_load_lds_dwordx4 %0, off offset:2048" :: "v"(s4) : "memory");
                if (full - win == 3072) asm volatile("global_load_lds_dwordx4 %0, off offset:3072" :: "v"(s4) : "memory");
            }
        }
    }
    asm volatile("s_mov_b32 m0, %0" :: "s"(keep) : "memory");
}



// ---- patch statistics, lean form -------------------------------------------------------------------------------------
// Same ten statistics as patch_statistics_regs<P, true> (Model.py:7-52) with two of the per-element accumulations removed:
//  * sum x^2 = sum (x - mean)^2 + P mean^2 (both terms non-negative: no cancellation), so rms comes from the second pass;
//  * sum |x| = +-sum x when the patch does not change sign (min >= 0 or max <= 0: every dataset the reference wires is scaled to
//    [0, 1]); a wavefront with a mixed-sign patch takes the explicit sum (wave-uniform branch).
template <int P>
__device__ __forceinline__ void patch_load(const float* pp, float (&v)[P]) {
    static_assert(P % 2 == 0 && P <= 64, "even patch sizes that fit the register budget");
    const float2* p2 = reinterpret_cast<const float2*>(pp);
#pragma unroll
    for (int i = 0; i < P / 2; ++i) { const float2 q = p2[i]; v[2 * i] = q.x; v[2 * i + 1] = q.y; }
}
template <int P>
__device__ __forceinline__ void patch_statistics_lean(const float (&v)[P], float (&st)[F]) {
#pragma clang fp contract(off)          // see patch_statistics_regs: a constant patch must give the exact 0 deviation
    float s = 0.f, mx = -INFINITY, mn = INFINITY;
#pragma unroll
    for (int i = 0; i < P; i += 2) {
        s += v[i] + v[i + 1];
        mx = vmax3(mx, v[i], v[i + 1]);
        mn = vmin3(mn, v[i], v[i + 1]);
    }
    float sa = mn >= 0.f ? s : -s;
    if (__builtin_amdgcn_ballot_w64(mn < 0.f && mx > 0.f) != 0) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < P; i += 2) t += __builtin_fabsf(v[i]) + __builtin_fabsf(v[i + 1]);
        sa = t;
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
    const float var = m2 * (1.0f / (float)(P - 1));
    const float sd = __builtin_amdgcn_sqrtf(var);
    const float isd = __builtin_amdgcn_rcpf(sd);       // sd == 0 -> inf; 0 * inf = NaN like the reference's 0/0
    const float isd2 = isd * isd;
    st[0] = mx;
    st[1] = mn;
    st[2] = mx - mn;
    st[3] = var;
    st[4] = sd;
    st[5] = mean;
    st[6] = __builtin_amdgcn_sqrtf(fmaf(mean, mean, m2 * invP));
    st[7] = sa * invP;
    st[8] = (m3 * invP) * (isd2 * isd);
    st[9] = (m4 * invP) * (isd2 * isd2) - 3.0f;
}

}  // namespace rulgnn
