// The large scaled GEMM on PRE-SPLIT operands (round 6; interface: sgemm_mfma.hpp, sgemm_planes*).
//
// C[m][n] (+)= sum_k A(m,k) B(n,k) for fp32 operands whose largest magnitudes the caller knows (the partial-maxima rows of
// csrc/sgemm.hip's two-plane f16 form), as THREE v_mfma_f32_32x32x16_f16 per 32 x 32 x 16 product block -- the same arithmetic as
// sgemm_f16x2v_kernel (a = hi + lo, hi = top 11 significant bits of a s, lo = f16(a s - hi); lo hi + hi lo + hi hi, fp32 accumulate,
// unscaled exactly) -- but with the split taken OUT of the product loop:
//
//   1. split pass (split_planes_kernel): each operand is read once and written as two k-contiguous f16 planes  hi[row][k], lo[row][k]
//      (a transposing pass through LDS when the source is row-contiguous), scaled by the power of two that puts its largest finite
//      element in [2^11, 2^12); the scale is left beside the planes.  theta used to be re-split by 40 workgroups, A.X by 4.
//   2. product (sgemm_planes_kernel): 160 x 256 (or 128 x 256) output tiles, 8 wavefronts = 4 column strips of 64 x 2 halves of every
//      32-deep k stage (the pair's accumulators meet once, in the epilogue, through LDS).  Operand tiles go HBM -> LDS by
//      global_load_lds_dwordx4 (no VGPR round trip, no VALU, no ds_write) into a THREE-stage ring: one workgroup barrier per 32 k,
//      loads two stages ahead, vmcnt counted by hand (nothing else in the loop touches vector memory).  Rows are 64 bytes in LDS;
//      the 16-byte k slot of row r sits at slot ^ ((r >> 2) & 3) -- applied on the DMA's per-lane SOURCE address and on the
//      fragment read -- which makes every ds_read_b128 lane group of the 32-row fragment reads conflict-free.
//      [10 240 x 1024] . [1024 x 1024] is 64 x 4 = 256 tiles: one per CU (the 256 x 256 kernel: 160).  Workgroups are numbered so
//      that the four column tiles of a row panel run on one XCD (they share the panel's A planes in that XCD's L2).
//
// Shapes: M a multiple of 160 or 128, N of 256, every k slice of 32; anything else stays on sgemm_f16x2v_kernel (sgemm.hip).
#include "sgemm_mfma.hpp"

namespace rulgnn {

typedef _Float16 pl_f16x8 __attribute__((ext_vector_type(8)));
typedef float pl_f32x16 __attribute__((ext_vector_type(16)));
typedef float pl_f32x4 __attribute__((ext_vector_type(4)));

// ---- 1. split pass ------------------------------------------------------------------------------------------------------------
static __device__ __forceinline__ float pl_scale_of(float amax) {
    const unsigned m = __builtin_bit_cast(unsigned, amax);
    const int e = (int)((m >> 23) & 0xFFu);                          // biased exponent of the largest finite magnitude
    if (!(amax > 0.f) || e == 0 || e == 255) return 1.0f;
    int se = 127 + 11 - (e - 127);
    se = se < 1 ? 1 : (se > 254 ? 254 : se);
    return __builtin_bit_cast(float, (unsigned)se << 23);
}
// the operand's scale from its producers' partial maxima (one workgroup: a few thousand floats out of L2)
static __device__ __forceinline__ float pl_block_scale(const float* __restrict__ amax, int n) {
    __shared__ float part[16];
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, amax[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    m = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, part[w]);
    return pl_scale_of(m);
}
static __device__ __forceinline__ void pl_split(float a, _Float16& h, _Float16& l) {
    const float ha = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a) & 0xFFFFE000u);
    h = (_Float16)ha;
    l = (_Float16)(a - ha);
}
struct SplitArgs {
    const float* src;
    int64_t s_row, s_k;          // element strides of src(row, k)
    int rows, K;
    const float* amax;
    int n_amax;
    _Float16* hi;                // [rows][K]
    _Float16* lo;
    float* scale;                // [1]: the power of two the planes were multiplied by
};
// k contiguous in the source: a thread splits 8 consecutive k of one row (two 16-byte loads, one 16-byte store per plane)
static __global__ __launch_bounds__(256) void split_planes_direct_kernel(SplitArgs a) {
    const float s = pl_block_scale(a.amax, a.n_amax);
    if (blockIdx.x == 0 && threadIdx.x == 0) a.scale[0] = s;
    const int64_t oct = (int64_t)a.rows * (a.K >> 3);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < oct; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / (a.K >> 3);
        const int k = (int)(i - r * (a.K >> 3)) * 8;
        const float* p = a.src + r * a.s_row + k;
        const pl_f32x4 v0 = *reinterpret_cast<const pl_f32x4*>(p), v1 = *reinterpret_cast<const pl_f32x4*>(p + 4);
        pl_f16x8 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            _Float16 hh, ll;
            pl_split(v0[e] * s, hh, ll); h[e] = hh; l[e] = ll;
            pl_split(v1[e] * s, hh, ll); h[4 + e] = hh; l[4 + e] = ll;
        }
        *reinterpret_cast<pl_f16x8*>(a.hi + r * a.K + k) = h;
        *reinterpret_cast<pl_f16x8*>(a.lo + r * a.K + k) = l;
    }
}
// rows contiguous in the source (src(row, k) = src[k * s_k + row]): 64 x 64 tiles through LDS, read along rows, written along k
static __global__ __launch_bounds__(256) void split_planes_transposed_kernel(SplitArgs a) {
    __shared__ unsigned tile[64][65];                 // [k][row]: (hi | lo << 16)
    const float s = pl_block_scale(a.amax, a.n_amax);
    if (blockIdx.x == 0 && threadIdx.x == 0) a.scale[0] = s;
    const int tr = a.rows >> 6, tk = a.K >> 6;
    const int tid = threadIdx.x;
    for (int t = blockIdx.x; t < tr * tk; t += gridDim.x) {
        const int r0 = (t % tr) * 64, k0 = (t / tr) * 64;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {                 // thread: 4 consecutive rows of k = (tid >> 4) + 16 e
            const int k = (tid >> 4) + 16 * e, r = 4 * (tid & 15);
            const pl_f32x4 v = *reinterpret_cast<const pl_f32x4*>(a.src + (int64_t)(k0 + k) * a.s_k + r0 + r);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                _Float16 hh, ll;
                pl_split(v[q] * s, hh, ll);
                tile[k][r + q] = (unsigned)__builtin_bit_cast(unsigned short, hh) | ((unsigned)__builtin_bit_cast(unsigned short, ll) << 16);
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 2; ++e) {                 // thread: 8 consecutive k of row (tid >> 3) + 32 e
            const int r = (tid >> 3) + 32 * e, k = 8 * (tid & 7);
            pl_f16x8 h, l;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const unsigned w = tile[k + q][r];
                h[q] = __builtin_bit_cast(_Float16, (unsigned short)(w & 0xFFFFu));
                l[q] = __builtin_bit_cast(_Float16, (unsigned short)(w >> 16));
            }
            *reinterpret_cast<pl_f16x8*>(a.hi + (int64_t)(r0 + r) * a.K + k0 + k) = h;
            *reinterpret_cast<pl_f16x8*>(a.lo + (int64_t)(r0 + r) * a.K + k0 + k) = l;
        }
    }
}

// ---- 2. product ---------------------------------------------------------------------------------------------------------------
struct PlaneGemmArgs {
    const _Float16* Ahi; const _Float16* Alo;     // [M][K]
    const _Float16* Bhi; const _Float16* Blo;     // [N][K]
    const float* scale_a; const float* scale_b;   // [1] each
    float* C; int64_t ldc;                        // slice z at C + z * M * ldc
    int M, N, K;
    int kchunk;                                   // k per slice (multiple of 32)
    int tiles_m, tiles_n, slices;
    int accumulate;
};

// one 1-KB piece (16 rows x 64 bytes of one plane) HBM -> LDS: lane i writes LDS bytes [16 i, 16 i + 16) of the piece and fetches
// row i >> 2, k slot (i & 3) ^ ((i >> 4) & 3) of it (`voff`, the same for every piece); `sbase` = the piece's first row at the stage's k
static __device__ __forceinline__ void pl_dma_piece(unsigned voff, const void* sbase, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %3\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_byte)
                 : "memory");
}
template <int N>
static __device__ __forceinline__ void pl_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// workgroup barrier that the compiler may not move LDS accesses across (the bare builtin is IntrNoMem)
static __device__ __forceinline__ void pl_barrier() { asm volatile("s_barrier" ::: "memory"); }

template <int MB>
static __global__ __launch_bounds__(512) void sgemm_planes_kernel(PlaneGemmArgs g) {
    constexpr int BM = 32 * MB, BN = 256;
    constexpr int APL = BM * 64, BPL = BN * 64;               // bytes of one plane of a stage
    constexpr int STAGE = 2 * APL + 2 * BPL;
    constexpr int PA = BM / 16, PB = BN / 16;                 // pieces per plane
    constexpr int NP = 2 * PA + 2 * PB;                       // pieces per stage (52 | 48)
    constexpr int NW_LO = NP / 8, NW_HI = (NP + 7) / 8;       // pieces per wavefront: waves below NP % 8 issue one more
    extern __shared__ __attribute__((aligned(16))) unsigned char pl_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = wave & 3, kh = wave >> 2;
    // XCD-aware numbering: consecutive linear tiles (the column tiles of one row panel first) on ONE XCD (workgroup i runs on XCD i % 8)
    int lin;
    {
        const int total = gridDim.x, wg = blockIdx.x, xcd = wg & 7, q = total >> 3, r = total & 7;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
    }
    const int tn = lin % g.tiles_n, tm = (lin / g.tiles_n) % g.tiles_m, z = lin / (g.tiles_n * g.tiles_m);
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = z * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    const int nst = (kend - kbeg) >> 5;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)pl_lds);
    const int64_t ldb = (int64_t)g.K * 2;                     // row stride of every plane, bytes
    const unsigned voff = (unsigned)((lane >> 2) * (int)ldb + (((lane & 3) ^ ((lane >> 4) & 3)) << 4));
    const unsigned char* Ah = reinterpret_cast<const unsigned char*>(g.Ahi) + (int64_t)m0 * ldb;
    const unsigned char* Al = reinterpret_cast<const unsigned char*>(g.Alo) + (int64_t)m0 * ldb;
    const unsigned char* Bh = reinterpret_cast<const unsigned char*>(g.Bhi) + (int64_t)n0 * ldb;
    const unsigned char* Bl = reinterpret_cast<const unsigned char*>(g.Blo) + (int64_t)n0 * ldb;

    // this wavefront's pieces (p = wave, wave + 8, ...): source row block and LDS offset, fixed for the whole product
    const unsigned char* pb[NW_HI];
    unsigned pl[NW_HI];
#pragma unroll
    for (int e = 0; e < NW_HI; ++e) {
        int p = wave + 8 * e;                              // (wave-uniform)
        p = p < NP ? p : NP - 1;                           // (the slot past the end of the waves that issue NW_LO: never issued)
        const unsigned char* base = p < PA ? Ah : (p < 2 * PA ? Al : (p < 2 * PA + PB ? Bh : Bl));
        const int row16 = p < PA ? p : (p < 2 * PA ? p - PA : (p < 2 * PA + PB ? p - 2 * PA : p - 2 * PA - PB));
        pb[e] = base + (int64_t)row16 * 16 * ldb + (int64_t)kbeg * 2;
        pl[e] = (unsigned)p * 1024u;
    }
    const bool extra = wave < (NP & 7);                    // this wavefront issues NW_HI pieces per stage
    auto issue_piece = [&](int st, int e) {                // piece e of stage `st` into ring slot st % 3
        if (e < NW_LO || extra) pl_dma_piece(voff, pb[e] + (int64_t)st * 64, lds0 + (unsigned)(st % 3) * STAGE + pl[e]);
    };
    auto issue = [&](int st) {
#pragma unroll
        for (int e = 0; e < NW_HI; ++e) issue_piece(st, e);
    };

    pl_f32x16 acc[MB][2];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read: lane -> row (lane & 31) of a 32-row block, k slot 2 kh + (lane >> 5), un-swizzled by the row's bits 2..3
    const int frag = (lane & 31) * 64 + (((2 * kh + (lane >> 5)) ^ (((lane & 31) >> 2) & 3)) << 4);
    if (nst > 0) issue(0);
    if (nst > 1) issue(1);
    for (int t = 0; t < nst; ++t) {
        // my pieces of stage t have landed when at most the pieces of stage t + 1 are outstanding
        if (t + 1 < nst) {
            if (extra) pl_wait_vm<NW_HI>(); else pl_wait_vm<NW_LO>();
        } else pl_wait_vm<0>();
        pl_barrier();                                          // ... and everybody's; everybody is done reading slot (t - 1) % 3
        const bool more = t + 2 < nst;
        const unsigned char* sb = pl_lds + (t % 3) * STAGE;
        pl_f16x8 b[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int p = 0; p < 2; ++p) b[j][p] = *reinterpret_cast<const pl_f16x8*>(sb + 2 * APL + p * BPL + (nw * 64 + 32 * j) * 64 + frag);
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            pl_f16x8 a[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) a[p] = *reinterpret_cast<const pl_f16x8*>(sb + p * APL + i * 2048 + frag);
#pragma unroll
            for (int j = 0; j < 2; ++j) {                       // lo hi, hi lo, hi hi: smallest first
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[j][0], acc[i][j], 0, 0, 0);
            }
            // the next-but-one stage's pieces go out BETWEEN the row blocks' products, not in front of them (an LDS-DMA piece holds a
            // wavefront's issue for ~100 cycles; measured 65.8 -> 63.6 us at [10 240 x 1024] . [1024 x 1024]; staggering the two wavefronts
            // of a SIMD on top of that: +- 0)
            if (more) {
                constexpr int PER = (NW_HI + MB - 1) / MB;      // pieces per row block
#pragma unroll
                for (int q = 0; q < PER; ++q)
                    if (i * PER + q < NW_HI) issue_piece(t + 2, i * PER + q);
            }
        }
    }
    // epilogue: the k halves of a wavefront pair meet in LDS (two rounds: the ring holds 3 of the 5 row blocks of all four pairs at once),
    // then the kh = 0 wavefront unscales and stores its 32 MB x 64 strip
    const float unscale = 1.0f / (g.scale_a[0] * g.scale_b[0]);       // (exact: powers of two)
    float* Cz = g.C + (int64_t)z * g.M * g.ldc;
    constexpr int R1 = MB < 3 ? MB : 3;
    pl_f32x4* xch = reinterpret_cast<pl_f32x4*>(pl_lds) + (size_t)nw * (R1 * 2 * 4 * 64);      // this pair's exchange area (R1 x 2 blocks x 4 KB)
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        const int i0 = round == 0 ? 0 : R1, i1 = round == 0 ? R1 : MB;
        __syncthreads();                                       // (round 0: every wavefront has left the product loop; round 1: round 0's data was consumed)
        if (kh == 1) {
#pragma unroll
            for (int i = i0; i < i1; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const pl_f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        xch[(((i - i0) * 2 + j) * 4 + q) * 64 + lane] = v;
                    }
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int i = i0; i < i1; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const pl_f32x4 v = xch[(((i - i0) * 2 + j) * 4 + q) * 64 + lane];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const int r = 4 * q + c;
                            const int gm = m0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), gn = n0 + nw * 64 + 32 * j + (lane & 31);
                            float* cp = Cz + (int64_t)gm * g.ldc + gn;
                            const float val = (acc[i][j][r] + v[c]) * unscale;
                            *cp = g.accumulate ? *cp + val : val;
                        }
                    }
        }
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------------------
static inline int pl_pick_mb(int M) {
    const bool ok5 = M % 160 == 0, ok4 = M % 128 == 0;
    if (ok5 && ok4) {                                          // the one whose tile count wastes less of the last round of 256 CUs
        auto eff = [](int t) { return (double)t / (double)(((t + 255) / 256) * 256); };
        return eff(M / 160 * 4) >= eff(M / 128 * 4) ? 5 : 4;
    }
    return ok5 ? 5 : (ok4 ? 4 : 0);
}
// k slices of a split-K product on this kernel: tiles x slices ~ 256 workgroups, every slice a multiple of 32 k; 0 = does not apply
int sgemm_planes_slices(int M, int N, int K, bool want_split) {
    const int mb = pl_pick_mb(M);
    if (mb == 0 || N % 256 != 0 || K % 32 != 0 || K < 32) return 0;
    const int tiles = (M / (32 * mb)) * (N / 256);
    if (!want_split) return tiles >= 128 ? 1 : 0;
    int best = 0;
    for (int s = 1; s <= 64; ++s) {
        if (K % (32 * s) != 0 || K / s < 256) continue;
        if (tiles * s > 320) break;
        if (tiles * s >= 128) best = s;                        // the largest split that stays within one round (and a quarter)
        if (tiles * s >= 256) break;
    }
    return best;
}
size_t sgemm_planes_ws_bytes(int M, int N, int K) { return (size_t)4 * K * ((size_t)M + N) + 1024; }

static int pl_split_operand(const float* src, int64_t s_row, int64_t s_k, int rows, int K, const float* amax, int n_amax, _Float16* hi,
                            _Float16* lo, float* scale, hipStream_t st) {
    SplitArgs a{src, s_row, s_k, rows, K, amax, n_amax, hi, lo, scale};
    if (s_k == 1 && s_row % 4 == 0 && K % 8 == 0) {
        const int64_t oct = (int64_t)rows * (K / 8);
        const int blocks = (int)((oct + 255) / 256 < 4096 ? (oct + 255) / 256 : 4096);
        hipLaunchKernelGGL(split_planes_direct_kernel, dim3(blocks), dim3(256), 0, st, a);
        return RULGNN_OK;
    }
    if (s_row == 1 && s_k % 4 == 0 && rows % 64 == 0 && K % 64 == 0) {
        const int tiles = (rows / 64) * (K / 64);
        hipLaunchKernelGGL(split_planes_transposed_kernel, dim3(tiles < 4096 ? tiles : 4096), dim3(256), 0, st, a);
        return RULGNN_OK;
    }
    return RULGNN_EUNSUPPORTED;
}

// whether sgemm_planes() takes this product (shape, layout, alignment); `slices` from sgemm_planes_slices
bool sgemm_planes_ok(const float* A, int64_t sAm, int64_t sAk, const float* B, int64_t sBn, int64_t sBk, int M, int N, int K, int slices) {
    if (slices < 1 || pl_pick_mb(M) == 0 || N % 256 != 0 || K % (32 * slices) != 0) return false;
    if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) return false;
    auto lay = [](int64_t s_row, int64_t s_k, int rows, int K2) {
        return (s_k == 1 && s_row % 4 == 0 && K2 % 8 == 0) || (s_row == 1 && s_k % 4 == 0 && rows % 64 == 0 && K2 % 64 == 0);
    };
    return lay(sAm, sAk, M, K) && lay(sBn, sBk, N, K);
}

// C (slice z at C + z M ldc) (+)= the product, operands split into `ws` first.  The caller has checked sgemm_planes_ok.
// where a producer that splits operand A itself writes: planes [M][K] (k contiguous) and the power of two it multiplied by
void sgemm_planes_a_slots(void* ws, int M, int K, void** hi, void** lo, float** scale) {
    unsigned char* w = static_cast<unsigned char*>(ws);
    *scale = reinterpret_cast<float*>(w);
    *hi = w + 1024;
    *lo = w + 1024 + (size_t)M * K * sizeof(_Float16);
}
// (`a_presplit`: operand A already sits in `ws` as planes -- written by the kernel that produced it, sgemm_planes_a_slots -- and A /
// amax_a are ignored)
int sgemm_planes(const float* A, int64_t sAm, int64_t sAk, const float* B, int64_t sBn, int64_t sBk, float* C, int64_t ldc, int M, int N, int K,
                 bool accumulate, int slices, const float* amax_a, int amax_na, const float* amax_b, int amax_nb, void* ws, size_t ws_bytes,
                 hipStream_t st, bool a_presplit) {
    if (ws_bytes < sgemm_planes_ws_bytes(M, N, K)) return RULGNN_EWORKSPACE;
    unsigned char* w = static_cast<unsigned char*>(ws);
    float* scales = reinterpret_cast<float*>(w);
    _Float16* Ahi = reinterpret_cast<_Float16*>(w + 1024);
    _Float16* Alo = Ahi + (size_t)M * K;
    _Float16* Bhi = Alo + (size_t)M * K;
    _Float16* Blo = Bhi + (size_t)N * K;
    (void)hipGetLastError();
    int rc = a_presplit ? RULGNN_OK : pl_split_operand(A, sAm, sAk, M, K, amax_a, amax_na, Ahi, Alo, scales, st);
    if (rc != RULGNN_OK) return rc;
    rc = pl_split_operand(B, sBn, sBk, N, K, amax_b, amax_nb, Bhi, Blo, scales + 64, st);
    if (rc != RULGNN_OK) return rc;
    const int mb = pl_pick_mb(M);
    PlaneGemmArgs g{Ahi, Alo, Bhi, Blo, scales, scales + 64, C, ldc, M, N, K, K / slices, M / (32 * mb), N / 256, slices, accumulate ? 1 : 0};
    const int grid = g.tiles_m * g.tiles_n * slices;
    auto go = [&](auto kernel, size_t lds) {
        static bool raised = false;                              // once per instantiation and process
        if (!raised) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            raised = true;
        }
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(512), lds, st, g);
    };
    if (mb == 5) go(sgemm_planes_kernel<5>, (size_t)3 * (2 * 160 * 64 + 2 * 256 * 64));
    else go(sgemm_planes_kernel<4>, (size_t)3 * (2 * 128 * 64 + 2 * 256 * 64));
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

}  // namespace rulgnn
