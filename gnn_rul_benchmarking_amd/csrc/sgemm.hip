// SGEMM on the gfx950 matrix cores, shared by every family (interface: sgemm_mfma.hpp).  ONE translation unit: the kernels used
// to live in the header and were instantiated in every unit that included it (most of the library size and build time).
#include "sgemm_mfma.hpp"
#include "reduce_device.hpp"

namespace rulgnn {

// ------------------------------------------------------------------------------------------------
// SGEMM on the matrix cores: C[m][n] (+)= sum_k A(m,k) * B(n,k), generic strides, fp32 MFMA 16x16x4.
// 64x64 block tile, K step 16, 4 wavefronts each owning a 32x32 quadrant (2x2 MFMA tiles).
// ------------------------------------------------------------------------------------------------
struct GemmArgs {
    const float* A; int64_t sAm, sAk;
    const float* B; int64_t sBn, sBk;
    float* C; int64_t ldc;
    int M, N, K;
    int accumulate;      // C += instead of C =
    int kchunk;          // split-K: blockIdx.z owns k in [z*kchunk, (z+1)*kchunk) and writes slice z of C (stride M*ldc)
    // optional (both or neither): partial maxima of |A| and |B| over their finite elements (one float per workgroup of the kernels that
    // wrote the operands; their maximum is the tensor's).  With them the 256 x 256 kernel runs the two-plane f16 split
    // (sgemm_f16x2v_kernel) instead of the three-plane bf16 one.
    const float* amax_a;
    const float* amax_b;
    int amax_na, amax_nb;
};

// (bx, by, bz: the tile's column / row index and its k slice -- blockIdx of the plain launch, decoded from a linear index by the batched one)
static __device__ __forceinline__ void sgemm_mfma_body(GemmArgs g, int bx, int by, int bz) {
    __shared__ float As[16][64 + 4];      // [k][m]
    __shared__ float Bs[16][64 + 4];      // [k][n]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = by * 64, n0 = bx * 64;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    f32x4t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4t){0.f, 0.f, 0.f, 0.f};
    const int li = lane & 15, kq = lane >> 4;
    const int kbeg = bz * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    g.C += (int64_t)bz * g.M * g.ldc;
    // Cooperative tile load: 64 x 16 elements of A and of B (4 + 4 per thread).  The lane -> element mapping follows the
    // operand's contiguous dimension (k-fastest when the k stride is 1, else m-fastest) so that wavefront loads coalesce, and
    // the next k-step's elements are fetched into registers while the matrix cores work on the current one.
    const bool a_kfast = g.sAk == 1, b_kfast = g.sBk == 1;
    int am[4], ak[4], bm[4], bk[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int idx = tid + e * 256;
        am[e] = a_kfast ? idx >> 4 : idx & 63;
        ak[e] = a_kfast ? idx & 15 : idx >> 6;
        bm[e] = b_kfast ? idx >> 4 : idx & 63;
        bk[e] = b_kfast ? idx & 15 : idx >> 6;
    }
    float ra[4], rb[4];
    // (unconditional loads on clamped indices; the mask is applied when the values go to LDS, one iteration later.  As `ok ? p[..] : 0` each
    // of the eight loads sat in its own branch with waits between them, and with the select right behind the load the compiler waited for
    // the next step's operands BEFORE the current step's products instead of behind them)
    auto fetch = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int gm = m0 + am[e], gka = k0 + ak[e], gn = n0 + bm[e], gkb = k0 + bk[e];
            ra[e] = g.A[(gm < g.M ? gm : g.M - 1) * g.sAm + (gka < kend ? gka : kend - 1) * g.sAk];
            rb[e] = g.B[(gn < g.N ? gn : g.N - 1) * g.sBn + (gkb < kend ? gkb : kend - 1) * g.sBk];
        }
    };
    if (kbeg < kend) fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            As[ak[e]][am[e]] = (m0 + am[e] < g.M && k0 + ak[e] < kend) ? ra[e] : 0.f;
            Bs[bk[e]][bm[e]] = (n0 + bm[e] < g.N && k0 + bk[e] < kend) ? rb[e] : 0.f;
        }
        __syncthreads();
        if (k0 + 16 < kend) fetch(k0 + 16);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = As[4 * ks + kq][wm + 16 * i + li];
                b[i] = Bs[4 * ks + kq][wn + 16 * i + li];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // D layout: lane (g4 = lane>>4, col = lane&15), reg r -> row 4*g4 + r
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gm = m0 + wm + 16 * i + 4 * kq + r, gn = n0 + wn + 16 * j + li;
                if (gm < g.M && gn < g.N) {
                    float* c = g.C + (int64_t)gm * g.ldc + gn;
                    *c = g.accumulate ? *c + acc[i][j][r] : acc[i][j][r];
                }
            }
}
static __global__ __launch_bounds__(256) void sgemm_mfma_kernel(GemmArgs g) { sgemm_mfma_body(g, blockIdx.x, blockIdx.y, blockIdx.z); }
// several split-K products of the 64 x 64 tile kernel as ONE launch (blockIdx.x: the jobs' tiles x slices back to back) and their slice
// reductions as one more: the parameter gradients of a small model are five to ten launch pairs of 5-8 us each, at their latency floor
struct GemmBatch {
    GemmArgs g[GEMM_BATCH_MAX];          // C = the job's partial buffer, ldc = N, kchunk set
    int first[GEMM_BATCH_MAX + 1];       // first workgroup of each job
    int nx[GEMM_BATCH_MAX], ny[GEMM_BATCH_MAX];
    int n;
};
static __global__ __launch_bounds__(256) void sgemm_mfma_batch_kernel(GemmBatch b) {
    int j = 0;
#pragma unroll
    for (int q = 1; q < GEMM_BATCH_MAX; ++q)
        if (q < b.n && (int)blockIdx.x >= b.first[q]) j = q;
    const int local = blockIdx.x - b.first[j];
    const int bx = local % b.nx[j], by = (local / b.nx[j]) % b.ny[j], bz = local / (b.nx[j] * b.ny[j]);
    sgemm_mfma_body(b.g[j], bx, by, bz);
}

// ------------------------------------------------------------------------------------------------
// Large outputs (the hidden-1000 layers of SAGCN, the Chebyshev layers of STNet, the tiled ST_GCN path): 128x128 block tile, K step
// 16, 4 wavefronts each owning a 64x64 quadrant (4x4 MFMA tiles, 64 accumulator registers).  The 64x64 kernel above moves 8 KB
// through LDS per 131 kFLOP -- 16 FLOP per byte, i.e. the fp32 matrix peak would need ~10 TB/s out of L2 --; this one 32 FLOP per byte,
// with 16-byte global loads along whichever dimension of the operand is contiguous (template flags), the next K step's loads in
// flight under the 64 MFMAs of the current one, double-buffered LDS (one barrier per K step) and an LDS row stride of 128 + 16 floats
// (the four k-rows a wavefront's operand read touches land in disjoint bank halves).  Same k order per accumulator as the 64x64
// kernel.  Requirements (checked by sgemm_big_ok): each operand contiguous along k or along its row index, 16-byte aligned base,
// the other stride a multiple of 4.
// ------------------------------------------------------------------------------------------------
template <bool A_KFAST, bool B_KFAST, int KT>
static __global__ __launch_bounds__(256, 2) void sgemm_mfma128_kernel(GemmArgs g) {
    constexpr int LD = 128 + 16;
    extern __shared__ __attribute__((aligned(16))) float sgemm_big_lds[];
    float (*As)[KT][LD] = reinterpret_cast<float (*)[KT][LD]>(sgemm_big_lds);                    // [buffer][k][m]
    float (*Bs)[KT][LD] = reinterpret_cast<float (*)[KT][LD]>(sgemm_big_lds + 2 * KT * LD);      // [buffer][k][n]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    f32x4t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4t){0.f, 0.f, 0.f, 0.f};
    const int li = lane & 15, kq = lane >> 4;
    const int kbeg = blockIdx.z * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    g.C += (int64_t)blockIdx.z * g.M * g.ldc;
    constexpr int NF = KT / 8;          // float4 per thread and operand: 128 x KT floats over 256 threads
    f32x4t ra[NF], rb[NF];
    // one operand tile = 128 x 16 floats = 512 float4, two per thread: along k (row = idx >> 2, k = 4 (idx & 3)) when k is the
    // contiguous dimension, else along the row index (k = idx >> 5, row = 4 (idx & 31))
    auto fetch_one = [&](const float* __restrict__ P, int64_t s_row, int64_t s_k, int rows, int r0, int k0, int idx, bool kfast) -> f32x4t {
        f32x4t v = {0.f, 0.f, 0.f, 0.f};
        if (kfast) {
            const int r = r0 + idx / (KT / 4), k = k0 + 4 * (idx % (KT / 4));
            if (r < rows) {
                const float* p = P + (int64_t)r * s_row + k;
                if (k + 3 < kend) v = *reinterpret_cast<const f32x4t*>(p);
                else {
                    if (k < kend) v[0] = p[0];
                    if (k + 1 < kend) v[1] = p[1];
                    if (k + 2 < kend) v[2] = p[2];
                }
            }
        } else {
            const int k = k0 + (idx >> 5), r = r0 + 4 * (idx & 31);
            if (k < kend) {
                const float* p = P + (int64_t)k * s_k + r;
                if (r + 3 < rows) v = *reinterpret_cast<const f32x4t*>(p);
                else {
                    if (r < rows) v[0] = p[0];
                    if (r + 1 < rows) v[1] = p[1];
                    if (r + 2 < rows) v[2] = p[2];
                }
            }
        }
        return v;
    };
    auto fetch = [&](int k0) {
#pragma unroll
        for (int e = 0; e < NF; ++e) {
            ra[e] = fetch_one(g.A, g.sAm, g.sAk, g.M, m0, k0, tid + e * 256, A_KFAST);
            rb[e] = fetch_one(g.B, g.sBn, g.sBk, g.N, n0, k0, tid + e * 256, B_KFAST);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int e = 0; e < NF; ++e) {
            const int idx = tid + e * 256;
            if (A_KFAST) {
#pragma unroll
                for (int j = 0; j < 4; ++j) As[buf][4 * (idx % (KT / 4)) + j][idx / (KT / 4)] = ra[e][j];
            } else {
                *reinterpret_cast<f32x4t*>(&As[buf][idx >> 5][4 * (idx & 31)]) = ra[e];
            }
            if (B_KFAST) {
#pragma unroll
                for (int j = 0; j < 4; ++j) Bs[buf][4 * (idx % (KT / 4)) + j][idx / (KT / 4)] = rb[e][j];
            } else {
                *reinterpret_cast<f32x4t*>(&Bs[buf][idx >> 5][4 * (idx & 31)]) = rb[e];
            }
        }
    };
    int buf = 0;
    if (kbeg < kend) {
        fetch(kbeg);
        stash(0);
    }
    __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += KT) {
        const bool more = k0 + KT < kend;
        if (more) fetch(k0 + KT);
#pragma unroll
        for (int ks = 0; ks < KT / 4; ++ks) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = As[buf][4 * ks + kq][wm + 16 * i + li];
                b[i] = Bs[buf][4 * ks + kq][wn + 16 * i + li];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) stash(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gm = m0 + wm + 16 * i + 4 * kq + r, gn = n0 + wn + 16 * j + li;
                if (gm < g.M && gn < g.N) {
                    float* c = g.C + (int64_t)gm * g.ldc + gn;
                    *c = g.accumulate ? *c + acc[i][j][r] : acc[i][j][r];
                }
            }
}

typedef __bf16 gemm_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned gemm_u32x4 __attribute__((ext_vector_type(4)));
// ------------------------------------------------------------------------------------------------
// The same 128x128 tile on the bf16 matrix cores with fp32-class accuracy ("bf16 x 3").  Every fp32 operand is split EXACTLY into
// three bf16 parts, x = h + m + l (h = the top 8 significant bits, m = the top 8 of x - h, l = x - h - m: 24 bits in all; the
// subtractions are exact in fp32), when its tile is written to LDS -- once per element, amortised over the 128 products it takes part
// in.  a b = ah bh + ah bm + am bh + ah bl + am bm + al bh + (terms below 2^-23 |a b|): six v_mfma_f32_32x32x16_bf16 with fp32
// accumulation, each bf16 x bf16 product exact.  The error per product is that of ONE fp32 rounding; the rate is a sixth of the bf16
// matrix peak = 2.6x the fp32 matrix peak (v_mfma_f32_16x16x4_f32 runs at the VALU FMA rate).  bf16 keeps fp32's exponent range, so no
// scaling is needed (an f16 split would need two parts and three products but underflows on gradient-sized values).
// LDS: per buffer three bf16 planes per operand, rows of 16 k = 32 bytes padded to 48 (eight consecutive rows x 16-byte reads cover
// the 32 banks exactly once); a lane's MFMA operand is one ds_read_b128.
// ------------------------------------------------------------------------------------------------
typedef float f32x16t __attribute__((ext_vector_type(16)));

// one fp32 pair -> one dword (first value in the low half) of each of the three planes
static __device__ __forceinline__ void split_pair_bf16x3(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned ha = __builtin_bit_cast(unsigned, a) & 0xFFFF0000u, hb = __builtin_bit_cast(unsigned, b) & 0xFFFF0000u;
    const float ra = a - __builtin_bit_cast(float, ha), rb = b - __builtin_bit_cast(float, hb);
    const unsigned ma = __builtin_bit_cast(unsigned, ra) & 0xFFFF0000u, mb = __builtin_bit_cast(unsigned, rb) & 0xFFFF0000u;
    const float la = ra - __builtin_bit_cast(float, ma), lb = rb - __builtin_bit_cast(float, mb);
    h = __builtin_amdgcn_perm(hb, ha, 0x07060302u);
    m = __builtin_amdgcn_perm(mb, ma, 0x07060302u);
    l = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, lb), __builtin_bit_cast(unsigned, la), 0x07060302u);
}

// LDS layouts of one operand plane (6 KB reserved each):
//   operand contiguous along k   : [row][16 k] bf16, rows padded to 48 bytes; a thread's float4 (4 k of one row) is one 8-byte store,
//                                  a lane's MFMA operand (8 k of its row) one 16-byte read;
//   operand contiguous along rows: [k pair][128 rows] dwords (k even in the low half); a thread holds 4 rows x 2 k (two float4, k and
//                                  k + 1) = one 16-byte store, a lane's MFMA operand four 4-byte reads, consecutive lanes consecutive
//                                  dwords.  (Two-byte stores into the row-major form were 16-way bank conflicts: 56 TFLOP/s.)
template <bool A_KFAST, bool B_KFAST, bool GUARD>
static __device__ __forceinline__ void sgemm_bf16x3_body(GemmArgs g) {
    constexpr int ROWB = 48;                       // bytes per LDS row of the k-contiguous form
    constexpr int PLANE = 128 * ROWB;              // one bf16 plane of one operand tile
    constexpr int BUF = 6 * PLANE;                 // A: h, m, l ; B: h, m, l
    extern __shared__ __attribute__((aligned(16))) unsigned char sgemm_x3_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    f32x16t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int kbeg = blockIdx.z * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    g.C += (int64_t)blockIdx.z * g.M * g.ldc;
    f32x4t ra[2], rb[2];
    // k-contiguous: float4 e of a thread = row (tid + 256 e) >> 2, k = 4 ((tid + 256 e) & 3)
    // row-contiguous: float4 e of a thread = rows 4 (tid & 31) .. + 3 at k = 2 (tid >> 5) + e
    auto fetch_one = [&](const float* __restrict__ P, int64_t s_row, int64_t s_k, int rows, int r0, int k0, int e, bool kfast) -> f32x4t {
        f32x4t v = {0.f, 0.f, 0.f, 0.f};
        if (!GUARD && k0 + 16 <= kend) {          // interior tile, whole K step (wave-uniform): every 16-byte load is in bounds
            if (kfast) {
                const int idx = tid + e * 256;
                return *reinterpret_cast<const f32x4t*>(P + (int64_t)(r0 + (idx >> 2)) * s_row + k0 + 4 * (idx & 3));
            }
            return *reinterpret_cast<const f32x4t*>(P + (int64_t)(k0 + 2 * (tid >> 5) + e) * s_k + r0 + 4 * (tid & 31));
        }
        if (kfast) {
            const int idx = tid + e * 256;
            const int r = r0 + (idx >> 2), k = k0 + 4 * (idx & 3);
            if (r < rows) {
                const float* p = P + (int64_t)r * s_row + k;
                if (k + 3 < kend) v = *reinterpret_cast<const f32x4t*>(p);
                else {
                    if (k < kend) v[0] = p[0];
                    if (k + 1 < kend) v[1] = p[1];
                    if (k + 2 < kend) v[2] = p[2];
                }
            }
        } else {
            const int k = k0 + 2 * (tid >> 5) + e, r = r0 + 4 * (tid & 31);
            if (k < kend) {
                const float* p = P + (int64_t)k * s_k + r;
                if (r + 3 < rows) v = *reinterpret_cast<const f32x4t*>(p);
                else {
                    if (r < rows) v[0] = p[0];
                    if (r + 1 < rows) v[1] = p[1];
                    if (r + 2 < rows) v[2] = p[2];
                }
            }
        }
        return v;
    };
    auto fetch = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            ra[e] = fetch_one(g.A, g.sAm, g.sAk, g.M, m0, k0, e, A_KFAST);
            rb[e] = fetch_one(g.B, g.sBn, g.sBk, g.N, n0, k0, e, B_KFAST);
        }
    };
    auto stash_one = [&](unsigned char* base, const f32x4t (&v)[2], bool kfast) {
        if (kfast) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int idx = tid + e * 256;
                const float x0 = v[e][0], x1 = v[e][1], x2 = v[e][2], x3 = v[e][3];
                unsigned h0, m0_, l0, h1, m1, l1;
                split_pair_bf16x3(x0, x1, h0, m0_, l0);
                split_pair_bf16x3(x2, x3, h1, m1, l1);
                unsigned char* p = base + (idx >> 2) * ROWB + 8 * (idx & 3);
                *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(p + PLANE) = make_uint2(m0_, m1);
                *reinterpret_cast<uint2*>(p + 2 * PLANE) = make_uint2(l0, l1);
            }
        } else {
            unsigned h[4], m[4], l[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x0 = v[0][j], x1 = v[1][j];
                split_pair_bf16x3(x0, x1, h[j], m[j], l[j]);
            }
            unsigned char* p = base + ((tid >> 5) * 128 + 4 * (tid & 31)) * 4;
            *reinterpret_cast<uint4*>(p) = make_uint4(h[0], h[1], h[2], h[3]);
            *reinterpret_cast<uint4*>(p + PLANE) = make_uint4(m[0], m[1], m[2], m[3]);
            *reinterpret_cast<uint4*>(p + 2 * PLANE) = make_uint4(l[0], l[1], l[2], l[3]);
        }
    };
    auto stash = [&](int buf) {
        unsigned char* b = sgemm_x3_lds + buf * BUF;
        stash_one(b, ra, A_KFAST);
        stash_one(b + 3 * PLANE, rb, B_KFAST);
    };
    // the MFMA operand of this lane: 8 consecutive k (k half lane >> 5) of row `row0 + (lane & 31)` of one plane
    auto operand = [&](const unsigned char* plane, int row0, bool kfast) -> gemm_bf16x8 {
        if (kfast) return *reinterpret_cast<const gemm_bf16x8*>(plane + (row0 + (lane & 31)) * ROWB + (lane >> 5) * 16);
        const unsigned* q = reinterpret_cast<const unsigned*>(plane) + (4 * (lane >> 5)) * 128 + row0 + (lane & 31);
        const gemm_u32x4 v = {q[0], q[128], q[256], q[384]};
        return __builtin_bit_cast(gemm_bf16x8, v);
    };
    // Software pipeline, two K steps deep: while the matrix cores work on tile k (LDS buffer `buf`), the registers loaded during the
    // PREVIOUS iteration (tile k + 1: a full iteration of latency cover) are split and written to the other buffer, and the loads of
    // tile k + 2 are issued.  The split's VALU work is interleaved with the MFMAs by the scheduling hints at the end of the body: an
    // in-order wavefront hides ~5 other instructions behind each 32-cycle MFMA, or none at all if they sit behind the whole chain.
    int buf = 0;
    if (kbeg < kend) {
        fetch(kbeg);
        stash(0);
        fetch(kbeg + 16);                     // tile 1 (past the end the guarded loads return zeros)
    }
    __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
        const bool more = k0 + 16 < kend;
        f32x4t na[2], nb[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) { na[e] = ra[e]; nb[e] = rb[e]; }          // tile k + 1, loaded one iteration ago
        if (k0 + 32 < kend) fetch(k0 + 32);                                      // tile k + 2 into ra / rb
        const unsigned char* b = sgemm_x3_lds + buf * BUF;
        gemm_bf16x8 a[2][3], bb[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a[i][p] = operand(b + p * PLANE, wm + 32 * i, A_KFAST);
                bb[i][p] = operand(b + (3 + p) * PLANE, wn + 32 * i, B_KFAST);
            }
        // the six product terms, smallest first; consecutive MFMAs go to different accumulators (independent: back-to-back issue)
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[t]], bb[j][PB[t]], acc[i][j], 0, 0, 0);
        if (more) {
            unsigned char* nbuf = sgemm_x3_lds + (buf ^ 1) * BUF;
            stash_one(nbuf, na, A_KFAST);
            stash_one(nbuf + 3 * PLANE, nb, B_KFAST);
        }
#pragma unroll
        for (int q = 0; q < 24; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);      // five VALU (the split)
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // one LDS write
        }
        __syncthreads();
        buf ^= 1;
    }
    // D layout of the 32x32 result: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), gn = n0 + wn + 32 * j + (lane & 31);
                if (!GUARD || (gm < g.M && gn < g.N)) {
                    float* c = g.C + (int64_t)gm * g.ldc + gn;
                    *c = g.accumulate ? *c + acc[i][j][r] : acc[i][j][r];
                }
            }
}

template <bool A_KFAST, bool B_KFAST>
static __global__ __launch_bounds__(256, 2) void sgemm_bf16x3_kernel(GemmArgs g) {
    // interior tiles take the body whose whole K steps load without bounds checks (a third of its non-MFMA instructions were guards)
    const int kbeg = blockIdx.z * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    const bool interior = (int)blockIdx.y * 128 + 128 <= g.M && (int)blockIdx.x * 128 + 128 <= g.N && kend > kbeg;
    if (interior) sgemm_bf16x3_body<A_KFAST, B_KFAST, false>(g);
    else sgemm_bf16x3_body<A_KFAST, B_KFAST, true>(g);
}

// ------------------------------------------------------------------------------------------------
// The same arithmetic on a 256x256 block tile.  SQ counters of the 128x128 kernel: per K step a wavefront issues 154 VALU + 37
// scalar + 20 LDS instructions for its 24 MFMAs -- 1044 issue cycles against 768 MFMA cycles, two wavefronts per SIMD: the split is
// re-done by every tile that loads an element.  Doubling both tile dimensions halves the elements loaded (and split) per MFMA; a
// thread serves ONE operand (wavefronts 0-3: A, 4-7: B).  First built with 16 wavefronts of 64 x 64 (193 TFLOP/s at 4096^3): that
// form is bound by LDS READ bandwidth -- 12 KB of operand planes per 24 MFMAs = 512 B per MFMA, 64 B/clk per CU of the LDS's 128 B/clk
// before bank conflicts (a variant that split every operand once, in a pre-pass, ran no faster).  This one has 8 wavefronts of
// 64 x 128 (128 accumulator registers): 18 KB per 48 MFMAs = 384 B per MFMA, 203 TFLOP/s.
// ------------------------------------------------------------------------------------------------
template <bool A_KFAST, bool B_KFAST, bool GUARD>
static __device__ __forceinline__ void sgemm_bf16x3v_body(GemmArgs g) {
    constexpr int T = 256;                         // tile rows / columns
    constexpr int ROWB = 48;
    constexpr int PLANE = T * ROWB;                // 12 KB
    constexpr int BUF = 6 * PLANE;                 // 72 KB
    extern __shared__ __attribute__((aligned(16))) unsigned char sgemm_x3_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * T, n0 = blockIdx.x * T;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 128;          // 8 wavefronts, each 64 rows x 128 columns
    f32x16t acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int kbeg = blockIdx.z * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    g.C += (int64_t)blockIdx.z * g.M * g.ldc;
    // this thread's operand
    const bool mine_b = tid >= 256;
    const int t = tid & 255;
    const float* __restrict__ P = mine_b ? g.B : g.A;
    const int64_t s_row = mine_b ? g.sBn : g.sAm, s_k = mine_b ? g.sBk : g.sAk;
    const int rows = mine_b ? g.N : g.M, r0 = mine_b ? n0 : m0;
    const bool kfast = mine_b ? B_KFAST : A_KFAST;                  // wave-uniform
    f32x4t rr[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f32x4t v = {0.f, 0.f, 0.f, 0.f};
            if (kfast) {
                const int idx = t + e * 256;
                const int r = r0 + (idx >> 2), k = k0 + 4 * (idx & 3);
                const float* p = P + (int64_t)r * s_row + k;
                if (!GUARD && k0 + 16 <= kend) v = *reinterpret_cast<const f32x4t*>(p);
                else if (r < rows) {
                    if (k + 3 < kend) v = *reinterpret_cast<const f32x4t*>(p);
                    else {
                        if (k < kend) v[0] = p[0];
                        if (k + 1 < kend) v[1] = p[1];
                        if (k + 2 < kend) v[2] = p[2];
                    }
                }
            } else {
                const int k = k0 + 4 * (t >> 6) + e, r = r0 + 4 * (t & 63);
                const float* p = P + (int64_t)k * s_k + r;
                if (!GUARD && k0 + 16 <= kend) v = *reinterpret_cast<const f32x4t*>(p);
                else if (k < kend) {
                    if (r + 3 < rows) v = *reinterpret_cast<const f32x4t*>(p);
                    else {
                        if (r < rows) v[0] = p[0];
                        if (r + 1 < rows) v[1] = p[1];
                        if (r + 2 < rows) v[2] = p[2];
                    }
                }
            }
            rr[e] = v;
        }
    };
    auto stash = [&](unsigned char* bufp, const f32x4t (&v)[4]) {
        unsigned char* base = bufp + (mine_b ? 3 * PLANE : 0);
        if (kfast) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int idx = t + e * 256;
                const float x0 = v[e][0], x1 = v[e][1], x2 = v[e][2], x3 = v[e][3];
                unsigned h0, m0_, l0, h1, m1, l1;
                split_pair_bf16x3(x0, x1, h0, m0_, l0);
                split_pair_bf16x3(x2, x3, h1, m1, l1);
                unsigned char* p = base + (idx >> 2) * ROWB + 8 * (idx & 3);
                *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(p + PLANE) = make_uint2(m0_, m1);
                *reinterpret_cast<uint2*>(p + 2 * PLANE) = make_uint2(l0, l1);
            }
        } else {
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {                       // two k pairs per thread: (4 kq, 4 kq + 1) and (4 kq + 2, 4 kq + 3)
                unsigned h[4], m[4], l[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float x0 = v[2 * pp][j], x1 = v[2 * pp + 1][j];
                    split_pair_bf16x3(x0, x1, h[j], m[j], l[j]);
                }
                unsigned char* p = base + ((2 * (t >> 6) + pp) * T + 4 * (t & 63)) * 4;
                *reinterpret_cast<uint4*>(p) = make_uint4(h[0], h[1], h[2], h[3]);
                *reinterpret_cast<uint4*>(p + PLANE) = make_uint4(m[0], m[1], m[2], m[3]);
                *reinterpret_cast<uint4*>(p + 2 * PLANE) = make_uint4(l[0], l[1], l[2], l[3]);
            }
        }
    };
    auto operand = [&](const unsigned char* plane, int row0, bool kf) -> gemm_bf16x8 {
        if (kf) return *reinterpret_cast<const gemm_bf16x8*>(plane + (row0 + (lane & 31)) * ROWB + (lane >> 5) * 16);
        const unsigned* q = reinterpret_cast<const unsigned*>(plane) + (4 * (lane >> 5)) * T + row0 + (lane & 31);
        const gemm_u32x4 v = {q[0], q[T], q[2 * T], q[3 * T]};
        return __builtin_bit_cast(gemm_bf16x8, v);
    };
    int buf = 0;
    if (kbeg < kend) {
        fetch(kbeg);
        stash(sgemm_x3_lds, rr);
        fetch(kbeg + 16);
    }
    __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
        const bool more = k0 + 16 < kend;
        f32x4t nn[4] = {rr[0], rr[1], rr[2], rr[3]};                                         // tile k + 1, loaded one iteration ago
        if (k0 + 32 < kend) fetch(k0 + 32);
        const unsigned char* b = sgemm_x3_lds + buf * BUF;
        gemm_bf16x8 a[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) a[i][p] = operand(b + p * PLANE, wm + 32 * i, A_KFAST);
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            gemm_bf16x8 bb[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) bb[p] = operand(b + (3 + p) * PLANE, wn + 32 * j, B_KFAST);
#pragma unroll
            for (int tt = 0; tt < 6; ++tt)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[tt]], bb[PB[tt]], acc[i][j], 0, 0, 0);
        }
        if (more) stash(sgemm_x3_lds + (buf ^ 1) * BUF, nn);
#pragma unroll
        for (int q = 0; q < 48; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), gn = n0 + wn + 32 * j + (lane & 31);
                if (!GUARD || (gm < g.M && gn < g.N)) {
                    float* c = g.C + (int64_t)gm * g.ldc + gn;
                    *c = g.accumulate ? *c + acc[i][j][r] : acc[i][j][r];
                }
            }
}

template <bool A_KFAST, bool B_KFAST>
static __global__ __launch_bounds__(512) void sgemm_bf16x3v_kernel(GemmArgs g) {
    const int kbeg = blockIdx.z * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    const bool interior = (int)blockIdx.y * 256 + 256 <= g.M && (int)blockIdx.x * 256 + 256 <= g.N && kend > kbeg;
    if (interior) sgemm_bf16x3v_body<A_KFAST, B_KFAST, false>(g);
    else sgemm_bf16x3v_body<A_KFAST, B_KFAST, true>(g);
}

// ------------------------------------------------------------------------------------------------
// The 256 x 256 kernel on TWO f16 planes per operand: a = hi + lo with hi = the top 11 significant bits of a s (exact in f16) and
// lo = f16(a s - hi); a b = (hi hi + hi lo + lo hi) / (sa sb) + terms below 2^-22 |a b| -- three v_mfma_f32_32x32x16_f16 per fp32
// product instead of the six of the bf16 split, the arithmetic of the fused ST_GCN kernels (stgcn_mx.hpp).  f16 has five exponent bits:
// the caller passes max |A| and max |B| (GemmArgs::amax_*, produced by the kernels that wrote the operands) and each operand is scaled
// by a power of two so that its largest element lands in [2^11, 2^12); an element more than 2^15 below the largest keeps fewer than 22
// bits (its lo part goes subnormal: absolute error 2^-37 of the largest) -- invisible in a product that also contains the large ones,
// ~4e-5 relative in an output only such elements touch, and the reason this form is opt-in: the generic entry points keep the
// range-free bf16 split.  Same tiles, LDS layout (two planes: 96 KB double-buffered) and pipeline.
// ------------------------------------------------------------------------------------------------
typedef _Float16 gemm_f16x8 __attribute__((ext_vector_type(8)));
static __device__ __forceinline__ float sgemm_f16_scale_of(float amax) {
    const unsigned m = __builtin_bit_cast(unsigned, amax);
    const int e = (int)((m >> 23) & 0xFFu);                          // biased exponent of the largest finite magnitude
    if (!(amax > 0.f) || e == 0 || e == 255) return 1.0f;
    int se = 127 + 11 - (e - 127);
    se = se < 1 ? 1 : (se > 254 ? 254 : se);
    return __builtin_bit_cast(float, (unsigned)se << 23);
}
// the two operand scales from the producers' partial maxima (256 or 512 threads; a few thousand floats out of L2: ~1 us per workgroup)
static __device__ __forceinline__ void sgemm_f16_scales(const GemmArgs& g, float& sa, float& sb) {
    __shared__ float part[2][8];
    float ma = 0.f, mb = 0.f;
    const int nthr = blockDim.x, nwave = blockDim.x >> 6;
    for (int i = threadIdx.x; i < g.amax_na; i += nthr) ma = fmaxf(ma, g.amax_a[i]);
    for (int i = threadIdx.x; i < g.amax_nb; i += nthr) mb = fmaxf(mb, g.amax_b[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ma = fmaxf(ma, __shfl_xor(ma, o, 64));
        mb = fmaxf(mb, __shfl_xor(mb, o, 64));
    }
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = ma; part[1][threadIdx.x >> 6] = mb; }
    __syncthreads();
    ma = mb = 0.f;
    for (int w = 0; w < nwave; ++w) { ma = fmaxf(ma, part[0][w]); mb = fmaxf(mb, part[1][w]); }
    sa = sgemm_f16_scale_of(ma);
    sb = sgemm_f16_scale_of(mb);
}
// one fp32 pair (already scaled) -> one dword (first value in the low half) of each of the two planes
static __device__ __forceinline__ void split_pair_f16x2(float a, float b, unsigned& h, unsigned& l) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const float ha = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a) & 0xFFFFE000u);
    const float hb = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, b) & 0xFFFFE000u);
    const f2 hv = {ha, hb}, lv = {a - ha, b - hb};
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(hv, h2));
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(lv, h2));
}
template <bool A_KFAST, bool B_KFAST, bool GUARD>
static __device__ __forceinline__ void sgemm_f16x2v_body(GemmArgs g) {
    constexpr int T = 256;                         // tile rows / columns
    constexpr int ROWB = 48;
    constexpr int PLANE = T * ROWB;                // 12 KB
    constexpr int BUF = 4 * PLANE;                 // 48 KB: A hi, lo ; B hi, lo
    extern __shared__ __attribute__((aligned(16))) unsigned char sgemm_x3_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * T, n0 = blockIdx.x * T;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 128;          // 8 wavefronts, each 64 rows x 128 columns
    f32x16t acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int kbeg = blockIdx.z * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    g.C += (int64_t)blockIdx.z * g.M * g.ldc;
    // this thread's operand
    const bool mine_b = tid >= 256;
    const int t = tid & 255;
    const float* __restrict__ P = mine_b ? g.B : g.A;
    const int64_t s_row = mine_b ? g.sBn : g.sAm, s_k = mine_b ? g.sBk : g.sAk;
    const int rows = mine_b ? g.N : g.M, r0 = mine_b ? n0 : m0;
    const bool kfast = mine_b ? B_KFAST : A_KFAST;                  // wave-uniform
    // per-operand power-of-two scale: max |x| s in [2^11, 2^12) -- sixteen times below the f16 range, and an element down to 2^-25 of
    // the tensor's largest still has a normal hi part
    float sa, sb;
    sgemm_f16_scales(g, sa, sb);
    const float sc = mine_b ? sb : sa;
    const float unscale = 1.0f / (sa * sb);                         // (exact: powers of two)
    f32x4t rr[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f32x4t v = {0.f, 0.f, 0.f, 0.f};
            if (kfast) {
                const int idx = t + e * 256;
                const int r = r0 + (idx >> 2), k = k0 + 4 * (idx & 3);
                const float* p = P + (int64_t)r * s_row + k;
                if (!GUARD && k0 + 16 <= kend) v = *reinterpret_cast<const f32x4t*>(p);
                else if (r < rows) {
                    if (k + 3 < kend) v = *reinterpret_cast<const f32x4t*>(p);
                    else {
                        if (k < kend) v[0] = p[0];
                        if (k + 1 < kend) v[1] = p[1];
                        if (k + 2 < kend) v[2] = p[2];
                    }
                }
            } else {
                const int k = k0 + 4 * (t >> 6) + e, r = r0 + 4 * (t & 63);
                const float* p = P + (int64_t)k * s_k + r;
                if (!GUARD && k0 + 16 <= kend) v = *reinterpret_cast<const f32x4t*>(p);
                else if (k < kend) {
                    if (r + 3 < rows) v = *reinterpret_cast<const f32x4t*>(p);
                    else {
                        if (r < rows) v[0] = p[0];
                        if (r + 1 < rows) v[1] = p[1];
                        if (r + 2 < rows) v[2] = p[2];
                    }
                }
            }
            rr[e] = v;
        }
    };
    auto stash = [&](unsigned char* bufp, const f32x4t (&v)[4]) {
        unsigned char* base = bufp + (mine_b ? 2 * PLANE : 0);
        if (kfast) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int idx = t + e * 256;
                const float x0 = v[e][0], x1 = v[e][1], x2 = v[e][2], x3 = v[e][3];
                unsigned h0, l0, h1, l1;
                split_pair_f16x2(x0 * sc, x1 * sc, h0, l0);
                split_pair_f16x2(x2 * sc, x3 * sc, h1, l1);
                unsigned char* p = base + (idx >> 2) * ROWB + 8 * (idx & 3);
                *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(p + PLANE) = make_uint2(l0, l1);
            }
        } else {
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {                       // two k pairs per thread: (4 kq, 4 kq + 1) and (4 kq + 2, 4 kq + 3)
                unsigned h[4], l[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float x0 = v[2 * pp][j], x1 = v[2 * pp + 1][j];
                    split_pair_f16x2(x0 * sc, x1 * sc, h[j], l[j]);
                }
                unsigned char* p = base + ((2 * (t >> 6) + pp) * T + 4 * (t & 63)) * 4;
                *reinterpret_cast<uint4*>(p) = make_uint4(h[0], h[1], h[2], h[3]);
                *reinterpret_cast<uint4*>(p + PLANE) = make_uint4(l[0], l[1], l[2], l[3]);
            }
        }
    };
    auto operand = [&](const unsigned char* plane, int row0, bool kf) -> gemm_f16x8 {
        if (kf) return *reinterpret_cast<const gemm_f16x8*>(plane + (row0 + (lane & 31)) * ROWB + (lane >> 5) * 16);
        const unsigned* q = reinterpret_cast<const unsigned*>(plane) + (4 * (lane >> 5)) * T + row0 + (lane & 31);
        const gemm_u32x4 v = {q[0], q[T], q[2 * T], q[3 * T]};
        return __builtin_bit_cast(gemm_f16x8, v);
    };
    int buf = 0;
    if (kbeg < kend) {
        fetch(kbeg);
        stash(sgemm_x3_lds, rr);
        fetch(kbeg + 16);
    }
    __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
        const bool more = k0 + 16 < kend;
        f32x4t nn[4] = {rr[0], rr[1], rr[2], rr[3]};                                         // tile k + 1, loaded one iteration ago
        if (k0 + 32 < kend) fetch(k0 + 32);
        const unsigned char* b = sgemm_x3_lds + buf * BUF;
        gemm_f16x8 a[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 2; ++p) a[i][p] = operand(b + p * PLANE, wm + 32 * i, A_KFAST);
        constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};             // lo hi, hi lo, hi hi: smallest first
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            gemm_f16x8 bb[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) bb[p] = operand(b + (2 + p) * PLANE, wn + 32 * j, B_KFAST);
#pragma unroll
            for (int tt = 0; tt < 3; ++tt)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][PA[tt]], bb[PB[tt]], acc[i][j], 0, 0, 0);
        }
        if (more) stash(sgemm_x3_lds + (buf ^ 1) * BUF, nn);
#ifndef F16X2_VALU
#define F16X2_VALU 4
#endif
#if F16X2_VALU > 0
#pragma unroll
        for (int q = 0; q < 24; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, F16X2_VALU, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
#endif
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), gn = n0 + wn + 32 * j + (lane & 31);
                if (!GUARD || (gm < g.M && gn < g.N)) {
                    float* c = g.C + (int64_t)gm * g.ldc + gn;
                    *c = g.accumulate ? *c + acc[i][j][r] * unscale : acc[i][j][r] * unscale;
                }
            }
}

template <bool A_KFAST, bool B_KFAST>
static __global__ __launch_bounds__(512) void sgemm_f16x2v_kernel(GemmArgs g) {
    const int kbeg = blockIdx.z * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    const bool interior = (int)blockIdx.y * 256 + 256 <= g.M && (int)blockIdx.x * 256 + 256 <= g.N && kend > kbeg;
    if (interior) sgemm_f16x2v_body<A_KFAST, B_KFAST, false>(g);
    else sgemm_f16x2v_body<A_KFAST, B_KFAST, true>(g);
}

// The 128 x 128 kernel (sgemm_bf16x3_kernel) on two f16 planes, for scaled products with too few 256 x 256 tiles (the tiled ST_GCN path at
// the reference protocol's batch of 100: [1000 x 1024] x [1024 x 1024] as 64 tiles x 4 k slices).  256 threads here: the scale reduction
// strides accordingly.
template <bool A_KFAST, bool B_KFAST, bool GUARD>
static __device__ __forceinline__ void sgemm_f16x2_body(GemmArgs g) {
    constexpr int ROWB = 48;                       // bytes per LDS row of the k-contiguous form
    constexpr int PLANE = 128 * ROWB;              // one bf16 plane of one operand tile
    constexpr int BUF = 4 * PLANE;                 // A: h, l ; B: h, l
    extern __shared__ __attribute__((aligned(16))) unsigned char sgemm_x3_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    f32x16t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int kbeg = blockIdx.z * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    g.C += (int64_t)blockIdx.z * g.M * g.ldc;
    float sa, sb;
    sgemm_f16_scales(g, sa, sb);
    const float unscale = 1.0f / (sa * sb);
    f32x4t ra[2], rb[2];
    // k-contiguous: float4 e of a thread = row (tid + 256 e) >> 2, k = 4 ((tid + 256 e) & 3)
    // row-contiguous: float4 e of a thread = rows 4 (tid & 31) .. + 3 at k = 2 (tid >> 5) + e
    auto fetch_one = [&](const float* __restrict__ P, int64_t s_row, int64_t s_k, int rows, int r0, int k0, int e, bool kfast) -> f32x4t {
        f32x4t v = {0.f, 0.f, 0.f, 0.f};
        if (!GUARD && k0 + 16 <= kend) {          // interior tile, whole K step (wave-uniform): every 16-byte load is in bounds
            if (kfast) {
                const int idx = tid + e * 256;
                return *reinterpret_cast<const f32x4t*>(P + (int64_t)(r0 + (idx >> 2)) * s_row + k0 + 4 * (idx & 3));
            }
            return *reinterpret_cast<const f32x4t*>(P + (int64_t)(k0 + 2 * (tid >> 5) + e) * s_k + r0 + 4 * (tid & 31));
        }
        if (kfast) {
            const int idx = tid + e * 256;
            const int r = r0 + (idx >> 2), k = k0 + 4 * (idx & 3);
            if (r < rows) {
                const float* p = P + (int64_t)r * s_row + k;
                if (k + 3 < kend) v = *reinterpret_cast<const f32x4t*>(p);
                else {
                    if (k < kend) v[0] = p[0];
                    if (k + 1 < kend) v[1] = p[1];
                    if (k + 2 < kend) v[2] = p[2];
                }
            }
        } else {
            const int k = k0 + 2 * (tid >> 5) + e, r = r0 + 4 * (tid & 31);
            if (k < kend) {
                const float* p = P + (int64_t)k * s_k + r;
                if (r + 3 < rows) v = *reinterpret_cast<const f32x4t*>(p);
                else {
                    if (r < rows) v[0] = p[0];
                    if (r + 1 < rows) v[1] = p[1];
                    if (r + 2 < rows) v[2] = p[2];
                }
            }
        }
        return v;
    };
    auto fetch = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            ra[e] = fetch_one(g.A, g.sAm, g.sAk, g.M, m0, k0, e, A_KFAST);
            rb[e] = fetch_one(g.B, g.sBn, g.sBk, g.N, n0, k0, e, B_KFAST);
        }
    };
    auto stash_one = [&](unsigned char* base, const f32x4t (&v)[2], bool kfast, float sc) {
        if (kfast) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int idx = tid + e * 256;
                const float x0 = v[e][0], x1 = v[e][1], x2 = v[e][2], x3 = v[e][3];
                unsigned h0, l0, h1, l1;
                split_pair_f16x2(x0 * sc, x1 * sc, h0, l0);
                split_pair_f16x2(x2 * sc, x3 * sc, h1, l1);
                unsigned char* p = base + (idx >> 2) * ROWB + 8 * (idx & 3);
                *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(p + PLANE) = make_uint2(l0, l1);
            }
        } else {
            unsigned h[4], l[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x0 = v[0][j], x1 = v[1][j];
                split_pair_f16x2(x0 * sc, x1 * sc, h[j], l[j]);
            }
            unsigned char* p = base + ((tid >> 5) * 128 + 4 * (tid & 31)) * 4;
            *reinterpret_cast<uint4*>(p) = make_uint4(h[0], h[1], h[2], h[3]);
            *reinterpret_cast<uint4*>(p + PLANE) = make_uint4(l[0], l[1], l[2], l[3]);
        }
    };
    auto stash = [&](int buf) {
        unsigned char* b = sgemm_x3_lds + buf * BUF;
        stash_one(b, ra, A_KFAST, sa);
        stash_one(b + 2 * PLANE, rb, B_KFAST, sb);
    };
    // the MFMA operand of this lane: 8 consecutive k (k half lane >> 5) of row `row0 + (lane & 31)` of one plane
    auto operand = [&](const unsigned char* plane, int row0, bool kfast) -> gemm_f16x8 {
        if (kfast) return *reinterpret_cast<const gemm_f16x8*>(plane + (row0 + (lane & 31)) * ROWB + (lane >> 5) * 16);
        const unsigned* q = reinterpret_cast<const unsigned*>(plane) + (4 * (lane >> 5)) * 128 + row0 + (lane & 31);
        const gemm_u32x4 v = {q[0], q[128], q[256], q[384]};
        return __builtin_bit_cast(gemm_f16x8, v);
    };
    // Software pipeline, two K steps deep: while the matrix cores work on tile k (LDS buffer `buf`), the registers loaded during the
    // PREVIOUS iteration (tile k + 1: a full iteration of latency cover) are split and written to the other buffer, and the loads of
    // tile k + 2 are issued.  The split's VALU work is interleaved with the MFMAs by the scheduling hints at the end of the body: an
    // in-order wavefront hides ~5 other instructions behind each 32-cycle MFMA, or none at all if they sit behind the whole chain.
    int buf = 0;
    if (kbeg < kend) {
        fetch(kbeg);
        stash(0);
        fetch(kbeg + 16);                     // tile 1 (past the end the guarded loads return zeros)
    }
    __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
        const bool more = k0 + 16 < kend;
        f32x4t na[2], nb[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) { na[e] = ra[e]; nb[e] = rb[e]; }          // tile k + 1, loaded one iteration ago
        if (k0 + 32 < kend) fetch(k0 + 32);                                      // tile k + 2 into ra / rb
        const unsigned char* b = sgemm_x3_lds + buf * BUF;
        gemm_f16x8 a[2][2], bb[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                a[i][p] = operand(b + p * PLANE, wm + 32 * i, A_KFAST);
                bb[i][p] = operand(b + (2 + p) * PLANE, wn + 32 * i, B_KFAST);
            }
        // the three product terms, smallest first; consecutive MFMAs go to different accumulators (independent: back-to-back issue)
        constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][PA[t]], bb[j][PB[t]], acc[i][j], 0, 0, 0);
        if (more) {
            unsigned char* nbuf = sgemm_x3_lds + (buf ^ 1) * BUF;
            stash_one(nbuf, na, A_KFAST, sa);
            stash_one(nbuf + 2 * PLANE, nb, B_KFAST, sb);
        }
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);      // eight VALU (the split)
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // one LDS write
        }
        __syncthreads();
        buf ^= 1;
    }
    // D layout of the 32x32 result: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), gn = n0 + wn + 32 * j + (lane & 31);
                if (!GUARD || (gm < g.M && gn < g.N)) {
                    float* c = g.C + (int64_t)gm * g.ldc + gn;
                    *c = g.accumulate ? *c + acc[i][j][r] * unscale : acc[i][j][r] * unscale;
                }
            }
}

template <bool A_KFAST, bool B_KFAST>
static __global__ __launch_bounds__(256, 2) void sgemm_f16x2_kernel(GemmArgs g) {
    // interior tiles take the body whose whole K steps load without bounds checks (a third of its non-MFMA instructions were guards)
    const int kbeg = blockIdx.z * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    const bool interior = (int)blockIdx.y * 128 + 128 <= g.M && (int)blockIdx.x * 128 + 128 <= g.N && kend > kbeg;
    if (interior) sgemm_f16x2_body<A_KFAST, B_KFAST, false>(g);
    else sgemm_f16x2_body<A_KFAST, B_KFAST, true>(g);
}


// partial maxima of |x| over the finite elements, one float per workgroup (the operand scales of the f16 split for tensors whose
// producer is not one of this library's kernels)
static __global__ __launch_bounds__(256) void absmax_partials_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ part) {
    __shared__ float l4[4];
    float m = 0.f;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const float a = __builtin_fabsf(x[e]);
        m = fmaxf(m, a <= 3.0e38f ? a : 0.f);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) l4[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(l4[0], l4[1]), fmaxf(l4[2], l4[3]));
}
int absmax_partials(const float* x, int64_t n, float* part, int nparts, hipStream_t st) {
    if (nparts <= 0) return RULGNN_EINVAL;
    (void)hipGetLastError();
    hipLaunchKernelGGL(absmax_partials_kernel, dim3(nparts), dim3(256), 0, st, x, n, part);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

// 256x256 tiles when both output dimensions fill them and there are enough of them for one per CU
static inline bool sgemm_wide_ok(const GemmArgs& g, int slices) {
    if (g.M <= 192 || g.N <= 192) return false;
    return (int64_t)((g.M + 255) / 256) * ((g.N + 255) / 256) * slices >= 160;
}

// process-wide arithmetic of the big-tile GEMM: 1 = bf16 x 3 (default), 0 = fp32 matrix instructions (bit-compatible with the 64x64 kernel)
// (defined once, in rulgnn_api.hip: this header is included by several translation units)
int& sgemm_big_mode();

constexpr int SGEMM_BIG_KT = 16;      // k depth of an LDS stage of the 128x128 kernel
// the 128x128 kernel pays when both output dimensions fill most of a tile and there are enough tiles for the chip
static inline bool sgemm_big_ok(const GemmArgs& g, int slices) {
    if (g.M <= 96 || g.N <= 96 || g.K < 16) return false;
    const int64_t tiles = (int64_t)((g.M + 127) / 128) * ((g.N + 127) / 128) * slices;
    if (tiles < 96) return false;
    const bool ak = g.sAk == 1, am = g.sAm == 1, bk = g.sBk == 1, bn = g.sBn == 1;
    if (!(ak || am) || !(bk || bn)) return false;
    if ((reinterpret_cast<uintptr_t>(g.A) | reinterpret_cast<uintptr_t>(g.B)) & 15) return false;
    if ((ak ? g.sAm : g.sAk) % 4 != 0 || (bk ? g.sBn : g.sBk) % 4 != 0) return false;
    return g.kchunk % 4 == 0;
}

// the matrix-core GEMM of a (possibly split-K) problem: the 128x128 kernel where it pays, the 64x64 one otherwise
static inline void sgemm_launch_tiles(const GemmArgs& g, int slices, hipStream_t st) {
    if (sgemm_big_ok(g, slices)) {
        const dim3 grid((g.N + 127) / 128, (g.M + 127) / 128, slices);
        const bool ak = g.sAk == 1, bk = g.sBk == 1;
        if (sgemm_big_mode() >= 1) {                                 // (1: bf16 x 3, f16 x 2 where the caller passes scales; 2: bf16 x 3 only)
            constexpr size_t lx = (size_t)2 * 6 * 128 * 48;
            auto gox = [&](auto kernel) {
                static bool raised = false;                          // once per instantiation and process
                if (!raised) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lx);
                    raised = true;
                }
                hipLaunchKernelGGL(kernel, grid, dim3(256), lx, st, g);
            };
            if (sgemm_wide_ok(g, slices)) {
                const dim3 wgrid((g.N + 255) / 256, (g.M + 255) / 256, slices);
                constexpr size_t lw = (size_t)2 * 6 * 256 * 48;
                auto gow = [&](auto kernel) {
                    static bool raised = false;
                    if (!raised) {
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lw);
                        raised = true;
                    }
                    hipLaunchKernelGGL(kernel, wgrid, dim3(512), lw, st, g);
                };
                if (g.amax_a && g.amax_b && sgemm_big_mode() == 1) {
                    constexpr size_t lh = (size_t)2 * 4 * 256 * 48;
                    auto goh = [&](auto kernel) {
                        static bool raised = false;
                        if (!raised) {
                            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lh);
                            raised = true;
                        }
                        hipLaunchKernelGGL(kernel, wgrid, dim3(512), lh, st, g);
                    };
                    if (ak && bk) goh(sgemm_f16x2v_kernel<true, true>);
                    else if (ak) goh(sgemm_f16x2v_kernel<true, false>);
                    else if (bk) goh(sgemm_f16x2v_kernel<false, true>);
                    else goh(sgemm_f16x2v_kernel<false, false>);
                    return;
                }
                if (ak && bk) gow(sgemm_bf16x3v_kernel<true, true>);
                else if (ak) gow(sgemm_bf16x3v_kernel<true, false>);
                else if (bk) gow(sgemm_bf16x3v_kernel<false, true>);
                else gow(sgemm_bf16x3v_kernel<false, false>);
                return;
            }
            if (g.amax_a && g.amax_b && sgemm_big_mode() == 1) {
                constexpr size_t l2 = (size_t)2 * 4 * 128 * 48;
                auto goh = [&](auto kernel) {
                    static bool raised = false;
                    if (!raised) {
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
                        raised = true;
                    }
                    hipLaunchKernelGGL(kernel, grid, dim3(256), l2, st, g);
                };
                if (ak && bk) goh(sgemm_f16x2_kernel<true, true>);
                else if (ak) goh(sgemm_f16x2_kernel<true, false>);
                else if (bk) goh(sgemm_f16x2_kernel<false, true>);
                else goh(sgemm_f16x2_kernel<false, false>);
                return;
            }
            if (ak && bk) gox(sgemm_bf16x3_kernel<true, true>);
            else if (ak) gox(sgemm_bf16x3_kernel<true, false>);
            else if (bk) gox(sgemm_bf16x3_kernel<false, true>);
            else gox(sgemm_bf16x3_kernel<false, false>);
            return;
        }
        constexpr size_t lds = (size_t)4 * SGEMM_BIG_KT * (128 + 16) * sizeof(float);
        auto go = [&](auto kernel) {
            static bool raised = false;                              // once per instantiation and process
            if (lds > 48 * 1024 && !raised) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                raised = true;
            }
            hipLaunchKernelGGL(kernel, grid, dim3(256), lds, st, g);
        };
        if (ak && bk) go(sgemm_mfma128_kernel<true, true, SGEMM_BIG_KT>);
        else if (ak) go(sgemm_mfma128_kernel<true, false, SGEMM_BIG_KT>);
        else if (bk) go(sgemm_mfma128_kernel<false, true, SGEMM_BIG_KT>);
        else go(sgemm_mfma128_kernel<false, false, SGEMM_BIG_KT>);
        return;
    }
    hipLaunchKernelGGL(sgemm_mfma_kernel, dim3((g.N + 63) / 64, (g.M + 63) / 64, slices), dim3(256), 0, st, g);
}

// ------------------------------------------------------------------------------------------------
// Tall-and-skinny case: many rows, a small [K x N] weight matrix (the per-row projections of the graph models:
// [batch*patches*nodes, 16..64] x [16..64, 8..64]).  A 64x64 MFMA tile wastes most of its columns there and the launch is
// bandwidth-bound anyway: one thread per output row, the weights in LDS (broadcast reads), the row streamed with 16-byte
// loads, N accumulators in registers.
// ------------------------------------------------------------------------------------------------
template <int NT>
static __global__ __launch_bounds__(256) void sgemm_skinny_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float Bsk[];      // [K][NT]
    for (int i = threadIdx.x; i < g.K * NT; i += 256) {
        const int k = i / NT, n = i % NT;
        Bsk[i] = n < g.N ? g.B[n * g.sBn + k * g.sBk] : 0.f;
    }
    __syncthreads();
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= g.M) return;
    float acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = 0.f;
    const float* a = g.A + m * g.sAm;
    const bool vec = ((g.sAm & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0);
    int k = 0;
    if (vec) {
        for (; k + 4 <= g.K; k += 4) {
            const f32x4t av = *reinterpret_cast<const f32x4t*>(a + k);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[n] = fmaf(av[q], Bsk[(k + q) * NT + n], acc[n]);
        }
    }
    for (; k < g.K; ++k) {
        const float av = a[k];
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = fmaf(av, Bsk[k * NT + n], acc[n]);
    }
    float* c = g.C + m * g.ldc;
#pragma unroll
    for (int n = 0; n < NT; ++n)
        if (n < g.N) c[n] = g.accumulate ? c[n] + acc[n] : acc[n];
}

// ------------------------------------------------------------------------------------------------
// bf16 variant of the tall-and-skinny case (BASELINE.json config "FC_STGNN ... bf16"): C[m][n] (+)= sum_k A[m][k] * B(n,k) with both
// operands ROUNDED TO bf16 (v_cvt_pk_bf16_f32, round to nearest even) and fp32 accumulation on v_mfma_f32_16x16x32_bf16.
// One wavefront per 16-row tile: lane (kg = lane >> 4, m = lane & 15) reads the 8 consecutive k of its row (two 16-byte loads),
// the weight operand (N <= 32, K <= 128: NT x KS MFMA operands) stays in registers for the whole grid-stride loop, each of the
// 4 result registers is one 64-byte row segment.  The kernel streams: [M, K] in, [M, N] out, nothing else.
// ------------------------------------------------------------------------------------------------

static __device__ __forceinline__ unsigned gemm_pk_bf16(float a, float b) {
    unsigned u;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u) : "v"(a), "v"(b));
    return u;
}

template <int NT, int KS>
static __global__ __launch_bounds__(256) void sgemm_rows_bf16_kernel(GemmArgs g) {
    const int lane = threadIdx.x & 63, kg = lane >> 4, mi = lane & 15;
    gemm_u32x4 bop[NT][KS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            float w[8];
            const int n = nt * 16 + mi;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = ks * 32 + 8 * kg + i;
                w[i] = (n < g.N && k < g.K) ? g.B[n * g.sBn + k * g.sBk] : 0.f;
            }
            bop[nt][ks] = gemm_u32x4{gemm_pk_bf16(w[0], w[1]), gemm_pk_bf16(w[2], w[3]), gemm_pk_bf16(w[4], w[5]), gemm_pk_bf16(w[6], w[7])};
        }
    const bool vec = ((g.sAm & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0) && ((g.K & 7) == 0);
    const int64_t tiles = ((int64_t)g.M + 15) / 16;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t tile = wave0; tile < tiles; tile += nwaves) {
        const int64_t row = tile * 16 + mi;
        f32x4t acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k0 = ks * 32 + 8 * kg;
            float a[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = 0.f;
            if (row < g.M && k0 < g.K) {
                const float* ap = g.A + row * g.sAm + k0;
                if (vec) {
                    const f32x4t v0 = *reinterpret_cast<const f32x4t*>(ap), v1 = *reinterpret_cast<const f32x4t*>(ap + 4);
                    a[0] = v0[0]; a[1] = v0[1]; a[2] = v0[2]; a[3] = v0[3]; a[4] = v1[0]; a[5] = v1[1]; a[6] = v1[2]; a[7] = v1[3];
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) a[i] = (k0 + i < g.K) ? ap[i] : 0.f;
                }
            }
            const gemm_u32x4 aop = {gemm_pk_bf16(a[0], a[1]), gemm_pk_bf16(a[2], a[3]), gemm_pk_bf16(a[4], a[5]), gemm_pk_bf16(a[6], a[7])};
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(gemm_bf16x8, aop), __builtin_bit_cast(gemm_bf16x8, bop[nt][ks]),
                                                                  acc[nt], 0, 0, 0);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t gm = tile * 16 + 4 * kg + r;
                const int gn = nt * 16 + mi;
                if (gm < g.M && gn < g.N) {
                    float* c = g.C + gm * g.ldc + gn;
                    *c = g.accumulate ? *c + acc[nt][r] : acc[nt][r];
                }
            }
    }
}

static inline bool sgemm_rows_bf16_ok(int64_t sAk, int M, int N, int K) { return sAk == 1 && N <= 32 && K <= 128 && K >= 1 && M >= 16; }

template <int NT, int KS>
static int sgemm_rows_bf16_launch(const GemmArgs& g, hipStream_t st) {
    const int64_t tiles = ((int64_t)g.M + 15) / 16;
    int64_t blocks = (tiles + 3) / 4;
    if (blocks > 4096) blocks = 4096;                   // 16 waves per CU worth of persistent workgroups
    (void)hipGetLastError();
    hipLaunchKernelGGL((sgemm_rows_bf16_kernel<NT, KS>), dim3((unsigned)blocks), dim3(256), 0, st, g);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

static int sgemm_rows_bf16(const GemmArgs& g, hipStream_t st) {
    const int NT = g.N <= 16 ? 1 : 2, KS = g.K <= 32 ? 1 : (g.K <= 64 ? 2 : 4);
    if (NT == 1) return KS == 1 ? sgemm_rows_bf16_launch<1, 1>(g, st) : (KS == 2 ? sgemm_rows_bf16_launch<1, 2>(g, st) : sgemm_rows_bf16_launch<1, 4>(g, st));
    return KS == 1 ? sgemm_rows_bf16_launch<2, 1>(g, st) : (KS == 2 ? sgemm_rows_bf16_launch<2, 2>(g, st) : sgemm_rows_bf16_launch<2, 4>(g, st));
}

// (measured: at N = 50..64 the LDS broadcast reads bind and the MFMA tile wins -- ASTGCNN batch 65536: 8.5 vs 10.6 ms/step)
static inline bool sgemm_is_skinny(int64_t sAk, int M, int N, int K) { return sAk == 1 && N <= 32 && K <= 128 && M >= 2048; }

// `bf16` != 0: operands rounded to bf16 on the matrix cores where the shape qualifies (sgemm_rows_bf16_ok), fp32 paths otherwise
int sgemm(const float* A, int64_t sAm, int64_t sAk, const float* B, int64_t sBn, int64_t sBk, float* C, int64_t ldc,
                 int M, int N, int K, bool accumulate, hipStream_t st, int bf16, const float* amax_a, int amax_na, const float* amax_b, int amax_nb,
                 void* plane_ws, size_t plane_ws_bytes) {
    if (M <= 0 || N <= 0) return RULGNN_OK;
    if (amax_a && amax_b && plane_ws && sgemm_big_mode() == 1 && sgemm_planes_slices(M, N, K, false) == 1 &&
        plane_ws_bytes >= sgemm_planes_ws_bytes(M, N, K) && sgemm_planes_ok(A, sAm, sAk, B, sBn, sBk, M, N, K, 1))
        return sgemm_planes(A, sAm, sAk, B, sBn, sBk, C, ldc, M, N, K, accumulate, 1, amax_a, amax_na, amax_b, amax_nb, plane_ws, plane_ws_bytes, st);
    if (bf16 && sgemm_rows_bf16_ok(sAk, M, N, K)) {
        const GemmArgs g{A, sAm, sAk, B, sBn, sBk, C, ldc, M, N, K, accumulate ? 1 : 0, K};
        return sgemm_rows_bf16(g, st);
    }
    if (sgemm_is_skinny(sAk, M, N, K)) {
        GemmArgs g{A, sAm, sAk, B, sBn, sBk, C, ldc, M, N, K, accumulate ? 1 : 0, K > 0 ? K : 1};
        const int NT = N <= 8 ? 8 : (N <= 16 ? 16 : 32);
        const size_t lds = (size_t)(K > 0 ? K : 1) * NT * sizeof(float);
        const dim3 grid((unsigned)((M + 255) / 256));
        (void)hipGetLastError();
        if (NT == 8) hipLaunchKernelGGL(sgemm_skinny_kernel<8>, grid, dim3(256), lds, st, g);
        else if (NT == 16) hipLaunchKernelGGL(sgemm_skinny_kernel<16>, grid, dim3(256), lds, st, g);
        else hipLaunchKernelGGL(sgemm_skinny_kernel<32>, grid, dim3(256), lds, st, g);
        return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
    }
    GemmArgs g{A, sAm, sAk, B, sBn, sBk, C, ldc, M, N, K, accumulate ? 1 : 0, K > 0 ? (K + 15) & ~15 : 16, amax_a, amax_b, amax_na, amax_nb};
    (void)hipGetLastError();
    sgemm_launch_tiles(g, 1, st);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

// Split-K for reductions over a long K (weight gradients: K = batch * nodes) with few output tiles: `slices` partial
// products into `partial` ([slices][M][N], caller-provided), then a fixed-order sum -- deterministic, no atomics.
static __global__ __launch_bounds__(1024) void sgemm_reduce_slices_kernel(const float* __restrict__ partial, float* __restrict__ C,
                                                                          int64_t ldc, int M, int N, int slices, int accumulate) {
    // 64 outputs per workgroup, sixteen threads per output each summing every sixteenth slice (up to 256 slices: sixteen loads in a
    // thread's chain), combined in a fixed order
    __shared__ float part[16][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    float a = 0.f;
    if (e < M * N)
        for (int z = q; z < slices; z += 16) a += partial[(int64_t)z * M * N + e];
    part[q][lane] = a;
    __syncthreads();
    if (q == 0 && e < M * N) {
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r += 4) v += (part[r][lane] + part[r + 1][lane]) + (part[r + 2][lane] + part[r + 3][lane]);
        float* c = C + (int64_t)(e / N) * ldc + (e % N);
        *c = accumulate ? *c + v : v;
    }
}

// ... of several products in one launch (sgemm_splitk_batch)
static __global__ __launch_bounds__(1024) void sgemm_reduce_slices_batch_kernel(ReduceBatch b) {
    __shared__ float part[16][64];
    reduce_slices_batch_body(b, blockIdx.x, part);
}
size_t sgemm_splitk_batch_floats(const SplitKJob* jobs, int n) {
    size_t f = 0;
    for (int j = 0; j < n; ++j) f += (size_t)sgemm_splitk_slices(jobs[j].M, jobs[j].N, jobs[j].K) * jobs[j].M * jobs[j].N;
    return f;
}
int sgemm_splitk_batch(const SplitKJob* jobs, int n, float* partial, size_t partial_floats, hipStream_t st) {
    return sgemm_splitk_batch_products(jobs, n, partial, partial_floats, st, nullptr);
}
int sgemm_splitk_batch_products(const SplitKJob* jobs, int n, float* partial, size_t partial_floats, hipStream_t st, ReduceBatch* reduce_out) {
    if (n < 1 || n > GEMM_BATCH_MAX) return RULGNN_EINVAL;
    if (partial_floats < sgemm_splitk_batch_floats(jobs, n)) return RULGNN_EWORKSPACE;
    GemmBatch gb{};
    ReduceBatch rb{};
    gb.n = rb.n = n;
    size_t off = 0;
    int wg = 0, rwg = 0;
    for (int j = 0; j < n; ++j) {
        const SplitKJob& q = jobs[j];
        if (q.M <= 0 || q.N <= 0 || q.K <= 0) return RULGNN_EINVAL;
        const int slices = sgemm_splitk_slices(q.M, q.N, q.K);
        int kchunk = (q.K + slices - 1) / slices;
        kchunk = (kchunk + 15) & ~15;
        const int used = (q.K + kchunk - 1) / kchunk;
        gb.g[j] = GemmArgs{q.A, q.sAm, q.sAk, q.B, q.sBn, q.sBk, partial + off, q.N, q.M, q.N, q.K, 0, kchunk};
        gb.nx[j] = (q.N + 63) / 64; gb.ny[j] = (q.M + 63) / 64;
        gb.first[j] = wg;
        wg += gb.nx[j] * gb.ny[j] * used;
        rb.partial[j] = partial + off; rb.C[j] = q.C; rb.ldc[j] = q.ldc; rb.M[j] = q.M; rb.N[j] = q.N; rb.slices[j] = used;
        rb.first[j] = rwg;
        rwg += (q.M * q.N + 63) / 64;
        off += (size_t)used * q.M * q.N;
    }
    gb.first[n] = wg; rb.first[n] = rwg;
    (void)hipGetLastError();
    hipLaunchKernelGGL(sgemm_mfma_batch_kernel, dim3(wg), dim3(256), 0, st, gb);
    if (reduce_out) *reduce_out = rb;          // (the caller runs the slice sums inside a launch of its own: reduce_slices_batch_body)
    else hipLaunchKernelGGL(sgemm_reduce_slices_batch_kernel, dim3(rwg), dim3(1024), 0, st, rb);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

// ... for large contiguous outputs (M N a multiple of 4, ldc == N: the [N x N] weight gradients of the tiled ST_GCN path, 4 MB per slice):
// a thread owns four consecutive outputs and walks the slices with 16-byte loads, four slices in flight -- the form above issues one
// 4-byte load per thread and slice and ran at ~2 TB/s (35 us for 16 slices of [1024 x 1024]).  Fixed order: slice 0, 1, 2, ...
static __global__ __launch_bounds__(256) void sgemm_reduce_slices4_kernel(const float4* __restrict__ partial, float4* __restrict__ C, int64_t quads,
                                                                          int slices, int accumulate) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= quads) return;
    float4 v = accumulate ? C[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    int z = 0;
    for (; z + 3 < slices; z += 4) {
        const float4 p0 = partial[(int64_t)z * quads + e], p1 = partial[(int64_t)(z + 1) * quads + e];
        const float4 p2 = partial[(int64_t)(z + 2) * quads + e], p3 = partial[(int64_t)(z + 3) * quads + e];
        v.x = (((v.x + p0.x) + p1.x) + p2.x) + p3.x;
        v.y = (((v.y + p0.y) + p1.y) + p2.y) + p3.y;
        v.z = (((v.z + p0.z) + p1.z) + p2.z) + p3.z;
        v.w = (((v.w + p0.w) + p1.w) + p2.w) + p3.w;
    }
    for (; z < slices; ++z) {
        const float4 p = partial[(int64_t)z * quads + e];
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    C[e] = v;
}
static inline int sgemm_reduce_slices(const float* partial, float* C, int64_t ldc, int M, int N, int slices, bool accumulate, hipStream_t st) {
    const int64_t MN = (int64_t)M * N;
    (void)hipGetLastError();
    if (MN >= 65536 && MN % 4 == 0 && (ldc == N || M == 1) && ((reinterpret_cast<uintptr_t>(partial) | reinterpret_cast<uintptr_t>(C)) & 15) == 0)
        hipLaunchKernelGGL(sgemm_reduce_slices4_kernel, dim3((unsigned)((MN / 4 + 255) / 256)), dim3(256), 0, st,
                           reinterpret_cast<const float4*>(partial), reinterpret_cast<float4*>(C), MN / 4, slices, accumulate ? 1 : 0);
    else
        hipLaunchKernelGGL(sgemm_reduce_slices_kernel, dim3((M * N + 63) / 64), dim3(1024), 0, st, partial, C, ldc, M, N, slices, accumulate ? 1 : 0);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

// ------------------------------------------------------------------------------------------------
// Long reduction into a small output (the weight gradients of the per-row projections: M x N <= 1024 outputs, K = all rows
// of the batch).  The generic split-K path launches a 64x64 MFMA tile per slice for a handful of useful columns and then
// walks the slices; here every workgroup owns a contiguous run of k, stages 32 k-rows of both operands in LDS with
// coalesced loads, each thread keeps <= 4 outputs in registers, and one wavefront per output adds the per-workgroup
// partials in a fixed order (deterministic).
// ------------------------------------------------------------------------------------------------
// `ones` = 1: B gains a virtual last column of ones, i.e. output column N is the sum over k of A(m, k) -- the bias gradient that goes
// with a weight gradient over the same rows, in the same launch (g.N already counts that column; it is not read from memory).
static __global__ __launch_bounds__(256) void sgemm_longk_kernel(GemmArgs g, int kper, int ones) {
    extern __shared__ float sk_lds[];                 // As[SKT_ROWS][M] | Bs[SKT_ROWS][N] | red[256] (few outputs only)
    float* As = sk_lds;
    float* Bs = sk_lds + SKT_ROWS * g.M;
    const int O = g.M * g.N, NB = g.N - ones;          // NB: the columns B really has
    // O >= 256: thread t owns outputs t, t + 256, ... (<= 4), every k-row.  O < 256: 256 / O thread slices share each output,
    // slice s taking rows s, s + S, ... of a tile; the slices are combined through LDS in a fixed order at the end.
    const int S = O >= 256 ? 1 : 256 / O;
    const int sl = O >= 256 ? 0 : threadIdx.x / O;
    const bool active = O >= 256 || sl < S;
    int oi[4], oj[4];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int o = O >= 256 ? threadIdx.x + q * 256 : (q == 0 ? threadIdx.x % O : O);
        oi[q] = o < O ? o / g.N : 0;
        oj[q] = o < O ? o % g.N : 0;
    }
    const int kbeg = blockIdx.x * kper, kend = min(g.K, kbeg + kper);
    for (int k0 = kbeg; k0 < kend; k0 += SKT_ROWS) {
        const int nr = min(SKT_ROWS, kend - k0);
        __syncthreads();
        for (int e = threadIdx.x; e < SKT_ROWS * g.M; e += 256)
            As[e] = e < nr * g.M ? g.A[(e % g.M) * g.sAm + (int64_t)(k0 + e / g.M) * g.sAk] : 0.f;
        for (int e = threadIdx.x; e < SKT_ROWS * g.N; e += 256) {
            const int col = e % g.N;
            Bs[e] = e < nr * g.N ? (col < NB ? g.B[col * g.sBn + (int64_t)(k0 + e / g.N) * g.sBk] : 1.f) : 0.f;
        }
        __syncthreads();
        if (!active) continue;
        if (O >= 256) {
#pragma unroll 4
            for (int r = 0; r < SKT_ROWS; ++r) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = fmaf(As[r * g.M + oi[q]], Bs[r * g.N + oj[q]], acc[q]);
            }
        } else {
            // rows past nr are zero-filled: no bounds in the loop; four independent chains hide the LDS latency
            for (int r = sl; r < SKT_ROWS; r += 4 * S) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int rr = r + q * S;
                    if (rr < SKT_ROWS) acc[q] = fmaf(As[rr * g.M + oi[0]], Bs[rr * g.N + oj[0]], acc[q]);
                }
            }
        }
    }
    if (O >= 256) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int o = threadIdx.x + q * 256;
            if (o < O) g.C[(int64_t)blockIdx.x * O + o] = acc[q];      // g.C = the partial buffer here
        }
    } else {
        float* red = Bs + SKT_ROWS * g.N;
        __syncthreads();
        red[threadIdx.x] = active ? (acc[0] + acc[1]) + (acc[2] + acc[3]) : 0.f;
        __syncthreads();
        if ((int)threadIdx.x < O) {
            float a = 0.f;
            for (int q = 0; q < S; ++q) a += red[q * O + threadIdx.x];
            g.C[(int64_t)blockIdx.x * O + threadIdx.x] = a;
        }
    }
}

// The same partial products on the fp32 matrix cores, straight from global memory, for the layout every weight gradient of the families
// has: A = [K][M] and B = [K][N] row-major (sAm = sBn = 1), M <= 32, N <= 64.  One WAVEFRONT per k-range (what a workgroup was above:
// same number of partial rows, same reduce kernel), v_mfma_f32_16x16x4f32 over four rows per step -- lane (kq, li) reads A[k + kq][16 mt + li]
// and B[k + kq][16 nt + li], i.e. a wavefront's load is four rows of 64 contiguous bytes -- eight steps of loads in flight, no LDS, no
// barrier.  The LDS kernel above ran 18-44 us per launch beside the families' backward chains (FC_STGNN: five of them were the step's
// critical path); this one is bound by the rows it reads.
template <int MT, int NT>
static __device__ __forceinline__ void sgemm_longk_mfma_body(const GemmArgs& g, int kper, int ones, int nwaves) {
    const int lane = threadIdx.x & 63, wv = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wv >= nwaves) return;
    const int li = lane & 15, kq = lane >> 4;
    const int NB = g.N - ones;
    f32x4t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4t){0.f, 0.f, 0.f, 0.f};
    bool am[MT], bn[NT], b1[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) am[i] = 16 * i + li < g.M;
#pragma unroll
    for (int j = 0; j < NT; ++j) { bn[j] = 16 * j + li < NB; b1[j] = ones && 16 * j + li == NB; }
    const int kbeg = wv * kper, kend = min(g.K, kbeg + kper);
    // Eight steps (32 rows) of operands are requested before the first product: every load UNCONDITIONAL, on a clamped row / column, and
    // masked afterwards.  Written as `ok ? p[..] : 0` the compiler kept each load in its own branch with a full wait in front of the
    // products -- one memory round trip per four rows, 15-30 us for the ~170 000 rows of FC_STGNN's window blocks (five such launches were
    // the tail of its side stream).  Same products in the same order; the steps past kend multiply zeros.
    const float* pa[MT];
    const float* pb[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) pa[i] = g.A + (am[i] ? 16 * i + li : 0);
#pragma unroll
    for (int j = 0; j < NT; ++j) pb[j] = g.B + (bn[j] ? 16 * j + li : 0);
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
        float a[8][MT], b[8][NT];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + 4 * u + kq, kc = k < kend ? k : kend - 1;
#pragma unroll
            for (int i = 0; i < MT; ++i) a[u][i] = pa[i][(int64_t)kc * g.sAk];
#pragma unroll
            for (int j = 0; j < NT; ++j) b[u][j] = pb[j][(int64_t)kc * g.sBk];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool kok = k0 + 4 * u + kq < kend;
#pragma unroll
            for (int i = 0; i < MT; ++i) a[u][i] = (kok && am[i]) ? a[u][i] : 0.f;
#pragma unroll
            for (int j = 0; j < NT; ++j) b[u][j] = (kok && bn[j]) ? b[u][j] : ((kok && b1[j]) ? 1.f : 0.f);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i], b[u][j], acc[i][j], 0, 0, 0);
        }
    }
    float* out = g.C + (int64_t)wv * g.M * g.N;                 // g.C = the partial buffer: one row of M N values per wavefront
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = 16 * i + 4 * kq + r, n = 16 * j + li;
                if (m < g.M && n < g.N) out[m * g.N + n] = acc[i][j][r];
            }
}
template <int MT, int NT>
static __global__ __launch_bounds__(256) void sgemm_longk_mfma_kernel(GemmArgs g, int kper, int ones, int nwaves) {
    sgemm_longk_mfma_body<MT, NT>(g, kper, ones, nwaves);
}
// several such products of ONE tile shape in one launch (blockIdx.y = product): FC_STGNN's five weight-gradient pairs were ten launches at
// the tail of its side stream
constexpr int LONGK_BATCH_MAX = 6;
struct LongkBatch {
    GemmArgs g[LONGK_BATCH_MAX];
    int kper[LONGK_BATCH_MAX], nwaves[LONGK_BATCH_MAX];
};
template <int MT, int NT>
static __global__ __launch_bounds__(256) void sgemm_longk_mfma_batch_kernel(LongkBatch b, int ones) {
    const int j = blockIdx.y;
    sgemm_longk_mfma_body<MT, NT>(b.g[j], b.kper[j], ones, b.nwaves[j]);
}
static bool sgemm_longk_mfma_ok(const GemmArgs& g) { return g.sAm == 1 && g.sBn == 1 && g.M <= 32 && g.N <= 64; }
static void sgemm_longk_mfma_launch(const GemmArgs& g, int kper, int ones, int nwaves, hipStream_t st) {
    const int MT = (g.M + 15) / 16, NT = (g.N + 15) / 16;
    const dim3 grid((nwaves + 3) / 4), block(256);
#define RULGNN_LK(mt, nt) hipLaunchKernelGGL((sgemm_longk_mfma_kernel<mt, nt>), grid, block, 0, st, g, kper, ones, nwaves)
    if (MT == 1) { if (NT == 1) RULGNN_LK(1, 1); else if (NT == 2) RULGNN_LK(1, 2); else if (NT == 3) RULGNN_LK(1, 3); else RULGNN_LK(1, 4); }
    else { if (NT == 1) RULGNN_LK(2, 1); else if (NT == 2) RULGNN_LK(2, 2); else if (NT == 3) RULGNN_LK(2, 3); else RULGNN_LK(2, 4); }
#undef RULGNN_LK
}

// (N counts the virtual ones column when colsum != nullptr: its sums go to colsum[m], not into C)
static __global__ __launch_bounds__(256) void sgemm_longk_reduce_kernel(const float* __restrict__ partial, float* __restrict__ C,
                                                                        int64_t ldc, int M, int N, int nblk, int accumulate,
                                                                        float* __restrict__ colsum) {
    const int o = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (o >= M * N) return;
    float a = 0.f;
    for (int b = lane; b < nblk; b += 64) a += partial[(int64_t)b * M * N + o];
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) a += __shfl_xor(a, m, 64);
    if (lane == 0) {
        const int mi = o / N, nj = o % N;
        if (colsum && nj == N - 1) {
            colsum[mi] = a;
        } else {
            float* c = C + (int64_t)mi * ldc + nj;
            *c = accumulate ? *c + a : a;
        }
    }
}


// C[m][n] (+)= sum_k A[m][k] B[n][k] for a SMALL output over a SHORT reduction (M N <= 1024, K M N <= 131072: the weight gradients of the
// families' MLP heads over a batch): one workgroup, 1024 / (M N) threads per output over interleaved k, combined in fixed order through
// LDS -- one launch of ~6 us instead of a split-K pair (two launches, 16-19 + 5-7 us).
static __global__ __launch_bounds__(1024) void sgemm_tiny_kernel(GemmArgs g, int accumulate) {
    __shared__ float red[1024];
    const int O = g.M * g.N, stripes = 1024 / O, e = threadIdx.x % O, sidx = threadIdx.x / O;
    const int m = e / g.N, n = e - m * g.N;
    float a = 0.f;
    if (sidx < stripes) {
        const float* ap = g.A + (int64_t)m * g.sAm;
        const float* bp = g.B + (int64_t)n * g.sBn;
        // eight products' operands requested before the first is used (one memory round trip per eight, same summation order)
        int k = sidx;
        for (; k + 7 * stripes < g.K; k += 8 * stripes) {
            float av[8], bv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                av[u] = ap[(int64_t)(k + u * stripes) * g.sAk];
                bv[u] = bp[(int64_t)(k + u * stripes) * g.sBk];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) a = fmaf(av[u], bv[u], a);
        }
        for (; k < g.K; k += stripes) a = fmaf(ap[(int64_t)k * g.sAk], bp[(int64_t)k * g.sBk], a);
    }
    red[threadIdx.x] = sidx < stripes ? a : 0.f;
    __syncthreads();
    if ((int)threadIdx.x < O) {
        float v = 0.f;
        for (int q = 0; q < stripes; ++q) v += red[q * O + threadIdx.x];
        float* c = g.C + (int64_t)m * g.ldc + n;
        *c = accumulate ? *c + v : v;
    }
}

// One workgroup: a thread walks K M N / 1024 products, eight operand pairs per memory round trip.  Measured before that bound existed
// (STNet, K = batch x patches = 2000): 29 us for 300 outputs, 240-540 us for ~1000 outputs -- far above the split-K pair it replaces.
// Up to 128 products per thread it is a 5-11 us launch (FC_STGNN's heads at batch 256); at 224 (STGNN's head: 896 outputs x 256
// rows) 25 us against 12 us for the pair.
static inline bool sgemm_tiny_ok(int M, int N, int K) { return (int64_t)M * N <= 1024 && (int64_t)K * M * N <= 131072; }

// ------------------------------------------------------------------------------------------------
// One output ROW over a long K with B contiguous along n: C[n] = sum_k a[k] B[k][n] -- the bias gradients (a = ones) and vector-matrix
// weight gradients (fc2 of the tiled ST_GCN path) over batch-sized row counts.  As a 64 x 64 matrix-core tile with one useful row in 64
// plus the slice reduction this was 27 + 15 us per product at N = 1024 (four of them per step of the XJTU-SY wiring); it is a
// memory-bound column sum: lanes along n (coalesced), a wavefront per row slice, eight loads in flight per thread, `ks` row chunks
// across blockIdx.y, the chunks' partial rows combined by rows_sum_kernel in a fixed order (deterministic).
// ------------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void sgemm_vecmat_kernel(const float* __restrict__ a, int64_t sAk, const float* __restrict__ B, int64_t sBk,
                                                                  float* __restrict__ partial, int N, int K, int kper) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    const int k0 = blockIdx.y * kper, k1 = min(K, k0 + kper);
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.f;
    if (n < N) {
        const float* bp = B + n;
        int k = k0 + wave;
        for (; k + 28 < k1; k += 32) {
            float bv[8], av[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                bv[u] = bp[(int64_t)(k + 4 * u) * sBk];
                av[u] = a[(int64_t)(k + 4 * u) * sAk];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] = fmaf(av[u], bv[u], acc[u]);
        }
        for (; k < k1; k += 4) acc[0] = fmaf(a[(int64_t)k * sAk], bp[(int64_t)k * sBk], acc[0]);
    }
    red[wave][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (wave == 0 && n < N) partial[(int64_t)blockIdx.y * N + n] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}
static inline bool sgemm_vecmat_ok(int M, int N, int K, int64_t sBn) { return M == 1 && sBn == 1 && N >= 64 && K >= 64; }

int sgemm_splitk(const float* A, int64_t sAm, int64_t sAk, const float* B, int64_t sBn, int64_t sBk, float* C, int64_t ldc,
                        int M, int N, int K, bool accumulate, float* partial, hipStream_t st, const float* amax_a, int amax_na, const float* amax_b,
                        int amax_nb, void* plane_ws, size_t plane_ws_bytes) {
    if (M <= 0 || N <= 0) return RULGNN_OK;
    // (also in front of the one-workgroup kernel: ASTGCNN's d fc.weight, [1 x 512] . [512 x 64], took 23.6 us there)
    if (!accumulate && (sgemm_tiny_ok(M, N, K) || sgemm_longk_blocks(M, N, K) == 0) && sgemm_vecmat_ok(M, N, K, sBn)) {
        // (the row chunks fit the scratch the caller sized with sgemm_splitk_need_floats: never more chunks than the tile path's slices)
        int ks = sgemm_splitk_slices(M, N, K);
        const int want = (512 + (N + 63) / 64 - 1) / ((N + 63) / 64);                  // ~512 workgroups
        ks = ks < want ? ks : want;
        ks = ks > 32 ? 32 : ks;                                                        // (rows_sum_kernel: 32 row slices)
        const int fit = (int)(sgemm_splitk_need_floats(M, N, K) / (size_t)N);          // (whichever path sized the caller's scratch)
        ks = ks > fit ? (fit > 0 ? fit : 1) : ks;
        if (K < 512) ks = 1;                                                           // (a batch-100-sized reduction: one launch straight into C)
        int kper = (K + ks - 1) / ks;
        kper = (kper + 31) & ~31;
        ks = (K + kper - 1) / kper;
        (void)hipGetLastError();
        hipLaunchKernelGGL(sgemm_vecmat_kernel, dim3((N + 63) / 64, ks), dim3(256), 0, st, A, sAk, B, sBk, ks > 1 ? partial : C, N, K, kper);
        if (hipGetLastError() != hipSuccess) return RULGNN_EHIP;
        return ks > 1 ? rows_sum(partial, ks, N, N, C, st) : RULGNN_OK;
    }
    if (sgemm_tiny_ok(M, N, K)) {
        GemmArgs g{A, sAm, sAk, B, sBn, sBk, C, ldc, M, N, K, 0, K};
        (void)hipGetLastError();
        hipLaunchKernelGGL(sgemm_tiny_kernel, dim3(1), dim3(1024), 0, st, g, accumulate ? 1 : 0);
        return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
    }
    if (sgemm_longk_blocks(M, N, K) > 0) {
        int nblk = sgemm_longk_blocks(M, N, K);
        int kper = (K + nblk - 1) / nblk;
        kper = (kper + SKT_ROWS - 1) / SKT_ROWS * SKT_ROWS;
        nblk = (K + kper - 1) / kper;
        GemmArgs g{A, sAm, sAk, B, sBn, sBk, partial, N, M, N, K, 0, kper};
        (void)hipGetLastError();
        if (sgemm_longk_mfma_ok(g)) sgemm_longk_mfma_launch(g, kper, 0, nblk, st);
        else
        hipLaunchKernelGGL(sgemm_longk_kernel, dim3(nblk), dim3(256), ((size_t)SKT_ROWS * (M + N) + 256) * sizeof(float), st, g, kper, 0);
        hipLaunchKernelGGL(sgemm_longk_reduce_kernel, dim3((M * N + 3) / 4), dim3(256), 0, st, (const float*)partial, C, ldc, M, N, nblk,
                           accumulate ? 1 : 0, (float*)nullptr);
        return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
    }
    const int slices = sgemm_splitk_slices(M, N, K);
    if (slices <= 1) return sgemm(A, sAm, sAk, B, sBn, sBk, C, ldc, M, N, K, accumulate, st, 0, amax_a, amax_na, amax_b, amax_nb, plane_ws, plane_ws_bytes);
    if (amax_a && amax_b && plane_ws && sgemm_big_mode() == 1) {
        // the pre-split kernel with its own slice count (never more slices than `partial` was sized for)
        const int ps = sgemm_planes_slices(M, N, K, true);
        if (ps >= 2 && ps <= slices && plane_ws_bytes >= sgemm_planes_ws_bytes(M, N, K) && sgemm_planes_ok(A, sAm, sAk, B, sBn, sBk, M, N, K, ps)) {
            const int rc = sgemm_planes(A, sAm, sAk, B, sBn, sBk, partial, N, M, N, K, false, ps, amax_a, amax_na, amax_b, amax_nb, plane_ws,
                                        plane_ws_bytes, st);
            if (rc != RULGNN_OK) return rc;
            return sgemm_reduce_slices(partial, C, ldc, M, N, ps, accumulate, st);
        }
    }
    int kchunk = (K + slices - 1) / slices;
    kchunk = (kchunk + 15) & ~15;
    const int used = (K + kchunk - 1) / kchunk;
    GemmArgs g{A, sAm, sAk, B, sBn, sBk, partial, N, M, N, K, 0, kchunk, amax_a, amax_b, amax_na, amax_nb};
    (void)hipGetLastError();
    sgemm_launch_tiles(g, used, st);
    if (hipGetLastError() != hipSuccess) return RULGNN_EHIP;
    return sgemm_reduce_slices(partial, C, ldc, M, N, used, accumulate, st);
}

// sgemm_splitk plus colsum[m] = sum_k A(m, k) from the same pass (a weight gradient and the bias gradient that goes with it): on the
// long-k path the sums are one more output column against a virtual column of ones; elsewhere two calls (`ones`: K ones, stride 0).
int sgemm_splitk_colsum(const float* A, int64_t sAm, int64_t sAk, const float* B, int64_t sBn, int64_t sBk, float* C, int64_t ldc,
                        int M, int N, int K, float* colsum, const float* ones, float* partial, hipStream_t st) {
    if (M <= 0 || N <= 0) return RULGNN_OK;
    const int NE = N + 1;
    if (!sgemm_tiny_ok(M, N, K) && sgemm_longk_blocks(M, NE, K) > 0) {
        int nblk = sgemm_longk_blocks(M, NE, K);
        int kper = (K + nblk - 1) / nblk;
        kper = (kper + SKT_ROWS - 1) / SKT_ROWS * SKT_ROWS;
        nblk = (K + kper - 1) / kper;
        GemmArgs g{A, sAm, sAk, B, sBn, sBk, partial, NE, M, NE, K, 0, kper};
        (void)hipGetLastError();
        if (sgemm_longk_mfma_ok(g)) sgemm_longk_mfma_launch(g, kper, 1, nblk, st);
        else
        hipLaunchKernelGGL(sgemm_longk_kernel, dim3(nblk), dim3(256), ((size_t)SKT_ROWS * (M + NE) + 256) * sizeof(float), st, g, kper, 1);
        hipLaunchKernelGGL(sgemm_longk_reduce_kernel, dim3((M * NE + 3) / 4), dim3(256), 0, st, (const float*)partial, C, ldc, M, NE, nblk, 0,
                           colsum);
        return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
    }
    const int rc = sgemm_splitk(A, sAm, sAk, B, sBn, sBk, C, ldc, M, N, K, false, partial, st);
    if (rc != RULGNN_OK) return rc;
    return sgemm_splitk(ones, 0, 0, A, sAm, sAk, colsum, M, 1, M, K, false, partial, st);
}

struct LongkReduceBatch {
    const float* partial[LONGK_BATCH_MAX];
    float* C[LONGK_BATCH_MAX];
    float* colsum[LONGK_BATCH_MAX];
    int64_t ldc[LONGK_BATCH_MAX];
    int M[LONGK_BATCH_MAX], N[LONGK_BATCH_MAX], nblk[LONGK_BATCH_MAX];
};
static __global__ __launch_bounds__(256) void sgemm_longk_reduce_batch_kernel(LongkReduceBatch r) {
    const int j = blockIdx.y;
    const int M = r.M[j], N = r.N[j], nblk = r.nblk[j];
    const int o = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (o >= M * N) return;
    const float* partial = r.partial[j];
    float a = 0.f;
    for (int b = lane; b < nblk; b += 64) a += partial[(int64_t)b * M * N + o];                      // (same order as sgemm_longk_reduce_kernel)
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) a += __shfl_xor(a, m, 64);
    if (lane == 0) {
        const int mi = o / N, nj = o % N;
        if (nj == N - 1) r.colsum[j][mi] = a;
        else r.C[j][(int64_t)mi * r.ldc[j] + nj] = a;
    }
}
// sgemm_splitk_colsum for several products at once: where every one of them is a long-k product of the same matrix-core tile shape (and the
// scratch holds all their partial rows) two launches serve them all -- bit-identical to the separate calls; otherwise the calls one by one
size_t sgemm_splitk_colsum_batch_floats(const SplitKColsumJob* jobs, int n) {
    size_t tot = 0, mx = 0;
    for (int i = 0; i < n; ++i) {
        const size_t v = sgemm_splitk_need_floats(jobs[i].M, jobs[i].N + 1, jobs[i].K), w = sgemm_splitk_need_floats(jobs[i].M, jobs[i].N, jobs[i].K);
        const size_t u = sgemm_splitk_need_floats(jobs[i].M, 1, jobs[i].K);
        tot += v;
        mx = mx > v ? mx : v; mx = mx > w ? mx : w; mx = mx > u ? mx : u;
    }
    return tot > mx ? tot : mx;
}
int sgemm_splitk_colsum_batch(const SplitKColsumJob* jobs, int n, const float* ones, float* partial, size_t partial_floats, hipStream_t st) {
    bool batch = n >= 2 && n <= LONGK_BATCH_MAX;
    int MT0 = 0, NT0 = 0;
    size_t tot = 0;
    for (int i = 0; i < n && batch; ++i) {
        const SplitKColsumJob& q = jobs[i];
        const int NE = q.N + 1;
        if (q.M <= 0 || q.N <= 0 || sgemm_tiny_ok(q.M, q.N, q.K) || sgemm_longk_blocks(q.M, NE, q.K) <= 0 || q.sAm != 1 || q.sBn != 1 || q.M > 32 || NE > 64) {
            batch = false;
            break;
        }
        const int MT = (q.M + 15) / 16, NT = (NE + 15) / 16;
        if (i == 0) { MT0 = MT; NT0 = NT; }
        else if (MT != MT0 || NT != NT0) batch = false;
        tot += sgemm_splitk_need_floats(q.M, NE, q.K);
    }
    if (!batch || tot > partial_floats) {
        for (int i = 0; i < n; ++i) {
            const SplitKColsumJob& q = jobs[i];
            const int rc = sgemm_splitk_colsum(q.A, q.sAm, q.sAk, q.B, q.sBn, q.sBk, q.C, q.ldc, q.M, q.N, q.K, q.colsum, ones, partial, st);
            if (rc != RULGNN_OK) return rc;
        }
        return RULGNN_OK;
    }
    LongkBatch lb{};
    LongkReduceBatch rb{};
    float* part = partial;
    int maxw = 0, maxo = 0;
    for (int i = 0; i < n; ++i) {
        const SplitKColsumJob& q = jobs[i];
        const int NE = q.N + 1;
        int nblk = sgemm_longk_blocks(q.M, NE, q.K);
        int kper = (q.K + nblk - 1) / nblk;
        kper = (kper + SKT_ROWS - 1) / SKT_ROWS * SKT_ROWS;
        nblk = (q.K + kper - 1) / kper;
        lb.g[i] = GemmArgs{q.A, q.sAm, q.sAk, q.B, q.sBn, q.sBk, part, NE, q.M, NE, q.K, 0, kper};
        lb.kper[i] = kper; lb.nwaves[i] = nblk;
        rb.partial[i] = part; rb.C[i] = q.C; rb.colsum[i] = q.colsum; rb.ldc[i] = q.ldc; rb.M[i] = q.M; rb.N[i] = NE; rb.nblk[i] = nblk;
        part += sgemm_splitk_need_floats(q.M, NE, q.K);
        maxw = nblk > maxw ? nblk : maxw;
        maxo = q.M * NE > maxo ? q.M * NE : maxo;
    }
    (void)hipGetLastError();
    const dim3 grid((maxw + 3) / 4, n), block(256);
#define RULGNN_LKB(mt, nt) hipLaunchKernelGGL((sgemm_longk_mfma_batch_kernel<mt, nt>), grid, block, 0, st, lb, 1)
    if (MT0 == 1) { if (NT0 == 1) RULGNN_LKB(1, 1); else if (NT0 == 2) RULGNN_LKB(1, 2); else if (NT0 == 3) RULGNN_LKB(1, 3); else RULGNN_LKB(1, 4); }
    else { if (NT0 == 1) RULGNN_LKB(2, 1); else if (NT0 == 2) RULGNN_LKB(2, 2); else if (NT0 == 3) RULGNN_LKB(2, 3); else RULGNN_LKB(2, 4); }
#undef RULGNN_LKB
    hipLaunchKernelGGL(sgemm_longk_reduce_batch_kernel, dim3((maxo + 3) / 4, n), dim3(256), 0, st, rb);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

// out[e] = sum_r part[r][e] over `rows` per-workgroup partial rows of n values (row stride ld): 32 columns x 32 row slices per
// workgroup (twice the workgroups of the 64 x 16 form: 118 instead of 59 for the 3 750 TCN weights of ASTGCNN, which left three
// quarters of the CUs idle), four independent partial sums per thread so that four loads are in flight, the slices combined through
// LDS in a fixed order (deterministic).  One thread per column walking every row alone was 140-190 us in the finalize kernels of
// three families.
static __global__ __launch_bounds__(1024) void rows_sum_kernel(const float* __restrict__ part, int rows, int64_t ld, int n,
                                                               float* __restrict__ out) {
    __shared__ float red[32][33];
    const int lane = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + lane;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (e < n) {
        const float* p = part + e;
        int r = sl;
        for (; r + 96 < rows; r += 128) {
            a0 += p[(int64_t)r * ld];
            a1 += p[(int64_t)(r + 32) * ld];
            a2 += p[(int64_t)(r + 64) * ld];
            a3 += p[(int64_t)(r + 96) * ld];
        }
        for (; r < rows; r += 32) a0 += p[(int64_t)r * ld];
    }
    red[sl][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (sl == 0 && e < n) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 32; ++q) v += red[q][lane];
        out[e] = v;
    }
}
// up to three such sums in one launch (blockIdx.y picks the job; reduce_device.hpp)
static __global__ __launch_bounds__(1024) void rows_sum_multi_kernel(RowsSumJobs jb) {
    __shared__ float red[32][33];
    if ((int)blockIdx.x * 32 >= jb.n[blockIdx.y]) return;
    rows_sum_job_body(jb, blockIdx.y, blockIdx.x, red);
}
int rows_sum2(const float* partA, float* outA, const float* partB, float* outB, int rows, int64_t ld, int n, hipStream_t st) {
    return rows_sum3(partA, outA, partB, outB, rows, ld, n, nullptr, nullptr, nullptr, 0, 0, 0, st);
}
int rows_sum3(const float* partA, float* outA, const float* partB, float* outB, int rows, int64_t ld, int n, const float* partC, float* outC,
              float* outC2, int rowsC, int64_t ldC, int nC, hipStream_t st) {
    if (n <= 0) return RULGNN_OK;
    RowsSumJobs jb{};
    jb.part[0] = partA; jb.out[0] = outA; jb.rows[0] = rows; jb.n[0] = n; jb.ld[0] = ld;
    jb.part[1] = partB; jb.out[1] = outB; jb.rows[1] = rows; jb.n[1] = n; jb.ld[1] = ld;
    jb.part[2] = partC; jb.out[2] = outC; jb.out2[2] = outC2; jb.rows[2] = rowsC; jb.n[2] = nC; jb.ld[2] = ldC;
    const int nmax = n > nC ? n : nC;
    (void)hipGetLastError();
    hipLaunchKernelGGL(rows_sum_multi_kernel, dim3((nmax + 31) / 32, partC && nC > 0 ? 3 : 2), dim3(1024), 0, st, jb);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}
int rows_sum_three(const float* const part[3], float* const out[3], const int rows[3], const int64_t ld[3], const int n[3], hipStream_t st) {
    RowsSumJobs jb{};
    int nmax = 0;
    for (int j = 0; j < 3; ++j) {
        jb.part[j] = part[j]; jb.out[j] = out[j]; jb.rows[j] = rows[j]; jb.ld[j] = ld[j]; jb.n[j] = n[j] > 0 ? n[j] : 0;
        nmax = jb.n[j] > nmax ? jb.n[j] : nmax;
    }
    if (nmax <= 0) return RULGNN_OK;
    (void)hipGetLastError();
    hipLaunchKernelGGL(rows_sum_multi_kernel, dim3((nmax + 31) / 32, 3), dim3(1024), 0, st, jb);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}
int rows_sum(const float* part, int rows, int64_t ld, int n, float* out, hipStream_t st) {
    if (n <= 0) return RULGNN_OK;
    (void)hipGetLastError();
    hipLaunchKernelGGL(rows_sum_kernel, dim3((n + 31) / 32), dim3(1024), 0, st, part, rows, ld, n, out);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

// dst[c] = sum over the rows of src[r][c] for a SHORT matrix (bias gradients over batch-sized row counts): one workgroup, a thread per
// (row stripe, column), fixed-order sum of the stripes -- one ~5-us launch instead of a split-K pair (19 + 7 us)
static __global__ __launch_bounds__(1024) void cols_sum_small_kernel(const float* __restrict__ src, int rows, int C, float* __restrict__ dst) {
    __shared__ float red[1024];
    const int stripes = 1024 / C, c = threadIdx.x % C, sidx = threadIdx.x / C;
    float a = 0.f;
    if (sidx < stripes)
        for (int r = sidx; r < rows; r += stripes) a += src[(int64_t)r * C + c];
    red[threadIdx.x] = sidx < stripes ? a : 0.f;
    __syncthreads();
    if ((int)threadIdx.x < C) {
        float v = 0.f;
        for (int k = 0; k < stripes; ++k) v += red[k * C + threadIdx.x];
        dst[threadIdx.x] = v;
    }
}
int cols_sum_small(const float* src, int rows, int C, float* dst, hipStream_t st) {
    if (C == 1) return block_sum(src, rows, dst, st);              // (one column: a tree, not one thread adding 1024 stripes)
    (void)hipGetLastError();
    hipLaunchKernelGGL(cols_sum_small_kernel, dim3(1), dim3(1024), 0, st, src, rows, C, dst);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

// out[0] = sum(v[0..n)) with one workgroup: strided partial sums, then a fixed-order tree (deterministic).
static __global__ __launch_bounds__(1024) void block_sum_kernel(const float* __restrict__ v, int64_t n, float* __restrict__ out) {
    __shared__ float red[1024];
    float a = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) a += v[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int m = 512; m > 0; m >>= 1) {
        if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}


int block_sum(const float* v, int64_t n, float* out, hipStream_t st) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(block_sum_kernel, dim3(1), dim3(1024), 0, st, v, n, out);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

}  // namespace rulgnn
