// probe (gfx950), round 5: does matrix-core time hide under VALU time WITHIN one SIMD when every wavefront runs a MIXED stream?
//
// Asked by the round-4 review: profiles/r03_probe_issue.txt (hetero / stream rows) says the costs of an f16 16x16x32 MFMA and
// of the VALU instructions next to it ADD on a SIMD; /opt/skills/guides/MI355X_MICROARCH.md ("MFMA and VALU pipes are separate
// ... both = max, not sum") says they overlap.  The round-3 probe timed with wall time x an assumed clock, and its "C = 0"
// rows were common-subexpression-eliminated by the compiler (ONE v_mfma per 64 in the ISA).  This probe
//   * writes every instruction as `asm volatile` (nothing merged, nothing reordered; the order in the source IS the order issued),
//   * counts SHADER cycles with s_memtime inside the kernel (no clock assumption; a power throttle cannot fake a sum),
//   * runs the mix of the real kernel (stgcn_forward_mx: per f16 MFMA ~7 VALU of which ~1.6 v_cvt_pk, ~1.4 v_and, rest fp32),
//   * at 1 / 2 / 3 / 4 wavefronts per SIMD, MFMAs spread (1 MFMA, NV VALU) or clustered (4 MFMA, 4 NV VALU, the shape of a
//     software-pipelined tile pair), with and without s_setprio around the cluster,
//   * for the three MFMA classes in question: f16 16x16x32 (4 passes), f16 32x32x16 (8 passes), f32 16x16x4 (8 passes).
// Output: cycles per group and SIMD, beside max(MFMA-only, VALU-only) and their sum.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe_mixed.hip -o tools/probe_mixed.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define GROUPS 32          // groups per loop body (unrolled)

// MK: 0 none, 1 f16 16x16x32 C = D (4 rotating accumulators), 2 f16 16x16x32 C = inline 0, 3 f16 32x32x16 (2 rotating), 4 f32 16x16x4
// VM: VALU mix: 0 v_fma_f32 only, 1 the kernel's mix (of 7: 2 cvt_pk, 1 and, 4 fp32)
// PAT: 0 spread (1 MFMA then NV VALU), 1 clustered (4 MFMA then 4 NV VALU), 2 clustered with s_setprio 1 around the MFMA cluster
template <int MK>
__device__ __forceinline__ void one_mfma(int r, f4 (&acc)[4], f16v (&big)[2], const h8& a8, const h8& b8, float fa, float fb) {
    if (MK == 1) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[r & 3]) : "v"(a8), "v"(b8));
    if (MK == 2) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "+v"(acc[r & 3]) : "v"(a8), "v"(b8));
    if (MK == 3) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(big[r & 1]) : "v"(a8), "v"(b8));
    if (MK == 4) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[r & 3]) : "v"(fa), "v"(fb));
}
template <int VM>
__device__ __forceinline__ void one_valu(int j, float (&x)[8], unsigned (&u)[8], float fa, float fb) {
    const int i = j & 7;
    if (VM == 0) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(fa), "v"(fb)); return; }
    switch (j % 7) {
        case 0: case 3: asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(x[i]), "v"(x[(i + 1) & 7])); break;
        case 1: asm volatile("v_and_b32 %0, 0xffffe000, %1" : "=v"(u[i]) : "v"(u[(i + 1) & 7])); break;
        case 2: case 5: asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[i]) : "v"(fa)); break;
        default: asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(fa), "v"(fb)); break;
    }
}

template <int MK, int NV, int VM, int PAT>
__global__ __launch_bounds__(256) void mixed(unsigned long long* cyc, float* out, int iters, float fa, float fb) {
    extern __shared__ float lds[];
    float x[8]; unsigned u[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x + i; u[i] = threadIdx.x * 3 + i; }
    f4 acc[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
    f16v big[2] = {{0}, {0}};
    h8 a8, b8; for (int j = 0; j < 8; ++j) { a8[j] = (_Float16)(fa + j); b8[j] = (_Float16)(fb + j); }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (PAT == 0) {
#pragma unroll
            for (int r = 0; r < GROUPS; ++r) {
                one_mfma<MK>(r, acc, big, a8, b8, fa, fb);
#pragma unroll
                for (int f = 0; f < NV; ++f) one_valu<VM>(r * NV + f, x, u, fa, fb);
            }
        } else {
#pragma unroll
            for (int r = 0; r < GROUPS; r += 4) {
                if (PAT == 2) asm volatile("s_setprio 1");
#pragma unroll
                for (int q = 0; q < 4; ++q) one_mfma<MK>(r + q, acc, big, a8, b8, fa, fb);
                if (PAT == 2) asm volatile("s_setprio 0");
#pragma unroll
                for (int f = 0; f < 4 * NV; ++f) one_valu<VM>(r * NV + f, x, u, fa, fb);
            }
        }
    }
    asm volatile("s_nop 7\n s_nop 7\n s_nop 7" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + (float)u[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
    s += big[0][0] + big[1][5];
    if (s == 12345.678f) out[0] = s + lds[threadIdx.x];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// hetero: 512-thread workgroups, wavefronts 0-3 MFMA only, 4-7 VALU only (they pair up on the four SIMDs)
template <int MK, int NV, int VM>
__global__ __launch_bounds__(512) void hetero(unsigned long long* cyc, float* out, int iters, int roles, float fa, float fb) {
    extern __shared__ float lds[];
    float x[8]; unsigned u[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x + i; u[i] = threadIdx.x * 3 + i; }
    f4 acc[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
    f16v big[2] = {{0}, {0}};
    h8 a8, b8; for (int j = 0; j < 8; ++j) { a8[j] = (_Float16)(fa + j); b8[j] = (_Float16)(fb + j); }
    const int half = threadIdx.x >> 8;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (half == 0) {
        if (roles & 1)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < GROUPS; ++r) one_mfma<MK>(r, acc, big, a8, b8, fa, fb);
            }
    } else {
        if (roles & 2)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < GROUPS * NV; ++r) one_valu<VM>(r, x, u, fa, fb);
            }
    }
    asm volatile("s_nop 7\n s_nop 7\n s_nop 7" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + (float)u[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
    s += big[0][0] + big[1][5];
    if (s == 12345.678f) out[0] = s + lds[threadIdx.x];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

static unsigned long long* d_cyc; static float* d_out;
static const int ITERS = 400;
static size_t lds_for(int wg_per_cu) { return (size_t)(160 * 1024 / wg_per_cu) - 2048; }   // pins the workgroups per CU

// average shader cycles of a wavefront from launch start to end; wps wavefronts share a SIMD -> per-group SIMD time = cyc / (groups_per_wave * wps)
template <int MK, int NV, int VM, int PAT>
static double run_mixed(int wps) {
    const int blocks = 256 * wps;
    auto kern = mixed<MK, NV, VM, PAT>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_for(wps));
    kern<<<blocks, 256, lds_for(wps)>>>(d_cyc, d_out, 8, 0.999f, 0.001f); (void)hipDeviceSynchronize();
    kern<<<blocks, 256, lds_for(wps)>>>(d_cyc, d_out, ITERS, 0.999f, 0.001f); (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    (void)hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    return med / ((double)ITERS * GROUPS * wps);
}
template <int MK, int NV, int VM>
static void run_hetero(const char* mk, int wg_per_cu, double out3[3]) {
    const int blocks = 256 * wg_per_cu;
    auto kern = hetero<MK, NV, VM>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_for(wg_per_cu));
    for (int roles = 1; roles <= 3; ++roles) {
        kern<<<blocks, 512, lds_for(wg_per_cu)>>>(d_cyc, d_out, 8, roles, 0.999f, 0.001f); (void)hipDeviceSynchronize();
        kern<<<blocks, 512, lds_for(wg_per_cu)>>>(d_cyc, d_out, ITERS, roles, 0.999f, 0.001f); (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks * 8);
        (void)hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost);
        // the slower role's wavefronts bound the pair: take the maximum of the two role medians
        std::vector<unsigned long long> m, v;
        for (int b = 0; b < blocks; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : v).push_back(h[b * 8 + w]);
        std::sort(m.begin(), m.end()); std::sort(v.begin(), v.end());
        const double mm = (double)m[m.size() / 2], vv = (double)v[v.size() / 2];
        out3[roles - 1] = std::max((roles & 1) ? mm : 0.0, (roles & 2) ? vv : 0.0) / ((double)ITERS * GROUPS * wg_per_cu);
    }
    printf("hetero %-18s | %2d x %-5s pairs/SIMD=%d : mfma-only %6.2f  valu-only %6.2f  both %6.2f   (max %6.2f  sum %6.2f) cyc per (1 mfma, NV valu) and SIMD\n",
           mk, NV, VM ? "mix" : "fma", wg_per_cu, out3[0], out3[1], out3[2], std::max(out3[0], out3[1]), out3[0] + out3[1]);
}

static const char* MKN[] = {"none", "f16 16x16x32 C=D", "f16 16x16x32 C=0", "f16 32x32x16", "f32 16x16x4"};
template <int MK, int NV, int VM>
static void line(int wps) {
    const double m = run_mixed<MK, 0, VM, 0>(wps);
    const double v = run_mixed<0, NV, VM, 0>(wps);
    const double s = run_mixed<MK, NV, VM, 0>(wps);
    const double c = run_mixed<MK, NV, VM, 1>(wps);
    const double p = run_mixed<MK, NV, VM, 2>(wps);
    printf("mixed  %-18s + %2d x %-5s waves/SIMD=%d : mfma-only %6.2f  valu-only %6.2f | spread %6.2f  clustered %6.2f  clustered+setprio %6.2f   (max %6.2f  sum %6.2f)  hidden %5.1f%% of min\n",
           MKN[MK], NV, VM ? "mix" : "fma", wps, m, v, s, c, p, std::max(m, v), m + v,
           100.0 * (m + v - std::min(s, std::min(c, p))) / std::min(m, v));
    fflush(stdout);
}
template <int MK>
static void sweep() {
    for (int wps : {1, 2, 3, 4}) {
        line<MK, 2, 0>(wps); line<MK, 4, 0>(wps); line<MK, 7, 0>(wps); line<MK, 12, 0>(wps);
        line<MK, 7, 1>(wps); line<MK, 14, 1>(wps);
    }
}

__global__ void clk(unsigned long long* out, int iters, float a, float b) {
    float x[8]; for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 64; ++r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[r & 7]) : "v"(a), "v"(b));
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    float s = 0; for (int i = 0; i < 8; ++i) s += x[i];
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; }
    if (s == 12345.678f) out[2] = 1;
}

int main() {
    (void)hipMalloc(&d_cyc, 8 * 8 * 256 * 8); (void)hipMalloc(&d_out, 64);
    {
        clk<<<1024, 256>>>(d_cyc, 20000, 0.999f, 0.001f); (void)hipDeviceSynchronize();
        clk<<<1024, 256>>>(d_cyc, 200000, 0.999f, 0.001f); (void)hipDeviceSynchronize();
        unsigned long long h[2]; (void)hipMemcpy(h, d_cyc, 16, hipMemcpyDeviceToHost);
        int wc_khz = 0; (void)hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, 0);
        printf("s_memtime: %llu ticks in %llu wall ticks (wall clock %d kHz) -> %.1f MHz; 'cycles' below are s_memtime ticks\n", h[0], h[1], wc_khz,
               (double)h[0] / (double)h[1] * wc_khz / 1e3);
    }
    // calibration of the tick: a VALU-only stream of plain fp32 at 4 waves/SIMD must come out at the documented 2 cycles per instruction
    printf("calibration: v_fma_f32 only, 12 per group: waves/SIMD=1 %.2f  2 %.2f  4 %.2f ticks per instruction and SIMD\n",
           run_mixed<0, 12, 0, 0>(1) / 12, run_mixed<0, 12, 0, 0>(2) / 12, run_mixed<0, 12, 0, 0>(4) / 12);
    sweep<1>(); sweep<2>(); sweep<3>(); sweep<4>();
    double o[3];
    for (int w : {1, 2}) {
        run_hetero<1, 7, 0>(MKN[1], w, o); run_hetero<1, 7, 1>(MKN[1], w, o); run_hetero<2, 7, 0>(MKN[2], w, o);
        run_hetero<3, 12, 0>(MKN[3], w, o); run_hetero<3, 14, 1>(MKN[3], w, o); run_hetero<4, 12, 0>(MKN[4], w, o);
    }
    return 0;
}
