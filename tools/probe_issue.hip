// probe (gfx950): what shares the SIMD's issue port.
//   (1) two co-resident wavefronts of one SIMD, one issuing only matrix-core instructions, the other only VALU: do they overlap?
//   (2) one instruction stream: MFMA + NF independent fillers, for the f16 16x16x32 / 16x16x16 forms, C = register or inline 0
//   (3) issue rates of the candidates for a cheaper f16 split (v_dot2*_f32_f16, v_cvt_pkrtz, ...)
// build: hipcc --offload-arch=gfx950 -O3 tools/probe_issue.hip -o tools/probe_issue.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define REP 64

static double g_ghz = 2.4;

// ---- (1) heterogeneous pair ---------------------------------------------------------------------------------------------
// 512-thread workgroup: wavefronts 0-3 and 4-7 land pairwise on the four SIMDs.  role bit 0: first half runs, bit 1: second half runs.
// MK: 0 f16 16x16x32 C=reg, 1 f16 16x16x32 C=0, 2 f32 16x16x4.   FT: filler type.
template <int MK, int FT>
__global__ __launch_bounds__(512) void hetero(float* out, int iters, int roles, float fa, float fb) {
    const int half = threadIdx.x >> 8;
    float x[8]; unsigned u[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x + i; u[i] = threadIdx.x * 3 + i; }
    f4 acc[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
    h8 a8, b8; for (int j = 0; j < 8; ++j) { a8[j] = (_Float16)(fa + j); b8[j] = (_Float16)(fb + j); }
    const f4 zero = {0, 0, 0, 0};
    if (half == 0) {
        if (!(roles & 1)) return;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < REP; ++r) {
                if (MK == 0) acc[r & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[r & 3], 0, 0, 0);
                if (MK == 1) { acc[r & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, zero, 0, 0, 0); asm volatile("" : "+v"(acc[r & 3])); }
                if (MK == 2) acc[r & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[r & 3], 0, 0, 0);
            }
        }
    } else {
        if (!(roles & 2)) return;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < REP * 8; ++r) {
                const int i = r & 7;
                if (FT == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(fa), "v"(fb));
                if (FT == 1) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(x[i]), "v"(x[(i + 1) & 7]));
                if (FT == 2) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (FT == 3) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(u[i]) : "v"(u[(i + 1) & 7]));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + (float)u[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}

template <int MK, int FT>
static void run_hetero(const char* mk, const char* ft, int blocks_per_cu) {
    float* d; (void)hipMalloc(&d, 4);
    const int iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    double t[4] = {0, 0, 0, 0};
    for (int roles = 1; roles <= 3; ++roles) {
        hetero<MK, FT><<<256 * blocks_per_cu, 512>>>(d, 10, roles, 0.999f, 0.001f); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); hetero<MK, FT><<<256 * blocks_per_cu, 512>>>(d, iters, roles, 0.999f, 0.001f); (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        t[roles] = ms * 1e6 * g_ghz / ((double)iters * REP * blocks_per_cu);      // cycles per (1 MFMA | 8 fillers) group and SIMD
    }
    printf("hetero %-16s | 8 x %-12s  wg/CU=%d : mfma-only %6.2f  valu-only %6.2f  both %6.2f cyc per (1 mfma, 8 valu)\n", mk, ft, blocks_per_cu,
           t[1], t[2], t[3]);
    (void)hipFree(d);
}

// ---- (2) one stream ------------------------------------------------------------------------------------------------------
// KIND: 0 f16 16x16x32 C=reg, 1 f16 16x16x32 C=inline 0, 2 f16 16x16x16 C=reg, 3 f16 16x16x16 C=0, 4 f16 32x32x16
typedef float f16v __attribute__((ext_vector_type(16)));
template <int KIND, int NF, int FT>
__global__ __launch_bounds__(256) void stream(float* out, int iters, float fa, float fb) {
    float x[8]; unsigned u[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x + i; u[i] = threadIdx.x * 3 + i; }
    f4 acc[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
    f16v big[2] = {{0}, {0}};
    h8 a8, b8; for (int j = 0; j < 8; ++j) { a8[j] = (_Float16)(fa + j); b8[j] = (_Float16)(fb + j); }
    h4 a4, b4; for (int j = 0; j < 4; ++j) { a4[j] = (_Float16)(fa + j); b4[j] = (_Float16)(fb + j); }
    const f4 zero = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (KIND == 0) acc[r & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[r & 3], 0, 0, 0);
            if (KIND == 1) { acc[r & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, zero, 0, 0, 0); asm volatile("" : "+v"(acc[r & 3])); }
            if (KIND == 2) acc[r & 3] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[r & 3], 0, 0, 0);
            if (KIND == 3) { acc[r & 3] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, zero, 0, 0, 0); asm volatile("" : "+v"(acc[r & 3])); }
            if (KIND == 4) big[r & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, big[r & 1], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int i = (r * NF + f) & 7;
                if (FT == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(fa), "v"(fb));
                if (FT == 1) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(x[i]), "v"(x[(i + 1) & 7]));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + (float)u[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
    s += big[0][0] + big[1][5];
    if (s == 12345.678f) out[0] = s;
}
template <int KIND, int NF, int FT>
static void run_stream(int wps) {
    float* d; (void)hipMalloc(&d, 4);
    const int iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    stream<KIND, NF, FT><<<256 * wps, 256>>>(d, 10, 0.999f, 0.001f); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); stream<KIND, NF, FT><<<256 * wps, 256>>>(d, iters, 0.999f, 0.001f); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    static const char* kn[] = {"f16 16x16x32 C=reg", "f16 16x16x32 C=0", "f16 16x16x16 C=reg", "f16 16x16x16 C=0", "f16 32x32x16"};
    static const char* fn[] = {"v_fma_f32", "v_cvt_pk_f16"};
    printf("stream %-19s + %2d x %-12s waves/SIMD=%d : %6.2f cyc per group\n", kn[KIND], NF, fn[FT], wps,
           ms * 1e6 * g_ghz / ((double)iters * REP * wps));
    (void)hipFree(d);
}
template <int KIND> static void sweep(int w) {
    run_stream<KIND, 0, 0>(w); run_stream<KIND, 1, 0>(w); run_stream<KIND, 2, 0>(w); run_stream<KIND, 4, 0>(w); run_stream<KIND, 6, 0>(w);
    run_stream<KIND, 8, 0>(w); run_stream<KIND, 12, 0>(w); run_stream<KIND, 4, 1>(w);
}

// ---- (3) issue rates -----------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void rate(float* out, int iters, float a, float b) {
    float x[8]; unsigned u[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x + i; u[i] = threadIdx.x * 7 + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                if (MODE == 1) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(x[i]), "v"(a));
                if (MODE == 2) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(x[i]), "v"(a));
                if (MODE == 3) asm volatile("v_dot2_f32_f16 %0, %1, %2, %3" : "=v"(x[i]) : "v"(u[i]), "v"(u[(i + 1) & 7]), "v"(x[(i + 2) & 7]));
                if (MODE == 4) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(x[i]) : "v"(u[i]), "v"(u[(i + 1) & 7]));
                if (MODE == 5) asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
                if (MODE == 6) asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
                if (MODE == 7) asm volatile("v_and_b32 %0, 0xffffe000, %1" : "=v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (MODE == 8) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (MODE == 9) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(x[i]) : "v"(u[i]), "v"(a));
                if (MODE == 10) asm volatile("v_mad_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(x[i]) : "v"(u[i]), "v"(a));
                if (MODE == 11) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(x[i]) : "v"(u[i]));
                if (MODE == 12) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(x[i]) : "v"(u[i]));
                if (MODE == 13) asm volatile("v_sub_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "=v"(x[i]) : "v"(x[(i + 1) & 7]), "v"(a));
                if (MODE == 14) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (MODE == 15) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (MODE == 16) asm volatile("v_pk_max_f16 %0, %1, %2" : "=v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
                if (MODE == 17) asm volatile("v_add_f32 %0, |%0|, %0" : "+v"(x[i]));
                if (MODE == 18) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*reinterpret_cast<double*>(&x[i & 6])) : "v"(*reinterpret_cast<double*>(&x[(i + 2) & 6])), "v"(*reinterpret_cast<double*>(&x[(i + 4) & 6])));
                if (MODE == 19) asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]), "v"(u[(i + 3) & 7]));
                if (MODE == 20) asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]), "v"(u[(i + 3) & 7]));
                if (MODE == 21) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]), "v"(u[(i + 3) & 7]));
                if (MODE == 22) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                if (MODE == 23) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                if (MODE == 24) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (MODE == 25) asm volatile("v_dot2_f32_f16 %0, %1, %2, %3" : "=v"(x[i]) : "v"(u[i]), "s"(0xbc00u), "v"(x[(i + 2) & 7]));
            }
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += x[i] + (float)u[i];
    if (s == 12345.678f) out[0] = s;
}
template <int MODE> static void run_rate(const char* name, int wps) {
    float* d; (void)hipMalloc(&d, 4);
    const int iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    rate<MODE><<<256 * wps, 256>>>(d, 10, 0.999f, 0.001f); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); rate<MODE><<<256 * wps, 256>>>(d, iters, 0.999f, 0.001f); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("rate %-28s waves/SIMD=%d  %.2f cyc/inst\n", name, wps, ms * 1e6 * g_ghz / ((double)iters * 256 * wps));
    (void)hipFree(d);
}

// ---- clock: s_memtime ticks (shader clock) against wall time under a VALU load ------------------------------------------
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
    {
        unsigned long long* d; (void)hipMalloc(&d, 32);
        clk<<<1024, 256>>>(d, 20000, 0.999f, 0.001f); (void)hipDeviceSynchronize();
        clk<<<1024, 256>>>(d, 200000, 0.999f, 0.001f); (void)hipDeviceSynchronize();
        unsigned long long h[2]; (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        int wc_khz = 0; (void)hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, 0);
        const double secs = (double)h[1] / (wc_khz * 1e3);
        printf("clock: %llu shader ticks in %llu wall ticks (wall clock %d kHz) -> %.3f GHz under a VALU load\n", h[0], h[1], wc_khz, h[0] / secs * 1e-9);
        if (secs > 0 && h[0] / secs > 1e9 && h[0] / secs < 3e9) g_ghz = h[0] / secs * 1e-9;
        (void)hipFree(d);
    }
    printf("cycles below assume %.3f GHz\n", g_ghz);
    for (int w : {1, 2}) {
        run_hetero<0, 0>("f16 16x16x32", "v_fma_f32", w); run_hetero<0, 1>("f16 16x16x32", "v_cvt_pk_f16", w);
        run_hetero<0, 2>("f16 16x16x32", "v_and_b32", w); run_hetero<0, 3>("f16 16x16x32", "v_mov_dpp", w);
        run_hetero<1, 0>("f16 16x16x32 C=0", "v_fma_f32", w); run_hetero<2, 0>("f32 16x16x4", "v_fma_f32", w);
    }
    for (int w : {1, 2}) { sweep<0>(w); sweep<1>(w); sweep<2>(w); sweep<3>(w); sweep<4>(w); }
    for (int w : {1, 2, 4}) {
        run_rate<0>("v_fma_f32", w); run_rate<1>("v_cvt_pk_f16_f32", w); run_rate<2>("v_cvt_pkrtz_f16_f32", w); run_rate<3>("v_dot2_f32_f16", w);
        run_rate<4>("v_dot2c_f32_f16", w); run_rate<5>("v_pk_mul_f16", w); run_rate<6>("v_pk_add_f16", w); run_rate<7>("v_and_b32 literal", w);
        run_rate<8>("v_sub_f32", w); run_rate<9>("v_fma_mix_f32", w); run_rate<11>("v_cvt_f32_f16", w);
        run_rate<12>("v_cvt_f32_f16_sdwa hi", w); run_rate<13>("v_sub_f32_sdwa", w); run_rate<14>("v_mov_b32_dpp", w); run_rate<15>("v_max_f32", w);
        run_rate<16>("v_pk_max_f16", w); run_rate<17>("v_add_f32 |x|,x", w); run_rate<18>("v_pk_fma_f32", w); run_rate<19>("v_pk_fma_f16", w);
        run_rate<20>("v_bfi_b32", w); run_rate<21>("v_perm_b32", w); run_rate<22>("v_fmac_f32", w); run_rate<23>("v_max3_f32", w);
        run_rate<24>("v_mul_f32", w); run_rate<25>("v_dot2_f32_f16 sgpr", w);
    }
    return 0;
}
