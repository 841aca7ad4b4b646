// probe (gfx950): how many independent VALU instructions hide behind one MFMA of each class, one SIMD's view
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define REP 64
// KIND: 0 f16 16x16x32 (4 pass), 1 f32 16x16x4 (8 pass), 2 f32 16x16x1_4b, 3 f16 32x32x16 (8 pass);  NF fillers per MFMA; FT filler type
template <int KIND, int NF, int FT>
__global__ __launch_bounds__(256) void k(float* out, int iters, float fa, float fb) {
    float x[8]; unsigned u[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x + i; u[i] = threadIdx.x * 3 + i; }
    f4 acc[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
    f16v big[2] = {{0},{0}};
    h8 a8, b8; for (int j = 0; j < 8; ++j) { a8[j] = (_Float16)(fa + j); b8[j] = (_Float16)(fb + j); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (KIND == 0) acc[r & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[r & 3], 0, 0, 0);
            if (KIND == 1) acc[r & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[r & 3], 0, 0, 0);
            if (KIND == 2) big[r & 1] = __builtin_amdgcn_mfma_f32_16x16x1f32(fa, fb, big[r & 1], 0, 0, 0);
            if (KIND == 3) big[r & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, big[r & 1], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int i = (r * NF + f) & 7;
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
    s += big[0][0] + big[1][5];
    if (s == 12345.678f) out[0] = s;
}
template <int KIND, int NF, int FT>
void run(int wps) {
    float* d; (void)hipMalloc(&d, 4);
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<KIND, NF, FT><<<256 * wps, 256>>>(d, 10, 0.999f, 0.001f); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<KIND, NF, FT><<<256 * wps, 256>>>(d, iters, 0.999f, 0.001f); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    static const char* kn[] = {"f16 16x16x32", "f32 16x16x4", "f32 16x16x1_4b", "f16 32x32x16"};
    static const char* fn[] = {"v_fma_f32", "v_cvt_pk_f16", "v_and_b32", "v_mov_dpp"};
    printf("%-15s + %d x %-13s waves/SIMD=%d : %6.2f cyc per group\n", kn[KIND], NF, fn[FT], wps, ms * 1e6 / ((double)iters * REP * wps) * 2.4);
    (void)hipFree(d);
}
template <int KIND> void sweep(int w) {
    run<KIND, 0, 0>(w); run<KIND, 1, 0>(w); run<KIND, 2, 0>(w); run<KIND, 3, 0>(w); run<KIND, 4, 0>(w); run<KIND, 6, 0>(w); run<KIND, 8, 0>(w); run<KIND, 12, 0>(w);
    run<KIND, 4, 1>(w); run<KIND, 4, 2>(w); run<KIND, 4, 3>(w);
}
int main() {
    for (int w : {1, 3}) { sweep<0>(w); sweep<1>(w); sweep<2>(w); sweep<3>(w); }
    return 0;
}
