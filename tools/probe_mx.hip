// probe (gfx950): what the matrix-core eval forward (csrc/stgcn_forward_mx.hip) relies on
//   1. operand / result layout of v_mfma_f32_16x16x32_f16 and v_mfma_f32_16x16x16_f16
//   2. accuracy of the 2-way f16 split (hi*hi + hi*lo + lo*hi) and the bf16 split against fp64
//   3. issue cost of the MFMA forms and of the VALU ops of the split / pointwise stages
//   4. global_load_lds_dwordx4 (LDS-DMA) placement
// build: hipcc --offload-arch=gfx950 -O3 tools/probe_mx.hip -o tools/probe_mx.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

// ---- 1. layouts: A [16][K], B [K][16] in global memory, hypothesis: lane (kg = l>>4, i = l&15) holds K-slots kg*KPL + j
template <int K>
__global__ void layout_k(const float* A, const float* B, float* D) {
    const int l = threadIdx.x, kg = l >> 4, i = l & 15;
    f4 acc = {0, 0, 0, 0};
    if constexpr (K == 32) {
        h8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (_Float16)A[i * 32 + kg * 8 + j]; b[j] = (_Float16)B[(kg * 8 + j) * 16 + i]; }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    } else {
        h4 a, b;
        for (int j = 0; j < 4; ++j) { a[j] = (_Float16)A[i * 16 + kg * 4 + j]; b[j] = (_Float16)B[(kg * 4 + j) * 16 + i]; }
        acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[(kg * 4 + r) * 16 + i] = acc[r];      // hypothesis: row = 4*(l>>4) + r, col = l&15
}

// ---- 2. split accuracy: D = A.B with fp32 inputs through f16 / bf16 hi+lo operands
__device__ inline unsigned pk_f16(float a, float b) { unsigned r; asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ inline unsigned pk_f16_lo(unsigned hi, float a, float b) {     // f16(a - hi.lo), f16(b - hi.hi)
    unsigned r = 0;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(hi), "v"(a), "v"(b));
    return r;
}
__device__ inline unsigned pk_bf16(float a, float b) { unsigned r; asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ inline unsigned pk_bf16_lo(unsigned hi, float a, float b) {
    const float ha = __builtin_bit_cast(float, hi << 16), hb = __builtin_bit_cast(float, hi & 0xffff0000u);
    return pk_bf16(a - ha, b - hb);
}
typedef unsigned u4 __attribute__((ext_vector_type(4)));
template <int MODE>   // 0: f16 3 terms, 1: bf16 3 terms, 2: f16 hi only
__global__ void split_k(const float* A, const float* B, float* D) {
    const int l = threadIdx.x, kg = l >> 4, i = l & 15;
    float a[8], b[8];
    for (int j = 0; j < 8; ++j) { a[j] = A[i * 32 + kg * 8 + j]; b[j] = B[(kg * 8 + j) * 16 + i]; }
    u4 ah, al, bh, bl;
    for (int q = 0; q < 4; ++q) {
        if (MODE == 1) {
            ah[q] = pk_bf16(a[2 * q], a[2 * q + 1]); al[q] = pk_bf16_lo(ah[q], a[2 * q], a[2 * q + 1]);
            bh[q] = pk_bf16(b[2 * q], b[2 * q + 1]); bl[q] = pk_bf16_lo(bh[q], b[2 * q], b[2 * q + 1]);
        } else {
            ah[q] = pk_f16(a[2 * q], a[2 * q + 1]); al[q] = pk_f16_lo(ah[q], a[2 * q], a[2 * q + 1]);
            bh[q] = pk_f16(b[2 * q], b[2 * q + 1]); bl[q] = pk_f16_lo(bh[q], b[2 * q], b[2 * q + 1]);
        }
    }
    f4 acc = {0, 0, 0, 0};
    if (MODE == 1) {
        typedef __bf16 b8 __attribute__((ext_vector_type(8)));
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, al), __builtin_bit_cast(b8, bh), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, ah), __builtin_bit_cast(b8, bl), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, ah), __builtin_bit_cast(b8, bh), acc, 0, 0, 0);
    } else {
        if (MODE == 0) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, al), __builtin_bit_cast(h8, bh), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, ah), __builtin_bit_cast(h8, bl), acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, ah), __builtin_bit_cast(h8, bh), acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[(kg * 4 + r) * 16 + i] = acc[r];
}

// ---- 3. issue costs
#define REP 256
template <int MODE>
__global__ __launch_bounds__(256) void rate_k(float* out, int iters, float fa, float fb) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
    unsigned u[8];
    for (int i = 0; i < 8; ++i) u[i] = threadIdx.x * 7 + i;
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    h8 a8, b8;
    for (int j = 0; j < 8; ++j) { a8[j] = (_Float16)(fa + j); b8[j] = (_Float16)(fb + j); }
    h4 a4 = {a8[0], a8[1], a8[2], a8[3]}, b4 = {b8[0], b8[1], b8[2], b8[3]};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (MODE == 0) { _Pragma("unroll") for (int i = 0; i < 8; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[i & 3], 0, 0, 0); }
            if (MODE == 1) { _Pragma("unroll") for (int i = 0; i < 8; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[i & 3], 0, 0, 0); }
            if (MODE == 2) { _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(x[i]), "v"(x[(i + 1) & 7])); }
            if (MODE == 3) { _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(x[i])); }
            if (MODE == 4) { _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(fa), "v"(fb)); }
            if (MODE == 5) { _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("v_max_i32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 3) & 7])); }
            if (MODE == 6) { _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(u[i]) : "v"(u[(i + 1) & 7])); }
            if (MODE == 7) { _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(fa)); }
            if (MODE == 8) { _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(fa)); }
            if (MODE == 9) { _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(fa), "v"(fb)); }
            if (MODE == 10) { _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(x[i]), "v"(x[(i + 1) & 7])); }
            if (MODE == 11) { _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x[i]) : "v"(fa)); }
            if (MODE == 12) { _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("v_add_f32 %0, |%0|, %1" : "+v"(x[i]) : "v"(fa)); }
            if (MODE == 13) { _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(fa), "v"(fb)); }
            if (MODE == 14) {   // mfma with 4 VALU fillers per MFMA: do both pipes overlap inside ONE wave?
                _Pragma("unroll") for (int i = 0; i < 8; ++i) {
                    acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[i & 3], 0, 0, 0);
                    asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "v"(fa), "v"(fb));
                }
            }
            if (MODE == 15) {   // 8 VALU fillers per MFMA
                _Pragma("unroll") for (int i = 0; i < 8; ++i) {
                    acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[i & 3], 0, 0, 0);
                    asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                 "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(fa), "v"(fb));
                }
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + (float)u[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}
template <int MODE>
void rate(const char* name, int wps) {
    float* d; hipMalloc(&d, 4);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    rate_k<MODE><<<256 * wps, 256>>>(d, 10, 0.999f, 0.001f); hipDeviceSynchronize();
    hipEventRecord(e0); rate_k<MODE><<<256 * wps, 256>>>(d, iters, 0.999f, 0.001f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s waves/SIMD=%d  %.2f cyc per wave-instruction per SIMD @2.4GHz\n", name, wps, ms * 1e6 / ((double)iters * REP * wps) * 2.4);
    hipFree(d);
}

// ---- 4. LDS-DMA: every lane hands 16 bytes of global memory to LDS at M0-base + 16*lane
__global__ void ldsdma_k(const float* g, float* out, int nfloats) {
    __shared__ __attribute__((aligned(16))) float lds[2048];
    const int l = threadIdx.x;
    for (int i = l; i < 2048; i += 64) lds[i] = -1.f;
    __syncthreads();
    // two wave-instructions: floats [0,256) and [256,512); the second only on the lanes that have data
    __builtin_amdgcn_global_load_lds(g + 4 * l, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
    if (256 + 4 * l < nfloats)
        __builtin_amdgcn_global_load_lds(g + 256 + 4 * l, (__attribute__((address_space(3))) void*)(lds + 256), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = l; i < 2048; i += 64) out[i] = lds[i];
}

int main() {
    // 1. layouts
    for (int K : {32, 16}) {
        std::vector<float> A(16 * K), B(K * 16), D(256), R(256, 0.f);
        for (auto& v : A) v = (float)(rand() % 7 - 3);
        for (auto& v : B) v = (float)(rand() % 5 - 2);
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < K; ++k) R[i * 16 + j] += A[i * K + k] * B[k * 16 + j];
        float *dA, *dB, *dD; hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 1024);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        if (K == 32) layout_k<32><<<1, 64>>>(dA, dB, dD); else layout_k<16><<<1, 64>>>(dA, dB, dD);
        hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
        int bad = 0; for (int i = 0; i < 256; ++i) bad += D[i] != R[i];
        printf("layout 16x16x%d f16: %s (%d mismatches)\n", K, bad ? "WRONG" : "as assumed", bad);
    }
    // 2. split accuracy
    {
        std::vector<float> A(512), B(512), D(256);
        std::vector<double> R(256, 0.0), S(256, 0.0);
        for (auto& v : A) v = (float)((rand() / (double)RAND_MAX) * 2 - 1) * 3.f;
        for (auto& v : B) v = (float)((rand() / (double)RAND_MAX) * 2 - 1);
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 32; ++k) {
            R[i * 16 + j] += (double)A[i * 32 + k] * B[k * 16 + j]; S[i * 16 + j] += fabs((double)A[i * 32 + k] * B[k * 16 + j]); }
        float *dA, *dB, *dD; hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 1024);
        hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
        const char* names[3] = {"f16 hi*hi+hi*lo+lo*hi", "bf16 hi*hi+hi*lo+lo*hi", "f16 hi*hi only"};
        for (int m = 0; m < 3; ++m) {
            if (m == 0) split_k<0><<<1, 64>>>(dA, dB, dD); else if (m == 1) split_k<1><<<1, 64>>>(dA, dB, dD); else split_k<2><<<1, 64>>>(dA, dB, dD);
            hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
            double e = 0; for (int i = 0; i < 256; ++i) e = fmax(e, fabs(D[i] - R[i]) / S[i]);
            printf("split %-26s max |err| / sum|a.b| = %.3e\n", names[m], e);
        }
    }
    // 4. LDS-DMA
    {
        std::vector<float> G(512), O(2048);
        for (int i = 0; i < 512; ++i) G[i] = (float)i;
        float *dG, *dO; hipMalloc(&dG, 2048); hipMalloc(&dO, 8192);
        hipMemcpy(dG, G.data(), 2048, hipMemcpyHostToDevice);
        ldsdma_k<<<1, 64>>>(dG, dO, 400);
        hipMemcpy(O.data(), dO, 8192, hipMemcpyDeviceToHost);
        int bad = 0; for (int i = 0; i < 2048; ++i) bad += O[i] != (i < 400 ? (float)i : -1.f);
        printf("global_load_lds_dwordx4: %s (%d mismatches; lds[255..257] = %g %g %g, lds[399..401] = %g %g %g)\n", bad ? "NOT linear" : "linear copy, masked tail ok",
               bad, O[255], O[256], O[257], O[399], O[400], O[401]);
    }
    // 3. rates
    for (int w : {1, 2, 4}) {
        rate<0>("v_mfma_f32_16x16x32_f16", w); rate<1>("v_mfma_f32_16x16x16_f16", w);
        rate<14>("mfma16x16x32 + 4 v_fma (per group)", w); rate<15>("mfma16x16x32 + 8 v_fma (per group)", w);
    }
    for (int w : {4}) {
        rate<2>("v_cvt_pk_f16_f32", w); rate<10>("v_cvt_pk_bf16_f32", w); rate<3>("v_fma_mixlo_f16", w); rate<4>("v_max3_f32", w); rate<13>("v_min3_f32", w);
        rate<5>("v_max_i32", w); rate<6>("v_mov_b32_dpp row_shr", w); rate<7>("v_max_f32", w); rate<11>("v_min_f32", w); rate<8>("v_mul_f32", w);
        rate<9>("v_med3_f32", w); rate<12>("v_add_f32 |x|", w);
    }
    return 0;
}
