// microbenchmark 3 (gfx950): issue cost of the instructions a cheaper f16 split could be built from
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 256
template<int MODE> __global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float x[8]; unsigned u[8];
  for (int i=0;i<8;++i) { x[i]=threadIdx.x+i; u[i] = threadIdx.x * 7 + i; }
  for (int it=0; it<iters; ++it) {
    #pragma unroll
    for (int r=0;r<REP/8;++r) {
      #pragma unroll
      for (int i=0;i<8;++i) {
        if (MODE==0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        if (MODE==1) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(x[i]), "v"(a));
        if (MODE==2) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(u[i]) : "v"(u[(i+1)&7]), "v"(a));
        if (MODE==3) asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(u[i]) : "v"(u[(i+1)&7]), "v"(a));
        if (MODE==4) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(x[i]) : "v"(u[i]), "v"(a));
        if (MODE==5) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(u[i]) : "v"(x[i]));
        if (MODE==6) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(x[i]) : "v"(u[i]));
        if (MODE==7) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[i]) : "v"(u[(i+1)&7]), "v"(u[(i+2)&7]), "v"(u[(i+3)&7]));
        if (MODE==8) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(u[i]) : "v"(u[(i+1)&7]), "v"(u[(i+2)&7]), "v"(u[(i+3)&7]));
        if (MODE==9) asm volatile("v_pack_b32_f16 %0, %1, %2" : "=v"(u[i]) : "v"(u[(i+1)&7]), "v"(u[(i+2)&7]));
        if (MODE==10) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        if (MODE==11) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
        if (MODE==12) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[i]));
        if (MODE==13) asm volatile("v_mul_f32 %0, |%0|, %1" : "+v"(x[i]) : "v"(a));
        if (MODE==14) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&x[i & 6])) : "v"(*reinterpret_cast<double*>(&x[(i + 2) & 6])));
        if (MODE==15) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(x[i]), "v"(a));
        if (MODE==16) asm volatile("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(u[i]) : "v"(u[(i+1)&7]), "v"(u[(i+2)&7]));
        if (MODE==17) asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(u[i]) : "v"(u[(i+1)&7]), "v"(u[(i+2)&7]), "v"(u[(i+3)&7]));
        if (MODE==18) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        if (MODE==19) asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(u[i]) : "v"(u[(i+1)&7]), "v"(u[(i+2)&7]));
        if (MODE==20) asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(u[i]) : "v"(u[(i+1)&7]), "v"(u[(i+2)&7]), "v"(u[(i+3)&7]));
        if (MODE==21) asm volatile("v_pk_max_f16 %0, %1, %2" : "=v"(u[i]) : "v"(u[(i+1)&7]), "v"(u[(i+2)&7]));
        if (MODE==22) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
        if (MODE==23) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
        if (MODE==24) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i+1)&7]));
        if (MODE==25) asm volatile("v_add_f32 %0, |%0|, %0" : "+v"(x[i]));
        if (MODE==26) asm volatile("v_cvt_f16_f32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(u[i]) : "v"(x[i]));
        if (MODE==27) asm volatile("v_rsq_f32 %0, %0" : "+v"(x[i]));
        if (MODE==28) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        if (MODE==29) asm volatile("v_fmamk_f32 %0, %1, 0x3c23d70a, %0" : "+v"(x[i]) : "v"(a));
      }
    }
  }
  float s=0; for (int i=0;i<8;++i) s+=x[i] + (float)u[i];
  if (s == 12345.678f) out[0]=s;
}
template<int MODE> void run(const char* name, int waves_per_simd) {
  float *d; (void)hipMalloc(&d, 4);
  int iters = 2000; int blocks = 256 * waves_per_simd;
  hipEvent_t e0,e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE><<<blocks,256>>>(d, 10, 0.999f, 0.001f); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); k<MODE><<<blocks,256>>>(d, iters, 0.999f, 0.001f); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms,e0,e1);
  printf("%-26s waves/SIMD=%d  %.2f cyc/inst @2.3GHz\n", name, waves_per_simd, ms*1e6/((double)iters*REP*waves_per_simd)*2.3);
  (void)hipFree(d);
}
int main() {
  for (int w : {1,2,4}) {
    run<0>("v_fma_f32", w); run<1>("v_cvt_pk_f16_f32", w); run<2>("v_fma_mixlo_f16", w); run<3>("v_fma_mixhi_f16", w); run<4>("v_fma_mix_f32", w);
    run<5>("v_cvt_f16_f32", w); run<6>("v_cvt_f32_f16", w); run<7>("v_perm_b32", w); run<8>("v_and_or_b32", w); run<9>("v_pack_b32_f16", w);
    run<10>("v_med3_f32", w); run<11>("v_min_f32", w); run<12>("v_sqrt_f32", w); run<13>("v_mul_f32 |x|", w); run<14>("v_pk_add_f32", w);
    run<15>("v_cvt_pk_bf16_f32", w); run<16>("v_lshl_or_b32", w); run<17>("v_bfi_b32", w); run<18>("v_max3_f32", w); run<19>("v_pk_mul_f16", w);
    run<20>("v_pk_fma_f16", w); run<21>("v_pk_max_f16", w); run<22>("v_mul_f32", w); run<23>("v_sub_f32", w); run<24>("v_and_b32", w);
    run<25>("v_add_f32 |x|,x", w); run<26>("v_cvt_f16_f32_sdwa hi", w); run<27>("v_rsq_f32", w); run<28>("v_fmac_f32", w); run<29>("v_fmamk_f32", w);
  }
  return 0;
}
