// microbenchmark 2: per-wave64 issue cost of the instruction kinds the ST_GCN kernels are made of (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 256
#define X8(stmt) stmt(0) stmt(1) stmt(2) stmt(3) stmt(4) stmt(5) stmt(6) stmt(7)
template<int MODE> __global__ __launch_bounds__(256) void k(float* out, const float* __restrict__ w, int iters, float a, float b) {
  float x[8]; for (int i=0;i<8;++i) x[i]=threadIdx.x+i;
  float s0=w[0], s1=w[1], s2=w[2], s3=w[3];   // uniform -> SGPR
  for (int it=0; it<iters; ++it) {
    #pragma unroll
    for (int r=0;r<REP/8;++r) {
      if (MODE==0) { _Pragma("unroll") for (int i=0;i<8;++i) x[i]=fmaf(x[i],a,b); }                 // v_fma vgpr
      if (MODE==1) { _Pragma("unroll") for (int i=0;i<8;++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "s"(s0), "v"(a)); }   // v_fmac with SGPR src
      if (MODE==2) { _Pragma("unroll") for (int i=0;i<8;++i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a)); }
      if (MODE==3) { _Pragma("unroll") for (int i=0;i<8;++i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a)); }
      if (MODE==4) { _Pragma("unroll") for (int i=0;i<8;++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a)); }
      if (MODE==5) { _Pragma("unroll") for (int i=0;i<8;++i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a)); }
      if (MODE==6) { _Pragma("unroll") for (int i=0;i<8;++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i])); }
      if (MODE==7) { _Pragma("unroll") for (int i=0;i<8;i+=2) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x[i]), "+v"(x[i+1])); }
      if (MODE==8) { _Pragma("unroll") for (int i=0;i<8;++i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x[i]) : "v"(a)); }
      if (MODE==9) { _Pragma("unroll") for (int i=0;i<8;++i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a) : "vcc"); }
      if (MODE==10) { _Pragma("unroll") for (int i=0;i<8;++i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(a)); }
      if (MODE==11) { _Pragma("unroll") for (int i=0;i<8;++i) asm volatile("v_fmac_f32 %0, %1, %2\n s_mov_b32 s20, s21" : "+v"(x[i]) : "s"(s0), "v"(a) : "s20"); }
    }
  }
  float s=0; for (int i=0;i<8;++i) s+=x[i];
  if (s == 12345.678f) out[0]=s + s1+s2+s3;
}
template<int MODE> void run(const char* name, int waves_per_simd, double per) {
  float *d, *w; hipMalloc(&d, 4); hipMalloc(&w, 64); hipMemset(w, 0, 64);
  int iters = 4000; int blocks = 256 * waves_per_simd;
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks,256>>>(d, w, 10, 0.999f, 0.001f); hipDeviceSynchronize();
  hipEventRecord(e0); k<MODE><<<blocks,256>>>(d, w, iters, 0.999f, 0.001f); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms,e0,e1);
  double insts = (double)iters*REP*per;
  printf("%-22s waves/SIMD=%d  %.2f cyc/inst @2.3GHz\n", name, waves_per_simd, ms*1e6/(insts*waves_per_simd)*2.3);
  hipFree(d); hipFree(w);
}
int main() {
  for (int w : {1,4}) {
    run<0>("v_fma vgpr", w, 1); run<1>("v_fmac sgpr src", w, 1); run<2>("v_cndmask vcc", w, 1); run<3>("v_max", w, 1); run<4>("v_add", w, 1);
    run<5>("v_mul_lo_u32", w, 1); run<6>("v_rcp", w, 1); run<7>("v_permlane16_swap", w, 0.5); run<8>("v_mov_dpp", w, 1); run<9>("v_cmp+v_cndmask", w, 2); run<10>("v_xor", w, 1); run<11>("v_fmac+s_mov", w, 1);
  }
  return 0;
}
