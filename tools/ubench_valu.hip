// microbenchmark: per-wave64 issue cost of v_fma_f32 / v_pk_fma_f32 / v_fmac_f32_dpp / mfma f32 on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_ __attribute__((ext_vector_type(2)));
typedef float float4_ __attribute__((ext_vector_type(4)));
#define REP 512
template<int MODE> __global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float x0=threadIdx.x, x1=x0+1, x2=x0+2, x3=x0+3, x4=x0+4, x5=x0+5, x6=x0+6, x7=x0+7;
  float2_ p0={x0,x1}, p1={x2,x3}, p2={x4,x5}, p3={x6,x7}, p4={x1,x2}, p5={x3,x4}, p6={x5,x6}, p7={x7,x0};
  float2_ aa={a,a}, bb={b,b};
  float4_ acc0={0,0,0,0}, acc1={0,0,0,0}, acc2={0,0,0,0}, acc3={0,0,0,0};
  for (int i=0;i<iters;++i) {
    #pragma unroll
    for (int r=0;r<REP/8;++r) {
      if (MODE==0) { x0=fmaf(x0,a,b); x1=fmaf(x1,a,b); x2=fmaf(x2,a,b); x3=fmaf(x3,a,b); x4=fmaf(x4,a,b); x5=fmaf(x5,a,b); x6=fmaf(x6,a,b); x7=fmaf(x7,a,b); }
      if (MODE==1) { p0=__builtin_elementwise_fma(p0,aa,bb); p1=__builtin_elementwise_fma(p1,aa,bb); p2=__builtin_elementwise_fma(p2,aa,bb); p3=__builtin_elementwise_fma(p3,aa,bb);
                     p4=__builtin_elementwise_fma(p4,aa,bb); p5=__builtin_elementwise_fma(p5,aa,bb); p6=__builtin_elementwise_fma(p6,aa,bb); p7=__builtin_elementwise_fma(p7,aa,bb); }
      if (MODE==2) { asm volatile("v_fmac_f32_dpp %0, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %2, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %4, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %5, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %6, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %7, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                       : "+v"(x0),"+v"(x1),"+v"(x2),"+v"(x3),"+v"(x4),"+v"(x5),"+v"(x6),"+v"(x7) : "v"(a), "v"(b)); }
      if (MODE==3) { acc0=__builtin_amdgcn_mfma_f32_16x16x4f32(x0,x1,acc0,0,0,0); acc1=__builtin_amdgcn_mfma_f32_16x16x4f32(x2,x3,acc1,0,0,0); acc2=__builtin_amdgcn_mfma_f32_16x16x4f32(x4,x5,acc2,0,0,0); acc3=__builtin_amdgcn_mfma_f32_16x16x4f32(x6,x7,acc3,0,0,0);
                     acc0=__builtin_amdgcn_mfma_f32_16x16x4f32(x1,x0,acc0,0,0,0); acc1=__builtin_amdgcn_mfma_f32_16x16x4f32(x3,x2,acc1,0,0,0); acc2=__builtin_amdgcn_mfma_f32_16x16x4f32(x5,x4,acc2,0,0,0); acc3=__builtin_amdgcn_mfma_f32_16x16x4f32(x7,x6,acc3,0,0,0); }
      if (MODE==4) { x0=fmaxf(x0,a); x1=fmaxf(x1,a); x2=fmaxf(x2,a); x3=fmaxf(x3,a); x4=fmaxf(x4,a); x5=fmaxf(x5,a); x6=fmaxf(x6,a); x7=fmaxf(x7,a); 
                     x0=x0+b; x1=x1+b; x2=x2+b; x3=x3+b; x4=x4+b; x5=x5+b; x6=x6+b; x7=x7+b; }
    }
  }
  float s = x0+x1+x2+x3+x4+x5+x6+x7 + p0.x+p0.y+p1.x+p1.y+p2.x+p2.y+p3.x+p3.y+p4.x+p4.y+p5.x+p5.y+p6.x+p6.y+p7.x+p7.y + acc0.x+acc1.y+acc2.z+acc3.w;
  if (s == 12345.678f) out[0] = s;
}
template<int MODE> void run(const char* name, int waves_per_simd, double ops_per_inst) {
  float* d; hipMalloc(&d, 4);
  int iters = 2000; int blocks = 256 * waves_per_simd;   // 256 threads = 4 waves = 1 per SIMD per block
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks,256>>>(d, 10, 0.999f, 0.001f); hipDeviceSynchronize();
  hipEventRecord(e0); k<MODE><<<blocks,256>>>(d, iters, 0.999f, 0.001f); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms,e0,e1);
  double insts_per_wave = (double)iters*REP* (MODE==4?2:1);
  double ns_per_inst_per_simd = ms*1e6 / (insts_per_wave*waves_per_simd);
  printf("%-14s waves/SIMD=%d  %.3f ns per wave-inst per SIMD  (%.2f cyc @2.4GHz)  -> %.1f T(ops)/s\n", name, waves_per_simd, ns_per_inst_per_simd, ns_per_inst_per_simd*2.4,
         ops_per_inst*64*insts_per_wave*waves_per_simd*1024/ (ms*1e-3) /1e12);
  hipFree(d);
}
int main() {
  for (int w : {1,2,4}) {
    run<0>("v_fma_f32", w, 2); run<1>("v_pk_fma_f32", w, 4); run<2>("v_fmac_dpp", w, 2); run<3>("mfma16x16x4f32", w, 2*16*16*4/64.0); run<4>("v_max+v_add", w, 1);
  }
  return 0;
}
