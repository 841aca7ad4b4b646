#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define REP 256
template<int MODE> __global__ __launch_bounds__(256) void k(float* out, const float* __restrict__ w, int iters, float a) {
  f32x2 x[8]; for (int i=0;i<8;++i) x[i]=(f32x2){(float)threadIdx.x+i, 1.f};
  f32x2 s0 = *reinterpret_cast<const f32x2*>(w);
  f32x2 hv = {a, a};
  for (int it=0; it<iters; ++it) {
    #pragma unroll
    for (int r=0;r<REP/8;++r) {
      if (MODE==0) { _Pragma("unroll") for (int i=0;i<8;++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(x[i]) : "s"(s0), "v"(hv)); }
      if (MODE==1) { _Pragma("unroll") for (int i=0;i<8;++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(hv), "v"(hv)); }
    }
  }
  float s=0; for (int i=0;i<8;++i) s+=x[i].x+x[i].y;
  if (s == 12345.678f) out[0]=s;
}
template<int MODE> void run(const char* name, int wps) {
  float *d, *w; hipMalloc(&d, 4); hipMalloc(&w, 64); hipMemset(w, 0, 64);
  int iters = 4000;
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<256*wps,256>>>(d, w, 10, 0.999f); hipDeviceSynchronize();
  hipEventRecord(e0); k<MODE><<<256*wps,256>>>(d, w, iters, 0.999f); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms,e0,e1);
  printf("%-26s waves/SIMD=%d  %.2f cyc/inst @2.3GHz\n", name, wps, ms*1e6/((double)iters*REP*wps)*2.3);
}
int main() { for (int w : {1,4}) { run<0>("v_pk_fma sgpr-pair src", w); run<1>("v_pk_fma vgpr", w); } return 0; }
