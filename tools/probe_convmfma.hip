#include "../gnn_rul_benchmarking_amd/csrc/stgcn_device.hpp"
#include <cstdio>
#include <cstdlib>
using namespace rulgnn;
template<int D, bool T>
__global__ void k(const float* hin, const float* w, float* o_ref, float* o_mfma) {
  __shared__ __attribute__((aligned(16))) float tab[16*CONV_ROW];
  int lane = threadIdx.x, t = lane & 15;
  stage_conv_table(tab, w, T, threadIdx.x, 64);
  __syncthreads();
  float h[F], a[F], b[F];
  for (int c = 0; c < F; ++c) h[c] = hin[c*64 + lane];
  if (!T) causal_conv<16, D>(h, w, t, a);
  else {
    float dzs[F];
    for (int c = 0; c < F; ++c) dzs[c] = Row<16>::template shl<D>(h[c], t);
    for (int ci = 0; ci < F; ++ci) { float acc = 0.f; for (int co = 0; co < F; ++co) { acc = fmaf(w[(co*F+ci)*2+1], h[co], acc); acc = fmaf(w[(co*F+ci)*2+0], dzs[co], acc);} a[ci] = acc; }
  }
  causal_conv_mfma<D, T>(h, tab + t*CONV_ROW, b);
  for (int c = 0; c < F; ++c) { o_ref[c*64+lane] = a[c]; o_mfma[c*64+lane] = b[c]; }
}
template<int D, bool T> void run() {
  float hh[640], hw[200], r[640], m[640];
  for (auto& v : hh) v = rand() / (float)RAND_MAX - 0.5f;
  for (auto& v : hw) v = rand() / (float)RAND_MAX - 0.5f;
  float *dh, *dw, *dr, *dm; hipMalloc(&dh, 2560); hipMalloc(&dw, 800); hipMalloc(&dr, 2560); hipMalloc(&dm, 2560);
  hipMemcpy(dh, hh, 2560, hipMemcpyHostToDevice); hipMemcpy(dw, hw, 800, hipMemcpyHostToDevice);
  k<D,T><<<1,64>>>(dh, dw, dr, dm); hipDeviceSynchronize();
  hipMemcpy(r, dr, 2560, hipMemcpyDeviceToHost); hipMemcpy(m, dm, 2560, hipMemcpyDeviceToHost);
  double e = 0; int bad = 0;
  for (int i = 0; i < 640; ++i) { double d = fabs(r[i]-m[i]); if (d > e) e = d; if (d > 1e-5 && bad < 6) { printf("  c=%d lane=%d ref=%f mfma=%f\n", i/64, i%64, r[i], m[i]); ++bad; } }
  printf("D=%d T=%d max abs diff %g\n", D, (int)T, e);
}
int main() { run<1,false>(); run<2,false>(); run<1,true>(); run<2,true>(); return 0; }
