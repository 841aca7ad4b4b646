#include "../gnn_rul_benchmarking_amd/csrc/stgcn_device.hpp"
#include <cstdio>
using namespace rulgnn;
__global__ void k(float* o) {
  int l = threadIdx.x; int g = l >> 4;
  // register b, lane row g holds value 10*b + g
  float v0 = 0*10 + g, v1 = 1*10 + g, v2 = 2*10 + g, v3 = 3*10 + g;
  transpose_rows4(v0, v1, v2, v3);
  o[0*64+l] = v0; o[1*64+l] = v1; o[2*64+l] = v2; o[3*64+l] = v3;
}
int main() { float* d; hipMalloc(&d, 1024); k<<<1,64>>>(d); hipDeviceSynchronize(); float h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  for (int r = 0; r < 4; ++r) { printf("out reg %d: rows:", r); for (int g = 0; g < 4; ++g) printf(" %g", h[r*64 + g*16]); printf("   (expect %d %d %d %d)\n", 0*10+r, 10+r, 20+r, 30+r); } return 0; }
