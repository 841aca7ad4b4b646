// probe: D layout of v_mfma_f32_16x16x1_4b_f32 and semantics of v_permlane16_swap / v_permlane32_swap on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(float* out, unsigned* sw) {
  int l = threadIdx.x;
  f32x16 z = {0};
  f32x16 d1 = __builtin_amdgcn_mfma_f32_16x16x1f32((float)((l & 15) + 1), 1.0f, z, 0, 0, 0);
  f32x16 d2 = __builtin_amdgcn_mfma_f32_16x16x1f32(1.0f, (float)((l & 15) + 1), z, 0, 0, 0);
  f32x16 d3 = __builtin_amdgcn_mfma_f32_16x16x1f32(1.0f, (float)((l >> 4) + 1), z, 0, 0, 0);
  f32x16 d4 = __builtin_amdgcn_mfma_f32_16x16x1f32((float)((l >> 4) + 1), 1.0f, z, 0, 0, 0);
  for (int r = 0; r < 16; ++r) { out[(0*64 + l)*16 + r] = d1[r]; out[(1*64 + l)*16 + r] = d2[r]; out[(2*64 + l)*16 + r] = d3[r]; out[(3*64+l)*16 + r] = d4[r]; }
  unsigned a = 1000 + l, b = 2000 + l;
  u32x2 s16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  u32x2 s32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  sw[0*64 + l] = s16[0]; sw[1*64 + l] = s16[1]; sw[2*64 + l] = s32[0]; sw[3*64 + l] = s32[1];
}
int main() {
  float* d; unsigned* s; hipMalloc(&d, 4*64*16*4); hipMalloc(&s, 4*64*4);
  k<<<1,64>>>(d, s); hipDeviceSynchronize();
  static float h[4*64*16]; static unsigned hs[4*64];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hs, s, sizeof(hs), hipMemcpyDeviceToHost);
  printf("lane: for reg r: (row i from A, col j from B, Bblock, Ablock)\n");
  for (int l : {0, 1, 15, 16, 17, 32, 48, 63}) {
    printf("lane %2d:", l);
    for (int r = 0; r < 16; ++r) printf(" r%d(i%d,j%d,b%d,a%d)", r, (int)h[(0*64+l)*16+r]-1, (int)h[(1*64+l)*16+r]-1, (int)h[(2*64+l)*16+r]-1, (int)h[(3*64+l)*16+r]-1);
    printf("\n");
  }
  printf("permlane16_swap(a=1000+l, b=2000+l): lane -> (ret0, ret1)\n");
  for (int l = 0; l < 64; l += 8) printf("  l%2d: %u %u |", l, hs[l], hs[64+l]);
  printf("\npermlane32_swap: \n");
  for (int l = 0; l < 64; l += 8) printf("  l%2d: %u %u |", l, hs[128+l], hs[192+l]);
  printf("\n");
  return 0;
}
