// Streaming-pattern microbenchmark (development aid): the F_{2l+1}-like pattern "read two saved tensors, write two" with
//   A: dword per lane, packed rows of 56 lanes (what the phase kernels do), B: dword per lane, 64-lane rows,
//   C: lane-major dwordx4 + dwordx4 + dwordx2, D: plain float4 copy of the same byte count.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_stream.hip -o /tmp/ubench_stream && /tmp/ubench_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int F = 10;
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ in0, const float* __restrict__ in1, float* __restrict__ o0,
                                         float* __restrict__ o1, int ntiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int srow = lane >> 4, t = lane & 15;
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        float a[F], b[F];
        if (MODE == 0) {
            const size_t base = (size_t)tile * F * 56 + srow * 14 + t;
            for (int c = 0; c < F; ++c) a[c] = b[c] = 0.f;
            if (t < 14) {
#pragma unroll
                for (int c = 0; c < F; ++c) { a[c] = in0[base + c * 56]; b[c] = in1[base + c * 56]; }
            }
#pragma unroll
            for (int c = 0; c < F; ++c) { a[c] = fmaf(a[c], 1.0001f, b[c]); b[c] = a[c] * 0.5f; }
            if (t < 14) {
#pragma unroll
                for (int c = 0; c < F; ++c) { o0[base + c * 56] = a[c]; o1[base + c * 56] = b[c]; }
            }
        } else if (MODE == 1) {
            const size_t base = (size_t)tile * F * 64 + lane;
#pragma unroll
            for (int c = 0; c < F; ++c) { a[c] = in0[base + c * 64]; b[c] = in1[base + c * 64]; }
#pragma unroll
            for (int c = 0; c < F; ++c) { a[c] = fmaf(a[c], 1.0001f, b[c]); b[c] = a[c] * 0.5f; }
#pragma unroll
            for (int c = 0; c < F; ++c) { o0[base + c * 64] = a[c]; o1[base + c * 64] = b[c]; }
        } else if (MODE == 2) {
            // [tile]{[64][4] | [64][4] | [64][2]}
            const size_t base = (size_t)tile * F * 64;
            const float4* p0 = reinterpret_cast<const float4*>(in0 + base);
            const float4* p1 = reinterpret_cast<const float4*>(in1 + base);
            float4 x0 = p0[lane], x1 = p0[64 + lane], y0 = p1[lane], y1 = p1[64 + lane];
            float2 x2 = reinterpret_cast<const float2*>(in0 + base + 512)[lane], y2 = reinterpret_cast<const float2*>(in1 + base + 512)[lane];
            a[0] = x0.x; a[1] = x0.y; a[2] = x0.z; a[3] = x0.w; a[4] = x1.x; a[5] = x1.y; a[6] = x1.z; a[7] = x1.w; a[8] = x2.x; a[9] = x2.y;
            b[0] = y0.x; b[1] = y0.y; b[2] = y0.z; b[3] = y0.w; b[4] = y1.x; b[5] = y1.y; b[6] = y1.z; b[7] = y1.w; b[8] = y2.x; b[9] = y2.y;
#pragma unroll
            for (int c = 0; c < F; ++c) { a[c] = fmaf(a[c], 1.0001f, b[c]); b[c] = a[c] * 0.5f; }
            float4* q0 = reinterpret_cast<float4*>(o0 + base);
            float4* q1 = reinterpret_cast<float4*>(o1 + base);
            q0[lane] = make_float4(a[0], a[1], a[2], a[3]); q0[64 + lane] = make_float4(a[4], a[5], a[6], a[7]);
            q1[lane] = make_float4(b[0], b[1], b[2], b[3]); q1[64 + lane] = make_float4(b[4], b[5], b[6], b[7]);
            reinterpret_cast<float2*>(o0 + base + 512)[lane] = make_float2(a[8], a[9]);
            reinterpret_cast<float2*>(o1 + base + 512)[lane] = make_float2(b[8], b[9]);
        } else {
            const size_t base = (size_t)tile * F * 64;
            for (int i = lane; i < F * 16; i += 64) {
                float4 x = reinterpret_cast<const float4*>(in0 + base)[i], y = reinterpret_cast<const float4*>(in1 + base)[i];
                reinterpret_cast<float4*>(o0 + base)[i] = x;
                reinterpret_cast<float4*>(o1 + base)[i] = y;
            }
        }
    }
}
template <int MODE>
void run(const char* name, int ntiles, int blocks, float bytes_per_tile) {
    float *a, *b, *c, *d;
    const size_t n = (size_t)ntiles * F * 64;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&c, n * 4); hipMalloc(&d, n * 4);
    hipMemset(a, 0, n * 4); hipMemset(b, 0, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, a, b, c, d, ntiles);
    hipEventRecord(e0);
    const int it = 20;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, a, b, c, d, ntiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
    printf("%-28s tiles %7d blocks %5d: %8.1f us  %6.2f TB/s\n", name, ntiles, blocks, ms * 1e3, bytes_per_tile * ntiles / (ms * 1e-3) / 1e12);
    hipFree(a); hipFree(b); hipFree(c); hipFree(d);
}
int main() {
    for (int ntiles : {16384, 262144})
        for (int blocks : {1024, 1280, 2048, 4096}) {
            run<0>("A dword packed 56", ntiles, blocks, 4.f * F * 56 * 4);
            run<1>("B dword rows of 64", ntiles, blocks, 4.f * F * 64 * 4);
            run<2>("C lane-major x4 x4 x2", ntiles, blocks, 4.f * F * 64 * 4);
            run<3>("D float4 copy", ntiles, blocks, 4.f * F * 64 * 4);
        }
    return 0;
}
