// Hand-counted asynchronous global traffic for sequential (recurrent) kernels on gfx9 / CDNA.
//
// A recurrence step should wait for nothing but its own arithmetic, yet (a) __syncthreads() is `s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier`
// -- with tape stores and prefetched rows in flight every step waits for HBM round trips -- and (b) the compiler cannot count vmcnt
// through a loop whose loads are used iterations later: it waits for 0 in front of every use.  So the loops that use these helpers
// issue EVERY vector-memory instruction through them (the compiler then has nothing to wait for) and name the count themselves:
// vmcnt decrements in issue order on gfx9, a step issues the same instructions in every wavefront that matters, and
// wait_vm<N>() in front of a read names exactly the instructions issued after the request it needs.
//   * request:  dma_dword / dma_dword2 -- global_load_lds_dword: one dword per lane straight into LDS (no destination register the
//               compiler could touch before the data arrived).  ALL 64 lanes must be enabled; lanes without work pass a valid address.
//               M0 carries the LDS base: nothing else in these kernels uses it (LDS instructions do not need M0 on gfx9).
//   * read:     wait_vm<N>(), then an ordinary LDS read of the slot by the wavefront that requested it.
//   * store:    store_async -- global_store_dword, counted like a request.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace rulgnn {

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void store_async(float* dst, float v) { asm volatile("global_store_dword %0, %1, off" ::"v"(dst), "v"(v) : "memory"); }
// one dword per lane to LDS byte address lds_wave + 4 lane
__device__ __forceinline__ void dma_dword(const float* src, unsigned lds_wave) {
    asm volatile("s_mov_b32 m0, %1\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dword %0, off"
                 :
                 : "v"(src), "s"(lds_wave)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// LDS byte address of a __shared__ object (the low half of its flat address), wave-uniform
__device__ __forceinline__ unsigned lds_address(const void* p) { return __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)p); }

}  // namespace rulgnn
