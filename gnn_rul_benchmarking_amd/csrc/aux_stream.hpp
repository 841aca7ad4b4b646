// Fork / join of an optional second stream of the caller (the `aux_stream` field of a family's argument struct, include/rulgnn.h).
// The backward passes of the small-batch families are chains of launches at their 5-19 us latency floor; the weight / bias gradient
// GEMMs among them feed nothing downstream in the same call.  With a second stream they leave the critical path:
//     AuxFork fk(stream, args->aux_stream);
//     ...            fk.fork();                 // the aux stream waits for everything enqueued on `stream` so far
//     ...            gemm(..., fk.side());      // = the aux stream, or `stream` itself when the caller gave none
//     ...            rc = fk.join();            // `stream` waits for the aux stream; in front of the call's last kernel
// Same launches, same results.  Events come from a small per-thread ring, created on first use and kept: an event may be re-recorded
// once the waits enqueued on it have been issued to their streams, which hipStreamWaitEvent does before it returns.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "../../include/rulgnn.h"

namespace rulgnn {

// One ring per (thread, device): an event belongs to the device that was current when it was created, and recording it on another
// device's stream fails -- a thread that drives training on several GPUs (torch.cuda.device(1): ...) gets a ring for each.
inline hipEvent_t aux_pooled_event() {
    constexpr int N = 32, MAX_DEV = 16;
    static thread_local hipEvent_t ring[MAX_DEV][N] = {};
    static thread_local int next[MAX_DEV] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
    hipEvent_t& e = ring[dev][next[dev]];
    next[dev] = (next[dev] + 1) % N;
    if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) e = nullptr;
    return e;
}

// Events for fork_after(): signalled by a kernel's OWN completion (hipExtLaunchKernel's stopEvent).  hipEventRecord puts a barrier packet of
// its own into the main stream's queue, and the next kernel of the chain waits for the command processor to retire it: ~6 us of bubble
// per fork on chains whose kernels take 5-30 us (FC_STGNN: eight forks per step).  A separate ring: these carry timestamps.
inline hipEvent_t aux_pooled_stop_event() {
    constexpr int N = 32, MAX_DEV = 16;
    static thread_local hipEvent_t ring[MAX_DEV][N] = {};
    static thread_local int next[MAX_DEV] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
    hipEvent_t& e = ring[dev][next[dev]];
    next[dev] = (next[dev] + 1) % N;
    if (!e && hipEventCreate(&e) != hipSuccess) e = nullptr;
    return e;
}

struct AuxFork {
    hipStream_t st, wst;
    int rc = RULGNN_OK;
    bool forked = false;        // the side stream carries work the main stream has not waited for yet
    bool capturing = false;     // a hipGraph capture is in progress on the main stream: plain event records only
    AuxFork(hipStream_t stream, void* aux) : st(stream), wst(aux ? static_cast<hipStream_t>(aux) : stream) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (active() && (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone)) capturing = true;
    }
    // The fork point is "behind THIS kernel": stop_event() before the launch, the launch through RULGNN_LAUNCH_EV with it, fork_after()
    // behind it.  Without a side stream, or inside a capture, stop_event() is null and the three reduce to a plain launch + fork().
    hipEvent_t stop_event() { return (active() && !capturing && rc == RULGNN_OK) ? aux_pooled_stop_event() : nullptr; }
    void fork_after(hipEvent_t ev) {
        if (!ev) { fork(); return; }
        if (rc != RULGNN_OK) return;
        if (hipStreamWaitEvent(wst, ev, 0) != hipSuccess) rc = RULGNN_EHIP;
        forked = true;
    }
    AuxFork(const AuxFork&) = delete;
    AuxFork& operator=(const AuxFork&) = delete;
    // An early return between fork() and join() (a failed launch, RULGNN_EHIP) must not leave the side stream un-joined: a hipGraph
    // capture in progress would be invalidated, and the next call could reuse scratch the side stream is still working on.
    ~AuxFork() {
        if (forked) (void)join();
    }
    bool active() const { return wst != st; }
    hipStream_t side() const { return wst; }
    void order(hipStream_t after, hipStream_t waiter) {
        if (!active()) return;
        hipEvent_t ev = aux_pooled_event();
        if (!ev || hipEventRecord(ev, after) != hipSuccess || hipStreamWaitEvent(waiter, ev, 0) != hipSuccess) rc = RULGNN_EHIP;
    }
    void fork() {
        if (rc != RULGNN_OK) return;
        order(st, wst);
        forked = active();
    }
    int join() {
        if (forked) order(wst, st);        // also after an error elsewhere: the join itself must still happen
        forked = false;
        return rc;
    }
};

// launch on `stream` with `ev` (may be null) signalled by the kernel's completion
#define RULGNN_LAUNCH_EV(ev, kernel, grid, block, lds, stream, ...)                                                     \
    do {                                                                                                                \
        if (ev) hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, nullptr, ev, 0, __VA_ARGS__);                   \
        else hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                         \
    } while (0)

}  // namespace rulgnn
