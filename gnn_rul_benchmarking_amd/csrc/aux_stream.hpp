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

struct AuxFork {
    hipStream_t st, wst;
    int rc = RULGNN_OK;
    bool forked = false;        // the side stream carries work the main stream has not waited for yet
    AuxFork(hipStream_t stream, void* aux) : st(stream), wst(aux ? static_cast<hipStream_t>(aux) : stream) {}
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

}  // namespace rulgnn
