// One-shot all-reduce of a few doubles over peer-mapped mailboxes: the synchronised-BatchNorm collectives of a data-parallel step WITHOUT
// the host (round 6; SURVEY section 8e: "on the fully-connected 8-GPU xGMI node use one-shot (all-to-all direct) reduce for <= 256 KB").
//
// The BatchNorm reductions of a synchronised step are 4 L (ST_GCN: 8) all-reduces of 20 doubles between 25-50-us phase kernels.  Through
// the caller's collective library each of them is a host callback (ctypes -> Python -> torch.distributed -> RCCL launch: ~30 us of host
// time + ~20 us of RCCL latency).  Here a collective is ONE single-workgroup launch on the compute stream and no host work beyond it:
//
//   * every rank owns a MAILBOX in fine-grained device memory, exported by hipIpcGetMemHandle and mapped by every peer
//     (hipIpcOpenMemHandle): [2 parities][world slots] x { flag, data[PEER_MAX_COUNT] };
//   * collective number q: rank r PUSHES its contribution into slot r (parity q & 1) of EVERY rank's mailbox (plain remote stores over
//     xGMI, system scope), then -- behind a system-scope release fence -- the flag q; it then spins (bounded) on the world flags of its
//     OWN mailbox (local memory: no remote polling traffic) and sums the slots in rank order: the same order on every rank, so the
//     result is bit-identical across ranks and equal to a fixed-order sum (RCCL's ring / tree orders are not);
//   * two parities suffice: a rank can only enter collective q + 2 after it completed q + 1, which needed every peer's q + 1 push, which
//     a peer issues after it finished READING collective q.
//
// The entry rulgnn_peer_allreduce_f64 has the signature of rulgnn_allreduce_f64_fn: its address and the communicator are passed as the
// (callback, user) pair of the *_syncbn_* entries -- no Python frame between two phases of a step.  A spin is bounded by wall-clock
// (the communicator's timeout, in ticks of the 100 MHz counter; 20 s unless rulgnn_peer_comm_set_timeout_ms says otherwise -- long
// enough for a peer whose host thread was descheduled or is still loading code objects, short enough to end a job whose peer died): a
// missing peer ends as a sticky error word in the mailbox (reported by rulgnn_peer_comm_status) and NaN in the buffer, never as a hung
// GPU.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <new>

#include "../../include/rulgnn.h"

namespace rulgnn {
namespace {

constexpr int PEER_MAX_WORLD = 8;
constexpr int PEER_MAX_COUNT = RULGNN_PEER_MAX_COUNT;
constexpr unsigned long long PEER_TICKS_PER_MS = 100000ull;           // the 100 MHz wall clock
constexpr unsigned long long PEER_TIMEOUT_TICKS = 20000ull * PEER_TICKS_PER_MS;      // default: 20 s

struct PeerSlot {
    unsigned long long flag;                 // number of the last collective whose data is complete in this slot
    unsigned long long pad[7];
    double data[PEER_MAX_COUNT];
};
struct PeerMailbox {
    unsigned long long error;                // sticky: a collective of this rank timed out waiting for a peer
    unsigned long long pad[7];
    PeerSlot slot[2][PEER_MAX_WORLD];
};
struct PeerPtrs {
    PeerMailbox* box[PEER_MAX_WORLD];
};
struct PeerComm {
    int rank, world;
    PeerPtrs peers;                          // peers.box[rank] = the own mailbox
    unsigned long long seq;                  // collectives issued so far
    unsigned long long timeout_ticks;
};

__global__ __launch_bounds__(PEER_MAX_COUNT) void peer_allreduce_kernel(double* __restrict__ buf, int n, PeerPtrs p, int rank, int world,
                                                                         unsigned long long seq, unsigned long long timeout_ticks) {
    const int t = threadIdx.x, par = (int)(seq & 1ull);
    if (t < n) {
        const double v = buf[t];
        for (int q = 0; q < world; ++q) __hip_atomic_store(&p.box[q]->slot[par][rank].data[t], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();                              // (the data is visible to the peers before the flag)
    __syncthreads();
    if (t < world) __hip_atomic_store(&p.box[t]->slot[par][rank].flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    __shared__ int failed;
    if (t == 0) failed = 0;
    __syncthreads();
    if (t < world) {
        const unsigned long long t0 = wall_clock64();
        PeerMailbox* mine = p.box[rank];
        while (__hip_atomic_load(&mine->slot[par][t].flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > timeout_ticks) {
                failed = 1;
                __hip_atomic_store(&mine->error, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __syncthreads();
    if (t < n) {
        double s = 0.0;
        for (int q = 0; q < world; ++q) s += __hip_atomic_load(&p.box[rank]->slot[par][q].data[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        buf[t] = failed ? __builtin_nan("") : s;
    }
}

}  // namespace
}  // namespace rulgnn

using namespace rulgnn;

extern "C" {

size_t rulgnn_peer_mailbox_bytes(void) { return sizeof(PeerMailbox); }
size_t rulgnn_peer_handle_bytes(void) { return sizeof(hipIpcMemHandle_t); }

int rulgnn_peer_mailbox_alloc(void** mailbox, void* handle_out) {
    if (!mailbox || !handle_out) return RULGNN_EINVAL;
    void* p = nullptr;
    // fine-grained: stores of a peer and of this device's own kernels are visible inside a running kernel, not only at its end
    if (hipExtMallocWithFlags(&p, sizeof(PeerMailbox), hipDeviceMallocFinegrained) != hipSuccess) return RULGNN_EHIP;
    if (hipMemset(p, 0, sizeof(PeerMailbox)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(p); return RULGNN_EHIP; }
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, p) != hipSuccess) { (void)hipFree(p); return RULGNN_EHIP; }
    *reinterpret_cast<hipIpcMemHandle_t*>(handle_out) = h;
    *mailbox = p;
    return RULGNN_OK;
}
int rulgnn_peer_mailbox_open(const void* handle, void** mailbox) {
    if (!handle || !mailbox) return RULGNN_EINVAL;
    void* p = nullptr;
    if (hipIpcOpenMemHandle(&p, *reinterpret_cast<const hipIpcMemHandle_t*>(handle), hipIpcMemLazyEnablePeerAccess) != hipSuccess) return RULGNN_EHIP;
    *mailbox = p;
    return RULGNN_OK;
}
int rulgnn_peer_mailbox_close(void* mailbox) { return mailbox && hipIpcCloseMemHandle(mailbox) == hipSuccess ? RULGNN_OK : RULGNN_EHIP; }
int rulgnn_peer_mailbox_free(void* mailbox) { return mailbox && hipFree(mailbox) == hipSuccess ? RULGNN_OK : RULGNN_EHIP; }

void* rulgnn_peer_comm_create(int32_t rank, int32_t world, void* const* mailboxes) {
    if (rank < 0 || world < 1 || world > PEER_MAX_WORLD || rank >= world || !mailboxes) return nullptr;
    PeerComm* c = new (std::nothrow) PeerComm();
    if (!c) return nullptr;
    c->rank = rank; c->world = world; c->seq = 0; c->timeout_ticks = PEER_TIMEOUT_TICKS;
    for (int q = 0; q < PEER_MAX_WORLD; ++q) c->peers.box[q] = q < world ? static_cast<PeerMailbox*>(mailboxes[q]) : nullptr;
    for (int q = 0; q < world; ++q)
        if (!c->peers.box[q]) { delete c; return nullptr; }
    return c;
}
void rulgnn_peer_comm_destroy(void* comm) { delete static_cast<PeerComm*>(comm); }
int rulgnn_peer_comm_set_timeout_ms(void* comm, int64_t ms) {
    if (!comm || ms < 1 || ms > 600000) return RULGNN_EINVAL;
    static_cast<PeerComm*>(comm)->timeout_ticks = (unsigned long long)ms * PEER_TICKS_PER_MS;
    return RULGNN_OK;
}

// rulgnn_allreduce_f64_fn: device_buf[0..count) summed over the ranks in place, in stream order
int rulgnn_peer_allreduce_f64(void* comm, double* device_buf, int32_t count, void* stream) {
    PeerComm* c = static_cast<PeerComm*>(comm);
    if (!c || !device_buf || count < 0 || count > PEER_MAX_COUNT) return RULGNN_EINVAL;
    if (count == 0) return RULGNN_OK;
    ++c->seq;
    (void)hipGetLastError();
    hipLaunchKernelGGL(peer_allreduce_kernel, dim3(1), dim3(PEER_MAX_COUNT), 0, static_cast<hipStream_t>(stream), device_buf, (int)count, c->peers,
                       c->rank, c->world, c->seq, c->timeout_ticks);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}
int64_t rulgnn_peer_comm_collectives(void* comm) { return comm ? (int64_t)static_cast<PeerComm*>(comm)->seq : -1; }
// 0: every collective completed so far found its peers; otherwise the number of the (last) collective that timed out.  Synchronises
// with the device (a host read of the mailbox's error word).
int64_t rulgnn_peer_comm_status(void* comm) {
    PeerComm* c = static_cast<PeerComm*>(comm);
    if (!c) return -1;
    unsigned long long e = 0;
    if (hipMemcpy(&e, &c->peers.box[c->rank]->error, sizeof(e), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int64_t)e;
}

}  // extern "C"
