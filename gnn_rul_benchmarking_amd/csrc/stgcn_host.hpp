// Host-side launch geometry shared by the forward and training translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rulgnn.h"
#include "stgcn_device.hpp"

namespace rulgnn {

struct TileGeom {
    int RW;             // lanes per sample row: 16, 32 or 64
    int SPW;            // samples per wavefront
    int Ppad;           // LDS patch stride (floats): keeps per-lane patch reads bank-conflict free
    int vec4;           // tile copy may use 16-byte loads (given a 16-byte aligned x)
    int stage_floats;   // LDS staging floats per wavefront (multiple of 4)
    uint32_t magicP;    // ceil(2^32 / P) for fastdiv
    int64_t ntiles;     // wavefront tiles = ceil(batch / SPW)
};

inline int validate_shape(const rulgnn_stgcn_shape* s) {
    if (!s) return RULGNN_EINVAL;
    if (s->batch < 0 || s->num_patch < 2 || s->patch_size < 2 || s->num_layers < 1) return RULGNN_EINVAL;
    if (s->mpnn_k < 1) return RULGNN_EINVAL;
    // MPNN order (Model.py:74-90): every wiring of the reference leaves the constructor default 1; 2 and 3 run on the row-mapped fp32
    // kernels (num_patch <= 64), not on the matrix-core or tiled ones
    if (s->mpnn_k > MAX_MPNN_ORDER || (s->mpnn_k > 1 && s->num_patch > 64)) return RULGNN_EUNSUPPORTED;
    if (s->num_patch > 4096 || s->num_layers > 8 || s->patch_size > 4096) return RULGNN_EUNSUPPORTED;
    if (s->batch > (int64_t)400000000 / ((int64_t)F * s->num_patch)) return RULGNN_EUNSUPPORTED;  // 32-bit dropout counter
    return RULGNN_OK;
}

inline int tile_geometry(const rulgnn_stgcn_shape* s, TileGeom* g) {
    const int rc = validate_shape(s);
    if (rc != RULGNN_OK) return rc;
    const int N = s->num_patch, P = s->patch_size;
    if (N > 64) return RULGNN_EUNSUPPORTED;                 // fused row-mapped kernels; larger num_patch -> tiled path
    g->RW = N <= 16 ? 16 : (N <= 32 ? 32 : 64);
    g->SPW = 64 / g->RW;
    // even P: patches are read with ds_read_b64, conflict-free iff (stride/2) is odd.
    // odd P: ds_read_b32, conflict-free as stride is odd.
    g->Ppad = (P % 2 == 0 && (P / 2) % 2 == 0) ? P + 2 : P;
    g->vec4 = (g->Ppad == P) && (((int64_t)N * P) % 4 == 0);
    const int raw = g->SPW * N * g->Ppad;
    g->stage_floats = (raw + 3) & ~3;
    if (g->RW == 16 && g->stage_floats < PT_FLOATS) g->stage_floats = PT_FLOATS;   // the staging area doubles as the Pearson transpose tile
    if ((size_t)g->stage_floats * sizeof(float) > 36 * 1024) return RULGNN_EUNSUPPORTED;
    g->magicP = (uint32_t)((((uint64_t)1 << 32) + (uint64_t)P - 1) / (uint64_t)P);
    g->ntiles = (s->batch + g->SPW - 1) / g->SPW;
    return RULGNN_OK;
}

// Device-resident step state (include/rulgnn.h, RULGNN_STEP_STATE_BYTES): counters advanced on the device, plus what the
// compute kernels derive from them, so that a captured hipGraph replays with fresh dropout keys / Adam bias corrections.
struct StepState {
    uint64_t dropout_step;
    int64_t adam_step;
    float lr_over_bc1, inv_sqrt_bc2;
    uint32_t drop_key[8];
    uint32_t pad[2];
};
static_assert(sizeof(StepState) == RULGNN_STEP_STATE_BYTES, "step state layout");

// (seed, step, layer) -> 32-bit dropout key; bit-identical to oracle/stgcn_oracle.py::dropout_layer_key
__host__ __device__ inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ inline uint32_t dropout_layer_key(uint64_t seed, uint64_t step, int layer) {
    return (uint32_t)(splitmix64(seed ^ splitmix64(step * 64 + (uint64_t)layer + 1)) & 0xFFFFFFFFull);
}

// Persistent grid: exactly as many workgroups as are co-resident (occupancy API x CU count), never
// more than there are tiles; workgroups grid-stride over wavefront tiles.  A grid larger than
// the resident set would run a second, mostly empty round (measured: 1024 blocks on 768 slots
// cost 1.5x).
// `oversub`: launch that many times the resident set so that the hardware dispatcher balances the tail
// (measured on the fused forward at 1M samples: x4 -> -10 %; it hurts the short training phases).
template <typename K>
inline int persistent_grid(K kernel, int64_t ntiles, size_t lds_bytes, int oversub = 1) {
    int dev = 0, cus = 256, per_cu = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, BLOCK, lds_bytes) != hipSuccess || per_cu < 1)
        per_cu = 1;
    int64_t want = (ntiles + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    int64_t cap = (int64_t)cus * per_cu;
    if (oversub > 1 && want >= cap * oversub * 2) cap *= oversub;    // only when every wavefront still gets >= 8 tiles
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    return (int)want;
}

// Eval-forward path selection (include/rulgnn.h: RULGNN_EVAL_*): AUTO = matrix-core kernel when the shape qualifies.
enum { STGCN_EVAL_AUTO = RULGNN_EVAL_AUTO, STGCN_EVAL_EXACT = RULGNN_EVAL_EXACT, STGCN_EVAL_MX = RULGNN_EVAL_MX };
int stgcn_forward_eval(const rulgnn_stgcn_shape* s, const float* x, const float* prm, const float* bn, float* out,
                       hipStream_t stream, int path = STGCN_EVAL_AUTO);
// Matrix-core kernel (stgcn_forward_mx.hip); RULGNN_EUNSUPPORTED when the shape does not qualify (nothing launched).
// `taps`: optional debug buffer, MX_TAP_FLOATS floats (raw register dumps of the first tile, see the kernel).
int stgcn_forward_eval_mx(const rulgnn_stgcn_shape* s, const float* x, const float* prm, const float* bn, float* out,
                          hipStream_t stream, float* taps = nullptr);
int stgcn_forward_mx_tap_floats();
// The matrix-core eval forward for 16 <= num_patch <= 47 (stgcn_forward_mx.hip) and the scanning launch that follows it: every
// non-finite prediction is recomputed by the exact row-mapped routine (stgcn_forward.hip).
int stgcn_forward_eval_mxw(const rulgnn_stgcn_shape* s, const float* x, const float* prm, const float* bn, float* out, hipStream_t stream);
int stgcn_forward_fixup(const rulgnn_stgcn_shape* s, const float* x, const float* prm, const float* bn, float* out, hipStream_t stream);
// Training phase F_0 on the matrix cores (stgcn_forward_mx.hip); RULGNN_EUNSUPPORTED when the shape is outside its rules
int stgcn_train_f0_mx(const rulgnn_stgcn_shape* s, const float* x, const float* prm, float* cacheX, float* cacheA, float* H0, float* Z1,
                      double* cells_bn0, int cell_stride_doubles, int replicas, hipStream_t stream);
size_t stgcn_tiled_forward_workspace_bytes(const rulgnn_stgcn_shape* s);
int stgcn_tiled_forward_eval(const rulgnn_stgcn_shape* s, const float* x, const float* prm, const float* bn, float* pred,
                             void* workspace, size_t workspace_bytes, hipStream_t stream);
size_t stgcn_tiled_train_workspace_bytes(const rulgnn_stgcn_shape* s);
// optional "these gradients are final" callback of the tiled training step (include/rulgnn.h: rulgnn_grad_ready_fn)
struct GradReadyHook {
    rulgnn_grad_ready_fn fn;
    void* user;
};
int stgcn_tiled_train(const rulgnn_stgcn_shape* s, const rulgnn_stgcn_train_args* a, int mode, hipStream_t stream,
                      const GradReadyHook* ready = nullptr);
size_t stgcn_train_workspace_bytes(const rulgnn_stgcn_shape* s);
int stgcn_train_forward(const rulgnn_stgcn_shape* s, const rulgnn_stgcn_train_args* a, hipStream_t stream);
int stgcn_train_backward(const rulgnn_stgcn_shape* s, const rulgnn_stgcn_train_args* a, hipStream_t stream);
int stgcn_train_fwdbwd(const rulgnn_stgcn_shape* s, const rulgnn_stgcn_train_args* a, hipStream_t stream);
int stgcn_train_step(const rulgnn_stgcn_shape* s, const rulgnn_stgcn_train_args* a, const rulgnn_adam_args* opt,
                     hipStream_t stream, int path = RULGNN_STEP_AUTO);     // opt == nullptr: forward + backward only
int stgcn_train_phase(const rulgnn_stgcn_shape* s, const rulgnn_stgcn_train_args* a, int phase, hipStream_t stream,
                      int path = RULGNN_STEP_AUTO);
int stgcn_train_fwdbwd_syncbn(const rulgnn_stgcn_shape* s, const rulgnn_stgcn_train_args* a, float bn_param_grad_scale,
                              rulgnn_allreduce_f64_fn allreduce, void* user, hipStream_t stream, int path = RULGNN_STEP_CHAIN);
int stgcn_train_mx_kind(const rulgnn_stgcn_shape* s, const float* x);        // 0 fp32 phases, 1 matrix-core chain, 2 wide matrix-core chain
int64_t stgcn_train_guard_counter_offset(const rulgnn_stgcn_shape* s);
int64_t stmsgcn_param_count(const rulgnn_stmsgcn_shape* s);
size_t stmsgcn_workspace_bytes(const rulgnn_stmsgcn_shape* s);
int stmsgcn_features(const rulgnn_stmsgcn_shape* s, const float* x, const float* prm, float* features, hipStream_t stream);
int stmsgcn_run(const rulgnn_stmsgcn_shape* s, const rulgnn_stmsgcn_args* a, int mode, hipStream_t stream);
int64_t astgcnn_param_count(const rulgnn_astgcnn_shape* s);
size_t astgcnn_workspace_bytes(const rulgnn_astgcnn_shape* s);
struct BnSyncHook;
struct AdamFuse;          // adam_device.hpp
// (`bn_running_out` != nullptr, whole training steps with plain batch statistics: the finalize kernel also updates the running statistics;
// `adam` != nullptr, whole training steps: the constants of the optimizer update (adam_fuse_args) -- on return adam->gbase != nullptr says
// the step's last kernel applied it, else the caller launches adam_step)
int astgcnn_run(const rulgnn_astgcnn_shape* s, const rulgnn_astgcnn_args* a, int mode, hipStream_t stream, const BnSyncHook* sync = nullptr,
                float* bn_running_out = nullptr, float bn_momentum = 0.f, AdamFuse* adam = nullptr);
void adam_fuse_args(AdamFuse* t, float* p, float* m, float* v, const float* gbase, int64_t step, float lr, float beta1, float beta2, float eps,
                    float wd);
int astgcnn_bn_running_update(const rulgnn_astgcnn_shape* s, float* bn_stats, const float* bn_batch, int64_t count, float momentum,
                              int from_moments, hipStream_t stream);
int64_t fcstgnn_param_count(const rulgnn_fcstgnn_shape* s);
int64_t fcstgnn_bn_count(const rulgnn_fcstgnn_shape* s);
size_t fcstgnn_workspace_bytes(const rulgnn_fcstgnn_shape* s);
// synchronised BatchNorm hook of the FC_STGNN / ASTGCNN steps (include/rulgnn.h: rulgnn_*_fwdbwd_syncbn_f32)
struct BnSyncHook {
    rulgnn_allreduce_f64_fn fn;
    void* user;
    float bn_param_grad_scale;
};
typedef BnSyncHook FcstgnnSync;
// (`bn_running_out` != nullptr, whole training steps with plain batch statistics: the running-statistics update rides on the side stream
// behind the batch-statistics kernel instead of closing the step)
// (`fuse` != nullptr, whole training steps on one rank: the step's last launch -- the finalize kernel, then behind the join with the side
// stream -- applies torch.optim.Adam to every parameter: adam_device.hpp; no optimizer launch)
struct AdamFuse;
int fcstgnn_run(const rulgnn_fcstgnn_shape* s, const rulgnn_fcstgnn_args* a, int mode, hipStream_t stream, const FcstgnnSync* sync = nullptr,
                float* bn_running_out = nullptr, float bn_momentum = 0.f, const AdamFuse* fuse = nullptr);
int fcstgnn_bn_running_update(const rulgnn_fcstgnn_shape* s, float* bn_stats, const float* bn_batch, float momentum, int from_moments,
                              hipStream_t stream);
int64_t hagcn_graph_param_count(const rulgnn_hagcn_shape* s);
size_t hagcn_workspace_bytes(const rulgnn_hagcn_shape* s);
int hagcn_graph_forward(const rulgnn_hagcn_shape* s, const rulgnn_hagcn_args* a, hipStream_t stream);
int hagcn_graph_backward(const rulgnn_hagcn_shape* s, const rulgnn_hagcn_args* a, hipStream_t stream);
size_t bilstm_workspace_bytes(const rulgnn_bilstm_shape* s);
int bilstm_forward(const rulgnn_bilstm_shape* s, const rulgnn_bilstm_args* a, hipStream_t stream, int ndir = 2);
int bilstm_backward(const rulgnn_bilstm_shape* s, const rulgnn_bilstm_args* a, hipStream_t stream, int ndir = 2);
int64_t stconv_param_count(const rulgnn_stconv_shape* s);
size_t stconv_workspace_bytes(const rulgnn_stconv_shape* s);
int stconv_run(const rulgnn_stconv_shape* s, const rulgnn_astgcnn_args* a, int mode, hipStream_t stream);
int stconv_bn_running_update(const rulgnn_stconv_shape* s, float* bn_stats, const float* bn_batch, int64_t count, float momentum,
                             int from_moments, hipStream_t stream);
int adam_step(float* p, const float* g, float* m, float* v, int64_t n, int64_t step, float lr, float beta1, float beta2,
              float eps, float wd, float gscale, hipStream_t stream, void* step_state = nullptr, const float* guard = nullptr);
int step_state_set(void* state, uint64_t dropout_step, int64_t adam_step, hipStream_t stream);
int step_prepare_dropout(void* state, uint64_t seed, int num_layers, hipStream_t stream);   // ++dropout_step, keys
int step_prepare_adam(void* state, float lr, float beta1, float beta2, hipStream_t stream); // ++adam_step, bias corrections
int bn_running_update(float* bn, const float* batch, int num_layers, int64_t count, float momentum, int from_moments,
                      hipStream_t stream, const float* guard = nullptr);
// adam_step + bn_running_update in one launch (what follows a data-parallel bucket all-reduce); `guard` may be null
int adam_bn_step(float* p, const float* g, float* m, float* v, int64_t n, int64_t step, float lr, float beta1, float beta2, float eps,
                 float wd, float gscale, float* bn, const float* batch, int num_layers, int64_t count, float momentum, int from_moments,
                 const float* guard, hipStream_t stream);

size_t stgnn_workspace_bytes(const rulgnn_stgnn_shape* s);
int stgnn_terms(const rulgnn_stgnn_shape* s, const float* x, float* terms, float* adj, hipStream_t st);
int stgnn_cheb_forward(const rulgnn_stgnn_shape* s, const float* terms, const float* filters, float* out, hipStream_t st);
int stgnn_cheb_backward(const rulgnn_stgnn_shape* s, const float* terms, const float* dout, float* dfilters, void* workspace,
                        size_t workspace_bytes, hipStream_t st);
int64_t stnet_param_count(const rulgnn_stnet_shape* s);
size_t stnet_workspace_bytes(const rulgnn_stnet_shape* s);
int stnet_run(const rulgnn_stnet_shape* s, const rulgnn_stnet_args* a, int mode, hipStream_t st);
int64_t sagcn_param_count(const rulgnn_sagcn_shape* s);
size_t sagcn_workspace_bytes(const rulgnn_sagcn_shape* s);
int64_t sagcn_tap_offset(const rulgnn_sagcn_shape* s, int which);
int sagcn_run(const rulgnn_sagcn_shape* s, const rulgnn_sagcn_args* a, int mode, hipStream_t st);
int64_t stagnn_param_count(const rulgnn_stagnn_shape* s);
int64_t stagnn_bn_state_count(const rulgnn_stagnn_shape* s);
size_t stagnn_workspace_bytes(const rulgnn_stagnn_shape* s);
int64_t stagnn_tap_offset(const rulgnn_stagnn_shape* s, int which);
int stagnn_run(const rulgnn_stagnn_shape* s, const rulgnn_stagnn_args* a, int mode, hipStream_t st);
int64_t rgcnu_param_count(const rulgnn_rgcnu_shape* s);
size_t rgcnu_workspace_bytes(const rulgnn_rgcnu_shape* s);
int rgcnu_run(const rulgnn_rgcnu_shape* s, const rulgnn_rgcnu_args* a, int mode, hipStream_t st);
int64_t stgnn_param_count(const rulgnn_stgnn_shape* s);
size_t stgnn_step_workspace_bytes(const rulgnn_stgnn_shape* s);
int stgnn_run(const rulgnn_stgnn_shape* s, const rulgnn_stmsgcn_args* a, int mode, hipStream_t st);
size_t gru_workspace_bytes(const rulgnn_gru_shape* s);
int gru_forward(const rulgnn_gru_shape* s, const rulgnn_gru_args* a, hipStream_t st);
int gru_backward(const rulgnn_gru_shape* s, const rulgnn_gru_args* a, hipStream_t st);
size_t rul_metrics_workspace_bytes(int64_t n);
int rul_metrics(const float* pred, const float* real, int64_t n, float max_rul, double* out, void* workspace, size_t workspace_bytes,
                hipStream_t st, int raw = 0);

}  // namespace rulgnn
