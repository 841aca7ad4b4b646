// extern "C" surface of librulgnn.so (declared in include/rulgnn.h).
#include <initializer_list>

#include "sgemm_mfma.hpp"
#include "stgcn_host.hpp"
#include "adam_device.hpp"
#include "stgcn_train_mx.hpp"

using namespace rulgnn;

namespace rulgnn {
int& sgemm_big_mode() {
    static int mode = RULGNN_GEMM_BF16X3;
    return mode;
}
}  // namespace rulgnn

extern "C" {

int rulgnn_version(void) { return 100; }   // 0.1.0

const char* rulgnn_strerror(int code) {
    switch (code) {
        case RULGNN_OK: return "ok";
        case RULGNN_EINVAL: return "invalid argument (shape, null pointer or hyper-parameter)";
        case RULGNN_EUNSUPPORTED: return "shape not covered by the fused gfx950 kernels";
        case RULGNN_EWORKSPACE: return "workspace too small";
        case RULGNN_EHIP: return "HIP runtime error";
        case RULGNN_EALIGN: return "pointer not 4-byte aligned";
        case RULGNN_ECALLBACK: return "caller-supplied callback failed";
        default: return "unknown rulgnn error";
    }
}

int64_t rulgnn_stgcn_param_count(int32_t num_patch, int32_t num_layers) {
    if (num_patch < 1 || num_layers < 1) return -1;
    return param_count(num_patch, num_layers);
}
int64_t rulgnn_stgcn_param_count_order(int32_t num_patch, int32_t num_layers, int32_t mpnn_k) {
    if (num_patch < 1 || num_layers < 1 || mpnn_k < 1) return -1;
    return param_count(num_patch, num_layers, mpnn_k);
}

static int check_ptrs(std::initializer_list<const void*> ps) {
    for (const void* p : ps) {
        if (!p) return RULGNN_EINVAL;
        if (reinterpret_cast<uintptr_t>(p) & 3) return RULGNN_EALIGN;
    }
    return RULGNN_OK;
}

// Path selection.  The fused row-mapped kernels cover num_patch <= 64 as long as one wavefront's input tile
// fits its LDS staging area (and, for training, num_layers <= 3 / <= 2); everything else valid goes to the
// tiled path, which has no such limits.
static bool tiled_eval(const rulgnn_stgcn_shape* shape) {
    TileGeom g;
    return tile_geometry(shape, &g) != RULGNN_OK;
}
static bool tiled(const rulgnn_stgcn_shape* shape) { return stgcn_train_workspace_bytes(shape) == 0; }

// MPNN order k > 1 exists on the fused row-mapped kernels only: a shape they cannot hold (a window beyond a wavefront's LDS staging area,
// more layers than the phase chain is instantiated for) must NOT fall through to the tiled kernels, which read the order-1 layout.
static bool order_needs_tiled(const rulgnn_stgcn_shape* shape, bool train) {
    return shape->mpnn_k != 1 && (train ? tiled(shape) : tiled_eval(shape));
}

size_t rulgnn_stgcn_forward_workspace_bytes(const rulgnn_stgcn_shape* shape) {
    if (validate_shape(shape) != RULGNN_OK || order_needs_tiled(shape, false)) return 0;
    return tiled_eval(shape) ? stgcn_tiled_forward_workspace_bytes(shape) : 0;
}

int rulgnn_stgcn_forward_path_f32(const rulgnn_stgcn_shape* shape, const float* x, const float* params,
                                  const float* bn_stats, float* pred, void* workspace, size_t workspace_bytes,
                                  int path, void* stream) {
    int rc = validate_shape(shape);
    if (rc != RULGNN_OK) return rc;
    if (path != RULGNN_EVAL_AUTO && path != RULGNN_EVAL_EXACT && path != RULGNN_EVAL_MX) return RULGNN_EINVAL;
    if (order_needs_tiled(shape, false)) return RULGNN_EUNSUPPORTED;
    if (shape->batch == 0) return RULGNN_OK;
    rc = check_ptrs({x, params, bn_stats, pred});
    if (rc != RULGNN_OK) return rc;
    if (tiled_eval(shape))
        return stgcn_tiled_forward_eval(shape, x, params, bn_stats, pred, workspace, workspace_bytes,
                                        static_cast<hipStream_t>(stream));
    return stgcn_forward_eval(shape, x, params, bn_stats, pred, static_cast<hipStream_t>(stream), path);
}

int rulgnn_stgcn_forward_f32(const rulgnn_stgcn_shape* shape, const float* x, const float* params,
                             const float* bn_stats, float* pred, void* workspace, size_t workspace_bytes,
                             void* stream) {
    return rulgnn_stgcn_forward_path_f32(shape, x, params, bn_stats, pred, workspace, workspace_bytes, RULGNN_EVAL_AUTO, stream);
}

int rulgnn_stgcn_forward_mx_tap_floats(void) { return stgcn_forward_mx_tap_floats(); }

int rulgnn_stgcn_forward_mx_taps_f32(const rulgnn_stgcn_shape* shape, const float* x, const float* params,
                                     const float* bn_stats, float* pred, float* taps, void* stream) {
    int rc = validate_shape(shape);
    if (rc != RULGNN_OK) return rc;
    rc = check_ptrs({x, params, bn_stats, pred, taps});
    if (rc != RULGNN_OK) return rc;
    return stgcn_forward_eval_mx(shape, x, params, bn_stats, pred, static_cast<hipStream_t>(stream), taps);
}

static size_t train_ws_bytes(const rulgnn_stgcn_shape* shape) {
    if (order_needs_tiled(shape, true)) return 0;
    return tiled(shape) ? stgcn_tiled_train_workspace_bytes(shape) : stgcn_train_workspace_bytes(shape);
}

size_t rulgnn_stgcn_train_workspace_bytes(const rulgnn_stgcn_shape* shape) {
    if (validate_shape(shape) != RULGNN_OK) return 0;
    return train_ws_bytes(shape);
}

static int check_train(const rulgnn_stgcn_shape* shape, const rulgnn_stgcn_train_args* a, bool need_grads) {
    int rc = validate_shape(shape);
    if (rc != RULGNN_OK) return rc;
    if (!a) return RULGNN_EINVAL;
    if (shape->batch < 1) return RULGNN_EINVAL;                 // BatchNorm needs a batch
    if (a->global_batch < shape->batch || a->sample_offset < 0) return RULGNN_EINVAL;
    if (!(a->dropout_p >= 0.f && a->dropout_p < 1.f)) return RULGNN_EINVAL;
    if (!(a->bn_moment_weight >= 0.f)) return RULGNN_EINVAL;
    rc = check_ptrs({a->x, a->params, a->pred, a->bn_batch, a->workspace});
    if (rc != RULGNN_OK) return rc;
    if (need_grads) {
        rc = check_ptrs({a->grads});
        if (rc != RULGNN_OK) return rc;
        if (!a->dpred) {
            rc = check_ptrs({a->y, a->loss});
            if (rc != RULGNN_OK) return rc;
        } else if (reinterpret_cast<uintptr_t>(a->dpred) & 3) {
            return RULGNN_EALIGN;
        }
    }
    const size_t need = train_ws_bytes(shape);
    if (need == 0) return RULGNN_EUNSUPPORTED;
    if (a->workspace_bytes < need) return RULGNN_EWORKSPACE;
    return RULGNN_OK;
}

int rulgnn_stgcn_train_forward_f32(const rulgnn_stgcn_shape* shape, const rulgnn_stgcn_train_args* args, void* stream) {
    const int rc = check_train(shape, args, false);
    if (rc != RULGNN_OK) return rc;
    if (tiled(shape)) return stgcn_tiled_train(shape, args, 0, static_cast<hipStream_t>(stream));
    return stgcn_train_forward(shape, args, static_cast<hipStream_t>(stream));
}

int rulgnn_stgcn_train_backward_f32(const rulgnn_stgcn_shape* shape, const rulgnn_stgcn_train_args* args, void* stream) {
    const int rc = check_train(shape, args, true);
    if (rc != RULGNN_OK) return rc;
    if (tiled(shape)) return stgcn_tiled_train(shape, args, 1, static_cast<hipStream_t>(stream));
    return stgcn_train_backward(shape, args, static_cast<hipStream_t>(stream));
}

int rulgnn_stgcn_train_fwdbwd_f32(const rulgnn_stgcn_shape* shape, const rulgnn_stgcn_train_args* args, void* stream) {
    const int rc = check_train(shape, args, true);
    if (rc != RULGNN_OK) return rc;
    if (tiled(shape)) return stgcn_tiled_train(shape, args, 2, static_cast<hipStream_t>(stream));
    return stgcn_train_fwdbwd(shape, args, static_cast<hipStream_t>(stream));
}

int rulgnn_stgcn_train_fwdbwd_ready_f32(const rulgnn_stgcn_shape* shape, const rulgnn_stgcn_train_args* args, rulgnn_grad_ready_fn ready,
                                        void* user, void* stream) {
    const int rc = check_train(shape, args, true);
    if (rc != RULGNN_OK) return rc;
    if (!ready) return RULGNN_EINVAL;
    if (tiled(shape)) {
        const GradReadyHook hook = {ready, user};
        return stgcn_tiled_train(shape, args, 2, static_cast<hipStream_t>(stream), &hook);
    }
    return stgcn_train_fwdbwd(shape, args, static_cast<hipStream_t>(stream));     // buckets of a few KB: nothing to overlap
}

int rulgnn_stgcn_train_fwdbwd_syncbn_f32(const rulgnn_stgcn_shape* shape, const rulgnn_stgcn_train_args* args,
                                         float bn_param_grad_scale, rulgnn_allreduce_f64_fn allreduce, void* user, void* stream) {
    const int rc = check_train(shape, args, true);
    if (rc != RULGNN_OK) return rc;
    if (!(bn_param_grad_scale >= 0.f && bn_param_grad_scale <= 1.f) || !allreduce || args->bn_moment_weight != 0.f) return RULGNN_EINVAL;
    if (tiled(shape)) return RULGNN_EUNSUPPORTED;          // the tiled path keeps local statistics
    return stgcn_train_fwdbwd_syncbn(shape, args, bn_param_grad_scale, allreduce, user, static_cast<hipStream_t>(stream), RULGNN_STEP_CHAIN);
}

int rulgnn_stgcn_train_fwdbwd_syncbn_path_f32(const rulgnn_stgcn_shape* shape, const rulgnn_stgcn_train_args* args,
                                              float bn_param_grad_scale, rulgnn_allreduce_f64_fn allreduce, void* user, int32_t path,
                                              void* stream) {
    const int rc = check_train(shape, args, true);
    if (rc != RULGNN_OK) return rc;
    if (!(bn_param_grad_scale >= 0.f && bn_param_grad_scale <= 1.f) || !allreduce || args->bn_moment_weight != 0.f) return RULGNN_EINVAL;
    if (path != RULGNN_STEP_AUTO && path != RULGNN_STEP_CHAIN && path != RULGNN_STEP_MX) return RULGNN_EINVAL;
    if (tiled(shape)) return RULGNN_EUNSUPPORTED;
    return stgcn_train_fwdbwd_syncbn(shape, args, bn_param_grad_scale, allreduce, user, static_cast<hipStream_t>(stream), path);
}

int64_t rulgnn_stgcn_train_guard_counter_offset(const rulgnn_stgcn_shape* shape) {
    if (validate_shape(shape) != RULGNN_OK) return -1;
    return tiled(shape) ? -1 : stgcn_train_guard_counter_offset(shape);
}

size_t rulgnn_stgcn_train_args_size(void) { return sizeof(rulgnn_stgcn_train_args); }

size_t rulgnn_struct_size(int32_t which) {
    switch (which) {
    case RULGNN_STRUCT_STGCN_SHAPE: return sizeof(rulgnn_stgcn_shape);
    case RULGNN_STRUCT_STGCN_TRAIN_ARGS: return sizeof(rulgnn_stgcn_train_args);
    case RULGNN_STRUCT_ADAM_ARGS: return sizeof(rulgnn_adam_args);
    case RULGNN_STRUCT_STMSGCN_SHAPE: return sizeof(rulgnn_stmsgcn_shape);
    case RULGNN_STRUCT_STMSGCN_ARGS: return sizeof(rulgnn_stmsgcn_args);
    case RULGNN_STRUCT_ASTGCNN_SHAPE: return sizeof(rulgnn_astgcnn_shape);
    case RULGNN_STRUCT_ASTGCNN_ARGS: return sizeof(rulgnn_astgcnn_args);
    case RULGNN_STRUCT_FCSTGNN_SHAPE: return sizeof(rulgnn_fcstgnn_shape);
    case RULGNN_STRUCT_FCSTGNN_ARGS: return sizeof(rulgnn_fcstgnn_args);
    case RULGNN_STRUCT_RGCNU_SHAPE: return sizeof(rulgnn_rgcnu_shape);
    case RULGNN_STRUCT_RGCNU_ARGS: return sizeof(rulgnn_rgcnu_args);
    case RULGNN_STRUCT_STNET_SHAPE: return sizeof(rulgnn_stnet_shape);
    case RULGNN_STRUCT_STNET_ARGS: return sizeof(rulgnn_stnet_args);
    case RULGNN_STRUCT_SAGCN_SHAPE: return sizeof(rulgnn_sagcn_shape);
    case RULGNN_STRUCT_SAGCN_ARGS: return sizeof(rulgnn_sagcn_args);
    case RULGNN_STRUCT_STAGNN_SHAPE: return sizeof(rulgnn_stagnn_shape);
    case RULGNN_STRUCT_STAGNN_ARGS: return sizeof(rulgnn_stagnn_args);
    case RULGNN_STRUCT_HAGCN_SHAPE: return sizeof(rulgnn_hagcn_shape);
    case RULGNN_STRUCT_HAGCN_ARGS: return sizeof(rulgnn_hagcn_args);
    case RULGNN_STRUCT_BILSTM_SHAPE: return sizeof(rulgnn_bilstm_shape);
    case RULGNN_STRUCT_BILSTM_ARGS: return sizeof(rulgnn_bilstm_args);
    case RULGNN_STRUCT_STCONV_SHAPE: return sizeof(rulgnn_stconv_shape);
    case RULGNN_STRUCT_STGNN_SHAPE: return sizeof(rulgnn_stgnn_shape);
    case RULGNN_STRUCT_GRU_SHAPE: return sizeof(rulgnn_gru_shape);
    case RULGNN_STRUCT_GRU_ARGS: return sizeof(rulgnn_gru_args);
    default: return 0;
    }
}

int rulgnn_stgcn_train_step_f32(const rulgnn_stgcn_shape* shape, const rulgnn_stgcn_train_args* args,
                                const rulgnn_adam_args* opt, void* stream) {
    if (!opt) return RULGNN_EINVAL;
    return rulgnn_stgcn_train_step_path_f32(shape, args, opt, RULGNN_STEP_AUTO, stream);
}

int rulgnn_stgcn_train_step_path_f32(const rulgnn_stgcn_shape* shape, const rulgnn_stgcn_train_args* args,
                                     const rulgnn_adam_args* opt, int32_t path, void* stream) {
    int rc = check_train(shape, args, true);
    if (rc != RULGNN_OK) return rc;
    if (path != RULGNN_STEP_AUTO && path != RULGNN_STEP_CHAIN && path != RULGNN_STEP_COOP && path != RULGNN_STEP_MX && path != RULGNN_STEP_MX_PERSIST)
        return RULGNN_EINVAL;
    if (!opt) {                                            // forward + backward only
        if (tiled(shape)) return stgcn_tiled_train(shape, args, 2, static_cast<hipStream_t>(stream));
        return stgcn_train_step(shape, args, nullptr, static_cast<hipStream_t>(stream), path);
    }
    if ((opt->step < 1 && !opt->step_state) || args->dpred) return RULGNN_EINVAL;
    if (opt->params != args->params) return RULGNN_EINVAL;
    rc = check_ptrs({opt->params, opt->exp_avg, opt->exp_avg_sq});
    if (rc != RULGNN_OK) return rc;
    if (opt->bn_stats && (reinterpret_cast<uintptr_t>(opt->bn_stats) & 3)) return RULGNN_EALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (tiled(shape)) {                                    // tiled path: same call, optimizer (+ running statistics) as one more kernel
        rc = stgcn_tiled_train(shape, args, 2, st);
        if (rc != RULGNN_OK) return rc;
        if (opt->bn_stats && !opt->step_state)
            return adam_bn_step(opt->params, args->grads, opt->exp_avg, opt->exp_avg_sq, param_count(shape->num_patch, shape->num_layers),
                                opt->step, opt->lr, opt->beta1, opt->beta2, opt->eps, opt->weight_decay, 1.0f, opt->bn_stats, args->bn_batch,
                                shape->num_layers, shape->batch * (int64_t)shape->num_patch, opt->bn_momentum,
                                args->bn_moment_weight > 0.f ? 1 : 0, nullptr, st);
        rc = adam_step(opt->params, args->grads, opt->exp_avg, opt->exp_avg_sq,
                       param_count(shape->num_patch, shape->num_layers), opt->step, opt->lr, opt->beta1, opt->beta2, opt->eps,
                       opt->weight_decay, 1.0f, st, opt->step_state);
        if (rc != RULGNN_OK || !opt->bn_stats) return rc;
        return bn_running_update(opt->bn_stats, args->bn_batch, shape->num_layers, shape->batch * (int64_t)shape->num_patch,
                                 opt->bn_momentum, args->bn_moment_weight > 0.f ? 1 : 0, st);
    }
    return stgcn_train_step(shape, args, opt, st, path);
}

int rulgnn_stgcn_train_step_resolve(const rulgnn_stgcn_shape* shape, const float* x, int32_t path) {
    const int rc = validate_shape(shape);
    if (rc != RULGNN_OK) return rc;
    if (tiled(shape)) return RULGNN_EUNSUPPORTED;
    if (path == RULGNN_STEP_COOP && shape->mpnn_k != 1) return RULGNN_EUNSUPPORTED;      // the single launch is built for order 1
    if (path == RULGNN_STEP_CHAIN || path == RULGNN_STEP_COOP) return path;
    if (path != RULGNN_STEP_AUTO && path != RULGNN_STEP_MX && path != RULGNN_STEP_MX_PERSIST) return RULGNN_EINVAL;
    const int kind = stgcn_train_mx_kind(shape, x);
    if (path == RULGNN_STEP_MX_PERSIST)          // the small-batch single launch: the 4-sample-tile chain, two layers, a workgroup per CU
        return kind == 1 && stgcn_train_mx_persistent_grid(shape->batch, shape->num_layers, 2048) > 0 ? RULGNN_STEP_MX : RULGNN_EUNSUPPORTED;
    if (kind != 0) return RULGNN_STEP_MX;
    return path != RULGNN_STEP_AUTO ? RULGNN_EUNSUPPORTED : RULGNN_STEP_CHAIN;
}

int rulgnn_stgcn_train_phase_count(int32_t num_layers) { return num_layers >= 1 ? 4 * num_layers + 1 : -1; }

int rulgnn_stgcn_train_phase_f32(const rulgnn_stgcn_shape* shape, const rulgnn_stgcn_train_args* args, int32_t phase,
                                 void* stream) {
    const int rc = check_train(shape, args, true);
    if (rc != RULGNN_OK) return rc;
    if (tiled(shape)) return RULGNN_EUNSUPPORTED;          // the tiled path is not a phase chain
    return stgcn_train_phase(shape, args, phase, static_cast<hipStream_t>(stream));
}

int rulgnn_adam_step_f32(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t step,
                         float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                         void* stream) {
    if (n < 0 || step < 1) return RULGNN_EINVAL;
    const int rc = check_ptrs({params, grads, exp_avg, exp_avg_sq});
    if (rc != RULGNN_OK) return rc;
    return adam_step(params, grads, exp_avg, exp_avg_sq, n, step, lr, beta1, beta2, eps, weight_decay, grad_scale,
                     static_cast<hipStream_t>(stream));
}

int rulgnn_adam_step_guarded_f32(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t step,
                                 float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                                 const float* guard, void* stream) {
    if (n < 0 || step < 1) return RULGNN_EINVAL;
    const int rc = check_ptrs({params, grads, exp_avg, exp_avg_sq, guard});
    if (rc != RULGNN_OK) return rc;
    return adam_step(params, grads, exp_avg, exp_avg_sq, n, step, lr, beta1, beta2, eps, weight_decay, grad_scale,
                     static_cast<hipStream_t>(stream), nullptr, guard);
}

int rulgnn_bn_running_update_guarded_f32(float* bn_stats, const float* bn_batch, int32_t num_layers, int64_t count,
                                         float momentum, int32_t from_moments, const float* guard, void* stream) {
    if (num_layers < 1 || num_layers > 8 || count < 1) return RULGNN_EINVAL;
    const int rc = check_ptrs({bn_stats, bn_batch, guard});
    if (rc != RULGNN_OK) return rc;
    return bn_running_update(bn_stats, bn_batch, num_layers, count, momentum, from_moments, static_cast<hipStream_t>(stream), guard);
}

int rulgnn_adam_bn_step_f32(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t step, float lr,
                            float beta1, float beta2, float eps, float weight_decay, float grad_scale, float* bn_stats,
                            const float* bn_batch, int32_t num_layers, int64_t count, float momentum, int32_t from_moments,
                            const float* guard, void* stream) {
    if (n < 0 || step < 1 || num_layers < 1 || num_layers > 8 || count < 1) return RULGNN_EINVAL;
    const int rc = check_ptrs({params, grads, exp_avg, exp_avg_sq, bn_stats, bn_batch});
    if (rc != RULGNN_OK) return rc;
    return adam_bn_step(params, grads, exp_avg, exp_avg_sq, n, step, lr, beta1, beta2, eps, weight_decay, grad_scale, bn_stats,
                        bn_batch, num_layers, count, momentum, from_moments, guard, static_cast<hipStream_t>(stream));
}

int rulgnn_bn_running_update_f32(float* bn_stats, const float* bn_batch, int32_t num_layers, int64_t count,
                                 float momentum, int32_t from_moments, void* stream) {
    if (num_layers < 1 || num_layers > 8 || count < 1) return RULGNN_EINVAL;
    const int rc = check_ptrs({bn_stats, bn_batch});
    if (rc != RULGNN_OK) return rc;
    return bn_running_update(bn_stats, bn_batch, num_layers, count, momentum, from_moments, static_cast<hipStream_t>(stream));
}


// ---- STMSGCN ------------------------------------------------------------------------------------------
int64_t rulgnn_stmsgcn_param_count(const rulgnn_stmsgcn_shape* shape) { return stmsgcn_param_count(shape); }

size_t rulgnn_stmsgcn_workspace_bytes(const rulgnn_stmsgcn_shape* shape) { return stmsgcn_workspace_bytes(shape); }

int rulgnn_stmsgcn_features_f32(const rulgnn_stmsgcn_shape* shape, const float* x, const float* params, float* features,
                                void* stream) {
    if (!shape) return RULGNN_EINVAL;
    if (stmsgcn_param_count(shape) >= 0 && shape->batch == 0) return RULGNN_OK;
    const int rc = check_ptrs({x, params, features});
    if (rc != RULGNN_OK) return rc;
    return stmsgcn_features(shape, x, params, features, static_cast<hipStream_t>(stream));
}

static int check_stmsgcn(const rulgnn_stmsgcn_shape* shape, const rulgnn_stmsgcn_args* a, bool backward, bool need_target) {
    if (!shape || !a) return RULGNN_EINVAL;
    if (shape->batch < 1 || a->global_batch < shape->batch) return RULGNN_EINVAL;
    int rc = check_ptrs({a->x, a->params, a->pred, a->workspace});
    if (rc != RULGNN_OK) return rc;
    if (a->y && (reinterpret_cast<uintptr_t>(a->y) & 3)) return RULGNN_EALIGN;
    if (backward) {
        rc = check_ptrs({a->grads});
        if (rc != RULGNN_OK) return rc;
        if (a->dpred) {
            if (reinterpret_cast<uintptr_t>(a->dpred) & 3) return RULGNN_EALIGN;
        } else if (need_target) {
            rc = check_ptrs({a->y, a->loss});
            if (rc != RULGNN_OK) return rc;
        }
    }
    return RULGNN_OK;
}

int rulgnn_stmsgcn_forward_f32(const rulgnn_stmsgcn_shape* shape, const rulgnn_stmsgcn_args* args, void* stream) {
    const int rc = check_stmsgcn(shape, args, false, false);
    if (rc != RULGNN_OK) return rc;
    return stmsgcn_run(shape, args, 1, static_cast<hipStream_t>(stream));
}

int rulgnn_stmsgcn_backward_f32(const rulgnn_stmsgcn_shape* shape, const rulgnn_stmsgcn_args* args, void* stream) {
    const int rc = check_stmsgcn(shape, args, true, true);
    if (rc != RULGNN_OK) return rc;
    return stmsgcn_run(shape, args, 2, static_cast<hipStream_t>(stream));
}

int rulgnn_stmsgcn_fwdbwd_f32(const rulgnn_stmsgcn_shape* shape, const rulgnn_stmsgcn_args* args, const rulgnn_adam_args* opt,
                              void* stream) {
    int rc = check_stmsgcn(shape, args, true, true);
    if (rc != RULGNN_OK) return rc;
    if (args->dpred) return RULGNN_EINVAL;                 // the fused call is the MSE step
    if (opt) {
        if ((opt->step < 1 && !opt->step_state) || opt->params != args->params) return RULGNN_EINVAL;
        rc = check_ptrs({opt->params, opt->exp_avg, opt->exp_avg_sq});
        if (rc != RULGNN_OK) return rc;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    rc = stmsgcn_run(shape, args, 3, st);
    if (rc != RULGNN_OK || !opt) return rc;
    return adam_step(opt->params, args->grads, opt->exp_avg, opt->exp_avg_sq, stmsgcn_param_count(shape), opt->step, opt->lr,
                     opt->beta1, opt->beta2, opt->eps, opt->weight_decay, 1.0f, st, opt->step_state);
}


// ---- ASTGCNN ------------------------------------------------------------------------------------------
int64_t rulgnn_astgcnn_param_count(const rulgnn_astgcnn_shape* shape) { return astgcnn_param_count(shape); }

size_t rulgnn_astgcnn_workspace_bytes(const rulgnn_astgcnn_shape* shape) { return astgcnn_workspace_bytes(shape); }

static int check_astgcnn(const rulgnn_astgcnn_shape* shape, const rulgnn_astgcnn_args* a, bool forward, bool backward) {
    if (!shape || !a) return RULGNN_EINVAL;
    if (shape->batch < 1 || a->global_batch < shape->batch) return RULGNN_EINVAL;
    if (!(a->bn_moment_weight >= 0.f)) return RULGNN_EINVAL;
    int rc = check_ptrs({a->x, a->params, a->pred, a->workspace});
    if (rc != RULGNN_OK) return rc;
    if (forward && !a->training) {
        rc = check_ptrs({a->bn_stats});
        if (rc != RULGNN_OK) return rc;
    }
    for (const void* p : {(const void*)a->y, (const void*)a->dpred, (const void*)a->bn_batch, (const void*)a->loss})
        if (p && (reinterpret_cast<uintptr_t>(p) & 3)) return RULGNN_EALIGN;
    if (backward) {
        if (!a->training) return RULGNN_EINVAL;            // the backward is the train-mode (batch-statistics) one
        rc = check_ptrs({a->grads});
        if (rc != RULGNN_OK) return rc;
        if (!a->dpred) {
            rc = check_ptrs({a->y, a->loss});
            if (rc != RULGNN_OK) return rc;
        }
    }
    return RULGNN_OK;
}

int rulgnn_astgcnn_forward_f32(const rulgnn_astgcnn_shape* shape, const rulgnn_astgcnn_args* args, void* stream) {
    const int rc = check_astgcnn(shape, args, true, false);
    if (rc != RULGNN_OK) return rc;
    return astgcnn_run(shape, args, 1, static_cast<hipStream_t>(stream));
}

int rulgnn_astgcnn_backward_f32(const rulgnn_astgcnn_shape* shape, const rulgnn_astgcnn_args* args, void* stream) {
    const int rc = check_astgcnn(shape, args, false, true);
    if (rc != RULGNN_OK) return rc;
    return astgcnn_run(shape, args, 2, static_cast<hipStream_t>(stream));
}

int rulgnn_astgcnn_fwdbwd_f32(const rulgnn_astgcnn_shape* shape, const rulgnn_astgcnn_args* args, const rulgnn_adam_args* opt,
                              void* stream) {
    int rc = check_astgcnn(shape, args, true, true);
    if (rc != RULGNN_OK) return rc;
    if (args->dpred) return RULGNN_EINVAL;
    if (opt) {
        if ((opt->step < 1 && !opt->step_state) || opt->params != args->params) return RULGNN_EINVAL;
        rc = check_ptrs({opt->params, opt->exp_avg, opt->exp_avg_sq});
        if (rc != RULGNN_OK) return rc;
        if (opt->bn_stats && (!args->bn_batch || (reinterpret_cast<uintptr_t>(opt->bn_stats) & 3))) return RULGNN_EINVAL;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    // (plain batch statistics: the running-statistics update rides in the step's finalize kernel)
    const bool tail_bn = opt && opt->bn_stats && args->training && args->bn_moment_weight == 0.f;
    // (host-side step count: the step's last kernel may apply the optimizer itself -- adam_device.hpp)
    AdamFuse fuse{};
    const bool try_fuse = opt && !opt->step_state;
    if (try_fuse)
        adam_fuse_args(&fuse, opt->params, opt->exp_avg, opt->exp_avg_sq, nullptr, opt->step, opt->lr, opt->beta1, opt->beta2, opt->eps,
                       opt->weight_decay);
    rc = astgcnn_run(shape, args, 3, st, nullptr, tail_bn ? opt->bn_stats : nullptr, tail_bn ? opt->bn_momentum : 0.f, try_fuse ? &fuse : nullptr);
    if (rc != RULGNN_OK || !opt) return rc;
    if (!(try_fuse && fuse.gbase))
        rc = adam_step(opt->params, args->grads, opt->exp_avg, opt->exp_avg_sq, astgcnn_param_count(shape), opt->step, opt->lr,
                       opt->beta1, opt->beta2, opt->eps, opt->weight_decay, 1.0f, st, opt->step_state);
    if (rc != RULGNN_OK || !opt->bn_stats || tail_bn) return rc;
    return astgcnn_bn_running_update(shape, opt->bn_stats, args->bn_batch, shape->batch * (int64_t)shape->time_length,
                                     opt->bn_momentum, args->bn_moment_weight > 0.f ? 1 : 0, st);
}

int rulgnn_astgcnn_fwdbwd_syncbn_f32(const rulgnn_astgcnn_shape* shape, const rulgnn_astgcnn_args* args, float bn_param_grad_scale,
                                     rulgnn_allreduce_f64_fn allreduce, void* user, void* stream) {
    const int rc = check_astgcnn(shape, args, true, true);
    if (rc != RULGNN_OK) return rc;
    if (args->dpred || !allreduce || !(bn_param_grad_scale >= 0.f && bn_param_grad_scale <= 1.f)) return RULGNN_EINVAL;
    const BnSyncHook hook = {allreduce, user, bn_param_grad_scale};
    return astgcnn_run(shape, args, 3, static_cast<hipStream_t>(stream), &hook);
}

int rulgnn_astgcnn_bn_running_update_f32(const rulgnn_astgcnn_shape* shape, float* bn_stats, const float* bn_batch, int64_t count,
                                         float momentum, int32_t from_moments, void* stream) {
    if (!shape || count < 1) return RULGNN_EINVAL;
    const int rc = check_ptrs({bn_stats, bn_batch});
    if (rc != RULGNN_OK) return rc;
    return astgcnn_bn_running_update(shape, bn_stats, bn_batch, count, momentum, from_moments, static_cast<hipStream_t>(stream));
}


// ---- device step state (hipGraph-capturable training steps) -----------------------------------------------
int rulgnn_step_state_set(void* step_state, uint64_t dropout_step, int64_t adam_step, void* stream) {
    if (!step_state || adam_step < 0) return RULGNN_EINVAL;
    if (reinterpret_cast<uintptr_t>(step_state) & 7) return RULGNN_EALIGN;
    return step_state_set(step_state, dropout_step, adam_step, static_cast<hipStream_t>(stream));
}

int rulgnn_adam_step_dev_f32(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, void* step_state,
                             float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale, void* stream) {
    if (n < 0 || !step_state) return RULGNN_EINVAL;
    const int rc = check_ptrs({params, grads, exp_avg, exp_avg_sq});
    if (rc != RULGNN_OK) return rc;
    return adam_step(params, grads, exp_avg, exp_avg_sq, n, 1, lr, beta1, beta2, eps, weight_decay, grad_scale,
                     static_cast<hipStream_t>(stream), step_state);
}


// ---- FC_STGNN -----------------------------------------------------------------------------------------
int64_t rulgnn_fcstgnn_param_count(const rulgnn_fcstgnn_shape* shape) { return fcstgnn_param_count(shape); }
int64_t rulgnn_fcstgnn_bn_count(const rulgnn_fcstgnn_shape* shape) { return fcstgnn_bn_count(shape); }
size_t rulgnn_fcstgnn_workspace_bytes(const rulgnn_fcstgnn_shape* shape) { return fcstgnn_workspace_bytes(shape); }

static int check_fcstgnn(const rulgnn_fcstgnn_shape* shape, const rulgnn_fcstgnn_args* a, bool forward, bool backward) {
    if (!shape || !a) return RULGNN_EINVAL;
    if (shape->batch < 1 || a->global_batch < shape->batch || a->sample_offset < 0) return RULGNN_EINVAL;
    if (!(a->bn_moment_weight >= 0.f) || !(a->dropout_p >= 0.f && a->dropout_p < 1.f)) return RULGNN_EINVAL;
    int rc = check_ptrs({a->x, a->params, a->pred, a->workspace});
    if (rc != RULGNN_OK) return rc;
    if (forward && !a->training) {
        rc = check_ptrs({a->bn_stats});
        if (rc != RULGNN_OK) return rc;
    }
    for (const void* p : {(const void*)a->y, (const void*)a->dpred, (const void*)a->bn_batch, (const void*)a->loss})
        if (p && (reinterpret_cast<uintptr_t>(p) & 3)) return RULGNN_EALIGN;
    if (backward) {
        if (!a->training) return RULGNN_EINVAL;
        rc = check_ptrs({a->grads});
        if (rc != RULGNN_OK) return rc;
        if (!a->dpred) {
            rc = check_ptrs({a->y, a->loss});
            if (rc != RULGNN_OK) return rc;
        }
    }
    return RULGNN_OK;
}

int rulgnn_fcstgnn_forward_f32(const rulgnn_fcstgnn_shape* shape, const rulgnn_fcstgnn_args* args, void* stream) {
    const int rc = check_fcstgnn(shape, args, true, false);
    if (rc != RULGNN_OK) return rc;
    return fcstgnn_run(shape, args, 1, static_cast<hipStream_t>(stream));
}

int rulgnn_fcstgnn_backward_f32(const rulgnn_fcstgnn_shape* shape, const rulgnn_fcstgnn_args* args, void* stream) {
    const int rc = check_fcstgnn(shape, args, false, true);
    if (rc != RULGNN_OK) return rc;
    return fcstgnn_run(shape, args, 2, static_cast<hipStream_t>(stream));
}

int rulgnn_fcstgnn_fwdbwd_f32(const rulgnn_fcstgnn_shape* shape, const rulgnn_fcstgnn_args* args, const rulgnn_adam_args* opt,
                              void* stream) {
    int rc = check_fcstgnn(shape, args, true, true);
    if (rc != RULGNN_OK) return rc;
    if (args->dpred) return RULGNN_EINVAL;
    if (opt) {
        if ((opt->step < 1 && !opt->step_state) || opt->params != args->params) return RULGNN_EINVAL;
        rc = check_ptrs({opt->params, opt->exp_avg, opt->exp_avg_sq});
        if (rc != RULGNN_OK) return rc;
        if (opt->bn_stats && (!args->bn_batch || (reinterpret_cast<uintptr_t>(opt->bn_stats) & 3))) return RULGNN_EINVAL;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool tail_bn = opt && opt->bn_stats && args->training && args->bn_moment_weight == 0.f;
    // (host-side step count: the step's last kernel applies the optimizer itself -- adam_device.hpp; a device step state keeps the launch)
    AdamFuse fuse{};
    fuse.p = nullptr;
    const bool fused_adam = opt && !opt->step_state && opt->step >= 1;
    if (fused_adam)
        adam_fuse_args(&fuse, opt->params, opt->exp_avg, opt->exp_avg_sq, args->grads, opt->step, opt->lr, opt->beta1, opt->beta2, opt->eps,
                       opt->weight_decay);
    rc = fcstgnn_run(shape, args, 3, st, nullptr, tail_bn ? opt->bn_stats : nullptr, tail_bn ? opt->bn_momentum : 0.f, fused_adam ? &fuse : nullptr);
    if (rc != RULGNN_OK || !opt) return rc;
    if (!fused_adam)
        rc = adam_step(opt->params, args->grads, opt->exp_avg, opt->exp_avg_sq, fcstgnn_param_count(shape), opt->step, opt->lr,
                       opt->beta1, opt->beta2, opt->eps, opt->weight_decay, 1.0f, st, opt->step_state);
    if (rc != RULGNN_OK || !opt->bn_stats || tail_bn) return rc;
    return fcstgnn_bn_running_update(shape, opt->bn_stats, args->bn_batch, opt->bn_momentum, args->bn_moment_weight > 0.f ? 1 : 0, st);
}

int rulgnn_fcstgnn_fwdbwd_syncbn_f32(const rulgnn_fcstgnn_shape* shape, const rulgnn_fcstgnn_args* args, float bn_param_grad_scale,
                                     rulgnn_allreduce_f64_fn allreduce, void* user, void* stream) {
    const int rc = check_fcstgnn(shape, args, true, true);
    if (rc != RULGNN_OK) return rc;
    if (args->dpred || !allreduce || !(bn_param_grad_scale >= 0.f && bn_param_grad_scale <= 1.f)) return RULGNN_EINVAL;
    const BnSyncHook hook = {allreduce, user, bn_param_grad_scale};
    return fcstgnn_run(shape, args, 3, static_cast<hipStream_t>(stream), &hook);
}

int rulgnn_fcstgnn_bn_running_update_f32(const rulgnn_fcstgnn_shape* shape, float* bn_stats, const float* bn_batch, float momentum,
                                         int32_t from_moments, void* stream) {
    if (!shape) return RULGNN_EINVAL;
    const int rc = check_ptrs({bn_stats, bn_batch});
    if (rc != RULGNN_OK) return rc;
    return fcstgnn_bn_running_update(shape, bn_stats, bn_batch, momentum, from_moments, static_cast<hipStream_t>(stream));
}


// ---- HAGCN graph stack ----------------------------------------------------------------------------------
int64_t rulgnn_hagcn_graph_param_count(const rulgnn_hagcn_shape* shape) { return hagcn_graph_param_count(shape); }
size_t rulgnn_hagcn_workspace_bytes(const rulgnn_hagcn_shape* shape) { return hagcn_workspace_bytes(shape); }

int rulgnn_hagcn_graph_forward_f32(const rulgnn_hagcn_shape* shape, const rulgnn_hagcn_args* a, void* stream) {
    if (!shape || !a || shape->graphs < 1) return RULGNN_EINVAL;
    const int rc = check_ptrs({a->nodes, a->params, a->feats, a->kl, a->workspace});
    if (rc != RULGNN_OK) return rc;
    for (const void* p : {(const void*)a->topk, (const void*)a->forced_topk})
        if (p && (reinterpret_cast<uintptr_t>(p) & 3)) return RULGNN_EALIGN;
    return hagcn_graph_forward(shape, a, static_cast<hipStream_t>(stream));
}

int rulgnn_hagcn_graph_backward_f32(const rulgnn_hagcn_shape* shape, const rulgnn_hagcn_args* a, void* stream) {
    if (!shape || !a || shape->graphs < 1) return RULGNN_EINVAL;
    const int rc = check_ptrs({a->params, a->dfeats, a->dkl, a->dnodes, a->grads, a->workspace});
    if (rc != RULGNN_OK) return rc;
    return hagcn_graph_backward(shape, a, static_cast<hipStream_t>(stream));
}


// ---- bidirectional LSTM layer (HAGCN encoder) ---------------------------------------------------------------
size_t rulgnn_bilstm_workspace_bytes(const rulgnn_bilstm_shape* shape) { return bilstm_workspace_bytes(shape); }

int rulgnn_bilstm_forward_f32(const rulgnn_bilstm_shape* shape, const rulgnn_bilstm_args* a, void* stream) {
    if (!shape || !a) return RULGNN_EINVAL;
    const int rc = check_ptrs({a->x, a->w_ih[0], a->w_ih[1], a->w_hh[0], a->w_hh[1], a->b_ih[0], a->b_ih[1], a->b_hh[0], a->b_hh[1],
                               a->out, a->workspace});
    if (rc != RULGNN_OK) return rc;
    return bilstm_forward(shape, a, static_cast<hipStream_t>(stream));
}

int rulgnn_bilstm_backward_f32(const rulgnn_bilstm_shape* shape, const rulgnn_bilstm_args* a, void* stream) {
    if (!shape || !a) return RULGNN_EINVAL;
    const int rc = check_ptrs({a->x, a->w_ih[0], a->w_ih[1], a->w_hh[0], a->w_hh[1], a->dout, a->dw_ih[0], a->dw_ih[1], a->dw_hh[0],
                               a->dw_hh[1], a->db_ih[0], a->db_ih[1], a->db_hh[0], a->db_hh[1], a->workspace});
    if (rc != RULGNN_OK) return rc;
    if (a->dx && (reinterpret_cast<uintptr_t>(a->dx) & 3)) return RULGNN_EALIGN;
    return bilstm_backward(shape, a, static_cast<hipStream_t>(stream));
}


// ---- ST_Conv ------------------------------------------------------------------------------------------
int64_t rulgnn_stconv_param_count(const rulgnn_stconv_shape* shape) { return stconv_param_count(shape); }
size_t rulgnn_stconv_workspace_bytes(const rulgnn_stconv_shape* shape) { return stconv_workspace_bytes(shape); }

static int check_stconv(const rulgnn_stconv_shape* shape, const rulgnn_astgcnn_args* a, bool forward, bool backward) {
    if (!shape) return RULGNN_EINVAL;
    rulgnn_astgcnn_shape probe{shape->batch, 1, 1, 1, 1};          // the argument checks do not depend on the model dimensions
    return check_astgcnn(&probe, a, forward, backward);
}

int rulgnn_stconv_forward_f32(const rulgnn_stconv_shape* shape, const rulgnn_astgcnn_args* args, void* stream) {
    const int rc = check_stconv(shape, args, true, false);
    if (rc != RULGNN_OK) return rc;
    return stconv_run(shape, args, 1, static_cast<hipStream_t>(stream));
}

int rulgnn_stconv_backward_f32(const rulgnn_stconv_shape* shape, const rulgnn_astgcnn_args* args, void* stream) {
    const int rc = check_stconv(shape, args, false, true);
    if (rc != RULGNN_OK) return rc;
    return stconv_run(shape, args, 2, static_cast<hipStream_t>(stream));
}

int rulgnn_stconv_fwdbwd_f32(const rulgnn_stconv_shape* shape, const rulgnn_astgcnn_args* args, const rulgnn_adam_args* opt,
                             void* stream) {
    int rc = check_stconv(shape, args, true, true);
    if (rc != RULGNN_OK) return rc;
    if (args->dpred) return RULGNN_EINVAL;
    if (opt) {
        if ((opt->step < 1 && !opt->step_state) || opt->params != args->params) return RULGNN_EINVAL;
        rc = check_ptrs({opt->params, opt->exp_avg, opt->exp_avg_sq});
        if (rc != RULGNN_OK) return rc;
        if (opt->bn_stats && (!args->bn_batch || (reinterpret_cast<uintptr_t>(opt->bn_stats) & 3))) return RULGNN_EINVAL;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    rc = stconv_run(shape, args, 3, st);
    if (rc != RULGNN_OK || !opt) return rc;
    rc = adam_step(opt->params, args->grads, opt->exp_avg, opt->exp_avg_sq, stconv_param_count(shape), opt->step, opt->lr, opt->beta1,
                   opt->beta2, opt->eps, opt->weight_decay, 1.0f, st, opt->step_state);
    if (rc != RULGNN_OK || !opt->bn_stats) return rc;
    return stconv_bn_running_update(shape, opt->bn_stats, args->bn_batch, shape->batch * (int64_t)shape->time_length, opt->bn_momentum,
                                    args->bn_moment_weight > 0.f ? 1 : 0, st);
}

int rulgnn_stconv_bn_running_update_f32(const rulgnn_stconv_shape* shape, float* bn_stats, const float* bn_batch, int64_t count,
                                        float momentum, int32_t from_moments, void* stream) {
    if (!shape || count < 1) return RULGNN_EINVAL;
    const int rc = check_ptrs({bn_stats, bn_batch});
    if (rc != RULGNN_OK) return rc;
    return stconv_bn_running_update(shape, bn_stats, bn_batch, count, momentum, from_moments, static_cast<hipStream_t>(stream));
}

size_t rulgnn_stgnn_workspace_bytes(const rulgnn_stgnn_shape* shape) { return stgnn_workspace_bytes(shape); }

int rulgnn_stgnn_terms_f32(const rulgnn_stgnn_shape* shape, const float* x, float* terms, float* adj, void* stream) {
    if (!shape) return RULGNN_EINVAL;
    if (shape->batch > 0) {
        const int rc = check_ptrs({x, terms});
        if (rc != RULGNN_OK) return rc;
    }
    return stgnn_terms(shape, x, terms, adj, static_cast<hipStream_t>(stream));
}

int rulgnn_stgnn_cheb_forward_f32(const rulgnn_stgnn_shape* shape, const float* terms, const float* filters, float* out,
                                  void* stream) {
    if (!shape) return RULGNN_EINVAL;
    if (shape->batch > 0) {
        const int rc = check_ptrs({terms, filters, out});
        if (rc != RULGNN_OK) return rc;
    }
    return stgnn_cheb_forward(shape, terms, filters, out, static_cast<hipStream_t>(stream));
}

int rulgnn_stgnn_cheb_backward_f32(const rulgnn_stgnn_shape* shape, const float* terms, const float* dout, float* dfilters,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    if (!shape) return RULGNN_EINVAL;
    int rc = check_ptrs({dfilters, workspace});
    if (rc != RULGNN_OK) return rc;
    if (shape->batch > 0) {
        rc = check_ptrs({terms, dout});
        if (rc != RULGNN_OK) return rc;
    }
    return stgnn_cheb_backward(shape, terms, dout, dfilters, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

int64_t rulgnn_stgnn_param_count(const rulgnn_stgnn_shape* shape) { return stgnn_param_count(shape); }
size_t rulgnn_stgnn_step_workspace_bytes(const rulgnn_stgnn_shape* shape) { return stgnn_step_workspace_bytes(shape); }

static int check_stgnn(const rulgnn_stgnn_shape* shape, const rulgnn_stmsgcn_args* a, bool fwd, bool bwd) {
    if (!shape || !a) return RULGNN_EINVAL;
    if (stgnn_param_count(shape) < 0) return stgnn_step_workspace_bytes(shape) == 0 && shape->batch >= 0 ? RULGNN_EUNSUPPORTED : RULGNN_EINVAL;
    int rc = check_ptrs({a->params, a->workspace});
    if (rc != RULGNN_OK) return rc;
    if (shape->batch > 0) {
        rc = check_ptrs({a->x});
        if (rc != RULGNN_OK) return rc;
        if (fwd && (rc = check_ptrs({a->pred})) != RULGNN_OK) return rc;
    }
    if (bwd) {
        if ((rc = check_ptrs({a->grads})) != RULGNN_OK) return rc;
        if (shape->batch > 0 && !a->dpred && !(fwd && a->y)) return RULGNN_EINVAL;     // needs d pred, or targets with a forward
    }
    return RULGNN_OK;
}

int rulgnn_stgnn_forward_f32(const rulgnn_stgnn_shape* shape, const rulgnn_stmsgcn_args* args, void* stream) {
    const int rc = check_stgnn(shape, args, true, false);
    if (rc != RULGNN_OK) return rc;
    return stgnn_run(shape, args, 1, static_cast<hipStream_t>(stream));
}

int rulgnn_stgnn_backward_f32(const rulgnn_stgnn_shape* shape, const rulgnn_stmsgcn_args* args, void* stream) {
    const int rc = check_stgnn(shape, args, false, true);
    if (rc != RULGNN_OK) return rc;
    return stgnn_run(shape, args, 2, static_cast<hipStream_t>(stream));
}

int rulgnn_stgnn_fwdbwd_f32(const rulgnn_stgnn_shape* shape, const rulgnn_stmsgcn_args* args, const rulgnn_adam_args* opt, void* stream) {
    int rc = check_stgnn(shape, args, true, true);
    if (rc != RULGNN_OK) return rc;
    if (args->dpred) return RULGNN_EINVAL;
    if (opt) {
        if ((opt->step < 1 && !opt->step_state) || opt->params != args->params) return RULGNN_EINVAL;
        rc = check_ptrs({opt->params, opt->exp_avg, opt->exp_avg_sq});
        if (rc != RULGNN_OK) return rc;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    rc = stgnn_run(shape, args, 3, st);
    if (rc != RULGNN_OK || !opt) return rc;
    return adam_step(opt->params, args->grads, opt->exp_avg, opt->exp_avg_sq, stgnn_param_count(shape), opt->step, opt->lr, opt->beta1,
                     opt->beta2, opt->eps, opt->weight_decay, 1.0f, st, opt->step_state);
}

// ---- STNet ----------------------------------------------------------------------------------------------------------------------
int64_t rulgnn_stnet_param_count(const rulgnn_stnet_shape* shape) { return stnet_param_count(shape); }
size_t rulgnn_stnet_workspace_bytes(const rulgnn_stnet_shape* shape) { return stnet_workspace_bytes(shape); }

static int check_stnet(const rulgnn_stnet_shape* shape, const rulgnn_stnet_args* a, bool bwd) {
    if (!shape || !a) return RULGNN_EINVAL;
    if (stnet_param_count(shape) < 0) return RULGNN_EUNSUPPORTED;
    int rc = check_ptrs({a->params, a->workspace});
    if (rc != RULGNN_OK) return rc;
    if (shape->batch > 0) {
        rc = check_ptrs({a->x, a->pred});
        if (rc != RULGNN_OK) return rc;
    }
    if (bwd) {
        rc = check_ptrs({a->grads});
        if (rc != RULGNN_OK) return rc;
        if (!a->dpred && !a->y && shape->batch > 0) return RULGNN_EINVAL;
    }
    return RULGNN_OK;
}

int rulgnn_stnet_forward_f32(const rulgnn_stnet_shape* shape, const rulgnn_stnet_args* args, void* stream) {
    const int rc = check_stnet(shape, args, false);
    if (rc != RULGNN_OK) return rc;
    return stnet_run(shape, args, 1, static_cast<hipStream_t>(stream));
}

int rulgnn_stnet_backward_f32(const rulgnn_stnet_shape* shape, const rulgnn_stnet_args* args, void* stream) {
    const int rc = check_stnet(shape, args, true);
    if (rc != RULGNN_OK) return rc;
    return stnet_run(shape, args, 2, static_cast<hipStream_t>(stream));
}

int rulgnn_stnet_fwdbwd_f32(const rulgnn_stnet_shape* shape, const rulgnn_stnet_args* args, const rulgnn_adam_args* opt, void* stream) {
    int rc = check_stnet(shape, args, true);
    if (rc != RULGNN_OK) return rc;
    if (args->dpred || (!args->y && shape->batch > 0)) return RULGNN_EINVAL;
    if (opt) {
        if ((opt->step < 1 && !opt->step_state) || opt->params != args->params) return RULGNN_EINVAL;
        rc = check_ptrs({opt->params, opt->exp_avg, opt->exp_avg_sq});
        if (rc != RULGNN_OK) return rc;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    rc = stnet_run(shape, args, 3, st);
    if (rc != RULGNN_OK || !opt) return rc;
    // cnn.{weight, bias} (the first 3 entries) have no gradient in the reference (`grad is None`): torch's Adam leaves them untouched
    return adam_step(opt->params + 3, args->grads + 3, opt->exp_avg + 3, opt->exp_avg_sq + 3, stnet_param_count(shape) - 3, opt->step,
                     opt->lr, opt->beta1, opt->beta2, opt->eps, opt->weight_decay, 1.0f, st, opt->step_state);
}

// ---- SAGCN ----------------------------------------------------------------------------------------------------------------------
int64_t rulgnn_sagcn_param_count(const rulgnn_sagcn_shape* shape) { return sagcn_param_count(shape); }
size_t rulgnn_sagcn_workspace_bytes(const rulgnn_sagcn_shape* shape) { return sagcn_workspace_bytes(shape); }
int64_t rulgnn_sagcn_tap_offset(const rulgnn_sagcn_shape* shape, int32_t which) { return sagcn_tap_offset(shape, which); }

static int check_sagcn(const rulgnn_sagcn_shape* shape, const rulgnn_sagcn_args* a, bool bwd) {
    if (!shape || !a) return RULGNN_EINVAL;
    if (sagcn_param_count(shape) < 0) return RULGNN_EUNSUPPORTED;
    int rc = check_ptrs({a->params, a->workspace});
    if (rc != RULGNN_OK) return rc;
    if (shape->batch > 0) {
        rc = check_ptrs({a->x, a->pred});
        if (rc != RULGNN_OK) return rc;
    }
    if (bwd) {
        rc = check_ptrs({a->grads});
        if (rc != RULGNN_OK) return rc;
        if (!a->dpred && !a->y && shape->batch > 0) return RULGNN_EINVAL;
    }
    return RULGNN_OK;
}

int rulgnn_sagcn_forward_f32(const rulgnn_sagcn_shape* shape, const rulgnn_sagcn_args* args, void* stream) {
    const int rc = check_sagcn(shape, args, false);
    if (rc != RULGNN_OK) return rc;
    return sagcn_run(shape, args, 1, static_cast<hipStream_t>(stream));
}

int rulgnn_sagcn_backward_f32(const rulgnn_sagcn_shape* shape, const rulgnn_sagcn_args* args, void* stream) {
    const int rc = check_sagcn(shape, args, true);
    if (rc != RULGNN_OK) return rc;
    return sagcn_run(shape, args, 2, static_cast<hipStream_t>(stream));
}

int rulgnn_sagcn_fwdbwd_f32(const rulgnn_sagcn_shape* shape, const rulgnn_sagcn_args* args, const rulgnn_adam_args* opt, void* stream) {
    int rc = check_sagcn(shape, args, true);
    if (rc != RULGNN_OK) return rc;
    if (args->dpred || (!args->y && shape->batch > 0)) return RULGNN_EINVAL;
    if (opt) {
        if ((opt->step < 1 && !opt->step_state) || opt->params != args->params) return RULGNN_EINVAL;
        rc = check_ptrs({opt->params, opt->exp_avg, opt->exp_avg_sq});
        if (rc != RULGNN_OK) return rc;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    rc = sagcn_run(shape, args, 3, st);
    if (rc != RULGNN_OK || !opt) return rc;
    return adam_step(opt->params, args->grads, opt->exp_avg, opt->exp_avg_sq, sagcn_param_count(shape), opt->step, opt->lr, opt->beta1,
                     opt->beta2, opt->eps, opt->weight_decay, 1.0f, st, opt->step_state);
}

int rulgnn_sgemm_mode(int32_t mode) {
    const int prev = sgemm_big_mode();
    if (mode == RULGNN_GEMM_F32 || mode == RULGNN_GEMM_BF16X3 || mode == RULGNN_GEMM_BF16X3_ONLY) sgemm_big_mode() = mode;
    return prev;
}

// ---- STAGNN ---------------------------------------------------------------------------------------------------------------------
int64_t rulgnn_stagnn_param_count(const rulgnn_stagnn_shape* shape) { return stagnn_param_count(shape); }
int64_t rulgnn_stagnn_bn_state_count(const rulgnn_stagnn_shape* shape) { return stagnn_bn_state_count(shape); }
size_t rulgnn_stagnn_workspace_bytes(const rulgnn_stagnn_shape* shape) { return stagnn_workspace_bytes(shape); }
int64_t rulgnn_stagnn_tap_offset(const rulgnn_stagnn_shape* shape, int32_t which) { return stagnn_tap_offset(shape, which); }

static int check_stagnn(const rulgnn_stagnn_shape* shape, const rulgnn_stagnn_args* a, bool bwd) {
    if (!shape || !a) return RULGNN_EINVAL;
    if (stagnn_param_count(shape) < 0) return RULGNN_EUNSUPPORTED;
    int rc = check_ptrs({a->params, a->workspace, a->bn_state});
    if (rc != RULGNN_OK) return rc;
    if (shape->batch > 0) {
        rc = check_ptrs({a->x, a->pred});
        if (rc != RULGNN_OK) return rc;
    }
    if (bwd) {
        rc = check_ptrs({a->grads});
        if (rc != RULGNN_OK) return rc;
        if (!a->training) return RULGNN_EINVAL;
        if (!a->dpred && !a->y && shape->batch > 0) return RULGNN_EINVAL;
    }
    return RULGNN_OK;
}

int rulgnn_stagnn_forward_f32(const rulgnn_stagnn_shape* shape, const rulgnn_stagnn_args* args, void* stream) {
    const int rc = check_stagnn(shape, args, false);
    if (rc != RULGNN_OK) return rc;
    return stagnn_run(shape, args, 1, static_cast<hipStream_t>(stream));
}

int rulgnn_stagnn_backward_f32(const rulgnn_stagnn_shape* shape, const rulgnn_stagnn_args* args, void* stream) {
    const int rc = check_stagnn(shape, args, true);
    if (rc != RULGNN_OK) return rc;
    return stagnn_run(shape, args, 2, static_cast<hipStream_t>(stream));
}

int rulgnn_stagnn_fwdbwd_f32(const rulgnn_stagnn_shape* shape, const rulgnn_stagnn_args* args, const rulgnn_adam_args* opt, void* stream) {
    int rc = check_stagnn(shape, args, true);
    if (rc != RULGNN_OK) return rc;
    if (args->dpred || (!args->y && shape->batch > 0)) return RULGNN_EINVAL;
    if (opt) {
        if ((opt->step < 1 && !opt->step_state) || opt->params != args->params) return RULGNN_EINVAL;
        rc = check_ptrs({opt->params, opt->exp_avg, opt->exp_avg_sq});
        if (rc != RULGNN_OK) return rc;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    rc = stagnn_run(shape, args, 3, st);
    if (rc != RULGNN_OK || !opt) return rc;
    return adam_step(opt->params, args->grads, opt->exp_avg, opt->exp_avg_sq, stagnn_param_count(shape), opt->step, opt->lr, opt->beta1,
                     opt->beta2, opt->eps, opt->weight_decay, 1.0f, st, opt->step_state);
}

// ---- plain GEMM -------------------------------------------------------------------------------------------------------------------
int rulgnn_sgemm_f32(const float* A, int64_t sAm, int64_t sAk, const float* B, int64_t sBn, int64_t sBk, float* C, int64_t ldc, int32_t M,
                     int32_t N, int32_t K, int32_t accumulate, void* stream) {
    if (M < 0 || N < 0 || K < 0 || ldc < N) return RULGNN_EINVAL;
    if (M == 0 || N == 0) return RULGNN_OK;
    if (!A || !B || !C) return RULGNN_EINVAL;
    return sgemm(A, sAm, sAk, B, sBn, sBk, C, ldc, M, N, K, accumulate != 0, static_cast<hipStream_t>(stream));
}

int rulgnn_absmax_partials_f32(const float* x, int64_t n, float* partials, int32_t nparts, void* stream) {
    if (n < 0 || nparts <= 0 || nparts > 65535 || !partials || (n > 0 && !x)) return RULGNN_EINVAL;
    return absmax_partials(x, n, partials, nparts, static_cast<hipStream_t>(stream));
}
int rulgnn_sgemm_scaled_f32(const float* A, int64_t sAm, int64_t sAk, const float* B, int64_t sBn, int64_t sBk, float* C, int64_t ldc, int32_t M,
                            int32_t N, int32_t K, int32_t accumulate, const float* amax_a, int32_t amax_na, const float* amax_b, int32_t amax_nb,
                            void* stream) {
    if (M < 0 || N < 0 || K < 0 || ldc < N) return RULGNN_EINVAL;
    if (M == 0 || N == 0) return RULGNN_OK;
    if (!A || !B || !C || !amax_a || !amax_b || amax_na <= 0 || amax_nb <= 0) return RULGNN_EINVAL;
    return sgemm(A, sAm, sAk, B, sBn, sBk, C, ldc, M, N, K, accumulate != 0, static_cast<hipStream_t>(stream), 0, amax_a, amax_na, amax_b, amax_nb);
}

static size_t scaled_plane_bytes(int32_t M, int32_t N, int32_t K) { return (sgemm_planes_ws_bytes(M, N, K) + 255) & ~(size_t)255; }
size_t rulgnn_sgemm_scaled_workspace_bytes(int32_t M, int32_t N, int32_t K, int32_t split_k) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    return scaled_plane_bytes(M, N, K) + (split_k ? sgemm_splitk_need_floats(M, N, K) * sizeof(float) : 0);
}
int rulgnn_sgemm_scaled_ws_f32(const float* A, int64_t sAm, int64_t sAk, const float* B, int64_t sBn, int64_t sBk, float* C, int64_t ldc, int32_t M,
                               int32_t N, int32_t K, int32_t accumulate, const float* amax_a, int32_t amax_na, const float* amax_b, int32_t amax_nb,
                               int32_t split_k, void* workspace, size_t workspace_bytes, int32_t* used_planes, void* stream) {
    if (M < 0 || N < 0 || K < 0 || ldc < N) return RULGNN_EINVAL;
    if (used_planes) *used_planes = 0;
    if (M == 0 || N == 0) return RULGNN_OK;
    if (!A || !B || !C || !amax_a || !amax_b || amax_na <= 0 || amax_nb <= 0 || !workspace) return RULGNN_EINVAL;
    if (workspace_bytes < rulgnn_sgemm_scaled_workspace_bytes(M, N, K, split_k)) return RULGNN_EWORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t pb = scaled_plane_bytes(M, N, K);
    if (used_planes) {
        const int s = sgemm_planes_slices(M, N, K, split_k != 0);
        const bool split_ok = !split_k || (s >= 2 && s <= sgemm_splitk_slices(M, N, K));
        *used_planes = (sgemm_big_mode() == 1 && s >= 1 && split_ok && sgemm_planes_ok(A, sAm, sAk, B, sBn, sBk, M, N, K, s)) ? 1 : 0;
    }
    if (split_k)
        return sgemm_splitk(A, sAm, sAk, B, sBn, sBk, C, ldc, M, N, K, accumulate != 0, reinterpret_cast<float*>(static_cast<unsigned char*>(workspace) + pb),
                            st, amax_a, amax_na, amax_b, amax_nb, workspace, pb);
    return sgemm(A, sAm, sAk, B, sBn, sBk, C, ldc, M, N, K, accumulate != 0, st, 0, amax_a, amax_na, amax_b, amax_nb, workspace, pb);
}

static size_t splitk_part_floats(int32_t M, int32_t N, int32_t K) {
    const size_t a = sgemm_splitk_need_floats(M, N, K), b = sgemm_splitk_need_floats(M, N + 1, K), c = sgemm_splitk_need_floats(M, 1, K);
    const size_t v = a > b ? a : b;
    return ((v > c ? v : c) + 63) & ~(size_t)63;
}
size_t rulgnn_sgemm_splitk_workspace_bytes(int32_t M, int32_t N, int32_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    return (splitk_part_floats(M, N, K) + (size_t)K) * sizeof(float);
}
__global__ void splitk_ones_kernel(float* p, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 1.f;
}
int rulgnn_sgemm_splitk_f32(const float* A, int64_t sAm, int64_t sAk, const float* B, int64_t sBn, int64_t sBk, float* C, int64_t ldc, int32_t M,
                            int32_t N, int32_t K, float* colsum, void* workspace, size_t workspace_bytes, void* stream) {
    if (M < 0 || N < 0 || K < 1 || ldc < N) return RULGNN_EINVAL;
    if (M == 0 || N == 0) return RULGNN_OK;
    if (!A || !B || !C || !workspace) return RULGNN_EINVAL;
    if (workspace_bytes < rulgnn_sgemm_splitk_workspace_bytes(M, N, K)) return RULGNN_EWORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(workspace);
    if (!colsum) return sgemm_splitk(A, sAm, sAk, B, sBn, sBk, C, ldc, M, N, K, false, part, st);
    float* ones = part + splitk_part_floats(M, N, K);
    (void)hipGetLastError();
    hipLaunchKernelGGL(splitk_ones_kernel, dim3(64), dim3(256), 0, st, ones, K);
    if (hipGetLastError() != hipSuccess) return RULGNN_EHIP;
    return sgemm_splitk_colsum(A, sAm, sAk, B, sBn, sBk, C, ldc, M, N, K, colsum, ones, part, st);
}

// ---- RGCNU ----------------------------------------------------------------------------------------------------------------------
int64_t rulgnn_rgcnu_param_count(const rulgnn_rgcnu_shape* shape) { return rgcnu_param_count(shape); }
size_t rulgnn_rgcnu_workspace_bytes(const rulgnn_rgcnu_shape* shape) { return rgcnu_workspace_bytes(shape); }

static int check_rgcnu(const rulgnn_rgcnu_shape* shape, const rulgnn_rgcnu_args* a, bool fwd, bool bwd) {
    if (!shape || !a) return RULGNN_EINVAL;
    if (rgcnu_param_count(shape) < 0) return rgcnu_workspace_bytes(shape) == 0 && shape->batch >= 0 && shape->num_nodes >= 1 &&
                                                     shape->time_length >= 1 && shape->hidden_dim >= 1 && shape->encoder_hidden_dim >= 1 &&
                                                     shape->kernel_size >= 1
                                                 ? RULGNN_EUNSUPPORTED
                                                 : RULGNN_EINVAL;
    if (!(a->dropout_p >= 0.f && a->dropout_p < 1.f)) return RULGNN_EINVAL;
    int rc = check_ptrs({a->params, a->workspace});
    if (rc != RULGNN_OK) return rc;
    if (shape->batch > 0) {
        rc = check_ptrs({a->x, a->pred});
        if (rc != RULGNN_OK) return rc;
    }
    if (bwd) {
        rc = check_ptrs({a->grads});
        if (rc != RULGNN_OK) return rc;
        if (!a->dpred && !a->y && shape->batch > 0) return RULGNN_EINVAL;
    }
    (void)fwd;
    return RULGNN_OK;
}

int rulgnn_rgcnu_forward_f32(const rulgnn_rgcnu_shape* shape, const rulgnn_rgcnu_args* args, void* stream) {
    const int rc = check_rgcnu(shape, args, true, false);
    if (rc != RULGNN_OK) return rc;
    return rgcnu_run(shape, args, 1, static_cast<hipStream_t>(stream));
}

int rulgnn_rgcnu_backward_f32(const rulgnn_rgcnu_shape* shape, const rulgnn_rgcnu_args* args, void* stream) {
    const int rc = check_rgcnu(shape, args, false, true);
    if (rc != RULGNN_OK) return rc;
    return rgcnu_run(shape, args, 2, static_cast<hipStream_t>(stream));
}

int rulgnn_rgcnu_fwdbwd_f32(const rulgnn_rgcnu_shape* shape, const rulgnn_rgcnu_args* args, const rulgnn_adam_args* opt, void* stream) {
    int rc = check_rgcnu(shape, args, true, true);
    if (rc != RULGNN_OK) return rc;
    if (args->dpred || (!args->y && shape->batch > 0)) return RULGNN_EINVAL;
    if (opt) {
        if ((opt->step < 1 && !opt->step_state) || opt->params != args->params) return RULGNN_EINVAL;
        rc = check_ptrs({opt->params, opt->exp_avg, opt->exp_avg_sq});
        if (rc != RULGNN_OK) return rc;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    rc = rgcnu_run(shape, args, 3, st);
    if (rc != RULGNN_OK || !opt) return rc;
    // the second head (fc2, the last E*L + 1 entries) has no gradient in the reference (`grad is None`): torch's Adam leaves it untouched
    const int64_t live = rgcnu_param_count(shape) - ((int64_t)shape->encoder_hidden_dim * shape->time_length + 1);
    return adam_step(opt->params, args->grads, opt->exp_avg, opt->exp_avg_sq, live, opt->step, opt->lr, opt->beta1,
                     opt->beta2, opt->eps, opt->weight_decay, 1.0f, st, opt->step_state);
}

size_t rulgnn_gru_workspace_bytes(const rulgnn_gru_shape* shape) { return gru_workspace_bytes(shape); }

int rulgnn_gru_forward_f32(const rulgnn_gru_shape* shape, const rulgnn_gru_args* args, void* stream) {
    if (!shape || !args) return RULGNN_EINVAL;
    int rc = check_ptrs({args->w_ih, args->w_hh, args->b_ih, args->b_hh, args->workspace});
    if (rc != RULGNN_OK) return rc;
    if (shape->num_seq > 0) {
        rc = check_ptrs({args->x, args->out});
        if (rc != RULGNN_OK) return rc;
    }
    return gru_forward(shape, args, static_cast<hipStream_t>(stream));
}

int rulgnn_gru_backward_f32(const rulgnn_gru_shape* shape, const rulgnn_gru_args* args, void* stream) {
    if (!shape || !args) return RULGNN_EINVAL;
    int rc = check_ptrs({args->w_ih, args->w_hh, args->b_ih, args->b_hh, args->workspace, args->dw_ih, args->dw_hh, args->db_ih,
                         args->db_hh});
    if (rc != RULGNN_OK) return rc;
    if (shape->num_seq > 0) {
        rc = check_ptrs({args->x, args->dout});
        if (rc != RULGNN_OK) return rc;
    }
    return gru_backward(shape, args, static_cast<hipStream_t>(stream));
}

size_t rulgnn_rul_metrics_workspace_bytes(int64_t n) { return rul_metrics_workspace_bytes(n); }

int rulgnn_rul_metrics_f32(const float* pred, const float* real, int64_t n, float max_rul, double* out, void* workspace,
                           size_t workspace_bytes, void* stream) {
    if (n < 1) return RULGNN_EINVAL;
    const int rc = check_ptrs({pred, real, out, workspace});
    if (rc != RULGNN_OK) return rc;
    return rul_metrics(pred, real, n, max_rul, out, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

int rulgnn_rul_metric_sums_f32(const float* pred, const float* real, int64_t n, float max_rul, double* out, void* workspace,
                               size_t workspace_bytes, void* stream) {
    if (n < 1) return RULGNN_EINVAL;
    const int rc = check_ptrs({pred, real, out, workspace});
    if (rc != RULGNN_OK) return rc;
    return rul_metrics(pred, real, n, max_rul, out, workspace, workspace_bytes, static_cast<hipStream_t>(stream), 1);
}

}  // extern "C"
