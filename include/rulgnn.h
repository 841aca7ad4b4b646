/* rulgnn.h -- C-ABI of the MI355X (gfx950) ST_GCN hot path.
 *
 * The reference (Frank-Wang-oss/GNN_RUL_Benchmarking) is pure Python/PyTorch: it has no native
 * boundary of its own.  The functions below are what a binding for its hot path would call
 * instead of the ATen op chains cited per entry point; INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer borrowed from the caller
 *     (e.g. torch `tensor.data_ptr()`), fp32 unless stated; nothing is allocated or freed here;
 *   - `stream` is the caller's hipStream_t passed as void* (NULL = default stream); all work is
 *     enqueued asynchronously on it, no host synchronisation, safe under stream capture;
 *   - return value: RULGNN_OK (0) or a negative RULGNN_E* code; rulgnn_strerror() names it;
 *   - no global mutable state: re-entrant across streams and devices.
 *
 * Flat parameter buffer ("live" parameters, the ones that receive gradients), N = num_patch,
 * per layer l (stride N*N + N + 440 floats):
 *     theta.weight[N][N] | theta.bias[N] | conv_block1.0.weight[10][10][2] | bn1.weight[10] |
 *     bn1.bias[10] | conv_block2.0.weight[10][10][2] | bn2.weight[10] | bn2.bias[10]
 * then fc1.weight[N][N] | fc1.bias[N] | fc2.weight[N] | fc2.bias[1].
 * Gradient, Adam-m and Adam-v buffers use the same layout.
 * BatchNorm buffer: [num_layers][2 (conv_block1, conv_block2)][2 (mean, var)][10].
 */
#ifndef RULGNN_H
#define RULGNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RULGNN_OK            0
#define RULGNN_EINVAL       -1   /* bad shape / null pointer / unsupported hyper-parameter */
#define RULGNN_EUNSUPPORTED -2   /* shape outside what the fused kernels cover (e.g. num_patch > 64) */
#define RULGNN_EWORKSPACE   -3   /* workspace too small (see *_workspace_bytes) */
#define RULGNN_EHIP         -4   /* a HIP runtime call failed */
#define RULGNN_EALIGN       -5   /* pointer not 4-byte aligned */
#define RULGNN_ECALLBACK    -6   /* a caller-supplied callback (rulgnn_allreduce_f64_fn) returned non-zero */

#define RULGNN_NUM_STATS 10      /* statistics per patch == graph nodes == TCN channels */

typedef struct rulgnn_stgcn_shape {
    int64_t batch;        /* samples in this call (this rank's shard) */
    int32_t num_patch;    /* N: patches per sample (C-MAPSS view: sensors), 2..4096 (fused kernels up to 64) */
    int32_t patch_size;   /* P: samples per patch (C-MAPSS view: window), 2..4096 */
    int32_t num_layers;   /* L: SG_TCN layers, 1..8 (reference default 2) */
    int32_t mpnn_k;       /* MPNN order k (Model.py:74-90): 1 (the reference's default and every hparams row) on every path; 2 and 3 on the
                           * row-mapped fp32 kernels (num_patch <= 64 and a shape those kernels hold -- not the tiled fallback; eval forward, the phase chain incl.
                           * synchronised BatchNorm);
                           * RULGNN_EUNSUPPORTED beyond, and for RULGNN_STEP_MX / RULGNN_STEP_COOP / RULGNN_EVAL_MX at k > 1 */
} rulgnn_stgcn_shape;

/* Library / ABI version: major*10000 + minor*100 + patch. */
int rulgnn_version(void);
const char *rulgnn_strerror(int code);

/* Number of floats in the flat live-parameter buffer for (num_patch, num_layers) at MPNN order 1. */
int64_t rulgnn_stgcn_param_count(int32_t num_patch, int32_t num_layers);
/* The same for MPNN order mpnn_k (models/ST_GCN/Model.py:74-79: theta is a ModuleList of k Linear(num_patch, num_patch)): per layer
 * theta.0.weight | theta.0.bias | ... | theta.<k-1>.weight | theta.<k-1>.bias, then the convolution / BatchNorm tensors as at k = 1 --
 * the reference's named_parameters() order without the dead net0 / net1 branches. */
int64_t rulgnn_stgcn_param_count_order(int32_t num_patch, int32_t num_layers, int32_t mpnn_k);

/* Eval-mode forward: replaces ST_GCN_model.forward under model.eval()/no_grad
 * (reference models/ST_GCN/Model.py:208-222, called from trainer.py:144).
 *   x        [batch, num_patch*patch_size]  input windows, contiguous
 *   params   flat live parameters (layout above)
 *   bn_stats BatchNorm running statistics (layout above)
 *   pred     [batch] output (the reference returns [batch, 1])
 *   workspace  rulgnn_stgcn_forward_workspace_bytes(shape) bytes of scratch: 0 (NULL allowed) for
 *              num_patch <= 64, activation tensors for the tiled path (num_patch > 64)
 * num_patch <= 64: ONE fused kernel: patch statistics -> Pearson adjacency -> L x (A.X.W, causal TCN,
 * folded BN, residuals) -> channel max-pool -> fc1 -> fc2.  num_patch > 64 (PHM2012 Condition_2, XJTU-SY):
 * tiled path, theta / fc1 as MFMA GEMMs over [batch][10][num_patch] tensors in the workspace. */
size_t rulgnn_stgcn_forward_workspace_bytes(const rulgnn_stgcn_shape *shape);
int rulgnn_stgcn_forward_f32(const rulgnn_stgcn_shape *shape, const float *x, const float *params,
                             const float *bn_stats, float *pred, void *workspace, size_t workspace_bytes,
                             void *stream);

/* Same call with the kernel chosen by the caller (the default entry above is RULGNN_EVAL_AUTO).  For num_patch <= 64 there
 * are two fused kernels:
 *   RULGNN_EVAL_EXACT  every contraction in fp32 (VALU / DPP FMAs, f32 MFMA for the Pearson matrix): any num_patch <= 64;
 *   RULGNN_EVAL_MX     the contractions of the layers on the f16 matrix cores with 2-way split operands
 *                      (hi + lo, fp32 accumulate: fp32-class accuracy, measured 5e-8 of sum|a.b| per product); requires
 *                      num_patch <= 47, num_layers <= 3, num_patch*patch_size a multiple of 4 and a 16-byte aligned x
 *                      (else RULGNN_EUNSUPPORTED).  num_patch <= 15 (the C-MAPSS shapes): four samples per wavefront;
 *                      16 <= num_patch <= 47 (PHM2012: 40 x 64, reference configs/hparams.py:238): one sample per
 *                      wavefront in 2-3 column tiles, when its LDS need fits (else RULGNN_EUNSUPPORTED).  Samples whose
 *                      arithmetic leaves the f16 range, or whose statistics are NaN / Inf, are recomputed by the EXACT
 *                      arithmetic -- inside the same launch (num_patch <= 15) or by a second, scanning launch enqueued
 *                      behind the first (16..47) -- so both paths agree on where NaN appears.
 *   RULGNN_EVAL_AUTO   MX when the shape qualifies, EXACT otherwise.
 * num_patch > 64 ignores `path` (tiled kernels). */
#define RULGNN_EVAL_AUTO  0
#define RULGNN_EVAL_EXACT 1
#define RULGNN_EVAL_MX    2
int rulgnn_stgcn_forward_path_f32(const rulgnn_stgcn_shape *shape, const float *x, const float *params,
                                  const float *bn_stats, float *pred, void *workspace, size_t workspace_bytes,
                                  int path, void *stream);

/* Verification aid for the MX kernel (tests/test_forward_mx_gpu.py): runs it on `shape` and additionally dumps the raw
 * registers of the FIRST 4-sample tile after every stage into `taps` (rulgnn_stgcn_forward_mx_tap_floats() floats,
 * [slot][64 lanes]; slot map in csrc/stgcn_forward_mx.hip), so that each layout conversion and each matrix-core product can
 * be checked against the oracle's intermediates separately.  Separate template instance: the production kernel carries
 * no tap code. */
int rulgnn_stgcn_forward_mx_tap_floats(void);
int rulgnn_stgcn_forward_mx_taps_f32(const rulgnn_stgcn_shape *shape, const float *x, const float *params,
                                     const float *bn_stats, float *pred, float *taps, void *stream);

/* Bytes of scratch the training entry points need for `shape` (features/adjacency cache,
 * per-block gradient partials, BatchNorm reduction cells).  Independent of pointer values. */
size_t rulgnn_stgcn_train_workspace_bytes(const rulgnn_stgcn_shape *shape);

typedef struct rulgnn_stgcn_train_args {
    const float *x;          /* [batch, N*P] */
    const float *y;          /* [batch] regression targets (RUL / max_rul); may be NULL if dpred given */
    const float *dpred;      /* optional [batch] upstream gradient d(loss)/d(pred); NULL -> MSE vs y */
    const float *params;     /* flat live parameters */
    float *grads;            /* out: flat gradient, same layout (overwritten, not accumulated) */
    float *pred;             /* out: [batch] train-mode predictions */
    float *loss;             /* out: 1 float, sum over this shard of (pred-y)^2 / global_batch */
    float *bn_batch;         /* out: batch statistics [L][2][2][10] = (mean, biased var) of this shard;
                                if bn_moment_weight > 0: w*(mean, mean of squares) instead */
    void *workspace;
    size_t workspace_bytes;
    int64_t global_batch;    /* MSE normaliser: batch summed over all data-parallel ranks (>= batch) */
    int64_t sample_offset;   /* index of this shard's first sample in the global batch (dropout stream) */
    float dropout_p;         /* 0 disables dropout */
    uint64_t seed;           /* dropout stream: (seed, step) -> per-layer keys */
    uint64_t step;
    float bn_moment_weight;  /* 0: plain statistics. w > 0 (data parallel, w = batch/global_batch): bn_batch holds
                                w*(E[z], E[z^2]) so that a SUM all-reduce over ranks yields the global-batch moments */
    void *step_state;        /* optional device step state (see rulgnn_step_state_set); NULL = use `step` above */
    uint32_t flags;          /* RULGNN_TRAIN_* bits, 0 = none */
    void *aux_stream;        /* optional second HIP stream of the caller (NULL: everything on `stream`).  The tiled path (num_patch > 64) runs its
                                parameter-gradient products (d fc1 / fc2, d theta of every layer, their bias sums) on it beside the
                                position-parallel backward chain and makes `stream` wait for it before the call's last kernel: the caller's
                                stream semantics do not change.  Ignored by the fused chains and when a gradient-ready callback is given.
                                (round 5; the struct grew: rulgnn_stgcn_train_args_size()) */
} rulgnn_stgcn_train_args;
/* rulgnn_stgcn_train_step*_f32 / _fwdbwd*_f32 on the matrix-core chain (RULGNN_STEP_MX, also through RULGNN_STEP_AUTO) end with a finalize
 * kernel that leaves the workspace's reduction cells ZERO.  A caller that sets this bit vouches that the previous call that used
 * args->workspace was such a step (same shape) and that nothing else has touched the workspace since: the step then runs without its
 * prepare launch (~5 us), the head-of-step scalars being written by its first phase.  Ignored where it does not apply (other chains,
 * a device step state, an empty shard).  Safety net: the first phase checks a token the finalize kernel left; without it the step ends
 * like one the f16 range guard rejected (NaN loss, parameters / optimizer state / running statistics untouched) and the workspace is
 * clean again afterwards. */
#define RULGNN_TRAIN_WS_CLEAN 1u

/* Train-mode forward only (BatchNorm batch statistics, dropout): fills pred, bn_batch and the
 * workspace cache.  Replaces model(X) under model.train() (algorithms/algorithms.py:482). */
int rulgnn_stgcn_train_forward_f32(const rulgnn_stgcn_shape *shape, const rulgnn_stgcn_train_args *args,
                                   void *stream);

/* Backward of the train-mode forward: gradient of sum(pred * dpred) (or of the MSE loss when
 * args->dpred is NULL) w.r.t. every live parameter.  Must follow rulgnn_stgcn_train_forward_f32
 * with the same args/workspace.  Replaces loss.backward() (algorithms/algorithms.py:488). */
int rulgnn_stgcn_train_backward_f32(const rulgnn_stgcn_shape *shape, const rulgnn_stgcn_train_args *args,
                                    void *stream);

/* Fused forward + MSE + backward: what ST_GCN.update does before optimizer.step()
 * (algorithms/algorithms.py:482-488), in one call with the loss kept on the device. */
int rulgnn_stgcn_train_fwdbwd_f32(const rulgnn_stgcn_shape *shape, const rulgnn_stgcn_train_args *args,
                                  void *stream);

/* Data parallel with SYNCHRONISED BatchNorm: rulgnn_stgcn_train_fwdbwd_f32 on this rank's shard, with every BatchNorm
 * normalising by the statistics of the GLOBAL batch -- the function the single-GPU step computes on the concatenated
 * batch (the reference has no data parallelism; this is what keeps an N-GPU run the same function as
 * algorithms/algorithms.py:481-490 on the whole batch).  After each of the 2L forward phases
 * ([sum z, sum z^2] of one BatchNorm) and each of the 2L backward phases that produce one ([sum dy, sum dy*xhat]) the
 * library calls `allreduce(user, buf, count, stream)` with a DEVICE pointer to `count` (= 20) contiguous doubles inside
 * args->workspace; the callback must enqueue (or perform) an in-place SUM over all ranks that is ordered after the
 * work already queued on `stream` and before anything queued on it later, and return 0.  4L callbacks per step, in the
 * same order on every rank.  Differences to the plain call: counts are global_batch * num_patch; args->bn_batch
 * receives the (mean, biased variance) of the GLOBAL batch (bn_moment_weight must be 0); the BatchNorm scale / shift
 * gradients, which come out of the all-reduced cells and are therefore already GLOBAL sums on every rank, are written to
 * args->grads multiplied by `bn_param_grad_scale` (in [0, 1]): pass 1 on exactly one rank and 0 on the others (or
 * 1/world_size everywhere when no shard is empty), so that a SUM of args->grads over the ranks is the gradient of the
 * global-batch loss for every parameter.  A rank with an empty shard makes no call; it must still take part in the 4L
 * all-reduces (with zeros).  num_patch <= 64 only (RULGNN_EUNSUPPORTED otherwise). */
typedef int (*rulgnn_allreduce_f64_fn)(void *user, double *device_buf, int32_t count, void *stream);
int rulgnn_stgcn_train_fwdbwd_syncbn_f32(const rulgnn_stgcn_shape *shape, const rulgnn_stgcn_train_args *args,
                                         float bn_param_grad_scale, rulgnn_allreduce_f64_fn allreduce, void *user,
                                         void *stream);
/* The same with the launch form chosen by the caller (RULGNN_STEP_AUTO / _CHAIN / _MX below; the entry above is RULGNN_STEP_CHAIN).
 * On the matrix-core chain the f16 range guard applies: see RULGNN_STEP_MX. */
int rulgnn_stgcn_train_fwdbwd_syncbn_path_f32(const rulgnn_stgcn_shape *shape, const rulgnn_stgcn_train_args *args,
                                              float bn_param_grad_scale, rulgnn_allreduce_f64_fn allreduce, void *user,
                                              int32_t path, void *stream);

/* Data parallel with a LARGE gradient bucket (num_patch > 64: theta and fc1 are num_patch x num_patch -- 12.6 MB at the reference's
 * XJTU-SY wiring, configs/hparams.py:349): rulgnn_stgcn_train_fwdbwd_f32 that reports gradient regions as they become final, so
 * that the caller can all-reduce them on another stream while the rest of the backward runs ("overlap with
 * backward").  `ready(user, grads, offset, count, stream)` is called from the launching thread right after the kernels that finalise
 * args->grads[offset, offset + count) were enqueued on `stream`; the callback must order whatever it starts after the work queued
 * on `stream` so far (record an event there and make its own stream wait for it) and must not touch the rest of args->grads.  Regions
 * are disjoint and reported in backward order: the head (fc1 | fc2.weight near the tail of the flat buffer; not fc2.bias, the last element) first, then theta (weight | bias) of
 * every layer but the first; everything that is not reported (the first layer, the convolution and BatchNorm gradients, the loss) is
 * final when the call returns and its work has drained, as with the plain entry.  Shapes of the fused kernels (num_patch <= 64:
 * buckets of a few KB) make no callback.  Returns RULGNN_ECALLBACK when `ready` returns non-zero. */
typedef int (*rulgnn_grad_ready_fn)(void *user, float *grads, int64_t offset, int64_t count, void *stream);
int rulgnn_stgcn_train_fwdbwd_ready_f32(const rulgnn_stgcn_shape *shape, const rulgnn_stgcn_train_args *args,
                                        rulgnn_grad_ready_fn ready, void *user, void *stream);

/* Single-GPU fast path: rulgnn_stgcn_train_fwdbwd_f32 with the optimizer folded into its last kernel --
 * the whole body of ST_GCN.update (algorithms/algorithms.py:482-489) in one call: the kernel that
 * reduces the per-block gradient partials applies torch.optim.Adam to each parameter as its gradient
 * becomes final, and updates the BatchNorm running statistics.  args->grads still receives the gradient. */
typedef struct rulgnn_adam_args {
    float *params;           /* flat live parameters, updated in place (same buffer as args->params) */
    float *exp_avg;          /* Adam first moment, flat */
    float *exp_avg_sq;       /* Adam second moment, flat */
    float *bn_stats;         /* BatchNorm running statistics, updated in place (may be NULL) */
    int64_t step;            /* 1-based step count after this update */
    float lr, beta1, beta2, eps, weight_decay;
    float bn_momentum;       /* 0.1 for nn.BatchNorm1d */
    void *step_state;        /* optional device step state; NULL = use `step` above */
} rulgnn_adam_args;
int rulgnn_stgcn_train_step_f32(const rulgnn_stgcn_shape *shape, const rulgnn_stgcn_train_args *args,
                                const rulgnn_adam_args *opt, void *stream);

/* The step with the launch form chosen by the caller (the entries above are RULGNN_STEP_AUTO); `opt` may be NULL (= forward +
 * backward only, rulgnn_stgcn_train_fwdbwd_f32).  For num_patch <= 64 there are two forms of the same fp32 arithmetic and, for the
 * C-MAPSS shapes, a matrix-core form:
 *   RULGNN_STEP_CHAIN  prepare | 2L forward phase kernels | head | 2L backward phase kernels | finalize: 4L + 3 launches; any batch;
 *   RULGNN_STEP_COOP   the same phase bodies inside ONE launch, the BatchNorm reductions behind device-side grid barriers -- for
 *                      batches whose every 4-sample tile (1-sample for num_patch > 16) gets its own resident wavefront: at most
 *                      4 x #CUs tiles (4096 samples on an MI355X at num_patch <= 16); RULGNN_EUNSUPPORTED otherwise.  Bit-identical
 *                      results to the chain (same per-workgroup partial sums, same finalize order).  Built for the reference
 *                      protocol's regime (batch_size 100, configs/hparams.py:223) and measured SLOWER there (97 vs 89 us per step:
 *                      a phase costs its prologue and one single-wavefront pass, not its launch), so it is an explicit option
 *                      only.  The workgroups spin on a counter in the workspace: do not share the device with a kernel that
 *                      waits on this stream.
 *   RULGNN_STEP_MX     the chain with every phase on the f16 matrix cores (2-way split operands, fp32 accumulation:
 *                      csrc/stgcn_train_mx.hip) and RECOMPUTATION instead of saved activations: a phase re-derives what it needs from
 *                      the layer input (X_l + the 55-entry adjacency), so only layer inputs and two gradient tensors cross HBM between
 *                      phases (13.5 KB per sample at 14 x 30 against 26.0 KB).  num_patch <= 15 and num_layers <= 3 (four samples per
 *                      wavefront), or 16 <= num_patch <= 47 and num_layers <= 2 (csrc/stgcn_train_mxw.hip: one sample per wavefront in
 *                      two or three column tiles -- PHM2012's 40 x 64); num_patch x patch_size a multiple of 4 (and at most 10240),
 *                      16-byte aligned x, MSE steps (args->y; not args->dpred): RULGNN_EUNSUPPORTED otherwise.  fp32-class
 *                      results (same gates as the fp32 chain), not bit-identical to it.  f16 RANGE GUARD: activations are not
 *                      rescaled; a value beyond the f16 range (inputs far from O(1): every dataset the reference wires is scaled to
 *                      [0, 1] or [-1, 1]) ends as Inf / NaN in a sum or a gradient row, the step's status word is raised and the
 *                      finalize kernel then leaves parameters, optimizer state and running statistics UNTOUCHED and reports a NaN
 *                      loss (and NaN-free untouched state): repeat that step with RULGNN_STEP_CHAIN.
 *                      A rejected step also increments a STICKY counter inside the workspace
 *                      (rulgnn_stgcn_train_guard_counter_offset) that no kernel ever clears: a caller that does not read the loss
 *                      back every step checks it at its own pace (once per epoch) and so cannot lose a step silently.
 *   RULGNN_STEP_MX_PERSIST  RULGNN_STEP_MX with F_1 .. G_0 as ONE launch behind F_0, the BatchNorm reductions behind arrival counters
 *                      instead of kernel boundaries (no fence: the fp64 cells are agent-scope atomics on both sides, a tile stays on
 *                      the wavefront that owns it) -- same arithmetic, partial rows and finalize as RULGNN_STEP_MX.  num_patch <= 15,
 *                      two layers, every workgroup of the phases' grid on a CU of its own (up to 16 x #CUs samples), not under
 *                      synchronised BatchNorm: RULGNN_EUNSUPPORTED otherwise.  Built for the reference protocol's batch (100) and
 *                      measured SLOWER there (85 us per step against 77 us for the ten launches: the in-kernel breakdown is in
 *                      profiles/r06_notes.md section 4), so it is an explicit option only.  Its workgroups wait for each other
 *                      (bounded: ~2^22 polls, then the step is rejected like a guard trip).
 *   RULGNN_STEP_AUTO   = RULGNN_STEP_MX where it applies, else RULGNN_STEP_CHAIN.
 * num_patch > 64 (tiled path) ignores `path`.
 *
 * CONTRACT (round 5): only THIS entry (explicit `path`), rulgnn_stgcn_train_step_f32 (a whole step: the library owns the optimizer
 * and honours the guard itself) and rulgnn_stgcn_train_fwdbwd_syncbn_path_f32 can run the matrix-core chain.  The split entries --
 * rulgnn_stgcn_train_fwdbwd_f32, _fwdbwd_ready_f32, _fwdbwd_syncbn_f32, after which the CALLER runs rulgnn_adam_step_f32 /
 * rulgnn_bn_running_update_f32 -- always run the fp32 phases, so the documented three-call flow never applies a rejected step's
 * gradients.  A caller that passes RULGNN_STEP_AUTO / _MX with opt == NULL must use the guarded optimizer entries
 * (rulgnn_adam_step_guarded_f32, rulgnn_bn_running_update_guarded_f32, rulgnn_adam_bn_step_f32 with guard = args->loss). */
#define RULGNN_STEP_AUTO  0
#define RULGNN_STEP_CHAIN 1
#define RULGNN_STEP_COOP  2
#define RULGNN_STEP_MX    3
#define RULGNN_STEP_MX_PERSIST 4
int rulgnn_stgcn_train_step_path_f32(const rulgnn_stgcn_shape *shape, const rulgnn_stgcn_train_args *args,
                                     const rulgnn_adam_args *opt, int32_t path, void *stream);
/* Which form a whole MSE step (args->y) with `path` runs for this shape and input pointer: RULGNN_STEP_CHAIN, RULGNN_STEP_COOP or
 * RULGNN_STEP_MX (also for RULGNN_STEP_MX_PERSIST: the same chain and guard protocol); RULGNN_EUNSUPPORTED for the tiled path (num_patch > 64) and for an explicit form the shape does not allow.  No launch. */
int rulgnn_stgcn_train_step_resolve(const rulgnn_stgcn_shape *shape, const float *x, int32_t path);
/* Byte offset, inside a training workspace of this shape, of a uint32 that counts the steps the f16 range guard rejected since the
 * caller last zeroed it (the library only ever adds to it; a caller that wants the count zeroes these four bytes when it allocates the
 * workspace).  -1: the shape runs on the tiled path (no matrix-core chain, no guard). */
int64_t rulgnn_stgcn_train_guard_counter_offset(const rulgnn_stgcn_shape *shape);
/* sizeof(rulgnn_stgcn_train_args) as this library was built: the struct has grown by trailing fields (`flags`, round 4); a caller
 * built against another header compares this with its own sizeof before the first call. */
size_t rulgnn_stgcn_train_args_size(void);
/* sizeof() of every struct of this header as the library was built, by index (0 for an unknown index).  Argument structs grow by
 * TRAILING fields (`flags`, `aux_stream`, ...) which the library reads unconditionally: a caller built against another header compares
 * these with its own sizeof before the first call (the Python binding does, at load time: gnn_rul_benchmarking_amd/_lib.py), and every
 * caller ZERO-INITIALISES an argument struct before filling it in (a zero / NULL trailing field always means "the behaviour before the
 * field existed"). */
#define RULGNN_STRUCT_STGCN_SHAPE 0
#define RULGNN_STRUCT_STGCN_TRAIN_ARGS 1
#define RULGNN_STRUCT_ADAM_ARGS 2
#define RULGNN_STRUCT_STMSGCN_SHAPE 3
#define RULGNN_STRUCT_STMSGCN_ARGS 4
#define RULGNN_STRUCT_ASTGCNN_SHAPE 5
#define RULGNN_STRUCT_ASTGCNN_ARGS 6
#define RULGNN_STRUCT_FCSTGNN_SHAPE 7
#define RULGNN_STRUCT_FCSTGNN_ARGS 8
#define RULGNN_STRUCT_RGCNU_SHAPE 9
#define RULGNN_STRUCT_RGCNU_ARGS 10
#define RULGNN_STRUCT_STNET_SHAPE 11
#define RULGNN_STRUCT_STNET_ARGS 12
#define RULGNN_STRUCT_SAGCN_SHAPE 13
#define RULGNN_STRUCT_SAGCN_ARGS 14
#define RULGNN_STRUCT_STAGNN_SHAPE 15
#define RULGNN_STRUCT_STAGNN_ARGS 16
#define RULGNN_STRUCT_HAGCN_SHAPE 17
#define RULGNN_STRUCT_HAGCN_ARGS 18
#define RULGNN_STRUCT_BILSTM_SHAPE 19
#define RULGNN_STRUCT_BILSTM_ARGS 20
#define RULGNN_STRUCT_STCONV_SHAPE 21
#define RULGNN_STRUCT_STGNN_SHAPE 22
#define RULGNN_STRUCT_GRU_SHAPE 23
#define RULGNN_STRUCT_GRU_ARGS 24
size_t rulgnn_struct_size(int32_t which);

/* Profiling aid: the training step is a chain of 4*num_layers+1 phase kernels:
 * phases 0..2L-1 = F_i (forward to BatchNorm i, batch statistics), 2L = TOP (prediction, loss, head
 * backward), 2L+1+j = G_{2L-1-j} (BatchNorm/conv/theta backward).  rulgnn_stgcn_train_phase_f32 launches
 * exactly ONE of them on `stream` with the state a previous full step left in the workspace, so a
 * harness can time a single kernel with HIP events.  The reduction cells are not cleared by a phase;
 * phase -1 launches the step's prepare kernel (clears them): a harness that runs -1, 0, 1, ... 4L in
 * order executes a numerically valid step, one launch at a time (with RULGNN_STEP_AUTO's choice of chain). */
int rulgnn_stgcn_train_phase_count(int32_t num_layers);
int rulgnn_stgcn_train_phase_f32(const rulgnn_stgcn_shape *shape, const rulgnn_stgcn_train_args *args,
                                 int32_t phase, void *stream);

/* torch.optim.Adam step over a flat buffer (L2 weight decay folded into the gradient, no amsgrad),
 * as configured at algorithms/algorithms.py:474-478.  `step` is the 1-based step count AFTER this
 * update.  grad_scale multiplies the gradient first (1/world_size after a sum all-reduce). */
int rulgnn_adam_step_f32(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n,
                         int64_t step, float lr, float beta1, float beta2, float eps, float weight_decay,
                         float grad_scale, void *stream);

/* The same step behind a guard: when *guard (a device float: the loss that follows the gradient in a data-parallel bucket) is not
 * finite, nothing is touched.  The matrix-core training chain (RULGNN_STEP_MX) reports an f16 range violation as a NaN loss; after the
 * bucket's all-reduce every rank then skips the optimizer step consistently and the caller repeats the step with RULGNN_STEP_CHAIN. */
int rulgnn_adam_step_guarded_f32(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n,
                                 int64_t step, float lr, float beta1, float beta2, float eps, float weight_decay,
                                 float grad_scale, const float *guard, void *stream);
int rulgnn_bn_running_update_guarded_f32(float *bn_stats, const float *bn_batch, int32_t num_layers, int64_t count,
                                         float momentum, int32_t from_moments, const float *guard, void *stream);

/* rulgnn_adam_step_f32 (or its guarded form, guard != NULL) and rulgnn_bn_running_update_f32 in ONE launch: the two kernels that follow
 * the bucket all-reduce of a data-parallel ST_GCN step (dp.py) -- same arithmetic per element, one launch latency less per step.
 * Replaces optimizer.step() + the running-statistics side effect of the training forward (algorithms/algorithms.py:474-478,
 * nn.BatchNorm1d in models/ST_GCN/Model.py). */
int rulgnn_adam_bn_step_f32(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n, int64_t step,
                            float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                            float *bn_stats, const float *bn_batch, int32_t num_layers, int64_t count, float momentum,
                            int32_t from_moments, const float *guard, void *stream);

/* nn.BatchNorm1d running-statistics update (momentum, unbiased running variance) from the batch
 * statistics produced by the training forward.  count = batch*num_patch values per channel.
 * from_moments != 0: bn_batch holds (E[z], E[z^2]) (the all-reduced form above) instead of (mean, var). */
int rulgnn_bn_running_update_f32(float *bn_stats, const float *bn_batch, int32_t num_layers, int64_t count,
                                 float momentum, int32_t from_moments, void *stream);

/* ------------------------------------------------------------------------------------------------
 * STMSGCN path (reference models/STMSGCN/Model.py, algorithms/algorithms.py:546-571).
 *
 * Per sample: num_patch patches of patch_size points -> per patch a graph of
 * nodes = (patch_size - interval) / band_width spectral bands (SED_features, Model.py:7-31) ->
 * stack of GCN layers with the adjacency x x^T of the current features (Model.py:34-49,96-100) ->
 * GRU over the patches per (sample, node) (Model.py:52-60) -> mean over nodes -> Linear.
 *
 * Flat parameter buffer (floats), dims = [1, gcn_dims...], C = sum(dims), H = gru_hidden:
 *     for each GCN layer l: linear.weight[dims[l+1]][dims[l]] | linear.bias[dims[l+1]]
 *     gru.weight_ih_l0[3H][C] | gru.weight_hh_l0[3H][H] | gru.bias_ih_l0[3H] | gru.bias_hh_l0[3H]
 *     fc.weight[num_patch*H] | fc.bias[1]
 * (6,818 floats for the reference's gcn_dims [16,64,16,1], H 8, num_patch 256.)
 */
#define RULGNN_STMSGCN_MAX_LAYERS 6

typedef struct rulgnn_stmsgcn_shape {
    int64_t batch;            /* samples in this call */
    int32_t num_patch;        /* patches per sample = GRU sequence length, 1..4096 */
    int32_t patch_size;       /* points per patch = DFT length, 2..512 */
    int32_t interval;         /* spectral-difference lag (Model.py:17-18) */
    int32_t band_width;       /* bins per band; (patch_size - interval) % band_width == 0; nodes <= 32 */
    int32_t num_gcn_layers;   /* 1..RULGNN_STMSGCN_MAX_LAYERS */
    int32_t gcn_dims[RULGNN_STMSGCN_MAX_LAYERS];   /* output width of each GCN layer, each 1..64, C <= 128 */
    int32_t gru_hidden;       /* H, 1..16 */
} rulgnn_stmsgcn_shape;

typedef struct rulgnn_stmsgcn_args {
    const float *x;           /* [batch, num_patch*patch_size] */
    const float *y;           /* [batch] targets; NULL when dpred is given / forward only */
    const float *dpred;       /* [batch] d loss / d pred (autograd backward); NULL = MSE against y */
    const float *params;      /* flat parameters (layout above) */
    float *grads;             /* flat gradient out (same layout) */
    float *pred;              /* [batch] */
    float *loss;              /* [1]: sum((pred - y)^2) / global_batch; may be NULL */
    void *workspace;          /* >= rulgnn_stmsgcn_workspace_bytes(shape) */
    size_t workspace_bytes;
    int64_t global_batch;     /* MSE denominator (data parallel: the global batch; else = batch) */
} rulgnn_stmsgcn_args;

int64_t rulgnn_stmsgcn_param_count(const rulgnn_stmsgcn_shape *shape);     /* < 0: invalid shape */
size_t rulgnn_stmsgcn_workspace_bytes(const rulgnn_stmsgcn_shape *shape);   /* 0: invalid / unsupported shape */

/* SED features + GCN stack only: features[batch*num_patch][nodes][C], the tensor the reference feeds to
 * its GRU before the transpose (Model.py:92-103).  No workspace. */
int rulgnn_stmsgcn_features_f32(const rulgnn_stmsgcn_shape *shape, const float *x, const float *params,
                                float *features, void *stream);

/* model(X): STMSGCN_model.forward (Model.py:84-112; trainer.py:144 in eval, algorithms.py:560 in training --
 * the model has no train/eval difference).  Uses x, params, pred, workspace; keeps in the workspace what
 * rulgnn_stmsgcn_backward_f32 needs. */
int rulgnn_stmsgcn_forward_f32(const rulgnn_stmsgcn_shape *shape, const rulgnn_stmsgcn_args *args, void *stream);

/* loss.backward() (algorithms.py:565): gradient of sum(pred*dpred), or of the MSE loss when dpred is NULL,
 * w.r.t. every parameter.  Must follow rulgnn_stmsgcn_forward_f32 with the same args/workspace. */
int rulgnn_stmsgcn_backward_f32(const rulgnn_stmsgcn_shape *shape, const rulgnn_stmsgcn_args *args, void *stream);

/* forward + MSE + backward in one call: STMSGCN.update up to optimizer.step() (algorithms.py:560-565).
 * With `opt` non-NULL also applies torch.optim.Adam (algorithms.py:566; opt->bn_stats is ignored). */
int rulgnn_stmsgcn_fwdbwd_f32(const rulgnn_stmsgcn_shape *shape, const rulgnn_stmsgcn_args *args,
                              const rulgnn_adam_args *opt, void *stream);

/* ------------------------------------------------------------------------------------------------
 * ASTGCNN path (reference models/ASTGCNN/Model.py, algorithms/algorithms.py:139-163): the model the reference
 * wires to C-MAPSS (14 sensors x 50 steps, configs/hparams.py:38) and N-CMAPSS (20 x 50, :202).
 *
 * x [batch, num_nodes, time_length] -> TCN over time with the nodes as channels (two causal Conv1d(k=6, dilation 1|2,
 * no bias) + BatchNorm1d + ReLU blocks with residuals, Model.py:72-146) -> tanh gate (Linear(T->E) + bias) times the TCN
 * output (:169-181; needs E == T) -> A = exp(-cdist(P x_i, P x_j)) (:184-195) -> Chebyshev graph convolution of order
 * K <= 3 with filters[K][E][O] (:198-230) -> mean over nodes -> Linear(O -> 1).
 *
 * Flat parameter buffer (floats), N = num_nodes, T = E = time_length, O = output_dim:
 *     conv_block1.0.weight[N][N][6] | bn1.weight[N] | bn1.bias[N] | conv_block2.0.weight[N][N][6] | bn2.weight[N] |
 *     bn2.bias[N] | gate.theta.weight[E][T] | gate.theta.bias[E] | gate.bias[E] | distance_module.P.weight[E][E] |
 *     chebnet.filters[K][E][O] | fc.weight[O] | fc.bias[1]
 * BatchNorm buffer: [2 (conv_block1, conv_block2)][2 (mean, var)][N].
 */
typedef struct rulgnn_astgcnn_shape {
    int64_t batch;
    int32_t num_nodes;        /* N, 1..25 (torch.cdist's exact path) */
    int32_t time_length;      /* T = encoder_out_dim, 1..64 */
    int32_t output_dim;       /* O, 1..256 */
    int32_t K;                /* Chebyshev order, 1..3 */
} rulgnn_astgcnn_shape;

typedef struct rulgnn_astgcnn_args {
    const float *x;           /* [batch, num_nodes*time_length] */
    const float *y;           /* [batch] targets, or NULL */
    const float *dpred;       /* [batch] d loss / d pred (autograd backward); NULL = MSE against y */
    const float *params;      /* flat parameters */
    float *grads;             /* flat gradient out */
    float *pred;              /* [batch] */
    float *loss;              /* [1]: sum((pred - y)^2) / global_batch; may be NULL */
    const float *bn_stats;    /* running statistics (read when training == 0) */
    float *bn_batch;          /* out, training: batch (mean, biased var) per BatchNorm/channel, or, with
                               * bn_moment_weight > 0, weight * (E[z], E[z^2]) (summable over ranks); may be NULL */
    void *workspace;
    size_t workspace_bytes;
    int64_t global_batch;     /* MSE denominator */
    float bn_moment_weight;
    int32_t training;         /* != 0: BatchNorm batch statistics (model.train()), else running statistics */
    void *aux_stream;         /* optional second HIP stream of the caller (see rulgnn_fcstgnn_args): the step's parameter-gradient products (one
                               * launch pair) then run beside the TCN backward.  Measured slower than NULL at the reference's shapes (the fork /
                               * join events cost more than the overlap returns): the Python model passes NULL. */
} rulgnn_astgcnn_args;

int64_t rulgnn_astgcnn_param_count(const rulgnn_astgcnn_shape *shape);      /* < 0: invalid / unsupported */
size_t rulgnn_astgcnn_workspace_bytes(const rulgnn_astgcnn_shape *shape);    /* 0: invalid / unsupported */

/* model(X): ASTGCNN_model.forward (Model.py:241-254) in eval (trainer.py:144) or train mode (algorithms.py:153). */
int rulgnn_astgcnn_forward_f32(const rulgnn_astgcnn_shape *shape, const rulgnn_astgcnn_args *args, void *stream);
/* loss.backward() (algorithms.py:159) after a train-mode forward with the same args/workspace. */
int rulgnn_astgcnn_backward_f32(const rulgnn_astgcnn_shape *shape, const rulgnn_astgcnn_args *args, void *stream);
/* ASTGCNN.update body (algorithms.py:153-161): train forward + MSE + backward; with `opt` also torch.optim.Adam and the
 * BatchNorm running-statistics update (opt->bn_stats, opt->bn_momentum). */
int rulgnn_astgcnn_fwdbwd_f32(const rulgnn_astgcnn_shape *shape, const rulgnn_astgcnn_args *args,
                              const rulgnn_adam_args *opt, void *stream);
/* nn.BatchNorm1d running-statistics update for this model's two BatchNorm layers (count = batch * time_length). */
/* rulgnn_astgcnn_fwdbwd_f32 on this rank's shard with both BatchNorm layers of the TemporalConvNet (Model.py:72-146) normalising by
 * the statistics of the GLOBAL batch, the contract of rulgnn_stgcn_train_fwdbwd_syncbn_f32: `allreduce` is
 * called 4 times per step (forward pairs of block 1, block 2, then backward pairs of block 2, block 1) with a device pointer to
 * 50 contiguous doubles inside args->workspace; counts are global_batch * time_length; args->bn_batch receives the global (mean,
 * biased variance) (bn_moment_weight must be 0); the BatchNorm scale / shift gradients are written multiplied by bn_param_grad_scale.
 * A rank with an empty shard makes no call and joins the 4 all-reduces with zeros.  training != 0, no dpred. */
int rulgnn_astgcnn_fwdbwd_syncbn_f32(const rulgnn_astgcnn_shape *shape, const rulgnn_astgcnn_args *args, float bn_param_grad_scale,
                                     rulgnn_allreduce_f64_fn allreduce, void *user, void *stream);
int rulgnn_astgcnn_bn_running_update_f32(const rulgnn_astgcnn_shape *shape, float *bn_stats, const float *bn_batch,
                                         int64_t count, float momentum, int32_t from_moments, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Device-resident step state, for capturing a whole training step in a hipGraph.
 *
 * A captured graph bakes its kernel arguments, so whatever changes from step to step (the dropout stream position, the
 * Adam step used for the bias corrections) must live in device memory.  The state is a caller-owned device buffer of
 * RULGNN_STEP_STATE_BYTES bytes.  When a call receives it (args->step_state / opt->step_state non-NULL) the host-side
 * `step` fields are ignored: a one-thread kernel at the head of the call advances the counter on the device and derives
 * the per-layer dropout keys / the Adam bias corrections, and the compute kernels read them from the buffer.  Every
 * such call therefore has identical arguments step after step and can be replayed from a graph.
 */
#define RULGNN_STEP_STATE_BYTES 64
/* (Re)initialise the counters: `dropout_step` training forwards and `adam_step` optimizer steps have happened so far. */
int rulgnn_step_state_set(void *step_state, uint64_t dropout_step, int64_t adam_step, void *stream);
/* rulgnn_adam_step_f32 with the step taken from (and advanced in) the device step state. */
int rulgnn_adam_step_dev_f32(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n,
                             void *step_state, float lr, float beta1, float beta2, float eps, float weight_decay,
                             float grad_scale, void *stream);

/* ------------------------------------------------------------------------------------------------
 * FC_STGNN path (reference models/FC_STGNN/Model.py + Model_Base.py, algorithms/algorithms.py:51-76): the model the
 * reference wires to C-MAPSS FD001-4 and N-CMAPSS with five different patchings (configs/hparams.py:32,69,109,149,196).
 *
 * x [batch, num_node, num_patch*patch_size] -> per (sample, patch, sensor) row: Conv1d(1->H1, k, pad k/2) + BN + ReLU ->
 * Conv1d(H1->CO, k, pad 1) + BN + ReLU -> Linear(CO*encoder_time_out -> 2*hidden) + BN -> + positional encoding, dropout
 * 0.1 (train) -> two window blocks (window 2, stride 1 and 2): per window a graph of 2*num_node nodes,
 * A = (softmax(leaky(M M^T - 1e8 I)) + I) * decay mask with M = Linear(X), X' = BN(X), leaky(BN(Linear(A X'))), mean
 * over the window -> concat -> MLP (2h, 2h, h, 1).
 *
 * Flat parameter buffer: the tensors in the order of the reference's named_parameters():
 *   conv1.weight[H1][1][k] | bn_a.weight | bn_a.bias | conv2.weight[CO][H1][k] | bn_b.{weight,bias} |
 *   nonlin_map2.0.{weight[2h][CO*TO], bias} | bn_c.{weight,bias} |
 *   for MPNN1, MPNN2: mapping.{weight[2h][2h], bias} | BN.{weight,bias} | theta.{weight[h][2h], bias} | bn1.{weight,bias} |
 *   fc1.{weight[2h][h*num_windows*num_node], bias} | fc2.{weight[2h][2h], bias} | fc3.{weight[h][2h], bias} | fc4.{weight[h], bias}
 * BatchNorm buffer: for the seven layers in that order: running_mean[C] | running_var[C].
 */
typedef struct rulgnn_fcstgnn_shape {
    int64_t batch;
    int32_t patch_size, num_patch, encoder_time_out, encoder_hidden_dim, encoder_out_dim, encoder_conv_kernel;
    int32_t hidden_dim, num_sequential, num_node, num_windows;   /* the reference's constructor arguments, Model.py:6-8 */
} rulgnn_fcstgnn_shape;

typedef struct rulgnn_fcstgnn_args {
    const float *x;           /* [batch, num_node * num_patch * patch_size] */
    const float *y;           /* [batch] targets, or NULL */
    const float *dpred;       /* [batch] d loss / d pred (autograd backward); NULL = MSE against y */
    const float *params;
    float *grads;
    float *pred;              /* [batch] */
    float *loss;              /* [1]; may be NULL */
    const float *bn_stats;    /* running statistics (read when training == 0) */
    float *bn_batch;          /* out, training: batch statistics in the BatchNorm-buffer layout (mean | biased var), or with
                               * bn_moment_weight > 0: weight * (E[z], E[z^2]); may be NULL */
    void *workspace;
    size_t workspace_bytes;
    int64_t global_batch;
    int64_t sample_offset;    /* index of this shard's first sample in the global batch (dropout stream) */
    float bn_moment_weight;
    float dropout_p;          /* positional-encoding dropout, 0.1 in the reference (Model.py:25); train mode only */
    uint64_t seed, step;      /* dropout stream: mask = f(seed, step, element index) */
    int32_t training;
    void *step_state;         /* optional device step state (rulgnn_step_state_set) */
    int32_t compute_dtype;    /* RULGNN_DTYPE_F32 (0, default): everything fp32.  RULGNN_DTYPE_BF16: the row projections that run as
                               * GEMM launches (Linear layers over the [rows, 8..24] activations, forward and data gradient) round
                               * both operands to bf16 and run on v_mfma_f32_16x16x32_bf16 with fp32 accumulation; the projections
                               * fused into other kernels (round 3: the encoder's last Linear, the MLP behind fc1, the theta Linear
                               * and the map products of the window blocks), BatchNorm statistics, the window graphs, the weight
                               * gradients, the loss and the optimizer stay fp32.  BASELINE.json config "FC_STGNN ... bf16":
                               * reported separately; tests/test_fcstgnn_gpu.py bounds its error (the parity claims are fp32) */
    void *aux_stream;         /* optional second HIP stream of the caller (NULL: everything on `stream`).  The backward then runs its
                               * weight / bias gradient GEMMs -- 17 latency-bound launches that nothing downstream waits for -- on it,
                               * forked behind events recorded on `stream` and joined before the call's last kernel, while the
                               * data-gradient chain continues on `stream`.  Same results (the GEMMs are the same launches). */
} rulgnn_fcstgnn_args;
#define RULGNN_DTYPE_F32  0
#define RULGNN_DTYPE_BF16 1

int64_t rulgnn_fcstgnn_param_count(const rulgnn_fcstgnn_shape *shape);     /* < 0: invalid / unsupported */
int64_t rulgnn_fcstgnn_bn_count(const rulgnn_fcstgnn_shape *shape);        /* floats in the BatchNorm buffer */
size_t rulgnn_fcstgnn_workspace_bytes(const rulgnn_fcstgnn_shape *shape);
/* model(X): FC_STGNN_RUL.forward (Model.py:46-84), eval (trainer.py:144) or train mode (algorithms.py:67). */
int rulgnn_fcstgnn_forward_f32(const rulgnn_fcstgnn_shape *shape, const rulgnn_fcstgnn_args *args, void *stream);
/* loss.backward() (algorithms.py:72) after a train-mode forward with the same args/workspace. */
int rulgnn_fcstgnn_backward_f32(const rulgnn_fcstgnn_shape *shape, const rulgnn_fcstgnn_args *args, void *stream);
/* FC_STGNN.update body (algorithms.py:67-74); with `opt` also Adam and the running statistics (opt->bn_stats). */
int rulgnn_fcstgnn_fwdbwd_f32(const rulgnn_fcstgnn_shape *shape, const rulgnn_fcstgnn_args *args,
                              const rulgnn_adam_args *opt, void *stream);
/* rulgnn_fcstgnn_fwdbwd_f32 on this rank's shard with all SEVEN BatchNorm layers (Model_Base.py:12-41,72-107,175-225; Model.py:19-22)
 * normalising by the statistics of the GLOBAL batch, the contract of rulgnn_stgcn_train_fwdbwd_syncbn_f32:
 * `allreduce` is called 14 times per step (forward pairs of layers 0,1,2,3,5,4,6, then backward pairs of 4,6,3,5,2,1,0, the same
 * order on every rank) with a device pointer to 128 contiguous doubles inside args->workspace; element counts are those of
 * args->global_batch; args->bn_batch receives the global (mean, biased variance) (bn_moment_weight must be 0); the BatchNorm scale /
 * shift gradients are written multiplied by bn_param_grad_scale.  A rank with an empty shard makes no call and joins the 14
 * all-reduces with zeros.  training != 0, no dpred. */
int rulgnn_fcstgnn_fwdbwd_syncbn_f32(const rulgnn_fcstgnn_shape *shape, const rulgnn_fcstgnn_args *args, float bn_param_grad_scale,
                                     rulgnn_allreduce_f64_fn allreduce, void *user, void *stream);
int rulgnn_fcstgnn_bn_running_update_f32(const rulgnn_fcstgnn_shape *shape, float *bn_stats, const float *bn_batch,
                                         float momentum, int32_t from_moments, void *stream);

/* ------------------------------------------------------------------------------------------------
 * RGCNU path (reference models/RGCNU/Model.py:96-119, algorithms/algorithms.py:250-296; SURVEY section 8f rank 3: a GCNLayer user
 * the reference wires to C-MAPSS FD001-4 and N-CMAPSS, configs/hparams.py:42,80,120,160,205).
 *
 * x [batch, num_nodes, time_length] -> learned adjacency A = relu(tanh(alpha (A1 A2^T - A2 A1^T))), A1/2 = tanh(alpha Linear(x))
 * (adj_construction :80-93) -> per (sample, time step) a graph of the num_nodes sensors with ONE feature, two GCN layers
 * (symmetric normalisation of A + I; 1 -> hidden -> hidden, ReLU), Dropout(0.5), 1x1 convolution back to one feature (SCL
 * :25-43) -> nn.LSTM(num_nodes -> encoder_hidden) over the time steps (TDL :46-53) -> 1x1 convolution of x + the LSTM output,
 * Conv1d(kernel_size, padding='same'), two Linear heads (FusionModule :56-78): pred [batch] and std [batch].  The loss of
 * RGCNU.update is the MSE of pred alone (:287-290): the std head (fc2) receives no gradient.
 * Quirk kept: the adjacency batch is tiled time_length times along the batch axis while the node signals are sample-major
 * (Model.py:104-106): graph (b, l) is convolved with the adjacency of sample (b * time_length + l) % batch.
 *
 * Flat parameter buffer in the order of the reference's named_parameters():
 *   adj.trainable_theta1.{weight[N][L], bias[N]} | adj.trainable_theta2.{...} | scl.gcn1.linear.{weight[H][1], bias[H]} |
 *   scl.gcn2.linear.{weight[H][H], bias[H]} | scl.conv1d.{weight[1][H][1], bias[1]} |
 *   tdl.lstm.{weight_ih_l0[4E][N], weight_hh_l0[4E][E], bias_ih_l0[4E], bias_hh_l0[4E]} |
 *   fusion.cnn1.{weight[E][N][1], bias[E]} | fusion.cnn2.{weight[E][E][k], bias[E]} | fusion.fc1.{weight[E*L], bias[1]} |
 *   fusion.fc2.{weight[E*L], bias[1]}
 * Limits: num_nodes <= 32, time_length <= 64, hidden_dim, encoder_hidden_dim <= 64, odd kernel_size <= 7, every tensor <= 4096
 * values (RULGNN_EUNSUPPORTED beyond; every wiring of the reference is 14|20 x 50, 32, 32, 3).
 */
typedef struct rulgnn_rgcnu_shape {
    int64_t batch;
    int32_t num_nodes, time_length, hidden_dim, encoder_hidden_dim, kernel_size;
    float alpha;
} rulgnn_rgcnu_shape;

typedef struct rulgnn_rgcnu_args {
    const float *x;           /* [batch, num_nodes * time_length] */
    const float *y;           /* [batch] targets, or NULL */
    const float *dpred;       /* [batch] d loss / d pred (autograd backward); NULL = MSE against y */
    const float *params;
    float *grads;             /* same layout as params; the fc2 entries are written as zeros */
    float *pred;              /* [batch] */
    float *std_pred;          /* [batch] second head (model(X, train=True)[1]); may be NULL */
    float *loss;              /* [1] sum over the shard of (pred - y)^2 / global_batch; may be NULL */
    void *workspace;
    size_t workspace_bytes;
    int64_t global_batch;
    int64_t sample_offset;    /* index of this shard's first sample in the global batch (dropout stream) */
    float dropout_p;          /* SCL's Dropout, 0.5 in the reference (Model.py:31); applied when training != 0 */
    uint64_t seed, step;      /* dropout stream: mask = f(seed, step, element index) */
    int32_t training;
} rulgnn_rgcnu_args;

int64_t rulgnn_rgcnu_param_count(const rulgnn_rgcnu_shape *shape);         /* < 0: invalid / unsupported */
size_t rulgnn_rgcnu_workspace_bytes(const rulgnn_rgcnu_shape *shape);
/* model(X) / model(X, train=True): RGCNU_model.forward (Model.py:104-119). */
int rulgnn_rgcnu_forward_f32(const rulgnn_rgcnu_shape *shape, const rulgnn_rgcnu_args *args, void *stream);
/* loss.backward() (algorithms.py:294) after a forward with the same args / workspace. */
int rulgnn_rgcnu_backward_f32(const rulgnn_rgcnu_shape *shape, const rulgnn_rgcnu_args *args, void *stream);
/* RGCNU.update body (algorithms.py:285-295); with `opt` also Adam on the flat parameter buffer. */
int rulgnn_rgcnu_fwdbwd_f32(const rulgnn_rgcnu_shape *shape, const rulgnn_rgcnu_args *args, const rulgnn_adam_args *opt, void *stream);

/* ------------------------------------------------------------------------------------------------
 * STNet path (reference models/STNet/Model.py:44-169, algorithms/algorithms.py:438-463; SURVEY section 8f rank 3: a ChebNet user the
 * reference wires to the bearing datasets, configs/hparams.py:236,267,303,347,382,416).
 *
 * x [batch, num_patch * patch_size] -> per (sample, patch): |STFT| with n_fft = hop = win = nperseg, periodic Hann window,
 * reflect-centred (torch.stft's defaults; Model.py:84-92) = num_nodes = nperseg/2 + 1 frequency nodes x input_dim = 1 +
 * patch_size/nperseg frames -> node weight = cnn(mean, max over the frames) (1x1 Conv2d, 2 -> 1) -> nodes above 0.7 are fully
 * connected (:104-110) -> num_cheb ChebNets, K = 3, no non-linearity in between (:114-115) -> Y_o [batch, num_patch,
 * num_nodes * C_last] -> encoder (Linear + ReLU x3, Linear) -> H [.., autoencoder_hidden] -> decoder (mirror) -> reconstruction
 * MSE(Y_o, decoder(H)) (:141) ; LSTM(autoencoder_hidden -> lstm_hidden) over the patches -> Linear(lstm_hidden * num_patch -> 1).
 * STNet.update: loss = MSE(pred, y) + reconstruction.  The threshold passes no gradient: cnn.{weight, bias} have grad None in the
 * reference (their entries of `grads` are written as zeros, and the optimizer must leave them alone -- they are the FIRST 3 floats).
 *
 * Flat parameter buffer in the order of the reference's named_parameters():
 *   cnn.weight[2] | cnn.bias[1] | chebnets.i.filters[3][C_i][C_i+1] ... | encoder.{0,2,4,6}.{weight[out][in], bias} |
 *   decoder.{0,2,4,6}.{weight, bias} | lstm.{weight_ih_l0[4E][A], weight_hh_l0[4E][E], bias_ih_l0[4E], bias_hh_l0[4E]} |
 *   linear.{weight[E * num_patch], bias[1]}
 * Limits: even nperseg <= 64 dividing patch_size, <= 64 frames, <= 4 ChebNets of <= 4096 channels, lstm_hidden <= 128,
 * autoencoder_hidden <= 1024 (RULGNN_EUNSUPPORTED beyond); num_nodes / input_dim must equal the STFT's shape (RULGNN_EINVAL).
 */
typedef struct rulgnn_stnet_shape {
    int64_t batch;
    int32_t num_patch, patch_size, num_nodes, nperseg, input_dim;
    int32_t num_cheb, cheb_layers[4];
    int32_t lstm_hidden_dim, autoencoder_hidden_dim;
} rulgnn_stnet_shape;

typedef struct rulgnn_stnet_args {
    const float *x;           /* [batch, num_patch * patch_size] */
    const float *y;           /* [batch] targets, or NULL */
    const float *dpred;       /* [batch] d loss / d pred (autograd backward); NULL = MSE against y.  The reconstruction term enters the
                               * backward with weight 1 (it is part of the reference's loss) unless recon_weight says otherwise */
    const float *params;
    float *grads;
    float *pred;              /* [batch] */
    float *recon;             /* [1] reconstruction MSE, averaged over the GLOBAL batch's elements (this shard's share); may be NULL */
    float *loss;              /* [1] this shard's share of MSE(pred, y) + reconstruction; may be NULL */
    void *workspace;
    size_t workspace_bytes;
    int64_t global_batch;
    const float *recon_weight; /* backward only: [1] device scalar d loss / d reconstruction (autograd: the reconstruction term may enter the
                               * objective with any weight, 0 when it is not used); NULL = 1, the reference's loss (algorithms.py:458) */
} rulgnn_stnet_args;

int64_t rulgnn_stnet_param_count(const rulgnn_stnet_shape *shape);         /* < 0: invalid / unsupported */
size_t rulgnn_stnet_workspace_bytes(const rulgnn_stnet_shape *shape);
int rulgnn_stnet_forward_f32(const rulgnn_stnet_shape *shape, const rulgnn_stnet_args *args, void *stream);
int rulgnn_stnet_backward_f32(const rulgnn_stnet_shape *shape, const rulgnn_stnet_args *args, void *stream);
/* STNet.update body (algorithms.py:455-462); with `opt` also Adam on the flat parameter buffer behind the cnn entries. */
int rulgnn_stnet_fwdbwd_f32(const rulgnn_stnet_shape *shape, const rulgnn_stnet_args *args, const rulgnn_adam_args *opt, void *stream);

/* ------------------------------------------------------------------------------------------------
 * SAGCN path (reference models/SAGCN/Model.py:6-156, algorithms/algorithms.py:412-436; the reference wires
 * it to the bearing datasets, configs/hparams.py:235,266,302,346,381,415).
 *
 * x [batch, num_patch * patch_size] -> per patch 12 temporal statistics (Model.py:17-34: max, min, std, rms, mean, peak-to-peak, var,
 * entropy of softmax, std of arcsin / arctan, kurtosis, skewness; std / var unbiased) + 8 spectral ones (:37-52: mean frequency,
 * frequency at the median rank of the sorted power, band power, occupied bandwidth, power bandwidth, max power, max amplitude, its
 * frequency; fs = 1) -> their cumulative form along the patches (:6-14) appended -> [num_patch, 40] scaled to unit Frobenius norm
 * (:66-69) -> cosine adjacency (:73-79) -> gcn1: ReLU(Linear_40->H(D^-1/2 (A + I) D^-1/2 X)) -> proj1, proj2: ReLU(Linear_H->H(
 * Linear_P->P over the node axis)) (:99-112) -> attention = softmax over the nodes of Linear_Ah->P(tanh(Linear_P->Ah(x^T))) (:115-125)
 * -> x * attention -> Linear(H * num_patch -> 1).  SAGCN.update: plain MSE.
 *
 * Tie rule of the two index-valued statistics (the spectrum of a real signal is mirrored exactly): the largest amplitude's bin is the
 * first one (torch.argmax); the bin at rank patch_size / 2 is taken from a STABLE ascending sort of the power (the reference's
 * torch.argsort is unstable: its CPU sort agrees for patch_size <= 16, beyond that its order among equal keys is unspecified).
 *
 * Flat parameter buffer in the order of the reference's named_parameters():
 *   gcn1.linear.{weight[H][40], bias[H]} | proj1.linear.{weight[H][H], bias[H]} | proj1.project_matrices.{weight[P][P], bias[P]} |
 *   proj2.(same) | attn.tanh_layer.{weight[Ah][P], bias[Ah]} | attn.softmax_layer.{weight[P][Ah], bias[P]} | fc.{weight[H * P], bias[1]}
 * Limits: num_patch <= 256, 2 <= patch_size <= 2048, hidden sizes <= 4096, batch * H * num_patch < 2^31 (RULGNN_EUNSUPPORTED beyond).
 */
typedef struct rulgnn_sagcn_shape {
    int64_t batch;
    int32_t num_patch, patch_size, gcn_hidden_dim, attention_hidden_dim;
} rulgnn_sagcn_shape;

typedef struct rulgnn_sagcn_args {
    const float *x;           /* [batch, num_patch * patch_size] */
    const float *y;           /* [batch] targets, or NULL */
    const float *dpred;       /* [batch] d loss / d pred (autograd backward); NULL = MSE against y */
    const float *params;
    float *grads;
    float *pred;              /* [batch] */
    float *loss;              /* [1] this shard's share of MSE(pred, y) over the GLOBAL batch; may be NULL */
    void *workspace;
    size_t workspace_bytes;
    int64_t global_batch;
} rulgnn_sagcn_args;

int64_t rulgnn_sagcn_param_count(const rulgnn_sagcn_shape *shape);         /* < 0: invalid / unsupported */
size_t rulgnn_sagcn_workspace_bytes(const rulgnn_sagcn_shape *shape);
/* workspace taps for the parity tests: float offsets of the normalised features [batch][P][40] and of A_hat X [P][batch][40] */
int64_t rulgnn_sagcn_tap_offset(const rulgnn_sagcn_shape *shape, int32_t which);   /* 0 features, 1 A_hat X, 2 h3 [P][batch][H], 3 attention */
int rulgnn_sagcn_forward_f32(const rulgnn_sagcn_shape *shape, const rulgnn_sagcn_args *args, void *stream);
int rulgnn_sagcn_backward_f32(const rulgnn_sagcn_shape *shape, const rulgnn_sagcn_args *args, void *stream);
/* SAGCN.update body (algorithms.py:427-435); with `opt` also Adam on the flat parameter buffer. */
int rulgnn_sagcn_fwdbwd_f32(const rulgnn_sagcn_shape *shape, const rulgnn_sagcn_args *args, const rulgnn_adam_args *opt, void *stream);

/* ------------------------------------------------------------------------------------------------
 * STAGNN path (reference models/STAGNN/Model.py:8-230, algorithms/algorithms.py:298-323; the reference wires
 * it to the aero-engine datasets, configs/hparams.py:43,82,122,162,206).
 *
 * x [batch, num_nodes, time_length] -> adj = (covariance of the sensor rows over the window, / (L - 1)) > threshold (Model.py:199-206;
 * forward-only) -> gcn1: leaky_relu_0.01(Linear_L->h(D^-1/2 (adj + I) D^-1/2 x)) -> gat1: mean over num_heads of
 * (softmax_j(leaky_relu_0.1(a . [Wh_i, Wh_j] + b)) * adj) Wh -- the mask is applied AFTER the softmax (:40-47) -> gcn2 (h -> h) -> gat2
 * -> [num_nodes, h] read as num_nodes channels of length h -> tcn1 (:85-159: Conv1d k=2 dilation 1, causal, no bias -> BatchNorm1d ->
 * ReLU; + downsample0 (1x1 Conv1d) of the input -> ReLU; Conv1d k=2 dilation 2 -> BatchNorm1d -> ReLU; + previous -> ReLU; channels
 * num_nodes -> h -> h) -> temporal_encoder1 (:163-180: per head softmax over the length of sigmoid(Linear over the channels); the mean
 * of the heads scales x) -> tcn2 (h -> output_dim -> output_dim) -> temporal_encoder2 -> Linear(output_dim * h -> 1).
 * The four BatchNorm1d layers use batch statistics when `training` (their backward is differentiated through those statistics) and
 * the running statistics otherwise.  The weight-normed `net0` / plain `net1` branches of TemporalConvNet are never called: dead
 * parameters, not part of the flat buffer.  STAGNN.update: plain MSE.
 *
 * Flat parameter buffer = the parameters that receive a gradient, in the order of the reference's named_parameters():
 *   gcn1.linear.{weight[h][L], bias[h]} | gat1.attention_i.{linear.weight[h][h], linear.bias[h], attention.weight[2h], attention.bias[1]}
 *   (i < num_heads) | gcn2.linear.{weight[h][h], bias} | gat2.(same) | tcn1.{downsample0.weight[h][N], downsample0.bias[h],
 *   conv_block1.0.weight[h][N][2], conv_block1.2.{weight, bias}[h], conv_block2.0.weight[h][h][2], conv_block2.2.{weight, bias}[h]} |
 *   temporal_encoder1.linears.i.{weight[h], bias[1]} | tcn2.(same with N -> h, h -> output_dim) | temporal_encoder2.linears.i.{weight[out],
 *   bias[1]} | fc.{weight[out * h], bias[1]}
 * bn_state: [running_mean | running_var] of tcn1.conv_block1.2, tcn1.conv_block2.2, tcn2.conv_block1.2, tcn2.conv_block2.2.
 * Limits: num_nodes <= 32, time_length <= 128, 3 <= hidden_dim <= 64, output_dim <= 16, num_heads <= 4, num_nodes != hidden_dim !=
 * output_dim (the residual 1x1 convolutions exist) -- RULGNN_EUNSUPPORTED beyond; every wiring of the reference is 14|20 x 50, 16|32|64,
 * 10, 3.
 */
typedef struct rulgnn_stagnn_shape {
    int64_t batch;
    int32_t num_nodes, time_length, hidden_dim, output_dim, num_heads;
    float threshold;
} rulgnn_stagnn_shape;

typedef struct rulgnn_stagnn_args {
    const float *x;           /* [batch, num_nodes, time_length] */
    const float *y;           /* [batch] targets, or NULL */
    const float *dpred;       /* [batch] d loss / d pred (autograd backward); NULL = MSE against y */
    const float *params;
    float *grads;
    float *pred;              /* [batch] */
    float *loss;              /* [1] this shard's share of MSE(pred, y) over the GLOBAL batch; may be NULL */
    float *bn_state;          /* running statistics, see above; read in eval mode, updated by a training forward when asked to */
    void *workspace;
    size_t workspace_bytes;
    int64_t global_batch;
    int32_t training;              /* != 0: batch statistics (model.train()) */
    int32_t update_running_stats;  /* != 0 with training: momentum-0.1 update of bn_state, unbiased variance (BatchNorm1d's defaults) */
} rulgnn_stagnn_args;

int64_t rulgnn_stagnn_param_count(const rulgnn_stagnn_shape *shape);         /* < 0: invalid / unsupported */
int64_t rulgnn_stagnn_bn_state_count(const rulgnn_stagnn_shape *shape);
size_t rulgnn_stagnn_workspace_bytes(const rulgnn_stagnn_shape *shape);
/* workspace taps for the parity tests (float offsets): 0 adjacency [batch][N][N], 1 graph output [batch][N][h], 2 tcn1 output, 3
 * temporal_encoder1 output [batch][h][h], 4 tcn2 output, 5 temporal_encoder2 output [batch][out][h] */
int64_t rulgnn_stagnn_tap_offset(const rulgnn_stagnn_shape *shape, int32_t which);
int rulgnn_stagnn_forward_f32(const rulgnn_stagnn_shape *shape, const rulgnn_stagnn_args *args, void *stream);
/* loss.backward() after a TRAINING forward with the same args / workspace (RULGNN_EINVAL when args->training == 0). */
int rulgnn_stagnn_backward_f32(const rulgnn_stagnn_shape *shape, const rulgnn_stagnn_args *args, void *stream);
/* STAGNN.update body (algorithms.py:314-322); with `opt` also Adam on the flat parameter buffer. */
int rulgnn_stagnn_fwdbwd_f32(const rulgnn_stagnn_shape *shape, const rulgnn_stagnn_args *args, const rulgnn_adam_args *opt, void *stream);

/* ------------------------------------------------------------------------------------------------
 * One-shot all-reduce of up to RULGNN_PEER_MAX_COUNT doubles over peer-mapped mailboxes (round 6; csrc/peer_comm.hip): the
 * synchronised-BatchNorm collectives of a data-parallel step (sum x, sum x^2 / sum dy, sum dy xhat per BatchNorm
 * layer; the reference itself is single-process) as ONE single-workgroup launch each on the compute stream, no host work in between.
 * rulgnn_peer_allreduce_f64 has the signature of rulgnn_allreduce_f64_fn: pass its address and the communicator as the
 * (allreduce, user) pair of rulgnn_stgcn_train_fwdbwd_syncbn[_path]_f32 / rulgnn_fcstgnn_fwdbwd_syncbn_f32 /
 * rulgnn_astgcnn_fwdbwd_syncbn_f32.  The result is the sum in RANK ORDER on every rank (bit-identical across ranks).
 *
 * Set-up (once per process; these calls allocate -- the only entries of this header that do): every rank allocates a mailbox
 * (fine-grained device memory) and gets its IPC handle (rulgnn_peer_handle_bytes() bytes: hipIpcMemHandle_t), the ranks exchange the
 * handles by whatever means they have (torch.distributed.all_gather_object in gnn_rul_benchmarking_amd/dp.py), open each other's
 * mailboxes and build a communicator from the world's pointers (mailboxes[rank] = the own one).  Every rank must issue the same
 * sequence of collectives.  A collective whose peers do not arrive within the communicator's timeout (20 s unless
 * rulgnn_peer_comm_set_timeout_ms set another, 1 ms .. 10 min) leaves NaN in the buffer and a sticky error in the
 * mailbox (rulgnn_peer_comm_status) instead of spinning forever.  Requires peer access between the devices (one node, xGMI or PCIe
 * P2P) and HSA_ENABLE_IPC_MODE_LEGACY=0 on this image (dmabuf IPC). */
#define RULGNN_PEER_MAX_COUNT 128
size_t rulgnn_peer_mailbox_bytes(void);
size_t rulgnn_peer_handle_bytes(void);
int rulgnn_peer_mailbox_alloc(void **mailbox, void *handle_out);
int rulgnn_peer_mailbox_open(const void *handle, void **mailbox);
int rulgnn_peer_mailbox_close(void *mailbox);
int rulgnn_peer_mailbox_free(void *mailbox);
void *rulgnn_peer_comm_create(int32_t rank, int32_t world, void *const *mailboxes);
void rulgnn_peer_comm_destroy(void *comm);
int rulgnn_peer_comm_set_timeout_ms(void *comm, int64_t milliseconds);
int rulgnn_peer_allreduce_f64(void *comm, double *device_buf, int32_t count, void *stream);
int64_t rulgnn_peer_comm_collectives(void *comm);
int64_t rulgnn_peer_comm_status(void *comm);

/* ------------------------------------------------------------------------------------------------
 * The fp32 matrix product behind every nn.Linear / torch.matmul / torch.bmm of the reference models that runs as a GEMM here
 * (e.g. models/SAGCN/Model.py:107-108, models/STNet/Model.py:31-38, models/ST_GCN/Model.py:88): C[m][n] (+)= sum_k A(m,k) B(n,k) with
 * element strides (A(m,k) = A[m * sAm + k * sAk], B(n,k) = B[n * sBn + k * sBk], C row stride ldc), exact fp32 on the matrix cores
 * (v_mfma_f32_16x16x4_f32), fixed summation order.  Exposed for the parity tests and for timing the kernel alone.
 */
int rulgnn_sgemm_f32(const float *A, int64_t sAm, int64_t sAk, const float *B, int64_t sBn, int64_t sBk, float *C, int64_t ldc,
                     int32_t M, int32_t N, int32_t K, int32_t accumulate, void *stream);
/* The same product with per-operand scales (round 5): `amax_a` / `amax_b` = amax_na / amax_nb floats each whose maximum is max |A| / max |B|
 * over the FINITE elements (rulgnn_absmax_partials_f32 below, or the partial maxima a producing kernel left).  Outputs that fill 256 x 256 tiles
 * (M, N > 192 and enough tiles for the chip) then run the two-plane f16 split: every operand is scaled by a power of two so that its largest
 * element lands in [2^11, 2^12), split exactly into hi (11 significant bits) + lo (f16 of the rest), three f16 matrix instructions per
 * product block with fp32 accumulation, the result unscaled exactly -- 22 significant bits per operand, half the matrix work of
 * RULGNN_GEMM_BF16X3; an element more than 2^15 below its tensor's largest keeps fewer bits (its lo part goes subnormal: absolute error
 * 2^-37 of the largest -- invisible in a sum that also contains large elements, ~4e-5 relative in an output only such elements touch:
 * the reason this form is opt-in).  Other shapes
 * and RULGNN_GEMM_F32 ignore the scales.  This is the large contraction of the reference's ST_GCN wirings with num_patch > 64
 * (theta(A.X), models/ST_GCN/Model.py:88, configs/hparams.py:349: [batch * 10, 1024] x [1024, 1024]) and its two gradients. */
int rulgnn_sgemm_scaled_f32(const float *A, int64_t sAm, int64_t sAk, const float *B, int64_t sBn, int64_t sBk, float *C, int64_t ldc,
                            int32_t M, int32_t N, int32_t K, int32_t accumulate, const float *amax_a, int32_t amax_na, const float *amax_b,
                            int32_t amax_nb, void *stream);
/* The scaled product on PRE-SPLIT operands (round 6; csrc/sgemm_planes.hip): with `workspace` (rulgnn_sgemm_scaled_workspace_bytes) a
 * split pass writes each operand ONCE as two k-contiguous f16 planes (hi | lo, scaled as above; a transposing pass when the operand is
 * row-contiguous) and the product kernel copies operand tiles HBM -> LDS by DMA (global_load_lds_dwordx4: no VGPR round trip, no split in
 * the product loop) into a three-stage ring, one workgroup barrier per 32 k, 160 x 256 or 128 x 256 output tiles (one per CU for
 * [10 240 x 1024] . [1024 x 1024]).  Same arithmetic and error class as rulgnn_sgemm_scaled_f32.  Shapes: M a multiple of 160 or 128,
 * N of 256, K of 32 (split_k: of 32 x the slice count), 16-byte aligned operands with unit stride along k or along the row index;
 * other shapes (and RULGNN_GEMM_F32 / _BF16X3_ONLY) run exactly what rulgnn_sgemm_scaled_f32 / rulgnn_sgemm_splitk_f32 run.
 * split_k != 0: the product as a deterministic split-K reduction (the weight gradient d theta = dH^T (A.X): K = batch x 10 rows);
 * the workspace then also holds the partial slices.  Returns in *used_planes (may be NULL) whether the pre-split kernel ran. */
size_t rulgnn_sgemm_scaled_workspace_bytes(int32_t M, int32_t N, int32_t K, int32_t split_k);
int rulgnn_sgemm_scaled_ws_f32(const float *A, int64_t sAm, int64_t sAk, const float *B, int64_t sBn, int64_t sBk, float *C, int64_t ldc,
                               int32_t M, int32_t N, int32_t K, int32_t accumulate, const float *amax_a, int32_t amax_na, const float *amax_b,
                               int32_t amax_nb, int32_t split_k, void *workspace, size_t workspace_bytes, int32_t *used_planes, void *stream);
/* partials[i] = max |x[e]| over the finite elements e = i * 256 + t (mod nparts * 256 strides) of x[0..n): nparts (1..65535) partial maxima. */
int rulgnn_absmax_partials_f32(const float *x, int64_t n, float *partials, int32_t nparts, void *stream);
/* The same product as a deterministic split-K reduction -- the weight gradients of the families: C[m][n] = sum over the K rows of the
 * batch of dY(k, m) X(k, n), e.g. nn.Linear's weight.grad (models/FC_STGNN/Model_Base.py:44-107, models/HAGCN/Model.py:26-73) -- and,
 * with `colsum` != NULL, colsum[m] = sum_k A(m, k) from the same pass (the bias gradient over the same rows).  `workspace`: at least
 * rulgnn_sgemm_splitk_workspace_bytes(M, N, K) bytes (partial products and, for the column sums, K ones).  Small outputs over long
 * reductions with row-major operands (sAm = sBn = 1, M <= 32, N <= 64) run one wavefront per k-range on the fp32 matrix cores straight from
 * global memory; other shapes take the tile kernels.  Fixed summation order.  Exposed for the parity tests. */
size_t rulgnn_sgemm_splitk_workspace_bytes(int32_t M, int32_t N, int32_t K);
int rulgnn_sgemm_splitk_f32(const float *A, int64_t sAm, int64_t sAk, const float *B, int64_t sBn, int64_t sBk, float *C, int64_t ldc,
                            int32_t M, int32_t N, int32_t K, float *colsum, void *workspace, size_t workspace_bytes, void *stream);
/* Arithmetic of the large-tile GEMM (outputs of at least ~100 x 100 with enough tiles to fill the chip), process-wide:
 * RULGNN_GEMM_BF16X3 (default): every fp32 operand split exactly into three bf16 parts, six bf16 matrix instructions per product
 * block with fp32 accumulation -- the error per product is that of one fp32 rounding (terms below 2^-23 relative are dropped), at up
 * to 2.6x the fp32 matrix peak; RULGNN_GEMM_F32: fp32 matrix instructions (v_mfma_f32_16x16x4_f32), bit-compatible with the small-tile
 * kernel.  Returns the previous mode; any other value only queries. */
#define RULGNN_GEMM_F32 0
#define RULGNN_GEMM_BF16X3 1
/* ... RULGNN_GEMM_BF16X3_ONLY: the same, and products whose caller passes operand scales (rulgnn_sgemm_scaled_f32; the tiled ST_GCN path's
 * five large contractions) ALSO stay on the three-plane bf16 split instead of the two-plane f16 one: range-free fp32-class products at
 * twice the matrix work. */
#define RULGNN_GEMM_BF16X3_ONLY 2
int rulgnn_sgemm_mode(int32_t mode);

/* ------------------------------------------------------------------------------------------------
 * HAGCN graph stack (reference models/HAGCN/Model.py:164-183: cosine_distance, GINLayer x3, SAGPool x3, node means).
 * The Bi-LSTM stack in front of it (Model.py:26-73) and the two-layer fc behind it stay with the vendor libraries on the
 * Python side (strictly sequential recurrence over batch*nodes, not a graph kernel).
 *
 * One graph = one (sample, patch): nodes [num_node, enc_dim] -> A = cosine similarity -> three levels of
 *   GIN: g = MLP(A x + (1 + eps) x);  SAGPool: xo = leaky(Linear(A g)), P = softmax_nodes(mlp(g)),
 *   score = softmax_nodes(rank(A g)), kl += KL(score || P), keep the k = 10, 5, 1 best-scored nodes (rows of xo, rows and
 *   columns of A)
 * -> feats = [mean of the kept nodes of level 1 | level 2 | level 3]  (3 * hidden_dim), kl = sum of the three terms with
 * 'batchmean' over the graphs.
 *
 * Flat parameter buffer, per level l = 1..3 (fin = enc_dim for l = 1, hidden_dim after), h = hidden_dim:
 *   gin.eps[1] | gin.mlp.0.weight[h][fin] | gin.mlp.0.bias[h] | gin.mlp.2.weight[h][h] | gin.mlp.2.bias[h] |
 *   gnn.rank.weight[h] | gnn.rank.bias[1] | gnn.model.weight[h][h] | gnn.model.bias[h] |
 *   gnn.mlp.0.weight[h/2][h] | gnn.mlp.0.bias[h/2] | gnn.mlp.2.weight[h/2] | gnn.mlp.2.bias[1]
 */
#define RULGNN_HAGCN_TOPK_SLOTS 16   /* ints per graph in the index arrays: 10 (level 1) | 5 (level 2) | 1 (level 3) */

typedef struct rulgnn_hagcn_shape {
    int64_t graphs;           /* batch * num_patch */
    int32_t num_node;         /* 10..20 */
    int32_t enc_dim;          /* LSTM output width (encoder_hidden_dim), <= 64 */
    int32_t hidden_dim;       /* even, <= 64 */
} rulgnn_hagcn_shape;

typedef struct rulgnn_hagcn_args {
    const float *nodes;       /* [graphs, num_node, enc_dim] */
    const float *params;      /* flat graph-stack parameters */
    float *feats;             /* out [graphs, 3 * hidden_dim] */
    float *kl;                /* out [1] */
    int32_t *topk;            /* out, optional: node indices kept per level [graphs, RULGNN_HAGCN_TOPK_SLOTS] */
    const int32_t *forced_topk; /* optional: impose these selections instead of the kernel's own ranking (parity checks
                               * under score ties: the scores of this model are equal to within fp32 rounding) */
    const float *dfeats;      /* backward in  [graphs, 3 * hidden_dim] */
    const float *dkl;         /* backward in  [1] (device scalar): d loss / d kl */
    float *dnodes;            /* backward out [graphs, num_node, enc_dim] */
    float *grads;             /* backward out, flat (parameter layout) */
    void *workspace;          /* >= rulgnn_hagcn_workspace_bytes; carries the tape from forward to backward */
    size_t workspace_bytes;
} rulgnn_hagcn_args;

int64_t rulgnn_hagcn_graph_param_count(const rulgnn_hagcn_shape *shape);   /* < 0: invalid / unsupported */
size_t rulgnn_hagcn_workspace_bytes(const rulgnn_hagcn_shape *shape);
int rulgnn_hagcn_graph_forward_f32(const rulgnn_hagcn_shape *shape, const rulgnn_hagcn_args *args, void *stream);
int rulgnn_hagcn_graph_backward_f32(const rulgnn_hagcn_shape *shape, const rulgnn_hagcn_args *args, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Bidirectional LSTM layer with summed halves: out = LSTM_forward(x) + LSTM_backward(x)
 * (reference Bi_LSTM_Standard, models/HAGCN/Model.py:58-61,64-66: nn.LSTM(bidirectional=True, batch_first=True) followed by
 * split + add; torch gate order i, f, g, o; h0 = c0 = 0).  HAGCN calls it with num_seq = num_patch and
 * seq_len = batch_size * num_node (Model.py:153-157): a few very long sequences -- one persistent workgroup per
 * (direction, sequence) runs the whole recurrence without per-step launches.
 * Tensors: x [num_seq, seq_len, input_dim]; w_ih[d] [4H, input_dim]; w_hh[d] [4H, H]; b_ih[d], b_hh[d] [4H] (d = 0 forward,
 * 1 reverse: torch's *_l0 and *_l0_reverse); out, dout [num_seq, seq_len, H]; gradients in the shapes of their tensors.
 */
typedef struct rulgnn_bilstm_shape {
    int64_t seq_len;
    int32_t num_seq;          /* <= 64 */
    int32_t input_dim;        /* <= 1024 */
    int32_t hidden_dim;       /* H <= 128 */
} rulgnn_bilstm_shape;

typedef struct rulgnn_bilstm_args {
    const float *x;
    const float *w_ih[2], *w_hh[2], *b_ih[2], *b_hh[2];
    float *out;               /* forward out */
    const float *dout;        /* backward in */
    float *dx;                /* backward out; may be NULL (first layer) */
    float *dw_ih[2], *dw_hh[2], *db_ih[2], *db_hh[2];
    void *workspace;          /* carries the tape (gates, cell states) from forward to backward */
    size_t workspace_bytes;
    void *aux_stream;         /* backward, optional second stream of the caller (NULL: everything on `stream`): the parameter-gradient GEMMs
                                 (dw_ih, dw_hh, db_ih, db_hh -- nothing downstream in the call needs them) run there beside whatever the
                                 caller enqueues on `stream` next: in a stack of layers, layer l's weight gradients under layer l-1's BPTT,
                                 whose persistent recurrence occupies 2 * num_seq of the 256 CUs.  The call does NOT join: the CALLER must make
                                 `stream` wait for `aux_stream` before it reads a parameter gradient, reuses the workspace or frees either. */
} rulgnn_bilstm_args;

size_t rulgnn_bilstm_workspace_bytes(const rulgnn_bilstm_shape *shape);     /* 0: invalid / unsupported */
int rulgnn_bilstm_forward_f32(const rulgnn_bilstm_shape *shape, const rulgnn_bilstm_args *args, void *stream);
int rulgnn_bilstm_backward_f32(const rulgnn_bilstm_shape *shape, const rulgnn_bilstm_args *args, void *stream);

/* ------------------------------------------------------------------------------------------------
 * ST_Conv path (reference models/ST_Conv/Model.py, algorithms/algorithms.py:195-220): the
 * reference's natively C-MAPSS / N-CMAPSS wired spatio-temporal convolution (configs/hparams.py:40,203).
 *
 * x [batch, num_nodes, time_length] -> g = leaky(Linear_T->T(pcc(x) x)) -> c = relu(BN(Conv1d_same(g, k=6))) and, in
 * parallel, t = TemporalConvNet(x) (the block of ASTGCNN) -> tanh(th1 t + th2 c) * sigmoid(th3 t + th4 c) + x ->
 * Linear(num_nodes*time_length -> 1).  The reference evaluates both branches with the "_1" modules (Model.py:196-206);
 * the "_2" modules are dead parameters and every BatchNorm runs twice per training forward.
 *
 * Flat parameter buffer: theta1..4 [4] | gcn_layer_1.theta.0.weight[T][T] | .bias[T] | cnn_layer_1.conv.weight[N][N][6] |
 *   .bias[N] | cnn_layer_1.bn.weight[N] | .bias[N] | tcn conv_block1.0.weight[N][N][6] | bn1.weight | bn1.bias |
 *   conv_block2.0.weight | bn2.weight | bn2.bias | fc.weight[N*T] | fc.bias[1]
 * BatchNorm buffer: [3 (tcn conv_block1, tcn conv_block2, cnn)][2 (mean, var)][N].
 * The argument struct is the ASTGCNN one, rulgnn_astgcnn_args: same fields, same meaning.
 */
typedef struct rulgnn_stconv_shape {
    int64_t batch;
    int32_t num_nodes;        /* N <= 25 */
    int32_t time_length;      /* T <= 64 */
    int32_t kernel_size;      /* 6 (every reference wiring) */
} rulgnn_stconv_shape;

int64_t rulgnn_stconv_param_count(const rulgnn_stconv_shape *shape);
size_t rulgnn_stconv_workspace_bytes(const rulgnn_stconv_shape *shape);
int rulgnn_stconv_forward_f32(const rulgnn_stconv_shape *shape, const rulgnn_astgcnn_args *args, void *stream);
int rulgnn_stconv_backward_f32(const rulgnn_stconv_shape *shape, const rulgnn_astgcnn_args *args, void *stream);
int rulgnn_stconv_fwdbwd_f32(const rulgnn_stconv_shape *shape, const rulgnn_astgcnn_args *args, const rulgnn_adam_args *opt,
                             void *stream);
/* running statistics of the three BatchNorms after one training forward (the momentum update is applied twice, as the
 * reference's double use of each module does); count = batch * time_length. */
int rulgnn_stconv_bn_running_update_f32(const rulgnn_stconv_shape *shape, float *bn_stats, const float *bn_batch, int64_t count,
                                        float momentum, int32_t from_moments, void *stream);

/* ------------------------------------------------------------------------------------------------
 * STGNN graph path -- models/STGNN/Model.py.
 *
 * The model input carries no gradient and the adjacency depends on the input alone: graph construction is forward-only,
 * ChebNet's backward is the filter gradient.  G = batch * num_patch graphs of num_nodes nodes with patch_size features.
 */
typedef struct rulgnn_stgnn_shape {
    int64_t batch;
    int32_t num_nodes;   /* <= 32 */
    int32_t num_patch;
    int32_t patch_size;  /* <= 128 */
    int32_t hidden_dim;
    int32_t K;           /* Chebyshev order, <= 4 */
    int32_t top_k;       /* <= num_nodes */
} rulgnn_stgnn_shape;

size_t rulgnn_stgnn_workspace_bytes(const rulgnn_stgnn_shape *shape);      /* 0: invalid / unsupported */
/* compute_adjacency_matrix (Model.py:8-25) + the Chebyshev terms of ChebNet.forward (:49-59) for x [batch, num_nodes,
 * num_patch*patch_size]: terms [G*num_nodes, K*patch_size] (row = graph*num_nodes + node, column = k*patch_size + j);
 * adj [G, num_nodes, num_nodes] is optional (NULL to skip). */
int rulgnn_stgnn_terms_f32(const rulgnn_stgnn_shape *shape, const float *x, float *terms, float *adj, void *stream);
/* ChebNet.forward (:52-61): out [G*num_nodes, hidden_dim] = sum_k T_k filters[k], filters [K, patch_size, hidden_dim]. */
int rulgnn_stgnn_cheb_forward_f32(const rulgnn_stgnn_shape *shape, const float *terms, const float *filters, float *out,
                                  void *stream);
/* Its backward: dfilters [K, patch_size, hidden_dim] = T_k^T dout (deterministic split-K reduction). */
int rulgnn_stgnn_cheb_backward_f32(const rulgnn_stgnn_shape *shape, const float *terms, const float *dout, float *dfilters,
                                   void *workspace, size_t workspace_bytes, void *stream);

/* The whole model on one flat parameter buffer -- chebnet.filters [K, patch_size, H] | gru.weight_ih_l0 [3H, H] |
 * gru.weight_hh_l0 [3H, H] | gru.bias_ih_l0 [3H] | gru.bias_hh_l0 [3H] | fc.weight [num_nodes*num_patch*H] | fc.bias [1]
 * (the reference's state_dict order, Model.py:69-72).  The argument struct is STMSGCN's (rulgnn_stmsgcn_args: same fields,
 * same meaning; x is [batch, num_nodes, num_patch*patch_size]).  forward = STGNN_model.forward (Model.py:75-107);
 * backward needs the workspace of the forward of the same batch; fwdbwd = STGNN.update up to and, with opt, including
 * optimizer.step() (algorithms.py:399-408). */
int64_t rulgnn_stgnn_param_count(const rulgnn_stgnn_shape *shape);          /* < 0: invalid / unsupported */
size_t rulgnn_stgnn_step_workspace_bytes(const rulgnn_stgnn_shape *shape);  /* 0: invalid / unsupported */
int rulgnn_stgnn_forward_f32(const rulgnn_stgnn_shape *shape, const rulgnn_stmsgcn_args *args, void *stream);
int rulgnn_stgnn_backward_f32(const rulgnn_stgnn_shape *shape, const rulgnn_stmsgcn_args *args, void *stream);
int rulgnn_stgnn_fwdbwd_f32(const rulgnn_stgnn_shape *shape, const rulgnn_stmsgcn_args *args, const rulgnn_adam_args *opt,
                            void *stream);

/* ------------------------------------------------------------------------------------------------
 * One-layer GRU over many short sequences -- nn.GRU(input_dim, hidden_dim, batch_first=True), h0 = 0, gate order (r, z, n);
 * STGNN's recurrent part (models/STGNN/Model.py:71,97-98: batch*nodes sequences of num_patch steps).
 *
 * Tensors: x [num_seq, seq_len, input_dim]; w_ih [3H, input_dim]; w_hh [3H, H]; b_ih, b_hh [3H] (torch's *_l0);
 * out, dout [num_seq, seq_len, H]; gradients in the shapes of their tensors.  The workspace carries the tape (input and
 * recurrent projections, previous states) from forward to backward.
 */
typedef struct rulgnn_gru_shape {
    int64_t num_seq;
    int32_t seq_len;
    int32_t input_dim;
    int32_t hidden_dim;
} rulgnn_gru_shape;

typedef struct rulgnn_gru_args {
    const float *x;
    const float *w_ih, *w_hh, *b_ih, *b_hh;
    float *out;               /* forward out */
    const float *dout;        /* backward in */
    float *dx;                /* backward out; may be NULL */
    float *dw_ih, *dw_hh, *db_ih, *db_hh;
    void *workspace;
    size_t workspace_bytes;
} rulgnn_gru_args;

size_t rulgnn_gru_workspace_bytes(const rulgnn_gru_shape *shape);           /* 0: invalid / unsupported */
int rulgnn_gru_forward_f32(const rulgnn_gru_shape *shape, const rulgnn_gru_args *args, void *stream);
int rulgnn_gru_backward_f32(const rulgnn_gru_shape *shape, const rulgnn_gru_args *args, void *stream);

/* ------------------------------------------------------------------------------------------------
 * RUL test metrics on the device.
 *
 * Replaces _calc_metrics (utils.py:191-201) and the per-batch device-to-host copies in front of it (trainer.py:148-152):
 * out[0..3] = Score_v1 (utils.py:136-146, sum), Score_v2 (utils.py:157-169, mean), MAE * max_rul, RMSE * max_rul, all
 * accumulated in fp64 with a fixed summation order.  pred / real: n floats in device memory; out: 4 doubles in device memory.
 */
size_t rulgnn_rul_metrics_workspace_bytes(int64_t n);
int rulgnn_rul_metrics_f32(const float *pred, const float *real, int64_t n, float max_rul, double *out, void *workspace,
                           size_t workspace_bytes, void *stream);
/* The same pass that stops at the four SUMS: out[0..3] = sum of the Score_v1 terms, sum of the Score_v2 terms, sum |real - pred|,
 * sum (real - pred)^2 over this call's n samples -- one rank's contribution when a test set is sharded across the GPUs (SURVEY section 8f
 * rank 4): SUM the four doubles and n over the ranks (one 5-double all-reduce), then Score_v1 = S0, Score_v2 = S1 / n,
 * MAE = S2 / n * max_rul, RMSE = sqrt(S3 / n) * max_rul (utils.py:136-169).  Same workspace as rulgnn_rul_metrics_f32. */
int rulgnn_rul_metric_sums_f32(const float *pred, const float *real, int64_t n, float max_rul, double *out, void *workspace,
                               size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RULGNN_H */
