"""ctypes binding of librulgnn.so -- the only door between Python and the HIP kernels.

There is deliberately no fallback: if the library is missing or a call fails, a RuntimeError is
raised.  Signatures mirror include/rulgnn.h one to one."""
from __future__ import annotations

import ctypes as C
import os

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RULGNN_LIB") or os.path.join(_PKG_DIR, "librulgnn.so")    # RULGNN_LIB: development override

OK = 0
EINVAL, EUNSUPPORTED, EWORKSPACE, EHIP, EALIGN, ECALLBACK = -1, -2, -3, -4, -5, -6      # include/rulgnn.h RULGNN_E*
EVAL_AUTO, EVAL_EXACT, EVAL_MX = 0, 1, 2      # include/rulgnn.h RULGNN_EVAL_*
STEP_AUTO, STEP_CHAIN, STEP_COOP, STEP_MX, STEP_MX_PERSIST = 0, 1, 2, 3, 4    # include/rulgnn.h RULGNN_STEP_*
TRAIN_WS_CLEAN = 1                                        # include/rulgnn.h RULGNN_TRAIN_WS_CLEAN
NUM_STATS = 10


class StgcnShape(C.Structure):
    _fields_ = [("batch", C.c_int64), ("num_patch", C.c_int32), ("patch_size", C.c_int32),
                ("num_layers", C.c_int32), ("mpnn_k", C.c_int32)]


class StgcnTrainArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("dpred", C.c_void_p), ("params", C.c_void_p),
                ("grads", C.c_void_p), ("pred", C.c_void_p), ("loss", C.c_void_p), ("bn_batch", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
                ("global_batch", C.c_int64), ("sample_offset", C.c_int64),
                ("dropout_p", C.c_float), ("seed", C.c_uint64), ("step", C.c_uint64),
                ("bn_moment_weight", C.c_float), ("step_state", C.c_void_p), ("flags", C.c_uint32), ("aux_stream", C.c_void_p)]


class AdamArgs(C.Structure):
    _fields_ = [("params", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("bn_stats", C.c_void_p),
                ("step", C.c_int64), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float), ("bn_momentum", C.c_float), ("step_state", C.c_void_p)]


STMSGCN_MAX_LAYERS = 6
STEP_STATE_BYTES = 64


class StmsgcnShape(C.Structure):
    _fields_ = [("batch", C.c_int64), ("num_patch", C.c_int32), ("patch_size", C.c_int32), ("interval", C.c_int32),
                ("band_width", C.c_int32), ("num_gcn_layers", C.c_int32), ("gcn_dims", C.c_int32 * STMSGCN_MAX_LAYERS),
                ("gru_hidden", C.c_int32)]


class StmsgcnArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("dpred", C.c_void_p), ("params", C.c_void_p),
                ("grads", C.c_void_p), ("pred", C.c_void_p), ("loss", C.c_void_p), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_size_t), ("global_batch", C.c_int64)]


class AstgcnnShape(C.Structure):
    _fields_ = [("batch", C.c_int64), ("num_nodes", C.c_int32), ("time_length", C.c_int32), ("output_dim", C.c_int32),
                ("K", C.c_int32)]


class AstgcnnArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("dpred", C.c_void_p), ("params", C.c_void_p), ("grads", C.c_void_p),
                ("pred", C.c_void_p), ("loss", C.c_void_p), ("bn_stats", C.c_void_p), ("bn_batch", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("global_batch", C.c_int64),
                ("bn_moment_weight", C.c_float), ("training", C.c_int32), ("aux_stream", C.c_void_p)]


class StconvShape(C.Structure):
    _fields_ = [("batch", C.c_int64), ("num_nodes", C.c_int32), ("time_length", C.c_int32), ("kernel_size", C.c_int32)]


class StgnnShape(C.Structure):
    _fields_ = [("batch", C.c_int64)] + [(k, C.c_int32) for k in ("num_nodes", "num_patch", "patch_size", "hidden_dim", "K", "top_k")]


class GruShape(C.Structure):
    _fields_ = [("num_seq", C.c_int64), ("seq_len", C.c_int32), ("input_dim", C.c_int32), ("hidden_dim", C.c_int32)]


class GruArgs(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("x", "w_ih", "w_hh", "b_ih", "b_hh", "out", "dout", "dx", "dw_ih", "dw_hh", "db_ih", "db_hh",
                                          "workspace")] + [("workspace_bytes", C.c_size_t)]


class FcstgnnShape(C.Structure):
    _fields_ = [("batch", C.c_int64)] + [(k, C.c_int32) for k in (
        "patch_size", "num_patch", "encoder_time_out", "encoder_hidden_dim", "encoder_out_dim", "encoder_conv_kernel",
        "hidden_dim", "num_sequential", "num_node", "num_windows")]


class FcstgnnArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("dpred", C.c_void_p), ("params", C.c_void_p), ("grads", C.c_void_p),
                ("pred", C.c_void_p), ("loss", C.c_void_p), ("bn_stats", C.c_void_p), ("bn_batch", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("global_batch", C.c_int64),
                ("sample_offset", C.c_int64), ("bn_moment_weight", C.c_float), ("dropout_p", C.c_float),
                ("seed", C.c_uint64), ("step", C.c_uint64), ("training", C.c_int32), ("step_state", C.c_void_p),
                ("compute_dtype", C.c_int32), ("aux_stream", C.c_void_p)]


class StnetShape(C.Structure):
    _fields_ = [("batch", C.c_int64), ("num_patch", C.c_int32), ("patch_size", C.c_int32), ("num_nodes", C.c_int32), ("nperseg", C.c_int32),
                ("input_dim", C.c_int32), ("num_cheb", C.c_int32), ("cheb_layers", C.c_int32 * 4), ("lstm_hidden_dim", C.c_int32),
                ("autoencoder_hidden_dim", C.c_int32)]


class StnetArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("dpred", C.c_void_p), ("params", C.c_void_p), ("grads", C.c_void_p),
                ("pred", C.c_void_p), ("recon", C.c_void_p), ("loss", C.c_void_p), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_size_t), ("global_batch", C.c_int64), ("recon_weight", C.c_void_p)]


class SagcnShape(C.Structure):
    _fields_ = [("batch", C.c_int64), ("num_patch", C.c_int32), ("patch_size", C.c_int32), ("gcn_hidden_dim", C.c_int32),
                ("attention_hidden_dim", C.c_int32)]


class SagcnArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("dpred", C.c_void_p), ("params", C.c_void_p), ("grads", C.c_void_p),
                ("pred", C.c_void_p), ("loss", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
                ("global_batch", C.c_int64)]


class StagnnShape(C.Structure):
    _fields_ = [("batch", C.c_int64), ("num_nodes", C.c_int32), ("time_length", C.c_int32), ("hidden_dim", C.c_int32),
                ("output_dim", C.c_int32), ("num_heads", C.c_int32), ("threshold", C.c_float)]


class StagnnArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("dpred", C.c_void_p), ("params", C.c_void_p), ("grads", C.c_void_p),
                ("pred", C.c_void_p), ("loss", C.c_void_p), ("bn_state", C.c_void_p), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_size_t), ("global_batch", C.c_int64), ("training", C.c_int32), ("update_running_stats", C.c_int32)]


class RgcnuShape(C.Structure):
    _fields_ = [("batch", C.c_int64), ("num_nodes", C.c_int32), ("time_length", C.c_int32), ("hidden_dim", C.c_int32),
                ("encoder_hidden_dim", C.c_int32), ("kernel_size", C.c_int32), ("alpha", C.c_float)]


class RgcnuArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("dpred", C.c_void_p), ("params", C.c_void_p), ("grads", C.c_void_p),
                ("pred", C.c_void_p), ("std_pred", C.c_void_p), ("loss", C.c_void_p), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_size_t), ("global_batch", C.c_int64), ("sample_offset", C.c_int64), ("dropout_p", C.c_float),
                ("seed", C.c_uint64), ("step", C.c_uint64), ("training", C.c_int32)]


DTYPE_F32, DTYPE_BF16 = 0, 1      # include/rulgnn.h RULGNN_DTYPE_*
GEMM_F32, GEMM_BF16X3, GEMM_BF16X3_ONLY = 0, 1, 2      # include/rulgnn.h RULGNN_GEMM_*
HAGCN_TOPK_SLOTS = 16


class HagcnShape(C.Structure):
    _fields_ = [("graphs", C.c_int64), ("num_node", C.c_int32), ("enc_dim", C.c_int32), ("hidden_dim", C.c_int32)]


class HagcnArgs(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("params", C.c_void_p), ("feats", C.c_void_p), ("kl", C.c_void_p), ("topk", C.c_void_p),
                ("forced_topk", C.c_void_p), ("dfeats", C.c_void_p), ("dkl", C.c_void_p), ("dnodes", C.c_void_p),
                ("grads", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)]


class BilstmShape(C.Structure):
    _fields_ = [("seq_len", C.c_int64), ("num_seq", C.c_int32), ("input_dim", C.c_int32), ("hidden_dim", C.c_int32)]


class BilstmArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w_ih", C.c_void_p * 2), ("w_hh", C.c_void_p * 2), ("b_ih", C.c_void_p * 2), ("b_hh", C.c_void_p * 2),
                ("out", C.c_void_p), ("dout", C.c_void_p), ("dx", C.c_void_p), ("dw_ih", C.c_void_p * 2), ("dw_hh", C.c_void_p * 2),
                ("db_ih", C.c_void_p * 2), ("db_hh", C.c_void_p * 2), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
                ("aux_stream", C.c_void_p)]


# rulgnn_allreduce_f64_fn: int (*)(void *user, double *device_buf, int32_t count, void *stream)
ALLREDUCE_F64_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p)
GRAD_READY_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p)

_SIGNATURES = {
    "rulgnn_stnet_param_count": (C.c_int64, [C.POINTER(StnetShape)]),
    "rulgnn_stnet_workspace_bytes": (C.c_size_t, [C.POINTER(StnetShape)]),
    "rulgnn_stnet_forward_f32": (C.c_int, [C.POINTER(StnetShape), C.POINTER(StnetArgs), C.c_void_p]),
    "rulgnn_stnet_backward_f32": (C.c_int, [C.POINTER(StnetShape), C.POINTER(StnetArgs), C.c_void_p]),
    "rulgnn_stnet_fwdbwd_f32": (C.c_int, [C.POINTER(StnetShape), C.POINTER(StnetArgs), C.POINTER(AdamArgs), C.c_void_p]),
    "rulgnn_sagcn_param_count": (C.c_int64, [C.POINTER(SagcnShape)]),
    "rulgnn_sagcn_workspace_bytes": (C.c_size_t, [C.POINTER(SagcnShape)]),
    "rulgnn_sagcn_tap_offset": (C.c_int64, [C.POINTER(SagcnShape), C.c_int32]),
    "rulgnn_sagcn_forward_f32": (C.c_int, [C.POINTER(SagcnShape), C.POINTER(SagcnArgs), C.c_void_p]),
    "rulgnn_sagcn_backward_f32": (C.c_int, [C.POINTER(SagcnShape), C.POINTER(SagcnArgs), C.c_void_p]),
    "rulgnn_sagcn_fwdbwd_f32": (C.c_int, [C.POINTER(SagcnShape), C.POINTER(SagcnArgs), C.POINTER(AdamArgs), C.c_void_p]),
    "rulgnn_sgemm_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                   C.c_int32, C.c_int32, C.c_void_p]),
    "rulgnn_sgemm_mode": (C.c_int, [C.c_int32]),
    "rulgnn_sgemm_scaled_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                          C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "rulgnn_sgemm_scaled_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "rulgnn_sgemm_scaled_ws_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                             C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t,
                                             C.POINTER(C.c_int32), C.c_void_p]),
    "rulgnn_absmax_partials_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]),
    "rulgnn_sgemm_splitk_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "rulgnn_sgemm_splitk_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                          C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rulgnn_stagnn_param_count": (C.c_int64, [C.POINTER(StagnnShape)]),
    "rulgnn_stagnn_bn_state_count": (C.c_int64, [C.POINTER(StagnnShape)]),
    "rulgnn_stagnn_workspace_bytes": (C.c_size_t, [C.POINTER(StagnnShape)]),
    "rulgnn_stagnn_tap_offset": (C.c_int64, [C.POINTER(StagnnShape), C.c_int32]),
    "rulgnn_stagnn_forward_f32": (C.c_int, [C.POINTER(StagnnShape), C.POINTER(StagnnArgs), C.c_void_p]),
    "rulgnn_stagnn_backward_f32": (C.c_int, [C.POINTER(StagnnShape), C.POINTER(StagnnArgs), C.c_void_p]),
    "rulgnn_stagnn_fwdbwd_f32": (C.c_int, [C.POINTER(StagnnShape), C.POINTER(StagnnArgs), C.POINTER(AdamArgs), C.c_void_p]),
    "rulgnn_rgcnu_param_count": (C.c_int64, [C.POINTER(RgcnuShape)]),
    "rulgnn_rgcnu_workspace_bytes": (C.c_size_t, [C.POINTER(RgcnuShape)]),
    "rulgnn_rgcnu_forward_f32": (C.c_int, [C.POINTER(RgcnuShape), C.POINTER(RgcnuArgs), C.c_void_p]),
    "rulgnn_rgcnu_backward_f32": (C.c_int, [C.POINTER(RgcnuShape), C.POINTER(RgcnuArgs), C.c_void_p]),
    "rulgnn_rgcnu_fwdbwd_f32": (C.c_int, [C.POINTER(RgcnuShape), C.POINTER(RgcnuArgs), C.POINTER(AdamArgs), C.c_void_p]),
    "rulgnn_peer_mailbox_bytes": (C.c_size_t, []),
    "rulgnn_peer_handle_bytes": (C.c_size_t, []),
    "rulgnn_peer_mailbox_alloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p]),
    "rulgnn_peer_mailbox_open": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "rulgnn_peer_mailbox_close": (C.c_int, [C.c_void_p]),
    "rulgnn_peer_mailbox_free": (C.c_int, [C.c_void_p]),
    "rulgnn_peer_comm_create": (C.c_void_p, [C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "rulgnn_peer_comm_destroy": (None, [C.c_void_p]),
    "rulgnn_peer_comm_set_timeout_ms": (C.c_int, [C.c_void_p, C.c_int64]),
    "rulgnn_peer_allreduce_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "rulgnn_peer_comm_collectives": (C.c_int64, [C.c_void_p]),
    "rulgnn_peer_comm_status": (C.c_int64, [C.c_void_p]),
    "rulgnn_version": (C.c_int, []),
    "rulgnn_strerror": (C.c_char_p, [C.c_int]),
    "rulgnn_stgcn_param_count": (C.c_int64, [C.c_int32, C.c_int32]),
    "rulgnn_stgcn_param_count_order": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "rulgnn_stgcn_forward_workspace_bytes": (C.c_size_t, [C.POINTER(StgcnShape)]),
    "rulgnn_stgcn_forward_path_f32": (C.c_int, [C.POINTER(StgcnShape), C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "rulgnn_stgcn_forward_mx_tap_floats": (C.c_int, []),
    "rulgnn_stgcn_forward_mx_taps_f32": (C.c_int, [C.POINTER(StgcnShape), C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "rulgnn_stgcn_forward_f32": (C.c_int, [C.POINTER(StgcnShape), C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rulgnn_stgcn_train_workspace_bytes": (C.c_size_t, [C.POINTER(StgcnShape)]),
    "rulgnn_stgcn_train_forward_f32": (C.c_int, [C.POINTER(StgcnShape), C.POINTER(StgcnTrainArgs), C.c_void_p]),
    "rulgnn_stgcn_train_backward_f32": (C.c_int, [C.POINTER(StgcnShape), C.POINTER(StgcnTrainArgs), C.c_void_p]),
    "rulgnn_stgcn_train_fwdbwd_f32": (C.c_int, [C.POINTER(StgcnShape), C.POINTER(StgcnTrainArgs), C.c_void_p]),
    "rulgnn_stgcn_train_fwdbwd_ready_f32": (C.c_int, [C.POINTER(StgcnShape), C.POINTER(StgcnTrainArgs), GRAD_READY_FN, C.c_void_p, C.c_void_p]),
    "rulgnn_stgcn_train_fwdbwd_syncbn_f32": (C.c_int, [C.POINTER(StgcnShape), C.POINTER(StgcnTrainArgs), C.c_float, ALLREDUCE_F64_FN,
                                                        C.c_void_p, C.c_void_p]),
    "rulgnn_stgcn_train_step_f32": (C.c_int, [C.POINTER(StgcnShape), C.POINTER(StgcnTrainArgs), C.POINTER(AdamArgs),
                                               C.c_void_p]),
    "rulgnn_stgcn_train_step_path_f32": (C.c_int, [C.POINTER(StgcnShape), C.POINTER(StgcnTrainArgs), C.POINTER(AdamArgs), C.c_int32,
                                                    C.c_void_p]),
    "rulgnn_stgcn_train_step_resolve": (C.c_int, [C.POINTER(StgcnShape), C.c_void_p, C.c_int32]),
    "rulgnn_stgcn_train_fwdbwd_syncbn_path_f32": (C.c_int, [C.POINTER(StgcnShape), C.POINTER(StgcnTrainArgs), C.c_float, ALLREDUCE_F64_FN,
                                                             C.c_void_p, C.c_int32, C.c_void_p]),
    "rulgnn_stgcn_train_guard_counter_offset": (C.c_int64, [C.POINTER(StgcnShape)]),
    "rulgnn_stgcn_train_args_size": (C.c_size_t, []),
    "rulgnn_struct_size": (C.c_size_t, [C.c_int32]),
    "rulgnn_stgcn_train_phase_count": (C.c_int, [C.c_int32]),
    "rulgnn_stgcn_train_phase_f32": (C.c_int, [C.POINTER(StgcnShape), C.POINTER(StgcnTrainArgs), C.c_int32, C.c_void_p]),
    "rulgnn_adam_step_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                        C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                        C.c_void_p]),
    "rulgnn_bn_running_update_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_float,
                                                C.c_int32, C.c_void_p]),
    "rulgnn_adam_step_guarded_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                                C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                                C.c_void_p, C.c_void_p]),
    "rulgnn_adam_bn_step_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                           C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                           C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_int32, C.c_void_p, C.c_void_p]),
    "rulgnn_bn_running_update_guarded_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_float,
                                                        C.c_int32, C.c_void_p, C.c_void_p]),
    "rulgnn_step_state_set": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int64, C.c_void_p]),
    "rulgnn_stgnn_workspace_bytes": (C.c_size_t, [C.POINTER(StgnnShape)]),
    "rulgnn_stgnn_terms_f32": (C.c_int, [C.POINTER(StgnnShape), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rulgnn_stgnn_cheb_forward_f32": (C.c_int, [C.POINTER(StgnnShape), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rulgnn_stgnn_cheb_backward_f32": (C.c_int, [C.POINTER(StgnnShape), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                                 C.c_void_p]),
    "rulgnn_stgnn_param_count": (C.c_int64, [C.POINTER(StgnnShape)]),
    "rulgnn_stgnn_step_workspace_bytes": (C.c_size_t, [C.POINTER(StgnnShape)]),
    "rulgnn_stgnn_forward_f32": (C.c_int, [C.POINTER(StgnnShape), C.POINTER(StmsgcnArgs), C.c_void_p]),
    "rulgnn_stgnn_backward_f32": (C.c_int, [C.POINTER(StgnnShape), C.POINTER(StmsgcnArgs), C.c_void_p]),
    "rulgnn_stgnn_fwdbwd_f32": (C.c_int, [C.POINTER(StgnnShape), C.POINTER(StmsgcnArgs), C.c_void_p, C.c_void_p]),
    "rulgnn_gru_workspace_bytes": (C.c_size_t, [C.POINTER(GruShape)]),
    "rulgnn_gru_forward_f32": (C.c_int, [C.POINTER(GruShape), C.POINTER(GruArgs), C.c_void_p]),
    "rulgnn_gru_backward_f32": (C.c_int, [C.POINTER(GruShape), C.POINTER(GruArgs), C.c_void_p]),
    "rulgnn_rul_metrics_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "rulgnn_rul_metrics_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rulgnn_rul_metric_sums_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rulgnn_adam_step_dev_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_float,
                                            C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "rulgnn_fcstgnn_param_count": (C.c_int64, [C.POINTER(FcstgnnShape)]),
    "rulgnn_fcstgnn_bn_count": (C.c_int64, [C.POINTER(FcstgnnShape)]),
    "rulgnn_fcstgnn_workspace_bytes": (C.c_size_t, [C.POINTER(FcstgnnShape)]),
    "rulgnn_fcstgnn_forward_f32": (C.c_int, [C.POINTER(FcstgnnShape), C.POINTER(FcstgnnArgs), C.c_void_p]),
    "rulgnn_fcstgnn_backward_f32": (C.c_int, [C.POINTER(FcstgnnShape), C.POINTER(FcstgnnArgs), C.c_void_p]),
    "rulgnn_fcstgnn_fwdbwd_f32": (C.c_int, [C.POINTER(FcstgnnShape), C.POINTER(FcstgnnArgs), C.POINTER(AdamArgs), C.c_void_p]),
    "rulgnn_fcstgnn_fwdbwd_syncbn_f32": (C.c_int, [C.POINTER(FcstgnnShape), C.POINTER(FcstgnnArgs), C.c_float, ALLREDUCE_F64_FN,
                                                    C.c_void_p, C.c_void_p]),
    "rulgnn_fcstgnn_bn_running_update_f32": (C.c_int, [C.POINTER(FcstgnnShape), C.c_void_p, C.c_void_p, C.c_float, C.c_int32,
                                                        C.c_void_p]),
    "rulgnn_hagcn_graph_param_count": (C.c_int64, [C.POINTER(HagcnShape)]),
    "rulgnn_hagcn_workspace_bytes": (C.c_size_t, [C.POINTER(HagcnShape)]),
    "rulgnn_hagcn_graph_forward_f32": (C.c_int, [C.POINTER(HagcnShape), C.POINTER(HagcnArgs), C.c_void_p]),
    "rulgnn_hagcn_graph_backward_f32": (C.c_int, [C.POINTER(HagcnShape), C.POINTER(HagcnArgs), C.c_void_p]),
    "rulgnn_bilstm_workspace_bytes": (C.c_size_t, [C.POINTER(BilstmShape)]),
    "rulgnn_bilstm_forward_f32": (C.c_int, [C.POINTER(BilstmShape), C.POINTER(BilstmArgs), C.c_void_p]),
    "rulgnn_bilstm_backward_f32": (C.c_int, [C.POINTER(BilstmShape), C.POINTER(BilstmArgs), C.c_void_p]),
    "rulgnn_stconv_param_count": (C.c_int64, [C.POINTER(StconvShape)]),
    "rulgnn_stconv_workspace_bytes": (C.c_size_t, [C.POINTER(StconvShape)]),
    "rulgnn_stconv_forward_f32": (C.c_int, [C.POINTER(StconvShape), C.POINTER(AstgcnnArgs), C.c_void_p]),
    "rulgnn_stconv_backward_f32": (C.c_int, [C.POINTER(StconvShape), C.POINTER(AstgcnnArgs), C.c_void_p]),
    "rulgnn_stconv_fwdbwd_f32": (C.c_int, [C.POINTER(StconvShape), C.POINTER(AstgcnnArgs), C.POINTER(AdamArgs), C.c_void_p]),
    "rulgnn_stconv_bn_running_update_f32": (C.c_int, [C.POINTER(StconvShape), C.c_void_p, C.c_void_p, C.c_int64, C.c_float,
                                                       C.c_int32, C.c_void_p]),
    "rulgnn_astgcnn_param_count": (C.c_int64, [C.POINTER(AstgcnnShape)]),
    "rulgnn_astgcnn_workspace_bytes": (C.c_size_t, [C.POINTER(AstgcnnShape)]),
    "rulgnn_astgcnn_forward_f32": (C.c_int, [C.POINTER(AstgcnnShape), C.POINTER(AstgcnnArgs), C.c_void_p]),
    "rulgnn_astgcnn_backward_f32": (C.c_int, [C.POINTER(AstgcnnShape), C.POINTER(AstgcnnArgs), C.c_void_p]),
    "rulgnn_astgcnn_fwdbwd_f32": (C.c_int, [C.POINTER(AstgcnnShape), C.POINTER(AstgcnnArgs), C.POINTER(AdamArgs), C.c_void_p]),
    "rulgnn_astgcnn_fwdbwd_syncbn_f32": (C.c_int, [C.POINTER(AstgcnnShape), C.POINTER(AstgcnnArgs), C.c_float, ALLREDUCE_F64_FN,
                                                    C.c_void_p, C.c_void_p]),
    "rulgnn_astgcnn_bn_running_update_f32": (C.c_int, [C.POINTER(AstgcnnShape), C.c_void_p, C.c_void_p, C.c_int64, C.c_float,
                                                        C.c_int32, C.c_void_p]),
    "rulgnn_stmsgcn_param_count": (C.c_int64, [C.POINTER(StmsgcnShape)]),
    "rulgnn_stmsgcn_workspace_bytes": (C.c_size_t, [C.POINTER(StmsgcnShape)]),
    "rulgnn_stmsgcn_features_f32": (C.c_int, [C.POINTER(StmsgcnShape), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rulgnn_stmsgcn_forward_f32": (C.c_int, [C.POINTER(StmsgcnShape), C.POINTER(StmsgcnArgs), C.c_void_p]),
    "rulgnn_stmsgcn_backward_f32": (C.c_int, [C.POINTER(StmsgcnShape), C.POINTER(StmsgcnArgs), C.c_void_p]),
    "rulgnn_stmsgcn_fwdbwd_f32": (C.c_int, [C.POINTER(StmsgcnShape), C.POINTER(StmsgcnArgs), C.POINTER(AdamArgs),
                                             C.c_void_p]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

# index -> ctypes mirror, in the order of the RULGNN_STRUCT_* constants of include/rulgnn.h (rulgnn_struct_size)
STRUCTS = (StgcnShape, StgcnTrainArgs, AdamArgs, StmsgcnShape, StmsgcnArgs, AstgcnnShape, AstgcnnArgs, FcstgnnShape, FcstgnnArgs, RgcnuShape, RgcnuArgs, StnetShape, StnetArgs, SagcnShape, SagcnArgs, StagnnShape, StagnnArgs, HagcnShape, HagcnArgs, BilstmShape, BilstmArgs, StconvShape, StgnnShape, GruShape, GruArgs)

_lib = None


def _preload_torch_hip_runtime() -> None:
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.  The kernels must run on THE SAME HIP
    runtime instance that owns torch's streams and allocations, so torch's copy has to be mapped
    before librulgnn.so resolves its libamdhip64 dependency (otherwise /opt/rocm's copy is loaded as
    a second runtime and every launch on a torch stream fails with hipErrorInvalidValue-class errors)."""
    import torch  # noqa: F401  (maps torch/lib/libamdhip64.so via libtorch_hip)
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load() -> C.CDLL:
    """Load librulgnn.so (built by ``python -m gnn_rul_benchmarking_amd.build``); raise if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            "`python -m gnn_rul_benchmarking_amd.build` (needs hipcc; cross-compiles gfx950 without a GPU). "
            "There is no CPU fallback for the ST_GCN path.")
    _preload_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so is stale: loud by design
        fn.restype = res
        fn.argtypes = args
    for which, mirror in enumerate(STRUCTS):       # a stale .so (or a stale binding) would read past the end of a shorter struct
        if lib.rulgnn_struct_size(which) != C.sizeof(mirror):
            raise RuntimeError(f"{LIB_PATH}: sizeof(struct #{which}) is {lib.rulgnn_struct_size(which)} in the library, {C.sizeof(mirror)} in "
                               f"this binding ({mirror.__name__}): rebuild with `python -m gnn_rul_benchmarking_amd.build --force`")
    _lib = lib
    return lib


def allreduce_callback(allreduce, ws):
    """(callback, user, failure) for the `allreduce` / `user` arguments of the *_syncbn_* entries (rulgnn_allreduce_f64_fn).
    ``allreduce`` is either a Python callable taking a float64 view of the reduction cells inside the workspace ``ws`` (summed over
    the ranks in place, in stream order: torch.distributed), or an object with ``c_callback`` / ``c_user`` -- the address of a C function
    of that type and its communicator (dp.PeerAllReduce: rulgnn_peer_allreduce_f64, no Python frame between the phases of a step).
    ``failure`` collects an exception raised inside the Python form (it must not cross the C frame)."""
    failure = []
    if hasattr(allreduce, "c_callback"):
        return C.cast(allreduce.c_callback, ALLREDUCE_F64_FN), C.c_void_p(allreduce.c_user), failure
    import torch
    base = ws.data_ptr()

    def hook(_user, buf, count, _stream):
        try:
            off = int(buf) - base
            allreduce(ws[off:off + 8 * int(count)].view(torch.float64))
            return 0
        except BaseException as e:
            failure.append(e)
            return 1
    return ALLREDUCE_F64_FN(hook), None, failure


def strerror(code: int) -> str:
    return load().rulgnn_strerror(code).decode()


def check(code: int, what: str) -> None:
    if code != OK:
        raise RuntimeError(f"{what} failed: rulgnn error {code} ({strerror(code)})")
