// placeholder: training kernels land here next
#include "stgcn_host.hpp"
namespace rulgnn {
size_t stgcn_train_workspace_bytes(const rulgnn_stgcn_shape*) { return 0; }
int stgcn_train_forward(const rulgnn_stgcn_shape*, const rulgnn_stgcn_train_args*, hipStream_t) { return RULGNN_EUNSUPPORTED; }
int stgcn_train_backward(const rulgnn_stgcn_shape*, const rulgnn_stgcn_train_args*, hipStream_t) { return RULGNN_EUNSUPPORTED; }
int stgcn_train_fwdbwd(const rulgnn_stgcn_shape*, const rulgnn_stgcn_train_args*, hipStream_t) { return RULGNN_EUNSUPPORTED; }
}
