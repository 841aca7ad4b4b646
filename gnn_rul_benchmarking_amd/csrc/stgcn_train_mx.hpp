// Host interface of the matrix-core training chain (stgcn_train_mx.hip), called by the step driver in stgcn_train.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rulgnn.h"

namespace rulgnn {

struct MxTrainArgs {
    const float* prm;
    const float* y;
    float* pred;
    double* cells;         // reduction cells + step scratch (stgcn_train_layout.hpp)
    float* gpart;          // [grid][pcount] partial gradient rows
    float* xrec[3];        // X_l tiles [ntiles][10][4 N]; xrec[0] is written by F_0
    float* qrec[3];        // l >= 1: gated x-hat of BatchNorm 2l-1 (F_{2l} -> G_{2l})
    uint32_t* mrec[3];     // dropout mask bits of layer l, one word per lane and tile (F_{2l+2} / TOP -> G_{2l+1})
    float* arec;           // adjacency tiles [ntiles][4][55], written by F_0
    float* sb;             // d(x0 + H)
    float* dx;             // d X_l
    float* dtop;           // d X_L as (value, arg-max channel)
    int64_t B, global_batch, sample_offset;
    int N, L, pcount;
    float dropout_p, drop_scale;
    uint32_t drop_thr;
    int do_backward;
};

// shapes the chain covers: num_patch <= 15, num_layers <= 3, 16-byte window pieces (the rules of the eval kernel)
bool stgcn_train_mx_shape_ok(const rulgnn_stgcn_shape* s, const float* x);
// the power of two the gradients are carried multiplied by
float stgcn_train_mx_grad_scale(int64_t global_batch);
// one phase: kind 0 = F_idx (idx >= 1), 1 = TOP, 2 = G_idx
int stgcn_train_mx_phase(const MxTrainArgs& m, int kind, int idx, hipStream_t stream, int max_grid, int* grid_out);
// F_1 .. G_0 of a whole two-layer step as ONE launch, for batches whose common phase grid gives every workgroup a CU of its own
// (stgcn_train_mx.hip: "Small batches"; RULGNN_STEP_MX_PERSIST); RULGNN_EUNSUPPORTED (nothing launched) otherwise.  *grid_out: the
// workgroups = partial rows.  _grid: that grid for a batch, 0 where the form does not apply.
int stgcn_train_mx_persistent_grid(int64_t batch, int num_layers, int max_grid);
int stgcn_train_mx_persistent(const MxTrainArgs& m, hipStream_t stream, int max_grid, int* grid_out);
// F_0 of the chain (stgcn_forward_mx.hip): windows -> X_0 tiles, packed adjacency tiles, BatchNorm-0 sums
// (`head` != nullptr: the step runs without its prepare launch, workgroup 0 writes the head-of-step scalars: stgcn_train_layout.hpp)
struct HeadScalars;
int stgcn_train_f0_mx_packed(const rulgnn_stgcn_shape* s, const float* x, const float* prm, float* xrec0, float* arec, double* cells_bn0,
                             int cell_stride_doubles, int replicas, hipStream_t stream, const HeadScalars* head = nullptr);


// ---- the wide chain (stgcn_train_mxw.hip): 16 <= num_patch <= 47, one sample per wavefront iteration, records per SAMPLE:
// X_l / Q_l / sb / dx [10][N] at a stride of (10 N + 3) & ~3 floats, adjacency 56 floats, d X_L [2][N] at (2 N + 3) & ~3, mask bits 64 words
bool stgcn_train_mxw_shape_ok(const rulgnn_stgcn_shape* s, const float* x);
int stgcn_train_mxw_f0(const MxTrainArgs& m, const float* x, int patch_size, hipStream_t stream, const HeadScalars* head = nullptr);
int stgcn_train_mxw_phase(const MxTrainArgs& m, int kind, int idx, hipStream_t stream, int max_grid, int* grid_out);

}  // namespace rulgnn
