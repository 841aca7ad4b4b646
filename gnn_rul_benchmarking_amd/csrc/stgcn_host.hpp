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
    if (s->mpnn_k != 1) return RULGNN_EUNSUPPORTED;          // only the reference default k = 1
    if (s->num_patch > 64 || s->num_layers > 8 || s->patch_size > 4096) return RULGNN_EUNSUPPORTED;
    if (s->batch > (int64_t)400000000 / ((int64_t)F * s->num_patch)) return RULGNN_EUNSUPPORTED;  // 32-bit dropout counter
    return RULGNN_OK;
}

inline int tile_geometry(const rulgnn_stgcn_shape* s, TileGeom* g) {
    const int rc = validate_shape(s);
    if (rc != RULGNN_OK) return rc;
    const int N = s->num_patch, P = s->patch_size;
    g->RW = N <= 16 ? 16 : (N <= 32 ? 32 : 64);
    g->SPW = 64 / g->RW;
    // even P: patches are read with ds_read_b64, conflict-free iff (stride/2) is odd.
    // odd P: ds_read_b32, conflict-free as stride is odd.
    g->Ppad = (P % 2 == 0 && (P / 2) % 2 == 0) ? P + 2 : P;
    g->vec4 = (g->Ppad == P) && (((int64_t)N * P) % 4 == 0);
    const int raw = g->SPW * N * g->Ppad;
    g->stage_floats = (raw + 3) & ~3;
    if ((size_t)g->stage_floats * sizeof(float) > 36 * 1024) return RULGNN_EUNSUPPORTED;
    g->magicP = (uint32_t)((((uint64_t)1 << 32) + (uint64_t)P - 1) / (uint64_t)P);
    g->ntiles = (s->batch + g->SPW - 1) / g->SPW;
    return RULGNN_OK;
}

// Persistent grid: enough workgroups to fill 256 CUs at the LDS-limited occupancy, never more
// than there are tiles.  Workgroups grid-stride over wavefront tiles.
inline int grid_for_tiles(int64_t ntiles, size_t lds_bytes, int max_blocks_per_cu = 4) {
    int per_cu = lds_bytes ? (int)((160 * 1024) / lds_bytes) : max_blocks_per_cu;
    if (per_cu < 1) per_cu = 1;
    if (per_cu > max_blocks_per_cu) per_cu = max_blocks_per_cu;
    int64_t want = (ntiles + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    const int64_t cap = 256LL * per_cu;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    return (int)want;
}

int stgcn_forward_eval(const rulgnn_stgcn_shape* s, const float* x, const float* prm, const float* bn, float* out,
                       hipStream_t stream);
size_t stgcn_train_workspace_bytes(const rulgnn_stgcn_shape* s);
int stgcn_train_forward(const rulgnn_stgcn_shape* s, const rulgnn_stgcn_train_args* a, hipStream_t stream);
int stgcn_train_backward(const rulgnn_stgcn_shape* s, const rulgnn_stgcn_train_args* a, hipStream_t stream);
int stgcn_train_fwdbwd(const rulgnn_stgcn_shape* s, const rulgnn_stgcn_train_args* a, hipStream_t stream);
int adam_step(float* p, const float* g, float* m, float* v, int64_t n, int64_t step, float lr, float beta1, float beta2,
              float eps, float wd, float gscale, hipStream_t stream);
int bn_running_update(float* bn, const float* batch, int num_layers, int64_t count, float momentum, hipStream_t stream);

}  // namespace rulgnn
