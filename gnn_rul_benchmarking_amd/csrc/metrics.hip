// RUL test metrics on the device.
//
// Reference path replaced: _calc_metrics -- utils.py:191-201 (scoring_function :136-146, rmse_value :148-151,
// mae_value :153-155, scoring_function_v2 :157-169), called once per epoch and test set from
// trainer.py:calc_results_per_run after the predictions were copied to the host batch by batch (trainer.py:148-152).
// Here the predictions stay where the eval forward wrote them: one pass over (pred, real) accumulates the four sums in
// fp64, per-block partials are combined in a fixed order (deterministic), 32 bytes travel to the host.
#include "stgcn_host.hpp"

namespace rulgnn {

constexpr int MB = 256;                 // threads per block
constexpr int MAX_METRIC_BLOCKS = 1024;

__global__ __launch_bounds__(MB) void rul_metrics_partial_kernel(const float* __restrict__ pred, const float* __restrict__ real,
                                                                 int64_t n, double max_rul, double* __restrict__ partial) {
    __shared__ double red[4][MB];
    double s1 = 0.0, s2 = 0.0, sa = 0.0, sq = 0.0;
    const double ln2 = 0.6931471805599453;      // -log(0.5)
    for (int64_t i = (int64_t)blockIdx.x * MB + threadIdx.x; i < n; i += (int64_t)gridDim.x * MB) {
        const double p = (double)pred[i], r = (double)real[i];
        const double d = r - p;
        // Score_v1: late predictions (real <= pred) cost exp(d/10) - 1, early ones exp(d/13) - 1, d in cycles
        s1 += r > p ? exp(d * max_rul / 13.0) - 1.0 : exp(-d * max_rul / 10.0) - 1.0;
        // Score_v2: percentage error, exp(ln2 * |err| / 5) when late, exp(-ln2 * err / 20) when early
        const double err = d / (r + 1e-8) * 100.0;
        s2 += err <= 0.0 ? exp(ln2 * (err / 5.0)) : exp(-ln2 * (err / 20.0));
        sa += fabs(d);
        sq += d * d;
    }
    red[0][threadIdx.x] = s1; red[1][threadIdx.x] = s2; red[2][threadIdx.x] = sa; red[3][threadIdx.x] = sq;
    __syncthreads();
    for (int s = MB / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s)
            for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x < 4) partial[(size_t)blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x][0];
}

// raw != 0: out[0..3] = the four SUMS themselves (Score_v1 terms, Score_v2 terms, |d|, d^2) -- what a rank contributes when a test set
// is sharded over the GPUs (one all-reduce of the sums and the count, then the same closing arithmetic on every rank)
__global__ void rul_metrics_final_kernel(const double* __restrict__ partial, int nblocks, int64_t n, double max_rul,
                                         double* __restrict__ out, int raw) {
    const int k = threadIdx.x;
    if (k >= 4) return;
    double v = 0.0;
    for (int b = 0; b < nblocks; ++b) v += partial[(size_t)b * 4 + k];
    if (raw) { out[k] = v; return; }
    const double dn = (double)n;
    if (k == 0) out[0] = v;                                 // Score_v1 (sum)
    if (k == 1) out[1] = v / dn;                            // Score_v2 (mean)
    if (k == 2) out[2] = v / dn * max_rul;                  // MAE
    if (k == 3) out[3] = sqrt(v / dn) * max_rul;            // RMSE
}

static int metric_blocks(int64_t n) {
    int64_t b = (n + MB - 1) / MB;
    return (int)(b < 1 ? 1 : (b > MAX_METRIC_BLOCKS ? MAX_METRIC_BLOCKS : b));
}

size_t rul_metrics_workspace_bytes(int64_t n) { return n < 1 ? 0 : (size_t)metric_blocks(n) * 4 * sizeof(double); }

int rul_metrics(const float* pred, const float* real, int64_t n, float max_rul, double* out, void* workspace,
                size_t workspace_bytes, hipStream_t st, int raw) {
    if (!pred || !real || !out || n < 1) return RULGNN_EINVAL;
    if (!workspace || workspace_bytes < rul_metrics_workspace_bytes(n)) return RULGNN_EWORKSPACE;
    const int nb = metric_blocks(n);
    double* partial = static_cast<double*>(workspace);
    (void)hipGetLastError();
    hipLaunchKernelGGL(rul_metrics_partial_kernel, dim3(nb), dim3(MB), 0, st, pred, real, n, (double)max_rul, partial);
    hipLaunchKernelGGL(rul_metrics_final_kernel, dim3(1), dim3(64), 0, st, (const double*)partial, nb, n, (double)max_rul, out, raw);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}

}  // namespace rulgnn
