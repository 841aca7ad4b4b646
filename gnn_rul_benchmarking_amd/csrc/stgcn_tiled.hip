// "Tiled" ST_GCN path for num_patch > 64 (PHM2012 Condition_2: 160 patches, XJTU-SY: 1024 / 2048).
//
// At these sizes theta and fc1 are real [N, N] matrices (XJTU: 1M parameters each) and
// theta(A.X) = [batch*10, N] x [N, N] is a dense contraction (353 FLOP/B, MFMA-bound), so
// the layer is no longer fused into one register-resident kernel: activations live in HBM as
// [batch][10][N] tensors (the reference's own layout), the contraction is an LDS-tiled
// v_mfma_f32_16x16x4_f32 GEMM, and everything else is position-parallel: one thread per (sample, patch)
// holding the ten channels in registers; the causal taps t-1 / t-2 are read from the neighbour's
// column in memory (coalesced along t).  Reference: models/ST_GCN/Model.py:7-222.
#include "stgcn_device.hpp"
#include "stgcn_host.hpp"
#include "sgemm_mfma.hpp"
#include "aux_stream.hpp"

namespace rulgnn {

// ------------------------------------------------------------------------------------------------
// position-parallel kernels: thread = (sample b, patch t); tensors are [B][10][N]
// ------------------------------------------------------------------------------------------------
struct TArgs {
    int64_t B;
    int N, P, L;
};

// patch statistics -- Model.py:7-52.  x: [B][N][P]  ->  X0: [B][10][N]
__global__ void t_stats_kernel(const float* __restrict__ x, float* __restrict__ X0, TArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.B * a.N) return;
    const int64_t b = i / a.N;
    const int t = (int)(i % a.N);
    float st[F];
    patch_statistics(x + i * a.P, a.P, st);
#pragma unroll
    for (int c = 0; c < F; ++c) X0[(b * F + c) * a.N + t] = st[c];
}

// The same for the reference wirings' patch sizes (16: PHM2012 Condition_2 and XJTU-SY 2048 x 16; 32: XJTU-SY 1024 x 32).  A thread walking
// its own patch in global memory reads 8 bytes at a stride of P floats: every load instruction touches 64 cache lines, twice over the two
// passes (73 us for the 134 MB of XJTU batch 1024, texture-address bound).  Here a workgroup copies its 256 patches with coalesced
// 16-byte loads into LDS (rows padded by four floats: a 16-byte read per lane is conflict-free), and a thread takes its patch from there
// into registers once (patch_statistics_regs: same formulas, IEEE divisions and square roots).
template <int P>
__global__ __launch_bounds__(256) void t_stats_lds_kernel(const float* __restrict__ x, float* __restrict__ X0, TArgs a) {
    constexpr int RS = P + 4;
    __shared__ __attribute__((aligned(16))) float rows[256 * RS];
    const int64_t total = a.B * a.N, p0 = (int64_t)blockIdx.x * 256;
    const int64_t left = total - p0;
    const int cnt = left < 256 ? (int)left : 256;
    const float4* src = reinterpret_cast<const float4*>(x + p0 * P);
#pragma unroll
    for (int e = 0; e < P / 4; ++e) {
        const int q = threadIdx.x + 256 * e;                 // float4 index inside the chunk
        const int r = q / (P / 4), c = q % (P / 4);
        if (r < cnt) *reinterpret_cast<float4*>(&rows[r * RS + 4 * c]) = src[q];
    }
    __syncthreads();
    if ((int)threadIdx.x >= cnt) return;
    const int64_t i = p0 + threadIdx.x;
    const int64_t b = i / a.N;
    const int t = (int)(i - b * a.N);
    float st[F];
    patch_statistics_regs<P, false>(&rows[threadIdx.x * RS], st);
#pragma unroll
    for (int c = 0; c < F; ++c) X0[(b * F + c) * a.N + t] = st[c];
}

// Pearson adjacency -- Model.py:53-71.  One block per sample.  A: [B][10][10]
// Means first (two passes like the reference), then the Gram matrix of the centred rows on the fp32 matrix cores: D = Xc Xc^T is one
// 16 x 16 tile (ten live rows), K = N patches -- v_mfma_f32_16x16x4_f32 with the same register as A and B operand (lane (i, kq): row i at
// patch t0 + 4 kq + s in step s: a lane's four steps are one 16-byte load), a quarter of the patches per wavefront, the four partial tiles
// summed through LDS.  As 55 pair sums per thread with a 6-step shuffle reduction each it took 30 us at XJTU batch 1024.
typedef float tg_f4 __attribute__((ext_vector_type(4)));
// With `AX0`: the first layer's aggregation A.X0 of this sample right behind its adjacency (the rows are in L2 from the two passes above; it
// was a launch of its own re-reading them), and the sample's max |A.X0| as its entry of the operand-scale row `amax0`.
// With `amaxX`: the sample's max |X0| as its entry of the row the plane-writing aggregation takes its scale bound from.
__global__ __launch_bounds__(256) void t_gram_kernel(const float* __restrict__ X0, float* __restrict__ A, TArgs a, float* __restrict__ AX0,
                                                     float* __restrict__ amax0, float* __restrict__ amaxX = nullptr) {
    __shared__ float red[F][4];
    __shared__ float As[F * F];
    __shared__ float mean[16];
    __shared__ float part[4][16][17];
    __shared__ float dots[16][17];
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4, N = a.N;
    const float* xb = X0 + b * F * N;
    float s[F];
    float mxx = 0.f;
#pragma unroll
    for (int c = 0; c < F; ++c) s[c] = 0.f;
    for (int t = tid; t < N; t += 256)
#pragma unroll
        for (int c = 0; c < F; ++c) {
            const float xv = xb[c * N + t];
            s[c] += xv;
            const float ab = __builtin_fabsf(xv);
            mxx = fmaxf(mxx, ab <= 3.0e38f ? ab : 0.f);
        }
#pragma unroll
    for (int c = 0; c < F; ++c) {
        float v = s[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[c][wave] = v;
    }
    if (amaxX) {                                                            // (uniform over the launch)
        __shared__ float mxr[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mxx = fmaxf(mxx, __shfl_xor(mxx, o, 64));
        if (lane == 0) mxr[wave] = mxx;
        __syncthreads();
        if (tid == 0) amaxX[b] = fmaxf(fmaxf(mxr[0], mxr[1]), fmaxf(mxr[2], mxr[3]));
    }
    __syncthreads();
    if (tid < 16) mean[tid] = tid < F ? (red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3]) / (float)N : 0.f;
    __syncthreads();
    // this lane's row (rows 10..15 of the tile are zero), centred; patches in chunks of 16, the chunks dealt round-robin to the wavefronts
    const float mu = mean[li];
    const float* row = xb + (li < F ? li : 0) * N;
    tg_f4 acc = {0.f, 0.f, 0.f, 0.f};
    const bool vec = (N & 3) == 0;
    for (int t0 = 16 * wave; t0 < N; t0 += 64) {
        const int t = t0 + 4 * kq;
        float v[4];
        if (vec && t + 3 < N) {
            const float4 q = *reinterpret_cast<const float4*>(row + t);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = row[t + e < N ? t + e : N - 1];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float c = (li < F && t + e < N) ? v[e] - mu : 0.f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(c, c, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][4 * kq + r][li] = acc[r];
    __syncthreads();
    {
        const int p = tid >> 4, q = tid & 15;
        dots[p][q] = (part[0][p][q] + part[1][p][q]) + (part[2][p][q] + part[3][p][q]);
    }
    __syncthreads();
    if (tid < F * F) {
        const int p = tid / F, q = tid % F;
        // (the upper triangle's value for both orders: the tile is symmetric up to the matrix cores' summation order inside a row)
        const float d = p <= q ? dots[p][q] : dots[q][p];
        const float av = d / (sqrtf(dots[p][p]) * sqrtf(dots[q][q]));       // 0/0 -> NaN as the reference
        A[b * F * F + tid] = av;
        As[tid] = av;
    }
    if (!AX0) return;
    __syncthreads();
    float* ob = AX0 + b * F * N;
    float m = 0.f;
    for (int t = tid; t < N; t += 256) {                                    // (same sums in the same order as t_aggregate_kernel)
        float x[F];
#pragma unroll
        for (int c = 0; c < F; ++c) x[c] = xb[c * N + t];
#pragma unroll
        for (int c = 0; c < F; ++c) {
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < F; ++q) acc = fmaf(As[c * F + q], x[q], acc);
            ob[c * N + t] = acc;
            const float ab = __builtin_fabsf(acc);
            m = fmaxf(m, ab <= 3.0e38f ? ab : 0.f);
        }
    }
    if (amax0) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == 0) red[0][wave] = m;
        __syncthreads();
        if (tid == 0) amax0[b] = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    }
}

// The operands of the large GEMMs carry their largest finite magnitude with them (the two-plane f16 split of csrc/sgemm.hip scales each
// operand by a power of two): the kernel that writes an operand stores its workgroup's max |v| as one float of the tensor's partial-maximum
// row, and the GEMM's workgroups take the maximum of that row on their way in.  No atomics: an atomicMax per workgroup (256 on each of 16
// replica addresses at XJTU batch 1024) cost the aggregation kernel 40 us, a pre-checked one with an agent-scope load 45.
// 256 threads, whole wavefronts, every thread calls.
constexpr int T_AMAX_MAX = 4096;            // partial maxima per tensor: the workgroups of the producing launch (capped by its grid)
__device__ __forceinline__ float t_finite_abs(float v) {
    const float m = __builtin_fabsf(v);
    return m <= 3.0e38f ? m : 0.f;          // (NaN and Inf do not set the scale: they propagate through the product on their own)
}
__device__ __forceinline__ void t_amax_store(float m, float* part, float* lds4) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(lds4[0], lds4[1]), fmaxf(lds4[2], lds4[3]));
}
// ... and for a parameter matrix (theta of every layer: blockIdx.y = layer, gridDim.x partial maxima per layer)
// (+ `nzero` doubles at `zero` cleared by the first row of workgroups: the step's reduction cells, which were a memset launch of their own)
__global__ __launch_bounds__(256) void t_absmax_kernel(const float* __restrict__ prm, int64_t layer_stride_, int64_t off, int64_t n, float* part,
                                                       double* __restrict__ zero, int nzero) {
    __shared__ float l4[4];
    if (blockIdx.y == 0)
        for (int i = blockIdx.x * 256 + threadIdx.x; i < nzero; i += gridDim.x * 256) zero[i] = 0.0;
    const float* p = prm + blockIdx.y * layer_stride_ + off;
    float m = 0.f;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) m = fmaxf(m, t_finite_abs(p[e]));
    t_amax_store(m, part + (size_t)blockIdx.y * T_AMAX_MAX, l4);
}

// out[b][c][t] = sum_c' A[b][c][c'] in[b][c'][t] (+ add[b][c][t])          (torch.bmm(A, X), Model.py:87)
// (amax: slot block of `out` when it feeds a large GEMM, else null)
// (`sp_val` / `sp_arg`: the addend in its SPARSE form -- behind the last layer d X_L is d pooled at the arg-max channel and zero elsewhere:
// the value per position and the channel (row 0 of the last layer's output slot, t_tail_train_kernel) instead of a ten-row tensor that
// t_tail_bwd_kernel would write, nine rows of zeros, for this kernel to read back)
__global__ __launch_bounds__(256) void t_aggregate_kernel(const float* __restrict__ A, const float* __restrict__ in, const float* add, float* out,
                                                          TArgs a, float* amax, const float* __restrict__ sp_val = nullptr,
                                                          const float* __restrict__ sp_arg = nullptr) {      // add may alias out (in-place residual accumulation)
    __shared__ float l4[4];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float m = 0.f;
    if (i < a.B * a.N) {
    const int64_t b = i / a.N;
    const int t = (int)(i % a.N);
    const float* Ab = A + b * F * F;
    float x[F], r[F];
    // (every load before the first store: `add` may alias `out`, and a load behind a store of the same loop waits for it)
#pragma unroll
    for (int c = 0; c < F; ++c) x[c] = in[(b * F + c) * a.N + t];
#pragma unroll
    for (int c = 0; c < F; ++c) r[c] = 0.f;
    if (sp_val) {
        const float dp = sp_val[i];
        const int arg = __builtin_bit_cast(int, sp_arg[(b * F) * a.N + t]);
#pragma unroll
        for (int c = 0; c < F; ++c) r[c] = c == arg ? dp : 0.f;
    } else if (add) {
#pragma unroll
        for (int c = 0; c < F; ++c) r[c] = add[(b * F + c) * a.N + t];
    }
#pragma unroll
    for (int c = 0; c < F; ++c) {
        float acc = r[c];
#pragma unroll
        for (int q = 0; q < F; ++q) acc = fmaf(Ab[c * F + q], x[q], acc);
        r[c] = acc;
    }
#pragma unroll
    for (int c = 0; c < F; ++c) out[(b * F + c) * a.N + t] = r[c];
    if (amax) {                                                     // (uniform over the launch)
#pragma unroll
        for (int c = 0; c < F; ++c) m = fmaxf(m, t_finite_abs(r[c]));
    }
    }
    if (amax) t_amax_store(m, amax, l4);                            // (whole wavefronts: the lanes behind the last position carry 0)
}

// The same aggregation written straight as the GEMM's A operand: two f16 planes hi | lo of (A.X) s instead of fp32 (csrc/sgemm_planes.hip:
// no split pass, no fp32 copy of A.X).  The scale must be known BEFORE the first value is written, and the tensor's largest element is not:
// it comes from a BOUND -- |A.X| <= sum_c' |A[c][c']| |X[c']| <= 10 max |X| (Pearson entries lie in [-1, 1]; 16 covers their fp32
// round-off), with max |X| left behind as partial maxima by the launch that produced X (`amaxX`, n_amax floats).  s puts 16 max |X| into
// [2^11, 2^12): no element can leave the f16 range, and one within 2^-8 of the bound still has all 22 bits; smaller ones carry an absolute
// error of 2^-25 / s, i.e. <= 2^-33 of max |X| (fp32's own epsilon relative to it: 2^-24).  NaN / Inf entries propagate through hi.
__global__ __launch_bounds__(256) void t_aggregate_planes_kernel(const float* __restrict__ A, const float* __restrict__ in, _Float16* __restrict__ hi,
                                                                 _Float16* __restrict__ lo, float* __restrict__ scale_out,
                                                                 const float* __restrict__ amaxX, int n_amax, TArgs a) {
    __shared__ float l4[4];
    float m = 0.f;
    for (int e = threadIdx.x; e < n_amax; e += 256) m = fmaxf(m, amaxX[e]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) l4[threadIdx.x >> 6] = m;
    __syncthreads();
    m = 16.0f * fmaxf(fmaxf(l4[0], l4[1]), fmaxf(l4[2], l4[3]));
    float s = 1.0f;
    {
        const unsigned u = __builtin_bit_cast(unsigned, m);
        const int ex = (int)((u >> 23) & 0xFFu);
        if (m > 0.f && ex != 0 && ex != 255) {
            int se = 127 + 11 - (ex - 127);
            se = se < 1 ? 1 : (se > 254 ? 254 : se);
            s = __builtin_bit_cast(float, (unsigned)se << 23);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) scale_out[0] = s;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.B * a.N) return;
    const int64_t b = i / a.N;
    const int t = (int)(i % a.N);
    const float* Ab = A + b * F * F;
    float x[F];
#pragma unroll
    for (int c = 0; c < F; ++c) x[c] = in[(b * F + c) * a.N + t];
#pragma unroll
    for (int c = 0; c < F; ++c) {
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < F; ++q) acc = fmaf(Ab[c * F + q], x[q], acc);       // (same sums in the same order as t_aggregate_kernel)
        const float v = acc * s;
        const float hv = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & 0xFFFFE000u);
        hi[(b * F + c) * a.N + t] = (_Float16)hv;
        lo[(b * F + c) * a.N + t] = (_Float16)(v - hv);
    }
}

// eval-mode TCN block of one layer after the theta GEMM (BatchNorm folded), Model.py:134-170,187-195:
//   H = leaky(Hpre + bias[t]);  o0 = relu(relu(bn1(conv1(H))) + H);  o1 = relu(relu(bn2(conv2(o0))) + o0);  Xn = o1 + X
// The causal taps need H at t-1 and o0 at t-2, t-3: recomputed from Hpre of the neighbours (cheap, no halo exchange).
__device__ __forceinline__ void t_load_H(const float* __restrict__ Hpre, const float* __restrict__ tb, int64_t b, int t, int N,
                                         float (&H)[F]) {
    if (t < 0) {
#pragma unroll
        for (int c = 0; c < F; ++c) H[c] = 0.f;
        return;
    }
    const float bias = tb[t];
#pragma unroll
    for (int c = 0; c < F; ++c) H[c] = leaky(Hpre[(b * F + c) * N + t] + bias);
}

__device__ __forceinline__ void t_conv_point(const float (&h0)[F], const float (&h1)[F], const float* __restrict__ w, float (&z)[F]) {
    // z[co] = sum_ci w[co][ci][0] h0[ci] + w[co][ci][1] h1[ci]   (h0 = tap at t - d, h1 = tap at t)
#pragma unroll
    for (int co = 0; co < F; ++co) {
        float acc = 0.f;
#pragma unroll
        for (int ci = 0; ci < F; ++ci) {
            acc = fmaf(w[(co * F + ci) * 2 + 0], h0[ci], acc);
            acc = fmaf(w[(co * F + ci) * 2 + 1], h1[ci], acc);
        }
        z[co] = acc;
    }
}

__device__ __forceinline__ void t_o0_at(const float* __restrict__ Hpre, const float* __restrict__ tb, const float* __restrict__ w1,
                                        const float* __restrict__ sc1, const float* __restrict__ sh1, int64_t b, int t, int N,
                                        float (&o0)[F]) {
    if (t < 0) {
#pragma unroll
        for (int c = 0; c < F; ++c) o0[c] = 0.f;
        return;
    }
    float H[F], Hm[F], z[F];
    t_load_H(Hpre, tb, b, t, N, H);
    t_load_H(Hpre, tb, b, t - 1, N, Hm);
    t_conv_point(Hm, H, w1, z);
#pragma unroll
    for (int c = 0; c < F; ++c) o0[c] = relu(relu(fmaf(z[c], sc1[c], sh1[c])) + H[c]);
}

// (amaxX: the workgroup's max |Xout| as its entry of the next layer's scale-bound row, or null; 256 threads, whole wavefronts)
// conv_block2 reads o0 at t and t - 2: every position's o0 (a conv_block1 point: twenty loads, 200 FMAs) is computed ONCE, by its own
// thread, and handed to the thread two positions on through LDS; the two positions in front of the workgroup's 256 are computed by its
// first two lanes.  (Each thread computing both points itself was 600 FMAs and fifty loads per position: 49.5 us at XJTU-SY batch 1024.)
// `pooled` (behind the last layer): the channel max-pool of the position (AdaptiveMaxPool1d over the ten channels, NaN-propagating like torch: Model.py:218-219) INSTEAD of the layer's output tensor,
// which only the pool would read -- a 42-MB write and a launch that read it back at XJTU-SY batch 1024.
__global__ __launch_bounds__(256) void t_tcn_eval_kernel(const float* __restrict__ Hpre, const float* __restrict__ Xin, const float* __restrict__ prm_l,
                                                         const float* __restrict__ bnf, float* __restrict__ Xout, TArgs a, float* __restrict__ amaxX,
                                                         float* __restrict__ pooled = nullptr) {
    // prm_l: this layer's parameters (flat layout); bnf: [2][2][F] folded scale/shift of this layer
    __shared__ float l4[4];
    __shared__ float so[F][256 + 2];
    const int N = a.N;
    const int64_t total = a.B * a.N, i0 = (int64_t)blockIdx.x * 256, i = i0 + threadIdx.x;
    const float* tb = prm_l + off_theta_b(N);
    const float* w1 = prm_l + off_conv_w(N, 0);
    const float* w2 = prm_l + off_conv_w(N, 1);
    const bool in = i < total;
    const int64_t b = in ? i / N : 0;
    const int t = in ? (int)(i - b * N) : 0;
    float o0[F];
#pragma unroll
    for (int c = 0; c < F; ++c) o0[c] = 0.f;
    if (in) t_o0_at(Hpre, tb, w1, bnf, bnf + F, b, t, N, o0);
#pragma unroll
    for (int c = 0; c < F; ++c) so[c][threadIdx.x + 2] = o0[c];
    if (threadIdx.x < 2) {
        const int64_t ih = i0 - 2 + threadIdx.x;
        float oh[F];
#pragma unroll
        for (int c = 0; c < F; ++c) oh[c] = 0.f;
        if (ih >= 0 && ih < total) {
            const int64_t bh = ih / N;
            t_o0_at(Hpre, tb, w1, bnf, bnf + F, bh, (int)(ih - bh * N), N, oh);
        }
#pragma unroll
        for (int c = 0; c < F; ++c) so[c][threadIdx.x] = oh[c];
    }
    __syncthreads();
    float m = 0.f;
    if (in) {
        float o0m[F], z[F];
#pragma unroll
        for (int c = 0; c < F; ++c) o0m[c] = t >= 2 ? so[c][threadIdx.x] : 0.f;       // (position i - 2 is t - 2 of the same sample)
        t_conv_point(o0m, o0, w2, z);
        float xo[F];
#pragma unroll
        for (int c = 0; c < F; ++c) {
            const float o1 = relu(relu(fmaf(z[c], bnf[2 * F + c], bnf[3 * F + c])) + o0[c]);
            xo[c] = o1 + Xin[(b * F + c) * N + t];
            m = fmaxf(m, t_finite_abs(xo[c]));
        }
        if (pooled) {
            float pm = xo[0];
#pragma unroll
            for (int c = 1; c < F; ++c) pm = (xo[c] > pm || xo[c] != xo[c]) ? xo[c] : pm;
            pooled[i] = pm;
        } else {
#pragma unroll
            for (int c = 0; c < F; ++c) Xout[(b * F + c) * N + t] = xo[c];
        }
    }
    if (amaxX) t_amax_store(m, amaxX, l4);                          // (uniform over the launch)
}

__global__ void t_bnfold_kernel(const float* __restrict__ prm, const float* __restrict__ bn, float* __restrict__ bnf, int N, int L) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L * 2 * F) return;
    const int l = i / (2 * F), blk = (i / F) % 2, c = i % F, LS = layer_stride(N);
    const float mean = bn[((l * 2 + blk) * 2 + 0) * F + c], var = bn[((l * 2 + blk) * 2 + 1) * F + c];
    const float g = prm[l * LS + off_bn_g(N, blk) + c], be = prm[l * LS + off_bn_b(N, blk) + c];
    const float sc = g / sqrtf(var + BN_EPS);
    bnf[((l * 2 + blk) * 2 + 0) * F + c] = sc;
    bnf[((l * 2 + blk) * 2 + 1) * F + c] = be - mean * sc;
}

// head after the fc1 GEMM: pred[b] = fc2.b + sum_j fc2.w[j] relu(y1pre[b][j] + fc1.b[j]).  One block per sample.
__global__ __launch_bounds__(256) void t_head_kernel(const float* __restrict__ y1pre, const float* __restrict__ prm,
                                                     float* __restrict__ pred, TArgs a) {
    __shared__ float red[4];
    const int64_t b = blockIdx.x;
    const int N = a.N, L = a.L, tid = threadIdx.x;
    const float* b1 = prm + off_fc1_b(N, L);
    const float* w2 = prm + off_fc2_w(N, L);
    float s = 0.f;
    for (int j = tid; j < N; j += 256) s = fmaf(relu(y1pre[b * N + j] + b1[j]), w2[j], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) pred[b] = red[0] + red[1] + red[2] + red[3] + prm[off_fc2_b(N, L)];
}

// ------------------------------------------------------------------------------------------------
// host: eval forward
// ------------------------------------------------------------------------------------------------
static inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
// split-K scratch of the eval forward: fc1 ([batch x N] . [N x N]: 64 tiles of 128 x 128 at batch 1024 -- through the split-K pair like
// the training forward; as one launch of the 64 x 64 kernel it was 56 us of the 0.43-ms forward) and, at batch-100-sized row counts, theta
static inline size_t t_eval_split_floats(int N, int64_t B) {
    if (B <= 0) return 0;
    const int64_t R = B * F;
    const size_t a = R < 2048 ? sgemm_splitk_need_floats((int)R, N, N) : 0, b = sgemm_splitk_need_floats((int)B, N, N);
    return a > b ? a : b;
}
// scratch of the large products' pre-split operands (two f16 planes per operand, csrc/sgemm_planes.hip) for R = batch * 10 rows: the theta
// products hold [R x N] + [N x N], the weight gradient d theta = dH^T (A.X) two [N x R] operands (`with_grad`)
static inline size_t t_plane_bytes(int N, int64_t R, bool with_grad) {
    if (N % 256 != 0 || R < 2048 || R > (1 << 24)) return 0;
    const size_t fwd = sgemm_planes_ws_bytes((int)R, N, N), grad = with_grad ? sgemm_planes_ws_bytes(N, N, (int)R) : 0;
    return al256(fwd > grad ? fwd : grad);
}

size_t stgcn_tiled_forward_workspace_bytes(const rulgnn_stgcn_shape* s) {
    const size_t T = (size_t)s->batch * F * s->num_patch * sizeof(float);
    // X (ping), X (pong), AX, Hpre  +  A, pooled, y1pre, bnfold, the large GEMMs' operand scales (partial maxima of A.X and of theta_l)
    // (+ split-K scratch of the theta / fc1 products at batch-100-sized row counts: too few output tiles for the large kernels otherwise)
    const int64_t R = s->batch * F;
    const size_t sk = t_eval_split_floats(s->num_patch, s->batch);
    // (+ the pre-split operand planes of the theta products, csrc/sgemm_planes.hip)
    return t_plane_bytes(s->num_patch, R, false) + 4 * al256(T) + al256((size_t)s->batch * F * F * 4) + 2 * al256((size_t)s->batch * s->num_patch * 4) +
           al256((size_t)s->num_layers * 4 * F * 4) + al256((size_t)(2 + s->num_layers) * T_AMAX_MAX * sizeof(float)) + al256(sk * sizeof(float));
}

// (persistent kernels: at most T_PGRID workgroups)
#define T_LAUNCH_P(kern, n, ...)                                                                            \
    do {                                                                                                    \
        (void)hipGetLastError();                                                                            \
        hipLaunchKernelGGL(kern, dim3((unsigned)t_pgrid(n)), dim3(256), 0, stream, __VA_ARGS__);            \
        if (hipGetLastError() != hipSuccess) return RULGNN_EHIP;                                            \
    } while (0)
// patch statistics: the LDS-staged form for the wirings' patch sizes (x 16-byte aligned), the per-thread walk otherwise
#define T_STATS(xp, X0p)                                                                                    \
    do {                                                                                                    \
        if (a.P == 32 && (reinterpret_cast<uintptr_t>(xp) & 15) == 0) T_LAUNCH(t_stats_lds_kernel<32>, BN_, xp, X0p, a);      \
        else if (a.P == 16 && (reinterpret_cast<uintptr_t>(xp) & 15) == 0) T_LAUNCH(t_stats_lds_kernel<16>, BN_, xp, X0p, a); \
        else T_LAUNCH(t_stats_kernel, BN_, xp, X0p, a);                                                     \
    } while (0)
#define T_LAUNCH(kern, n, ...)                                                                              \
    do {                                                                                                    \
        (void)hipGetLastError();                                                                            \
        hipLaunchKernelGGL(kern, dim3((unsigned)(((n) + 255) / 256)), dim3(256), 0, stream, __VA_ARGS__);   \
        if (hipGetLastError() != hipSuccess) return RULGNN_EHIP;                                            \
    } while (0)

int stgcn_tiled_forward_eval(const rulgnn_stgcn_shape* s, const float* x, const float* prm, const float* bn, float* pred,
                             void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (!workspace || workspace_bytes < stgcn_tiled_forward_workspace_bytes(s)) return RULGNN_EWORKSPACE;
    const int N = s->num_patch, L = s->num_layers, LS = layer_stride(N);
    const int64_t B = s->batch, BN_ = B * N;
    TArgs a{B, N, s->patch_size, L};
    char* w = static_cast<char*>(workspace);
    const size_t T = al256((size_t)B * F * N * sizeof(float));
    float* Xa = reinterpret_cast<float*>(w); w += T;
    float* Xb = reinterpret_cast<float*>(w); w += T;
    float* AX = reinterpret_cast<float*>(w); w += T;
    float* Hpre = reinterpret_cast<float*>(w); w += T;
    float* A = reinterpret_cast<float*>(w); w += al256((size_t)B * F * F * 4);
    float* pooled = reinterpret_cast<float*>(w); w += al256((size_t)B * N * 4);
    float* y1pre = reinterpret_cast<float*>(w); w += al256((size_t)B * N * 4);
    float* bnf = reinterpret_cast<float*>(w); w += al256((size_t)L * 4 * F * 4);
    float* amax = reinterpret_cast<float*>(w); w += al256((size_t)(2 + L) * T_AMAX_MAX * sizeof(float));   // A.X of the current layer, theta of every layer, X of the current layer
    float* amaxX = amax + (size_t)(1 + L) * T_AMAX_MAX;
    float* split = reinterpret_cast<float*>(w); w += al256(t_eval_split_floats(N, B) * sizeof(float));
    const size_t plane_bytes = t_plane_bytes(N, B * F, false);
    void* planes = plane_bytes ? static_cast<void*>(w) : nullptr;
    const bool few_rows = B * F < 2048;
    const int n_pos = (int)((BN_ + 255) / 256), n_th = 256;
    const bool scaled = n_pos <= T_AMAX_MAX;

    T_LAUNCH(t_bnfold_kernel, L * 2 * F, prm, bn, bnf, N, L);
    if (scaled) {
        (void)hipGetLastError();
        hipLaunchKernelGGL(t_absmax_kernel, dim3(n_th, L), dim3(256), 0, stream, prm, (int64_t)LS, (int64_t)off_theta_w(N), (int64_t)N * N,
                           amax + T_AMAX_MAX, (double*)nullptr, 0);
        if (hipGetLastError() != hipSuccess) return RULGNN_EHIP;
    }
    // theta(A.X) on pre-split operands with the A planes written by the aggregation itself (t_aggregate_planes_kernel: no split pass, no
    // fp32 A.X; XJTU-SY batch 1024: 2 x (22 + 17) us of aggregation + split pass become 2 x ~25): needs max |X_l| from the launch in front
    const bool fused_planes = planes && scaled && !few_rows && B <= T_AMAX_MAX && sgemm_big_mode() == 1 &&
                              sgemm_planes_slices((int)(B * F), N, N, false) == 1 &&
                              sgemm_planes_ok(AX, N, 1, prm + off_theta_w(N), N, 1, (int)(B * F), N, N, 1);
    T_STATS(x, Xa);
    (void)hipGetLastError();
    hipLaunchKernelGGL(t_gram_kernel, dim3((unsigned)B), dim3(256), 0, stream, Xa, A, a, (float*)nullptr, (float*)nullptr,
                       fused_planes ? amaxX : (float*)nullptr);
    if (hipGetLastError() != hipSuccess) return RULGNN_EHIP;
    float* Xin = Xa;
    float* Xout = Xb;
    for (int l = 0; l < L; ++l) {
        const float* pl = prm + l * LS;
        int rc;                                                                                  // (A.X) theta^T
        if (fused_planes) {
            void *ph, *plo;
            float* psc;
            sgemm_planes_a_slots(planes, (int)(B * F), N, &ph, &plo, &psc);
            T_LAUNCH(t_aggregate_planes_kernel, BN_, A, Xin, static_cast<_Float16*>(ph), static_cast<_Float16*>(plo), psc, (const float*)amaxX,
                     l == 0 ? (int)B : n_pos, a);
            rc = sgemm_planes(nullptr, N, 1, pl + off_theta_w(N), N, 1, Hpre, N, (int)(B * F), N, N, false, 1, nullptr, 0,
                              amax + (size_t)(1 + l) * T_AMAX_MAX, n_th, planes, plane_bytes, stream, true);
            if (rc != RULGNN_OK) return rc;
            T_LAUNCH(t_tcn_eval_kernel, BN_, Hpre, Xin, pl, bnf + l * 4 * F, Xout, a, l + 1 < L ? amaxX : (float*)nullptr,
                     l + 1 == L ? pooled : (float*)nullptr);
            float* tmp = Xin; Xin = Xout; Xout = tmp;
            continue;
        }
        T_LAUNCH(t_aggregate_kernel, BN_, A, Xin, (const float*)nullptr, AX, a, scaled ? amax : (float*)nullptr);
        if (few_rows)
            rc = sgemm_splitk(AX, N, 1, pl + off_theta_w(N), N, 1, Hpre, N, (int)(B * F), N, N, false, split, stream, scaled ? amax : (float*)nullptr,
                              n_pos, scaled ? amax + (size_t)(1 + l) * T_AMAX_MAX : (float*)nullptr, n_th);
        else
            rc = sgemm(AX, N, 1, pl + off_theta_w(N), N, 1, Hpre, N, (int)(B * F), N, N, false, stream, 0, scaled ? amax : (float*)nullptr,
                       n_pos, scaled ? amax + (size_t)(1 + l) * T_AMAX_MAX : (float*)nullptr, n_th, planes, plane_bytes);
        if (rc != RULGNN_OK) return rc;
        T_LAUNCH(t_tcn_eval_kernel, BN_, Hpre, Xin, pl, bnf + l * 4 * F, Xout, a, (float*)nullptr, l + 1 == L ? pooled : (float*)nullptr);
        float* tmp = Xin; Xin = Xout; Xout = tmp;
    }
    int rc = sgemm_splitk(pooled, N, 1, prm + off_fc1_w(N, L), N, 1, y1pre, N, (int)B, N, N, false, split, stream);          // fc1
    if (rc != RULGNN_OK) return rc;
    (void)hipGetLastError();
    hipLaunchKernelGGL(t_head_kernel, dim3((unsigned)B), dim3(256), 0, stream, y1pre, prm, pred, a);
    return hipGetLastError() == hipSuccess ? RULGNN_OK : RULGNN_EHIP;
}


// ================================================================================================
// training (tiled path): activations of every layer are kept in the workspace, the backward reads them
// ================================================================================================
typedef float f32x4tt __attribute__((ext_vector_type(4)));
constexpr int TTS = 68, TTR = 30;       // per-wave LDS transpose tile of the conv weight gradient (see stgcn_train.hip)

struct TBn {            // BatchNorm constants of one layer-block, computed by every block from the reduction cells
    float mean[F], istd[F], sc[F], sh[F], gi[F], k1[F], k2[F];
};

// Reduction cells of the tiled path, replicated like the fused path's (4096 workgroups per kernel adding into the same 20
// addresses serialise for ~40 us): forward cells [T_REP][tc_sf(L)], then backward cells + loss [T_REP][tc_sb(L)] (loss at
// offset tc_sf(L) of a backward replica).  Workgroup b adds into replica b % T_REP; readers sum the replicas in a fixed order.
constexpr int T_REP = 16;
__host__ __device__ constexpr int tc_sf(int L) { return 2 * L * 2 * F; }
__host__ __device__ constexpr int tc_sb(int L) { return 2 * L * 2 * F + 8; }
__device__ __forceinline__ double tc_sum(const double* base, int stride, int idx) {
    double v = 0.0;
#pragma unroll
    for (int r = 0; r < T_REP; ++r) v += base[r * stride + idx];
    return v;
}

// (threads tid0 .. tid0 + F - 1 do the work: a kernel that needs two sets gives them to two wavefronts and pays the fp64 chain once)
__device__ __forceinline__ void t_bn_consts(const double* cells_fwd, const double* cells_bwd, const float* __restrict__ prm_l,
                                            int N, int L, int blk, int bn_index, double cnt, bool with_bwd, float* lds /* [7][F] */,
                                            int tid0 = 0) {
    if ((int)threadIdx.x >= tid0 && (int)threadIdx.x < tid0 + F) {
        const int c = threadIdx.x - tid0;
        const double mean = tc_sum(cells_fwd, tc_sf(L), (bn_index * 2 + 0) * F + c) / cnt;
        double var = tc_sum(cells_fwd, tc_sf(L), (bn_index * 2 + 1) * F + c) / cnt - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        const double istd = 1.0 / sqrt(var + (double)BN_EPS);
        const double g = prm_l[off_bn_g(N, blk) + c], be = prm_l[off_bn_b(N, blk) + c];
        lds[0 * F + c] = (float)mean;
        lds[1 * F + c] = (float)istd;
        lds[2 * F + c] = (float)(g * istd);
        lds[3 * F + c] = (float)(be - mean * g * istd);
        lds[4 * F + c] = (float)(g * istd);
        lds[5 * F + c] = with_bwd ? (float)(tc_sum(cells_bwd, tc_sb(L), (bn_index * 2 + 0) * F + c) / cnt) : 0.f;
        lds[6 * F + c] = with_bwd ? (float)(tc_sum(cells_bwd, tc_sb(L), (bn_index * 2 + 1) * F + c) / cnt) : 0.f;
    }
}

// block-wide sum of ten per-thread pairs -> one fp64 atomic per channel and pair member
__device__ __forceinline__ void t_pair_reduce(const float (&sa)[F], const float (&sb)[F], double* cell /* [2][F] */, float* lds /* [4][2F] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < F; ++c) {
        float va = sa[c], vb = sb[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { va += __shfl_xor(va, o, 64); vb += __shfl_xor(vb, o, 64); }
        if (lane == 0) { lds[wave * 2 * F + c] = va; lds[wave * 2 * F + F + c] = vb; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * F) {
        double v = 0.0;
        for (int w = 0; w < 4; ++w) v += (double)lds[w * 2 * F + threadIdx.x];
        atomicAdd(cell + threadIdx.x, v);
    }
}

// The kernels with a reduction behind them (BatchNorm sums, convolution weight-gradient partials) are PERSISTENT: at most T_PGRID
// workgroups (four per CU, what the weight-gradient kernels' LDS tiles allow), each walking chunks of 256 positions with its sums in
// registers, so that the BatchNorm constants (a chain of fp64 divisions and a square root over 16 cell replicas) and the block
// reduction + atomics are paid once per workgroup instead of once per 256 positions -- at XJTU batch 1024 (4096 chunks) those two were
// 20 and 33 us of t_conv2_bwd_kernel's 101.
#ifndef T_PGRID_V
#define T_PGRID_V 1024
#endif
constexpr int T_PGRID = T_PGRID_V;
__host__ __device__ inline int t_pgrid(int64_t positions) {
    const int64_t chunks = (positions + 255) / 256;
    return (int)(chunks < T_PGRID ? chunks : T_PGRID);
}

struct TTrain {
    int64_t B, sample_offset, global_batch;
    int N, P, L;
    float dropout_p, drop_scale;
    uint32_t drop_thr, drop_key;
    const uint32_t* key_dev;    // device step state: this layer's key (else drop_key)
    double cnt;
    double* cells_fwd;          // [T_REP][tc_sf(L)]
    double* cells_bwd;          // [T_REP][tc_sb(L)], the loss cell at offset tc_sf(L) of every replica
};

__device__ __forceinline__ void t_ld10(const float* __restrict__ T, int64_t b, int t, int N, float (&v)[F]) {
    if (t < 0 || t >= N) {
#pragma unroll
        for (int c = 0; c < F; ++c) v[c] = 0.f;
        return;
    }
#pragma unroll
    for (int c = 0; c < F; ++c) v[c] = T[(b * F + c) * N + t];
}
__device__ __forceinline__ void t_st10(float* __restrict__ T, int64_t b, int t, int N, const float (&v)[F]) {
#pragma unroll
    for (int c = 0; c < F; ++c) T[(b * F + c) * N + t] = v[c];
}

// ---- forward --------------------------------------------------------------------------------------
// H = leaky(Hpre + bias); z1 = conv1(H); sums of z1.  H is NOT stored: the layer keeps the product's output Hpre (the GEMM writes it into the
// layer's slot) and every reader re-derives H from it -- one add and one select per element instead of a 42-MB tensor per layer written
// here and read three times (XJTU-SY batch 1024).
__global__ __launch_bounds__(256) void t_conv1_train_kernel(const float* __restrict__ Hpre, const float* __restrict__ prm_l,
                                                            float* __restrict__ z1, int bn_index, TTrain a) {
    __shared__ float lds[4 * 2 * F];
    const int N = a.N;
    const int64_t total = a.B * a.N;
    float sa[F], sb[F];
#pragma unroll
    for (int c = 0; c < F; ++c) sa[c] = sb[c] = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / N;
        const int t = (int)(i - b * N);
        float h[F], hm[F], z[F];
        t_load_H(Hpre, prm_l + off_theta_b(N), b, t, N, h);
        t_load_H(Hpre, prm_l + off_theta_b(N), b, t - 1, N, hm);
        t_conv_point(hm, h, prm_l + off_conv_w(N, 0), z);
        t_st10(z1, b, t, N, z);
#pragma unroll
        for (int c = 0; c < F; ++c) { sa[c] += z[c]; sb[c] = fmaf(z[c], z[c], sb[c]); }
    }
    t_pair_reduce(sa, sb, a.cells_fwd + (blockIdx.x % T_REP) * tc_sf(a.L) + bn_index * 2 * F, lds);
}

// o0 = relu(relu(bn1(z1)) + H); z2 = conv2(o0); sums of z2
__device__ __forceinline__ void t_o0_from(const float* __restrict__ z1, const float* __restrict__ Hpre, const float* __restrict__ tb, const float* bnc,
                                          int64_t b, int t, int N, float (&o0)[F]) {
    if (t < 0 || t >= N) {
#pragma unroll
        for (int c = 0; c < F; ++c) o0[c] = 0.f;
        return;
    }
    const float bias = tb[t];
#pragma unroll
    for (int c = 0; c < F; ++c)
        o0[c] = relu(relu(fmaf(z1[(b * F + c) * N + t], bnc[2 * F + c], bnc[3 * F + c])) + leaky(Hpre[(b * F + c) * N + t] + bias));
}

__global__ __launch_bounds__(256) void t_conv2_train_kernel(const float* __restrict__ z1, const float* __restrict__ Hpre, const float* __restrict__ prm_l,
                                                            float* __restrict__ o0, float* __restrict__ z2, int bn_index, TTrain a) {
    __shared__ float lds[4 * 2 * F];
    __shared__ float bnc[7 * F];
    t_bn_consts(a.cells_fwd, a.cells_bwd, prm_l, a.N, a.L, 0, bn_index - 1, a.cnt, false, bnc);
    __syncthreads();
    const int N = a.N;
    const int64_t total = a.B * a.N;
    float sa[F], sb[F];
#pragma unroll
    for (int c = 0; c < F; ++c) sa[c] = sb[c] = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / N;
        const int t = (int)(i - b * N);
        float v[F], vm[F], z[F];
        t_o0_from(z1, Hpre, prm_l + off_theta_b(N), bnc, b, t, N, v);
        t_o0_from(z1, Hpre, prm_l + off_theta_b(N), bnc, b, t - 2, N, vm);
        t_conv_point(vm, v, prm_l + off_conv_w(N, 1), z);
        t_st10(o0, b, t, N, v);
        t_st10(z2, b, t, N, z);
#pragma unroll
        for (int c = 0; c < F; ++c) { sa[c] += z[c]; sb[c] = fmaf(z[c], z[c], sb[c]); }
    }
    t_pair_reduce(sa, sb, a.cells_fwd + (blockIdx.x % T_REP) * tc_sf(a.L) + bn_index * 2 * F, lds);
}

// Xout = dropout(relu(relu(bn2(z2)) + o0)) + Xin; and what the next stage reads of it, position by position: the next layer's
// aggregation A.Xout (AXnext) or, behind the last layer, the channel max-pool (pooled) -- both were launches that re-read Xout
__global__ __launch_bounds__(256) void t_tail_train_kernel(const float* __restrict__ z2, const float* __restrict__ o0, const float* __restrict__ Xin,
                                                           const float* __restrict__ prm_l, float* __restrict__ Xout, int bn_index, TTrain a,
                                                           const float* __restrict__ A, float* __restrict__ AXnext, float* __restrict__ pooled,
                                                           float* amax_next /* row of AXnext, or of pooled behind the last layer */) {
    __shared__ float bnc[7 * F];
    __shared__ float l4[4];
    t_bn_consts(a.cells_fwd, a.cells_bwd, prm_l, a.N, a.L, 1, bn_index, a.cnt, false, bnc);
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float mx = 0.f;
    if (i < a.B * a.N) {
    const int64_t b = i / a.N;
    const int t = (int)(i % a.N), N = a.N;
    const uint32_t ctr = (uint32_t)((a.sample_offset + b) * F) * (uint32_t)N + (uint32_t)t;
    const uint32_t key = a.key_dev ? *a.key_dev : a.drop_key;
    float zv[F], ov[F], xv[F];
#pragma unroll
    for (int c = 0; c < F; ++c) {
        const int64_t idx = (b * F + c) * N + t;
        zv[c] = z2[idx]; ov[c] = o0[idx]; xv[c] = Xin[idx];
    }
#pragma unroll
    for (int c = 0; c < F; ++c) {
        float o1 = relu(relu(fmaf(zv[c], bnc[2 * F + c], bnc[3 * F + c])) + ov[c]);
        if (a.dropout_p > 0.f) {
            const uint32_t h = lowbias32((ctr + (uint32_t)(c * N)) ^ key);
            o1 = h >= a.drop_thr ? o1 * a.drop_scale : 0.f;
        }
        xv[c] = o1 + xv[c];
        if (!pooled) Xout[(b * F + c) * N + t] = xv[c];
    }
    if (AXnext) {                                                    // (same sums, in the same order, as t_aggregate_kernel)
        const float* Ab = A + b * F * F;
#pragma unroll
        for (int c = 0; c < F; ++c) {
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < F; ++q) acc = fmaf(Ab[c * F + q], xv[q], acc);
            AXnext[(b * F + c) * N + t] = acc;
            mx = fmaxf(mx, t_finite_abs(acc));
        }
    }
    if (pooled) {                                                    // (the channel max-pool, as in t_tcn_eval_kernel)
        float m = xv[0];
#pragma unroll
        for (int c = 1; c < F; ++c) m = (xv[c] > m || xv[c] != xv[c]) ? xv[c] : m;
        pooled[i] = m;
        mx = t_finite_abs(m);
        // The last layer's output is read by nobody but the backward's arg-max (t_tail_bwd_kernel, top): the channel it routes d pooled
        // to travels as ONE row -- row 0 of the output tensor -- instead of the ten rows it would re-derive it from (76 MB less traffic
        // per step at XJTU-SY batch 1024).  Same rule as there: the first maximum, the first NaN.
        int arg = 0;
        float am = xv[0];
#pragma unroll
        for (int c = 1; c < F; ++c) {
            const bool take = (xv[c] > am) || (xv[c] != xv[c] && am == am);
            am = take ? xv[c] : am;
            arg = take ? c : arg;
        }
        Xout[(b * F) * N + t] = __builtin_bit_cast(float, arg);
    }
    }
    if (amax_next) t_amax_store(mx, amax_next, l4);                 // (whole wavefronts)
}

// head: y1 = relu(y1pre + b1) (stored), pred, loss, dpred, dy1pre.  One block per sample.
__global__ __launch_bounds__(256) void t_head_train_kernel(const float* __restrict__ y1pre, const float* __restrict__ prm, const float* __restrict__ gy,
                                                           int has_dpred, float* __restrict__ y1, float* __restrict__ pred,
                                                           float* __restrict__ dpred_out, float* __restrict__ dy1pre, TTrain a,
                                                           float* __restrict__ one, float* __restrict__ amax_dy1 /* [B] or null */) {
    __shared__ float red[4];
    __shared__ float dp;
    const int64_t b = blockIdx.x;
    const int N = a.N, L = a.L, tid = threadIdx.x;
    const float* b1 = prm + off_fc1_b(N, L);
    const float* w2 = prm + off_fc2_w(N, L);
    float s = 0.f;
    for (int j = tid; j < N; j += 256) {
        const float v = relu(y1pre[b * N + j] + b1[j]);
        y1[b * N + j] = v;
        s = fmaf(v, w2[j], s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        const float p = red[0] + red[1] + red[2] + red[3] + prm[off_fc2_b(N, L)];
        pred[b] = p;
        float d = 0.f;
        if (has_dpred == 1) d = gy[b];
        else if (has_dpred == 0) {
            const float diff = p - gy[b];
            d = 2.f * diff / (float)a.global_batch;
            atomicAdd(a.cells_bwd + (blockIdx.x % T_REP) * tc_sb(a.L) + tc_sf(a.L), (double)diff * (double)diff);
        }
        // d fc2.bias = sum_b dpred[b]: the cell behind the loss (it was a launch of its own, 13 us); and the constant the bias column sums read
        atomicAdd(a.cells_bwd + (blockIdx.x % T_REP) * tc_sb(a.L) + tc_sf(a.L) + 1, (double)d);
        if (blockIdx.x == 0) one[0] = 1.f;
        dpred_out[b] = d;
        dp = d;
    }
    __syncthreads();
    const float d = dp;
    float mx = 0.f;
    for (int j = tid; j < N; j += 256) {
        const float v = (y1[b * N + j] > 0.f) ? d * w2[j] : 0.f;
        dy1pre[b * N + j] = v;
        mx = fmaxf(mx, t_finite_abs(v));
    }
    if (amax_dy1) {
        __syncthreads();
        t_amax_store(mx, amax_dy1, red);
    }
}

// ---- backward --------------------------------------------------------------------------------------
// gradient entering the layer: top layer: dXn = scatter of dpooled to the arg-max channel; else dXn = dX buffer.
// gsum = dropmask(dXn) * [o1 > 0]; dy2 = gsum * [x1 > 0]; BatchNorm (conv_block2) backward sums
__global__ __launch_bounds__(256) void t_tail_bwd_kernel(const float* __restrict__ dpooled, const float* __restrict__ Xout, float* __restrict__ dXn,
                                                         const float* __restrict__ z2, const float* __restrict__ o0, const float* __restrict__ prm_l,
                                                         float* __restrict__ gsum, int bn_index, int top, TTrain a) {
    __shared__ float lds[4 * 2 * F];
    __shared__ float bnc[7 * F];
    t_bn_consts(a.cells_fwd, a.cells_bwd, prm_l, a.N, a.L, 1, bn_index, a.cnt, false, bnc);
    __syncthreads();
    const int N = a.N;
    const int64_t total = a.B * a.N;
    float sa[F], sb[F];
#pragma unroll
    for (int c = 0; c < F; ++c) sa[c] = sb[c] = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / N;
        const int t = (int)(i - b * N);
        // (all loads first, then the arithmetic, then the stores: written channel by channel the compiler kept one channel's loads
        // behind the previous channel's stores and waits -- some thirty dependent memory round trips per position)
        float gin[F], zv[F], ov[F];
#pragma unroll
        for (int c = 0; c < F; ++c) {
            const int64_t idx = (b * F + c) * N + t;
            gin[c] = top ? 0.f : dXn[idx];
            zv[c] = z2[idx];
            ov[c] = o0[idx];
        }
        if (top) {
            const int arg = __builtin_bit_cast(int, Xout[(b * F) * N + t]);      // (row 0 of the last layer's output: t_tail_train_kernel)
            const float dp = dpooled[b * N + t];
#pragma unroll
            for (int c = 0; c < F; ++c) gin[c] = (c == arg) ? dp : 0.f;      // (d X_L itself is not written: t_aggregate_kernel takes (d pooled, arg))
        }
        const uint32_t ctr = (uint32_t)((a.sample_offset + b) * F) * (uint32_t)N + (uint32_t)t;
        const uint32_t key = a.key_dev ? *a.key_dev : a.drop_key;
#pragma unroll
        for (int c = 0; c < F; ++c) {
            const int64_t idx = (b * F + c) * N + t;
            float g = gin[c];
            const float zz = zv[c];
            const float x1 = relu(fmaf(zz, bnc[2 * F + c], bnc[3 * F + c]));
            const float o1 = relu(x1 + ov[c]);
            if (a.dropout_p > 0.f) {
                const uint32_t h = lowbias32((ctr + (uint32_t)(c * N)) ^ key);
                g = h >= a.drop_thr ? g * a.drop_scale : 0.f;
            }
            g = (o1 > 0.f) ? g : 0.f;
            gsum[idx] = g;
            const float dy = (x1 > 0.f) ? g : 0.f;
            sa[c] += dy;
            sb[c] = fmaf(dy, (zz - bnc[0 * F + c]) * bnc[1 * F + c], sb[c]);
        }
    }
    t_pair_reduce(sa, sb, a.cells_bwd + (blockIdx.x % T_REP) * tc_sb(a.L) + bn_index * 2 * F, lds);
}

__device__ __forceinline__ void t_wgrad_mfma(float* T, const float (&dz)[F], const float (&h)[F], const float (&hs)[F], int lane,
                                             f32x4tt& acc0, f32x4tt& acc1) {
#pragma unroll
    for (int c = 0; c < F; ++c) {
        T[c * TTS + lane] = dz[c];
        T[(F + c) * TTS + lane] = h[c];
        T[(2 * F + c) * TTS + lane] = hs[c];
    }
    __builtin_amdgcn_wave_barrier();
    const int i = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
        const float4 av = *reinterpret_cast<const float4*>(&T[i * TTS + 16 * kq + 4 * jp]);
        const float4 b0 = *reinterpret_cast<const float4*>(&T[(F + i) * TTS + 16 * kq + 4 * jp]);
        const float4 b1 = *reinterpret_cast<const float4*>(&T[(i < 4 ? 2 * F + 6 + i : 3 * F - 1) * TTS + 16 * kq + 4 * jp]);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b0.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b1.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b0.y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b1.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b0.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b1.z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b0.w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b1.w, acc1, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
}

// per-block partial conv weight gradient row [200] from the four wavefronts' MFMA accumulators
__device__ __forceinline__ void t_wgrad_store(float* red /* [8][64] */, const f32x4tt& acc0, const f32x4tt& acc1, float* row) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                red[e * 64 + lane] = (w == 0 ? 0.f : red[e * 64 + lane]) + acc0[e];
                red[(4 + e) * 64 + lane] = (w == 0 ? 0.f : red[(4 + e) * 64 + lane]) + acc1[e];
            }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < 8 * 64; i += 256) {
        const int rg = (i / 64) % 4, which = i / 256, ln = i % 64;
        const int co = 4 * (ln >> 4) + rg, j = ln & 15;
        int ci, tap;
        if (which == 0) { ci = j < F ? j : j - F; tap = j < F ? 1 : 0; }
        else { ci = 6 + j; tap = 0; }
        if (co < F && ci < F && (which == 0 || j < 4)) row[(co * F + ci) * 2 + tap] = red[(which * 4 + rg) * 64 + ln];
    }
}

// dz2 at position t from the stored gsum / z2 (BatchNorm conv_block2 backward)
__device__ __forceinline__ void t_dz2_at(const float* __restrict__ gsum, const float* __restrict__ z2, const float* bnc, int64_t b, int t, int N,
                                         float (&dz)[F]) {
    if (t < 0 || t >= N) {
#pragma unroll
        for (int c = 0; c < F; ++c) dz[c] = 0.f;
        return;
    }
    // (both tensors loaded up front: with the load inside the select, `x1 > 0 ? gsum[idx] : 0`, the compiler issued it under a branch
    // behind the wait for z2 -- ten dependent memory round trips per call, two thirds of the kernel's wavefront time)
    float zv[F], gv[F];
#pragma unroll
    for (int c = 0; c < F; ++c) {
        zv[c] = z2[(b * F + c) * N + t];
        gv[c] = gsum[(b * F + c) * N + t];
    }
#pragma unroll
    for (int c = 0; c < F; ++c) {
        const float zz = zv[c];
        const float x1 = relu(fmaf(zz, bnc[2 * F + c], bnc[3 * F + c]));
        const float dy = (x1 > 0.f) ? gv[c] : 0.f;
        const float xh = (zz - bnc[0 * F + c]) * bnc[1 * F + c];
        dz[c] = bnc[4 * F + c] * (dy - bnc[5 * F + c] - xh * bnc[6 * F + c]);
    }
}

__device__ __forceinline__ void t_convT_point(const float (&dz0)[F], const float (&dzs)[F], const float* __restrict__ w, float (&dh)[F]) {
    // dh[ci] = sum_co w[co][ci][1] dz0[co] + w[co][ci][0] dzs[co]      (dz0 at t, dzs at t + d)
#pragma unroll
    for (int ci = 0; ci < F; ++ci) {
        float acc = 0.f;
#pragma unroll
        for (int co = 0; co < F; ++co) {
            acc = fmaf(w[(co * F + ci) * 2 + 1], dz0[co], acc);
            acc = fmaf(w[(co * F + ci) * 2 + 0], dzs[co], acc);
        }
        dh[ci] = acc;
    }
}

// conv_block2 backward: weight gradient partials, gsum0 = (convT2(dz2) + gsum) * [o0 > 0], BatchNorm (conv_block1) backward sums
__global__ __launch_bounds__(256) void t_conv2_bwd_kernel(const float* __restrict__ gsum, const float* __restrict__ z2, const float* __restrict__ o0,
                                                          const float* __restrict__ z1, const float* __restrict__ prm_l,
                                                          float* __restrict__ gsum0, float* __restrict__ gpart, int bn_index, TTrain a) {
    __shared__ __attribute__((aligned(16))) float tile[4][TTR * TTS];
    __shared__ float lds[4 * 2 * F];
    __shared__ float bnc2[7 * F], bnc1[7 * F];
    __shared__ float red[8 * 64];
    t_bn_consts(a.cells_fwd, a.cells_bwd, prm_l, a.N, a.L, 1, bn_index, a.cnt, true, bnc2, 0);
    t_bn_consts(a.cells_fwd, a.cells_bwd, prm_l, a.N, a.L, 0, bn_index - 1, a.cnt, false, bnc1, 64);
    __syncthreads();
    const int N = a.N;
    const int64_t total = a.B * a.N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float sa[F], sb[F];
#pragma unroll
    for (int c = 0; c < F; ++c) sa[c] = sb[c] = 0.f;
    f32x4tt acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int64_t base = (int64_t)blockIdx.x * 256; base < total; base += (int64_t)gridDim.x * 256) {       // (workgroup-uniform trip count)
    const int64_t i = base + threadIdx.x;
    const bool ok = i < total;
    const int64_t b = ok ? i / N : 0;
    const int t = ok ? (int)(i - b * N) : 0;
    float dz[F], dzs[F], h[F], hs[F];
#pragma unroll
    for (int c = 0; c < F; ++c) dz[c] = dzs[c] = h[c] = hs[c] = 0.f;
    if (ok) {
        t_dz2_at(gsum, z2, bnc2, b, t, N, dz);
        t_dz2_at(gsum, z2, bnc2, b, t + 2, N, dzs);
        t_ld10(o0, b, t, N, h);
        t_ld10(o0, b, t - 2, N, hs);
    }
    t_wgrad_mfma(tile[wave], dz, h, hs, lane, acc0, acc1);
    if (ok) {
        float d_o0[F];
        t_convT_point(dz, dzs, prm_l + off_conv_w(N, 1), d_o0);
#pragma unroll
        for (int c = 0; c < F; ++c) {
            const int64_t idx = (b * F + c) * N + t;
            float g = d_o0[c] + gsum[idx];
            g = (h[c] > 0.f) ? g : 0.f;                          // h = o0
            gsum0[idx] = g;
            const float zz = z1[idx];
            const float x0 = relu(fmaf(zz, bnc1[2 * F + c], bnc1[3 * F + c]));
            const float dy = (x0 > 0.f) ? g : 0.f;
            sa[c] += dy;
            sb[c] = fmaf(dy, (zz - bnc1[0 * F + c]) * bnc1[1 * F + c], sb[c]);
        }
    }
    }
    t_pair_reduce(sa, sb, a.cells_bwd + (blockIdx.x % T_REP) * tc_sb(a.L) + (bn_index - 1) * 2 * F, lds);
    __syncthreads();
    t_wgrad_store(red, acc0, acc1, gpart + (size_t)blockIdx.x * CONVW);
}

__device__ __forceinline__ void t_dz1_at(const float* __restrict__ gsum0, const float* __restrict__ z1, const float* bnc, int64_t b, int t, int N,
                                         float (&dz)[F]) {
    if (t < 0 || t >= N) {
#pragma unroll
        for (int c = 0; c < F; ++c) dz[c] = 0.f;
        return;
    }
    float zv[F], gv[F];                     // (loaded up front: see t_dz2_at)
#pragma unroll
    for (int c = 0; c < F; ++c) {
        zv[c] = z1[(b * F + c) * N + t];
        gv[c] = gsum0[(b * F + c) * N + t];
    }
#pragma unroll
    for (int c = 0; c < F; ++c) {
        const float zz = zv[c];
        const float x0 = relu(fmaf(zz, bnc[2 * F + c], bnc[3 * F + c]));
        const float dy = (x0 > 0.f) ? gv[c] : 0.f;
        const float xh = (zz - bnc[0 * F + c]) * bnc[1 * F + c];
        dz[c] = bnc[4 * F + c] * (dy - bnc[5 * F + c] - xh * bnc[6 * F + c]);
    }
}

// conv_block1 backward: weight gradient partials, dHpre = (convT1(dz1) + gsum0) * leaky'(H)
__global__ __launch_bounds__(256) void t_conv1_bwd_kernel(const float* __restrict__ gsum0, const float* __restrict__ z1, const float* __restrict__ Hpre,
                                                          const float* __restrict__ prm_l, float* __restrict__ dHpre, float* __restrict__ gpart,
                                                          int bn_index, TTrain a, float* amax_dh) {
    __shared__ __attribute__((aligned(16))) float tile[4][TTR * TTS];
    __shared__ float bnc[7 * F];
    __shared__ float red[8 * 64];
    t_bn_consts(a.cells_fwd, a.cells_bwd, prm_l, a.N, a.L, 0, bn_index, a.cnt, true, bnc);
    __syncthreads();
    const int N = a.N;
    const int64_t total = a.B * a.N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4tt acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    float dhmax = 0.f;
    for (int64_t base = (int64_t)blockIdx.x * 256; base < total; base += (int64_t)gridDim.x * 256) {
    const int64_t i = base + threadIdx.x;
    const bool ok = i < total;
    const int64_t b = ok ? i / N : 0;
    const int t = ok ? (int)(i - b * N) : 0;
    float dz[F], dzs[F], h[F], hs[F];
#pragma unroll
    for (int c = 0; c < F; ++c) dz[c] = dzs[c] = h[c] = hs[c] = 0.f;
    if (ok) {
        t_dz1_at(gsum0, z1, bnc, b, t, N, dz);
        t_dz1_at(gsum0, z1, bnc, b, t + 1, N, dzs);
        t_load_H(Hpre, prm_l + off_theta_b(N), b, t, N, h);               // (H = leaky(Hpre + bias): not stored, t_conv1_train_kernel)
        t_load_H(Hpre, prm_l + off_theta_b(N), b, t - 1, N, hs);
    }
    t_wgrad_mfma(tile[wave], dz, h, hs, lane, acc0, acc1);
    if (ok) {
        float dH[F];
        t_convT_point(dz, dzs, prm_l + off_conv_w(N, 0), dH);
#pragma unroll
        for (int c = 0; c < F; ++c) {
            const int64_t idx = (b * F + c) * N + t;
            const float g = dH[c] + gsum0[idx];
            const float dv = h[c] > 0.f ? g : g * LEAKY;
            dHpre[idx] = dv;
            dhmax = fmaxf(dhmax, t_finite_abs(dv));
        }
    }
    }
    if (amax_dh) t_amax_store(dhmax, amax_dh, red);
    __syncthreads();
    t_wgrad_store(red, acc0, acc1, gpart + (size_t)blockIdx.x * CONVW);
}

// finalize (tiled path): conv weight partial rows, BatchNorm gamma/beta from the cells, loss, batch statistics
struct TFin {
    const float* gpart;       // [2L][grid][200]
    const double* cells_fwd;
    const double* cells_bwd;
    float* grads;
    float* loss;
    float* bn_batch;
    int grid, N, L;
    int64_t global_batch;
    double cnt;
    float moment_weight;
    int write_loss, write_grads;
};
constexpr int TFIN_SUB = (CONVW + 31) / 32;     // workgroups per (layer, conv block): 32 weights each
__global__ __launch_bounds__(1024) void t_finalize_kernel(TFin f) {
    const int N = f.N, L = f.L, LS = layer_stride(N);
    // (layer, conv block) x sub: 200 conv weights -- lanes along the weights (the partial rows are read in 128-byte pieces), 32 row slices
    // across the workgroup with four loads in flight each, combined through LDS in a fixed order (a wavefront per weight with its lanes
    // striding over the rows was 18 us at the end of every step of the XJTU-SY wiring; one thread walking all rows 1.5 ms) -- and, in sub 0,
    // the 20 BatchNorm parameters
    __shared__ float red[32][33];
    const int bnidx = blockIdx.x / TFIN_SUB, sub = blockIdx.x % TFIN_SUB;
    const int l = bnidx / 2, blk = bnidx % 2;
    const float* gp = f.gpart + (size_t)bnidx * f.grid * CONVW;
    if (f.write_grads) {
        const int col = threadIdx.x & 31, sl = threadIdx.x >> 5;
        const int p = sub * 32 + col;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (p < CONVW) {
            const float* q = gp + p;
            int r = sl;
            for (; r + 96 < f.grid; r += 128) {
                a0 += q[(size_t)r * CONVW];
                a1 += q[(size_t)(r + 32) * CONVW];
                a2 += q[(size_t)(r + 64) * CONVW];
                a3 += q[(size_t)(r + 96) * CONVW];
            }
            for (; r < f.grid; r += 32) a0 += q[(size_t)r * CONVW];
        }
        red[sl][col] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (sl == 0 && p < CONVW) {
            float v = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < 32; ++s2) v += red[s2][col];
            f.grads[l * LS + off_conv_w(N, blk) + p] = v;
        }
    }
    if (sub != 0) return;
    if (threadIdx.x < F) {
        const int c = threadIdx.x;
        if (f.write_grads) {
            f.grads[l * LS + off_bn_g(N, blk) + c] = (float)tc_sum(f.cells_bwd, tc_sb(L), (bnidx * 2 + 1) * F + c);
            f.grads[l * LS + off_bn_b(N, blk) + c] = (float)tc_sum(f.cells_bwd, tc_sb(L), (bnidx * 2 + 0) * F + c);
        }
        const double mean = tc_sum(f.cells_fwd, tc_sf(L), (bnidx * 2 + 0) * F + c) / f.cnt;
        const double ex2 = tc_sum(f.cells_fwd, tc_sf(L), (bnidx * 2 + 1) * F + c) / f.cnt;
        double var = ex2 - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        if (f.moment_weight > 0.f) {
            f.bn_batch[(bnidx * 2 + 0) * F + c] = (float)(mean * (double)f.moment_weight);
            f.bn_batch[(bnidx * 2 + 1) * F + c] = (float)(ex2 * (double)f.moment_weight);
        } else {
            f.bn_batch[(bnidx * 2 + 0) * F + c] = (float)mean;
            f.bn_batch[(bnidx * 2 + 1) * F + c] = (float)var;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && f.write_loss)
        f.loss[0] = (float)(tc_sum(f.cells_bwd, tc_sb(L), tc_sf(L)) / (double)f.global_batch);
    if (blockIdx.x == 0 && threadIdx.x == 0 && f.write_grads) f.grads[off_fc2_b(N, L)] = (float)tc_sum(f.cells_bwd, tc_sb(L), tc_sf(L) + 1);
}

// ------------------------------------------------------------------------------------------------
// host: training forward / backward
// ------------------------------------------------------------------------------------------------
struct TWs {
    size_t T, total;
    size_t off_X, off_AX, off_H, off_z1, off_o0, off_z2;      // per layer, L (+1 for X) tensors each; off_H holds Hpre = theta(A.X) (H is re-derived)
    size_t off_gsum, off_gsum0, off_dH, off_dAX, off_dX;
    size_t off_A, off_pooled, off_y1pre, off_y1, off_dy1, off_dpool, off_dpred, off_cells, off_gpart, off_one, off_split, off_split2;
    size_t off_planes, off_planes2, plane_bytes;      // pre-split operand planes of the large products: main stream, side stream
    size_t off_amax;            // [3 L + 3][T_AMAX_MAX] floats: partial maxima of |A.X_l|, |theta_l| and |fc1.weight|, |d Hpre_l|, |pooled|, |d y1|
                                // (the operand scales of the GEMMs)
    size_t cells_bytes;
    size_t split_floats;
    int grid;
};
// Small wirings (PHM2012 c2: 160 patches): every parameter-gradient product of the step -- fc2.weight, fc1.weight, fc1.bias, theta and its
// bias of every layer -- is a few 64 x 64 tiles over a long reduction, a launch pair of 5-10 us each at its latency floor: 14 launches of
// a 44-launch step.  They feed nothing in the call and run as ONE split-K launch + ONE reduction behind the backward chain
// (sgemm_splitk_batch; 3 + 2 L jobs).  Larger wirings keep the scaled 128 / 256-tile kernels and the side stream.
static inline bool t_pgrad_batched(int N, int L) { return N <= 256 && 3 + 2 * L <= 10; }
// (M, N, K) of the jobs in launch order: fc2.weight, fc1.weight, fc1.bias, then (theta, theta bias) of layers L-1 .. 0
static int t_pgrad_dims(int N, int L, int Bi, int R, SplitKJob* j) {
    int n = 0;
    auto add = [&](int M, int Nn, int K) { j[n] = SplitKJob{}; j[n].M = M; j[n].N = Nn; j[n].K = K > 0 ? K : 1; ++n; };
    add(1, N, Bi); add(N, N, Bi); add(1, N, Bi);
    for (int l = 0; l < L; ++l) { add(N, N, R); add(1, N, R); }
    return n;
}

static void tws_layout(const rulgnn_stgcn_shape* s, TWs* w) {
    const int N = s->num_patch, L = s->num_layers;
    const int64_t B = s->batch;
    w->T = al256((size_t)B * F * N * sizeof(float));
    w->grid = t_pgrid(B * N);
    size_t o = 0;
    w->off_X = o; o += (size_t)(L + 1) * w->T;
    w->off_AX = o; o += (size_t)L * w->T;
    w->off_H = o; o += (size_t)L * w->T;
    w->off_z1 = o; o += (size_t)L * w->T;
    w->off_o0 = o; o += (size_t)L * w->T;
    w->off_z2 = o; o += (size_t)L * w->T;
    w->off_gsum = o; o += w->T;
    w->off_gsum0 = o; o += w->T;
    w->off_dH = o; o += (size_t)L * w->T;          // per layer: the side stream's d theta product of layer l reads it while layer l - 1 runs
    w->off_dAX = o; o += w->T;
    w->off_dX = o; o += w->T;
    const size_t BNb = al256((size_t)B * N * 4);
    w->off_A = o; o += al256((size_t)B * F * F * 4);
    w->off_pooled = o; o += BNb;
    w->off_y1pre = o; o += BNb;
    w->off_y1 = o; o += BNb;
    w->off_dy1 = o; o += BNb;
    w->off_dpool = o; o += BNb;
    w->off_dpred = o; o += al256((size_t)B * 4);
    w->cells_bytes = sizeof(double) * (size_t)T_REP * (tc_sf(L) + tc_sb(L));
    w->off_cells = o; o += al256(w->cells_bytes);
    w->off_gpart = o; o += al256((size_t)2 * L * w->grid * CONVW * 4);
    w->off_one = o; o += 256;
    w->off_amax = o; o += al256((size_t)(3 * L + 3) * T_AMAX_MAX * sizeof(float));
    // partial products of the split-K weight / bias gradient GEMMs (reductions over batch * 10 or batch rows)
    {
        const int R = (int)(B * F), Bi = (int)B;
        size_t need = 1024;
        if (B > 0)
            for (size_t v : {sgemm_splitk_need_floats(N, N, R), sgemm_splitk_need_floats(1, N, R), sgemm_splitk_need_floats(N, N, Bi),
                             sgemm_splitk_need_floats(1, N, Bi), sgemm_splitk_need_floats(1, 1, Bi), sgemm_splitk_need_floats(Bi, N, N),
                             R < 2048 ? sgemm_splitk_need_floats(R, N, N) : (size_t)0})
                need = v > need ? v : need;
        // (... or every parameter-gradient product of a step at once: t_pgrad_batched)
        if (B > 0 && t_pgrad_batched(N, L)) {
            SplitKJob dims[3 + 2 * 8];
            const int nj = t_pgrad_dims(N, L, Bi, R, dims);
            const size_t bf = sgemm_splitk_batch_floats(dims, nj);
            need = bf > need ? bf : need;
        }
        w->split_floats = need;
        w->off_split = o; o += al256(need * sizeof(float));
        w->off_split2 = o; o += al256(need * sizeof(float));          // the side stream's own (parameter-gradient products)
    }
    w->plane_bytes = t_plane_bytes(N, B * F, false);
    w->off_planes = o; o += w->plane_bytes;
    w->off_planes2 = o;                                                  // (the side stream's products stay on the in-loop split: no second area)
    w->total = o;
}

size_t stgcn_tiled_train_workspace_bytes(const rulgnn_stgcn_shape* s) {
    TWs w;
    tws_layout(s, &w);
    return w.total;
}


int stgcn_tiled_train(const rulgnn_stgcn_shape* s, const rulgnn_stgcn_train_args* ar, int mode /* 0 fwd, 1 bwd, 2 both */,
                      hipStream_t stream, const GradReadyHook* ready) {
    TWs w;
    tws_layout(s, &w);
    if (ar->workspace_bytes < w.total) return RULGNN_EWORKSPACE;
    const int N = s->num_patch, L = s->num_layers, LS = layer_stride(N);
    const int64_t B = s->batch, BN_ = B * N;
    char* ws = static_cast<char*>(ar->workspace);
    auto TP = [&](size_t off, int l) { return reinterpret_cast<float*>(ws + off + (size_t)l * w.T); };
    float* gsum = TP(w.off_gsum, 0); float* gsum0 = TP(w.off_gsum0, 0);
    float* dAX = TP(w.off_dAX, 0); float* dX = TP(w.off_dX, 0);
    float* A = reinterpret_cast<float*>(ws + w.off_A);
    float* pooled = reinterpret_cast<float*>(ws + w.off_pooled);
    float* y1pre = reinterpret_cast<float*>(ws + w.off_y1pre);
    float* y1 = reinterpret_cast<float*>(ws + w.off_y1);
    float* dy1 = reinterpret_cast<float*>(ws + w.off_dy1);
    float* dpool = reinterpret_cast<float*>(ws + w.off_dpool);
    float* dpredb = reinterpret_cast<float*>(ws + w.off_dpred);
    double* cells = reinterpret_cast<double*>(ws + w.off_cells);
    float* gpart = reinterpret_cast<float*>(ws + w.off_gpart);
    float* one = reinterpret_cast<float*>(ws + w.off_one);
    float* split = reinterpret_cast<float*>(ws + w.off_split);
    float* split2 = reinterpret_cast<float*>(ws + w.off_split2);
    void* planes = w.plane_bytes ? static_cast<void*>(ws + w.off_planes) : nullptr;
    void* planes2 = nullptr;
    // Which of the step's products run on pre-split operands (csrc/sgemm_planes.hip), measured at XJTU-SY batch 1024 on one box
    // (tools/time_tiled_step.py): none 1.1945 ms; theta(A.X) 1.1878; + d(A.X) 1.1857 (kept); + d theta 1.2109; + the fc1 products 1.2192.
    // The product kernel alone is 64 us against 105-111 (324 against 193 TFLOP/s), but every operand costs a split pass (17 us per
    // [10 240 x 1024] activation: 42 MB in, 42 MB out), d theta needs BOTH operands transposed, and the 256-workgroup / 156-KB kernel
    // leaves the side stream's product no CU to run beside it (the 160-workgroup round-5 kernel does).
    const size_t plane_bytes = w.plane_bytes;
    const size_t pb_fwd = plane_bytes, pb_dax = plane_bytes, pb_dth = 0, pb_fc = 0;
    const float* prm = ar->params;
    // Parameter-gradient products (d fc1 / fc2, d theta of every layer and their bias sums: ~290 us of matrix-core work per step at XJTU-SY
    // batch 1024) feed nothing downstream in this call: with a second stream of the caller (args->aux_stream) they run BESIDE the
    // position-parallel chain, which is bound by HBM while they are bound by the matrix pipe and LDS.  Not with a gradient-ready callback
    // (data parallel with overlap): its contract orders the caller's all-reduces against `stream`.
    // ... nor at batch-100-sized row counts: there every launch sits at its latency floor and the fork / join events cost more than the
    // overlap returns (XJTU-SY batch 100: 0.38 -> 0.40 ms with it, batch 1024: 1.24 -> 1.22 ms).
    // (every parameter-gradient product as one launch pair at the end of the chain where the wiring is small: no side stream then)
    const bool pg_batch = !ready && t_pgrad_batched(N, L);
    SplitKJob pjobs[3 + 2 * 8];
    int npj = 0;
    auto pjob = [&](const float* A, int64_t sAm, int64_t sAk, const float* Bm, int64_t sBn, int64_t sBk, float* C, int64_t ldc, int M, int Nn, int K) {
        pjobs[npj] = SplitKJob{A, sAm, sAk, Bm, sBn, sBk, C, ldc, M, Nn, K};
        ++npj;
    };
    AuxFork fk(stream, (ready || pg_batch || B * F < 2048) ? nullptr : ar->aux_stream);
    hipStream_t wst = fk.side();
    // operand scales of the large GEMMs: partial maxima rows, one float per workgroup of the producing launch.  The scaled form only where
    // the producers' grids fit a row (every reference wiring does: <= 4096 chunks of 256 positions)
    float* amax = reinterpret_cast<float*>(ws + w.off_amax);
    const int n_pos = (int)((BN_ + 255) / 256), n_dh = t_pgrid(BN_), n_th = 256;
    const bool scaled = n_pos <= T_AMAX_MAX;
    const int n_ax0 = (scaled && B <= T_AMAX_MAX) ? (int)B : 0;            // layer 0's row is written per SAMPLE by the Gram kernel (else per chunk)
    auto am_ax = [&](int l) { return scaled ? amax + (size_t)(0 * L + l) * T_AMAX_MAX : (float*)nullptr; };
    auto am_th = [&](int l) { return scaled ? amax + (size_t)(1 * L + l) * T_AMAX_MAX : (float*)nullptr; };          // (l = L: fc1.weight)
    auto am_dh = [&](int l) { return scaled ? amax + (size_t)(2 * L + 1 + l) * T_AMAX_MAX : (float*)nullptr; };
    float* am_pool = scaled ? amax + (size_t)(3 * L + 1) * T_AMAX_MAX : (float*)nullptr;                             // n_pos entries (tail kernel)
    float* am_dy1 = (scaled && B <= T_AMAX_MAX) ? amax + (size_t)(3 * L + 2) * T_AMAX_MAX : (float*)nullptr;         // B entries (head kernel)
    // rows of [batch * 10, N] too few for the large tiles (the reference protocol's batch of 100): the theta products through the split-K
    // pair as well, like fc1
    const bool few_rows = B * F < 2048;

    TArgs a{B, N, s->patch_size, L};
    TTrain t;
    t.B = B; t.sample_offset = ar->sample_offset; t.global_batch = ar->global_batch; t.N = N; t.P = s->patch_size; t.L = L;
    t.dropout_p = ar->dropout_p;
    t.drop_scale = ar->dropout_p > 0.f ? 1.0f / (1.0f - ar->dropout_p) : 1.0f;
    {
        double thr = (double)ar->dropout_p * 4294967296.0;
        const uint64_t ti = (uint64_t)((thr < 0 ? 0 : thr) + 0.5);
        t.drop_thr = ti > 4294967295ull ? 4294967295u : (uint32_t)ti;
    }
    t.drop_key = 0;
    t.key_dev = nullptr;
    const StepState* sstate = static_cast<const StepState*>(ar->step_state);
    t.cnt = (double)B * (double)N;
    t.cells_fwd = cells; t.cells_bwd = cells + T_REP * tc_sf(L);
    const int has_dpred = ar->dpred ? 1 : (ar->y ? 0 : 2);
    const float* gy = ar->dpred ? ar->dpred : ar->y;
    int rc = RULGNN_OK;

    if (mode == 0 || mode == 2) {
        if (ar->step_state) {
            rc = step_prepare_dropout(ar->step_state, ar->seed, L, stream);
            if (rc != RULGNN_OK) return rc;
        }
        if (!scaled && hipMemsetAsync(cells, 0, w.cells_bytes, stream) != hipSuccess) return RULGNN_EHIP;
        if (scaled) {
            (void)hipGetLastError();
            // (theta of every layer and, as "layer L" of the same stride, fc1.weight)
            static_assert(off_theta_w(1) == 0 && off_fc1_w(7, 3) == 3 * layer_stride(7), "fc1.weight sits where a layer-L theta would");
            hipLaunchKernelGGL(t_absmax_kernel, dim3(n_th, L + 1), dim3(256), 0, stream, prm, (int64_t)LS, (int64_t)off_theta_w(N), (int64_t)N * N, am_th(0),
                               cells, (int)(w.cells_bytes / sizeof(double)));
            if (hipGetLastError() != hipSuccess) return RULGNN_EHIP;
        }
        T_STATS(ar->x, TP(w.off_X, 0));
        (void)hipGetLastError();
        // (+ the first layer's aggregation and its operand-scale row, one entry per sample, where the batch fits a row)
        const bool agg0 = scaled && B <= T_AMAX_MAX;
        hipLaunchKernelGGL(t_gram_kernel, dim3((unsigned)B), dim3(256), 0, stream, TP(w.off_X, 0), A, a, agg0 ? TP(w.off_AX, 0) : (float*)nullptr,
                           agg0 ? am_ax(0) : (float*)nullptr);
        if (hipGetLastError() != hipSuccess) return RULGNN_EHIP;
        for (int l = 0; l < L; ++l) {
            const float* pl = prm + l * LS;
            t.drop_key = dropout_layer_key(ar->seed, ar->step, l);
            t.key_dev = sstate ? &sstate->drop_key[l] : nullptr;
            if (l == 0 && !agg0) T_LAUNCH(t_aggregate_kernel, BN_, A, TP(w.off_X, l), (const float*)nullptr, TP(w.off_AX, l), a, am_ax(0));
            float* const Hpl = TP(w.off_H, l);             // the layer's slot keeps the product's output: H = leaky(Hpre + bias) is re-derived by its readers
            if (few_rows)
                rc = sgemm_splitk(TP(w.off_AX, l), N, 1, pl + off_theta_w(N), N, 1, Hpl, N, (int)(B * F), N, N, false, split, stream, am_ax(l),
                                  l == 0 && agg0 ? (int)B : n_pos, am_th(l), n_th);
            else
            rc = sgemm(TP(w.off_AX, l), N, 1, pl + off_theta_w(N), N, 1, Hpl, N, (int)(B * F), N, N, false, stream, 0, am_ax(l),
                       l == 0 && agg0 ? (int)B : n_pos, am_th(l), n_th, pb_fwd ? planes : nullptr, pb_fwd);
            if (rc != RULGNN_OK) return rc;
            T_LAUNCH_P(t_conv1_train_kernel, BN_, Hpl, pl, TP(w.off_z1, l), 2 * l, t);
            T_LAUNCH_P(t_conv2_train_kernel, BN_, TP(w.off_z1, l), Hpl, pl, TP(w.off_o0, l), TP(w.off_z2, l), 2 * l + 1, t);
            // (+ the next layer's A.X, or the channel max-pool behind the last layer)
            T_LAUNCH(t_tail_train_kernel, BN_, TP(w.off_z2, l), TP(w.off_o0, l), TP(w.off_X, l), pl, TP(w.off_X, l + 1), 2 * l + 1, t,
                     (const float*)A, l + 1 < L ? TP(w.off_AX, l + 1) : (float*)nullptr, l + 1 < L ? (float*)nullptr : pooled,
                     l + 1 < L ? am_ax(l + 1) : am_pool);
        }
        // (through the split-K pair: [batch x N] has too few output tiles to fill the chip -- at XJTU batch 1024, 64 tiles of 128 x 128)
        rc = sgemm_splitk(pooled, N, 1, prm + off_fc1_w(N, L), N, 1, y1pre, N, (int)B, N, N, false, split, stream, am_pool, n_pos, am_th(L), n_th,
                          pb_fc ? planes : nullptr, pb_fc);
        if (rc != RULGNN_OK) return rc;
    } else {
        if (hipMemsetAsync(t.cells_bwd, 0, sizeof(double) * T_REP * tc_sb(L), stream) != hipSuccess) return RULGNN_EHIP;
    }
    // head (also recomputed by a backward-only call: cheap, gives dpred for the incoming gradient)
    (void)hipGetLastError();
    hipLaunchKernelGGL(t_head_train_kernel, dim3((unsigned)B), dim3(256), 0, stream, y1pre, prm, gy, has_dpred, y1, ar->pred, dpredb,
                       dy1, t, one, am_dy1);
    if (hipGetLastError() != hipSuccess) return RULGNN_EHIP;

    if (mode != 0) {
        float* g = ar->grads;
        fk.fork();
        // fc2: dW2[j] = sum_b dpred[b] y1[b][j];  db2 = sum_b dpred[b]
        if (pg_batch) pjob(dpredb, 0, 1, y1, 1, N, g + off_fc2_w(N, L), N, 1, N, (int)B);
        else rc = sgemm_splitk(dpredb, 0, 1, y1, 1, N, g + off_fc2_w(N, L), N, 1, N, (int)B, false, split2, wst);
        if (rc != RULGNN_OK) return rc;
        // (db2 = sum_b dpred[b]: accumulated by the head kernel, written by the finalize kernel)
        // fc1: dW1[j][t] = sum_b dy1[b][j] pooled[b][t];  db1[j] = sum_b dy1[b][j];  dpooled = dy1 . W1
        if (pg_batch) {
            pjob(dy1, 1, N, pooled, 1, N, g + off_fc1_w(N, L), N, N, N, (int)B);
            pjob(one, 0, 0, dy1, 1, N, g + off_fc1_b(N, L), N, 1, N, (int)B);
        } else {
        rc = sgemm_splitk(dy1, 1, N, pooled, 1, N, g + off_fc1_w(N, L), N, N, N, (int)B, false, split2, wst, am_dy1, (int)B,
                          am_dy1 ? am_pool : (float*)nullptr, n_pos, pb_fc ? planes2 : nullptr, pb_fc);
        if (rc != RULGNN_OK) return rc;
        rc = sgemm_splitk(one, 0, 0, dy1, 1, N, g + off_fc1_b(N, L), N, 1, N, (int)B, false, split2, wst);
        if (rc != RULGNN_OK) return rc;
        }
        // data parallel with overlap: the head's gradients (fc1 is N x N: 4 MB at XJTU-SY) are final here, with the whole layer
        // stack still to run -- the caller may start their all-reduce on another stream (include/rulgnn.h: rulgnn_grad_ready_fn)
        // (without fc2.bias, the last parameter: its gradient comes out of the finalize kernel with the convolution / BatchNorm ones)
        if (ready && ready->fn(ready->user, g, (int64_t)off_fc1_w(N, L), (int64_t)(off_fc2_b(N, L) - off_fc1_w(N, L)), wst) != 0)
            return RULGNN_ECALLBACK;
        rc = sgemm_splitk(dy1, N, 1, prm + off_fc1_w(N, L), 1, N, dpool, N, (int)B, N, N, false, split, stream, am_dy1, (int)B,
                          am_dy1 ? am_th(L) : (float*)nullptr, n_th, pb_fc ? planes : nullptr, pb_fc);
        if (rc != RULGNN_OK) return rc;
        for (int l = L - 1; l >= 0; --l) {
            const float* pl = prm + l * LS;
            float* gl = g + l * LS;
            t.drop_key = dropout_layer_key(ar->seed, ar->step, l);
            t.key_dev = sstate ? &sstate->drop_key[l] : nullptr;
            T_LAUNCH_P(t_tail_bwd_kernel, BN_, dpool, TP(w.off_X, l + 1), dX, TP(w.off_z2, l), TP(w.off_o0, l), pl, gsum, 2 * l + 1,
                     l == L - 1 ? 1 : 0, t);
            T_LAUNCH_P(t_conv2_bwd_kernel, BN_, gsum, TP(w.off_z2, l), TP(w.off_o0, l), TP(w.off_z1, l), pl, gsum0,
                     gpart + (size_t)(2 * l + 1) * w.grid * CONVW, 2 * l + 1, t);
            float* dHp = TP(w.off_dH, l);
            T_LAUNCH_P(t_conv1_bwd_kernel, BN_, gsum0, TP(w.off_z1, l), TP(w.off_H, l), pl, dHp,
                     gpart + (size_t)(2 * l) * w.grid * CONVW, 2 * l, t, am_dh(l));
            fk.fork();
            // (the main stream's next product first: the host enqueues in program order.  The side stream's d theta product then runs
            // beside d(A.X); started behind it instead -- beside the HBM-bound kernels of the layer below -- the step is 4 % SLOWER)
            if (l > 0) {
                if (few_rows)
                    rc = sgemm_splitk(dHp, N, 1, pl + off_theta_w(N), 1, N, dAX, N, (int)(B * F), N, N, false, split, stream, am_dh(l), n_dh, am_th(l), n_th);
                else
                rc = sgemm(dHp, N, 1, pl + off_theta_w(N), 1, N, dAX, N, (int)(B * F), N, N, false, stream, 0, am_dh(l), n_dh, am_th(l), n_th,
                           pb_dax ? planes : nullptr, pb_dax);      // dHpre . theta
                if (rc != RULGNN_OK) return rc;
            }
            // theta: dW[j][k] = sum_r dHpre[r][j] AX[r][k];  db[j] = sum_r dHpre[r][j]
            if (pg_batch) {
                pjob(dHp, 1, N, TP(w.off_AX, l), 1, N, gl + off_theta_w(N), N, N, N, (int)(B * F));
                pjob(one, 0, 0, dHp, 1, N, gl + off_theta_b(N), N, 1, N, (int)(B * F));
            } else {
            rc = sgemm_splitk(dHp, 1, N, TP(w.off_AX, l), 1, N, gl + off_theta_w(N), N, N, N, (int)(B * F), false, split2, wst, am_dh(l), n_dh, am_ax(l),
                              l == 0 && n_ax0 > 0 ? n_ax0 : n_pos, pb_dth ? planes2 : nullptr, pb_dth);
            if (rc != RULGNN_OK) return rc;
            // (the first layer's bias sums on the MAIN stream: it has nothing left to do there while the side stream works through that
            // layer's d theta, the last product of the step)
            if (l == 0) rc = sgemm_splitk(one, 0, 0, dHp, 1, N, gl + off_theta_b(N), N, 1, N, (int)(B * F), false, split, stream);
            else rc = sgemm_splitk(one, 0, 0, dHp, 1, N, gl + off_theta_b(N), N, 1, N, (int)(B * F), false, split2, wst);
            if (rc != RULGNN_OK) return rc;
            }
            // theta of this layer (weight + bias, contiguous at the head of the layer's block) is final; the convolution and
            // BatchNorm gradients behind it (2 x 220 floats) come out of t_finalize_kernel at the end of the step
            if (ready && l > 0 && ready->fn(ready->user, g, (int64_t)l * LS + off_theta_w(N), (int64_t)N * N + N, wst) != 0)
                return RULGNN_ECALLBACK;
            if (l > 0) {
                // (A^T dAX + dXn, A symmetric.  Folded into the tail kernel of the layer below it cost more there -- 100 adjacency loads and
                // 100 FMAs per position inside the persistent loop: +22 us -- than this launch takes)
                if (l == L - 1) T_LAUNCH(t_aggregate_kernel, BN_, A, dAX, (const float*)nullptr, dX, a, (float*)nullptr, (const float*)dpool, (const float*)TP(w.off_X, L));
                else T_LAUNCH(t_aggregate_kernel, BN_, A, dAX, dX, dX, a, (float*)nullptr);
            }
        }
    }
    if (npj > 0) {
        rc = sgemm_splitk_batch(pjobs, npj, split2, w.split_floats, stream);
        if (rc != RULGNN_OK) return rc;
    }
    TFin f;
    f.gpart = gpart; f.cells_fwd = t.cells_fwd; f.cells_bwd = t.cells_bwd;
    f.grads = ar->grads; f.loss = ar->loss; f.bn_batch = ar->bn_batch;
    f.grid = w.grid; f.N = N; f.L = L; f.global_batch = ar->global_batch; f.cnt = t.cnt;
    f.moment_weight = ar->bn_moment_weight;
    f.write_loss = (has_dpred == 0) && ar->loss && mode != 0;
    f.write_grads = mode != 0;          // forward only: just the batch statistics for the running-stat update
    (void)hipGetLastError();
    hipLaunchKernelGGL(t_finalize_kernel, dim3(2 * L * TFIN_SUB), dim3(1024), 0, stream, f);
    if (hipGetLastError() != hipSuccess) return RULGNN_EHIP;
    // (the finalize kernel reads the chain's partial rows and cells only: it runs beside the side stream's last products; the parameter
    // gradients are final in `stream` order from here)
    return fk.join();
}

}  // namespace rulgnn
