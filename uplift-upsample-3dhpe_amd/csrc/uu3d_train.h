// uu3d_train.h -- training-step kernels that need no back-propagation through the network:
// the MPJPE loss and its gradient w.r.t. the predictions (T1), the fused AdamW update with
// tensorflow-addons' decoupled-decay semantics (T3), and the EMA update (T4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

namespace uu3d {

// ---- T1: loss = w_c * sum||pred_c - gt_c|| / (BS*J) + w_s * sum||pred - gt|| / (BS*N*J) ------------
// One thread per (b, n, j) joint of the full output, plus one per (b, j) of the central output.
// Stage 1 writes one partial sum per workgroup (central and sequence separately); stage 2 adds the
// partials in index order.  Gradient of tf.norm(gt - pred): (pred - gt) / ||pred - gt|| (NaN at
// exactly zero distance, as in TensorFlow).
constexpr int kLossGrid = 1024;

static __global__ void __launch_bounds__(256)
mpjpe_loss_stage1(const float* __restrict__ pred_full, const float* __restrict__ pred_central,
                  const float* __restrict__ gt3d, const int B, const int N, const int J, const int root,
                  const float gscale_seq, const float gscale_cen,
                  float* __restrict__ grad_full, float* __restrict__ grad_central, float* __restrict__ partial)
{
    const int n_seq = (pred_full != nullptr) ? B * N * J : 0;
    const int n_cen = B * J;
    float s_seq = 0.f, s_cen = 0.f;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < n_seq + n_cen; idx += kLossGrid * 256) {
        const bool cen = idx >= n_seq;
        int b, n, j;
        if (!cen) { b = idx / (N * J); const int r = idx - b * N * J; n = r / J; j = r - n * J; }
        else { const int r = idx - n_seq; b = r / J; j = r - b * J; n = N / 2; }
        const float* g = gt3d + (((size_t)b * N + n) * J + j) * 3;
        const float* gr = gt3d + (((size_t)b * N + n) * J + root) * 3;
        const float* p = cen ? pred_central + ((size_t)b * J + j) * 3 : pred_full + (size_t)idx * 3;
        const float dx = (g[0] - gr[0]) - p[0], dy = (g[1] - gr[1]) - p[1], dz = (g[2] - gr[2]) - p[2];   // gt - pred
        const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
        if (cen) s_cen += dist; else s_seq += dist;
        float* gout = cen ? (grad_central ? grad_central + ((size_t)b * J + j) * 3 : nullptr)
                          : (grad_full ? grad_full + (size_t)idx * 3 : nullptr);
        if (gout) {
            const float sc = (cen ? gscale_cen : gscale_seq) / dist;
            gout[0] = -dx * sc; gout[1] = -dy * sc; gout[2] = -dz * sc;
        }
    }
    __shared__ float red[2][4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s_seq += __shfl_xor(s_seq, o); s_cen += __shfl_xor(s_cen, o); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s_seq; red[1][threadIdx.x >> 6] = s_cen; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partial[kLossGrid + blockIdx.x] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

static __global__ void __launch_bounds__(64)
mpjpe_loss_stage2(const float* __restrict__ partial, const float norm_seq, const float norm_cen,
                  const float w_center, const float w_seq, const int has_seq, float* __restrict__ loss_out)
{
    const int lane = threadIdx.x;
    float s_seq = 0.f, s_cen = 0.f;
    for (int i = lane; i < kLossGrid; i += 64) { s_seq += partial[i]; s_cen += partial[kLossGrid + i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s_seq += __shfl_xor(s_seq, o); s_cen += __shfl_xor(s_cen, o); }
    if (lane == 0) {
        const float central = s_cen / norm_cen;
        const float seq = has_seq ? s_seq / norm_seq : 0.f;
        loss_out[0] = has_seq ? (w_center * central) + (w_seq * seq) : (w_center + w_seq) * central;
        loss_out[1] = central;
        loss_out[2] = seq;
    }
}

// ---- T3: tfa AdamW dense update on a flat buffer --------------------------------------------------
// `#pragma clang fp contract(off)` keeps every operation separately rounded (no fma contraction), so
// the result is bit-identical to the float32 op sequence of TF's ApplyAdam functor after tfa's decay.
template <bool AMS>
__device__ __forceinline__ void adamw_one(float& var, float& m, float& v, float& vhat, const float g, const float wd,
                                          const float alpha, const float omb1, const float omb2, const float eps)
{
#pragma clang fp contract(off)
    var = var - wd * var;
    m = m + (g - m) * omb1;
    v = v + (g * g - v) * omb2;
    if (AMS) { vhat = fmaxf(vhat, v); var = var - (m * alpha) / (sqrtf(vhat) + eps); }     // TF ApplyAdamWithAmsgrad
    else var = var - (m * alpha) / (sqrtf(v) + eps);
}

template <bool AMS>
static __global__ void __launch_bounds__(256)
adamw_kernel(float* __restrict__ var, float* __restrict__ m, float* __restrict__ v, float* __restrict__ vhat, const float* __restrict__ grad,
             const long long n, const float wd, const float alpha, const float omb1, const float omb2, const float eps,
             const unsigned* __restrict__ skip)
{
    if (skip != nullptr && *skip != 0u) return;            // the backward pass flagged non-finite gradients: leave weights and moments alone
    const long long n4 = n >> 2;
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 w4 = reinterpret_cast<float4*>(var)[i], m4 = reinterpret_cast<float4*>(m)[i];
        float4 v4 = reinterpret_cast<float4*>(v)[i];
        float4 h4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (AMS) h4 = reinterpret_cast<float4*>(vhat)[i];
        const float4 g4 = reinterpret_cast<const float4*>(grad)[i];
        adamw_one<AMS>(w4.x, m4.x, v4.x, h4.x, g4.x, wd, alpha, omb1, omb2, eps);
        adamw_one<AMS>(w4.y, m4.y, v4.y, h4.y, g4.y, wd, alpha, omb1, omb2, eps);
        adamw_one<AMS>(w4.z, m4.z, v4.z, h4.z, g4.z, wd, alpha, omb1, omb2, eps);
        adamw_one<AMS>(w4.w, m4.w, v4.w, h4.w, g4.w, wd, alpha, omb1, omb2, eps);
        reinterpret_cast<float4*>(var)[i] = w4; reinterpret_cast<float4*>(m)[i] = m4; reinterpret_cast<float4*>(v)[i] = v4;
        if (AMS) reinterpret_cast<float4*>(vhat)[i] = h4;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {        // tail
        const long long i = (n4 << 2) + threadIdx.x;
        float w1 = var[i], m1 = m[i], v1 = v[i], h1 = AMS ? vhat[i] : 0.f;
        adamw_one<AMS>(w1, m1, v1, h1, grad[i], wd, alpha, omb1, omb2, eps);
        var[i] = w1; m[i] = m1; v[i] = v1;
        if (AMS) vhat[i] = h1;
    }
}

// ---- T4: ema -= (1 - decay) * (ema - w) -----------------------------------------------------------
static __global__ void __launch_bounds__(256)
ema_kernel(float* __restrict__ ema, const float* __restrict__ w, const long long n, const float one_minus_decay)
{
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
#pragma clang fp contract(off)
        const float e = ema[i];
        ema[i] = e - one_minus_decay * (e - w[i]);
    }
}

}  // namespace uu3d
