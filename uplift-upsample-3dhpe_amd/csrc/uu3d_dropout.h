// uu3d_dropout.h -- the Dropout layers of the reference in TRAINING mode (kl.Dropout: vision_transformer.py:57-58,63-67,87-90,
// 127-128,153-154; uplift_upsample_transformer.py:78-79,84-89,201,324).
//
// Keras' Dropout multiplies by a Bernoulli(1 - rate) mask scaled by 1 / (1 - rate); which elements are kept is the reference's
// private random stream and cannot be reproduced, so the mask here is a COUNTER-BASED function of (seed, site, element index):
// nothing is stored for the backward pass (it recomputes the mask), nothing has to be generated on the host, and the CPU oracle
// (oracle/dropout_oracle.py) evaluates the same integer function, which is what makes forward and gradients testable bit for bit
// against it.  keep(element) <=> u >= rate with u = (hash >> 8) * 2^-24 in [0, 1) -- the comparison Keras makes on its uniforms.
//
// Sites (one per Dropout layer instance; `index` = linear index into the tensor the layer sees, row-major):
//      1                     token_dropout          (B N, J, d_s)        after keypoint embedding + positional encoding
//     10 + 4 i + {0, 1, 3}   spatial block i        attention weights (B N, H, J, J) | projection output (B N J, d_s) | fc2 output
//    100 + 4 i + {0,1,2,3}   temporal block i       attention weights (B, H, N, N) | projection output | hidden (after ReLU) | fc2 output
//    200 + 4 j + {0,1,2,3}   strided block j        attention weights (B, H, L, L) | projection output | hidden | strided-conv output (B L_out, d_t)
// The spatial MLP has no inner dropout (TransformerBlock is built without inner_dropout there, u_u_t.py:233-235).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace uu3d {

struct DropCfg {
    float rate = 0.f;          // 0: no Dropout layer (Keras builds none for rate 0)
    float inv_keep = 1.f;      // 1 / (1 - rate)
    unsigned seed_lo = 0, seed_hi = 0, site = 0;
    __host__ __device__ bool on() const { return rate > 0.f; }
};

__host__ __device__ inline unsigned drop_hash(unsigned seed_lo, unsigned seed_hi, unsigned site, unsigned long long index) {
    unsigned h = (unsigned)index * 0x9E3779B1u + seed_lo;
    h ^= (unsigned)(index >> 32) * 0x85EBCA77u + site * 0xC2B2AE3Du + seed_hi;
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h;
}
// the factor an element is multiplied by: 1 / (1 - rate) if kept, 0 if dropped
__host__ __device__ inline float drop_factor(const DropCfg& d, unsigned long long index) {
    const float u = (float)(drop_hash(d.seed_lo, d.seed_hi, d.site, index) >> 8) * (1.0f / 16777216.0f);
    return u >= d.rate ? d.inv_keep : 0.f;
}
inline DropCfg drop_cfg(float rate, unsigned long long seed, unsigned site) {
    DropCfg d;
    if (rate > 0.f) { d.rate = rate; d.inv_keep = 1.0f / (1.0f - rate); d.seed_lo = (unsigned)seed; d.seed_hi = (unsigned)(seed >> 32); d.site = site; }
    return d;
}

// x[i] *= factor(i)  (forward: a layer output that lives in a buffer of its own; backward: the gradient with respect to it)
static __global__ void __launch_bounds__(256)
dropout_inplace_kernel(float* __restrict__ x, const long long n, const DropCfg d)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] *= drop_factor(d, (unsigned long long)i);
}
// out[i] = in[i] * factor(i)
static __global__ void __launch_bounds__(256)
dropout_copy_kernel(const float* __restrict__ in, float* __restrict__ out, const long long n, const DropCfg d)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i] * drop_factor(d, (unsigned long long)i);
}

}  // namespace uu3d
