// uu3d_spatial.h -- the whole per-frame spatial stack in ONE launch.
//
// Replaces UpliftUpsampleTransformer.spatial_transformation up to (not including)
// spatial_to_temporal_fc (uplift_upsample_transformer.py:313-330): keypoint_embedding
// Dense(2 -> d_s) + spatial_pe, `depth` vit.TransformerBlock (vision_transformer.py:176-195,
// pre-LN, 8 heads of d_h = d_s / 8, GELU(erf) MLP), spatial_norm (eps 1e-6), and the
// "(b n) p c -> b n (p c)" flatten.
//
// Mapping: one thread = one (frame, joint) token row; its d_s-wide activation stays in
// registers for the whole stack.  A 256-thread workgroup holds floor(256 / J) frames
// (15 for J = 17).  Only K and V cross threads: they are exchanged through an LDS tile
// (row stride 2*DS + 4 floats), read back as 16-byte, frame-broadcast fragments.
//
// The d_s x d_s / d_s x 2d_s products are tiny (K = 32..64) and every weight address is
// wave-uniform, so weights are read through the scalar cache (s_load) straight into the
// SGPR operand of v_fma_f32.  Each product is written either in "dot" form (loop over
// the OUTPUT index, reduce over a statically indexed register row) or in "axpy" form
// (loop over the INPUT index, update a statically indexed accumulator row), so no
// register array is ever indexed by a run-time value.  fc1 -> GELU -> fc2 and
// q -> attention -> projection are chained element by element without staging.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "uu3d_gemm.h"

namespace uu3d {

// Packed per-block weights (floats), see pack_spatial_block() in uu3d_api.hip.
template <int DS, int HS>
struct SpatialBlockLayout {
    static constexpr int ln1_g = 0;
    static constexpr int ln1_b = ln1_g + DS;
    static constexpr int wq_t = ln1_b + DS;          // [DS out][DS in]
    static constexpr int bq = wq_t + DS * DS;
    static constexpr int wk_t = bq + DS;
    static constexpr int bk = wk_t + DS * DS;
    static constexpr int wv_t = bk + DS;
    static constexpr int bv = wv_t + DS * DS;
    static constexpr int wp = bv + DS;               // [DS in][DS out]  (Keras layout)
    static constexpr int bp = wp + DS * DS;
    static constexpr int ln2_g = bp + DS;
    static constexpr int ln2_b = ln2_g + DS;
    static constexpr int w1_t = ln2_b + DS;          // [HS out][DS in]
    static constexpr int b1 = w1_t + HS * DS;
    static constexpr int w2 = b1 + HS;               // [HS in][DS out]  (Keras layout)
    static constexpr int b2 = w2 + HS * DS;
    static constexpr int size = b2 + DS;
};

__host__ __device__ inline constexpr size_t spatial_lds_bytes(int DS) { return (size_t)256 * (2 * DS + 4) * sizeof(float); }

struct SpatialParams {
    const float* __restrict__ embed_w;   // (2, DS)
    const float* __restrict__ embed_b;   // (DS)
    const float* __restrict__ pe;        // (J, DS)
    const float* __restrict__ blocks;    // depth * SpatialBlockLayout::size
    const float* __restrict__ norm_g;    // (DS)
    const float* __restrict__ norm_b;    // (DS)
    int depth;
    int total_frames;                    // B * N
    const int* __restrict__ frame_list;  // compacted valid frames + count at [total_frames]; nullptr = all frames
};

template <int DS>
__device__ __forceinline__ void ln_row(const float (&x)[DS], const float* __restrict__ g,
                                       const float* __restrict__ b, const float eps, float (&y)[DS])
{
    float mean = 0.f;
#pragma unroll
    for (int c = 0; c < DS; ++c) mean += x[c];
    mean *= (1.0f / DS);
    float var = 0.f;
#pragma unroll
    for (int c = 0; c < DS; ++c) { const float d = x[c] - mean; var = fmaf(d, d, var); }
    var *= (1.0f / DS);
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int c = 0; c < DS; ++c) {
        const float inv = rstd * g[c];
        y[c] = x[c] * inv + (b[c] - mean * inv);
    }
}

template <int DS>
__device__ __forceinline__ float dot_row(const float (&y)[DS], const float* __restrict__ w)
{
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int k = 0; k < DS; k += 4) {
        a0 = fmaf(y[k + 0], w[k + 0], a0);
        a1 = fmaf(y[k + 1], w[k + 1], a1);
        a2 = fmaf(y[k + 2], w[k + 2], a2);
        a3 = fmaf(y[k + 3], w[k + 3], a3);
    }
    return (a0 + a1) + (a2 + a3);
}

template <int J, int DS, int HS, int HEADS>
__global__ void __launch_bounds__(256)
spatial_stack_kernel(const float* __restrict__ kp2d, const SpatialParams p, float* __restrict__ out)
{
    using LY = SpatialBlockLayout<DS, HS>;
    constexpr int DH = DS / HEADS;
    static_assert(DH == 4, "spatial head dim must be 4 (d_s = 32, 8 heads)");
    constexpr int FPW = 256 / J;                 // frames per workgroup
    constexpr int LDR = 2 * DS + 4;              // K|V row stride in LDS
    extern __shared__ __attribute__((aligned(16))) float KV[];   // [256][LDR], 69,632 B (dynamic: > 64 KiB)

    const int tid = threadIdx.x;
    const int fl = tid / J;                      // local frame
    const int joint = tid - fl * J;
    const int frame = blockIdx.x * FPW + fl;
    const bool valid = (fl < FPW) && (frame < p.total_frames);
    const int fbase = fl * J;                    // first LDS row of this thread's frame

    float x[DS];
    {
        float kx = 0.f, ky = 0.f;
        if (valid) { const float2 k2 = *reinterpret_cast<const float2*>(kp2d + ((size_t)frame * J + joint) * 2); kx = k2.x; ky = k2.y; }
        const float* pe = p.pe + (valid ? joint : 0) * DS;
#pragma unroll
        for (int c = 0; c < DS; ++c)
            x[c] = (fmaf(ky, p.embed_w[DS + c], kx * p.embed_w[c]) + p.embed_b[c]) + pe[c];
    }

    const float sqrt_dh = sqrtf((float)DH);
    for (int blk = 0; blk < p.depth; ++blk) {
        const float* __restrict__ W = p.blocks + (size_t)blk * LY::size;
        float y[DS];
        ln_row<DS>(x, W + LY::ln1_g, W + LY::ln1_b, 1e-5f, y);

        // K, V rows -> LDS (dot form)
#pragma unroll 2
        for (int j = 0; j < DS; ++j) {
            KV[tid * LDR + j] = dot_row<DS>(y, W + LY::wk_t + j * DS) + W[LY::bk + j];
            KV[tid * LDR + DS + j] = dot_row<DS>(y, W + LY::wv_t + j * DS) + W[LY::bv + j];
        }
        __syncthreads();

        float acc[DS];
#pragma unroll
        for (int c = 0; c < DS; ++c) acc[c] = W[LY::bp + c];

        for (int h = 0; h < HEADS; ++h) {
            float q[DH];
#pragma unroll
            for (int cc = 0; cc < DH; ++cc)
                q[cc] = dot_row<DS>(y, W + LY::wq_t + (h * DH + cc) * DS) + W[LY::bq + h * DH + cc];
            float s[J];
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const float4 k4 = *reinterpret_cast<const float4*>(&KV[(fbase + j) * LDR + h * DH]);
                float d = q[0] * k4.x;
                d = fmaf(q[1], k4.y, d); d = fmaf(q[2], k4.z, d); d = fmaf(q[3], k4.w, d);
                s[j] = d / sqrt_dh;
                mx = fmaxf(mx, s[j]);
            }
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < J; ++j) { s[j] = expf(s[j] - mx); sum += s[j]; }
            float o[DH] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const float pj = s[j] / sum;
                const float4 v4 = *reinterpret_cast<const float4*>(&KV[(fbase + j) * LDR + DS + h * DH]);
                o[0] = fmaf(pj, v4.x, o[0]); o[1] = fmaf(pj, v4.y, o[1]);
                o[2] = fmaf(pj, v4.z, o[2]); o[3] = fmaf(pj, v4.w, o[3]);
            }
            // projection, axpy form: acc += o[cc] * Wp[h*DH + cc][:]
#pragma unroll
            for (int cc = 0; cc < DH; ++cc) {
                const float* __restrict__ wr = W + LY::wp + (h * DH + cc) * DS;
#pragma unroll
                for (int c = 0; c < DS; ++c) acc[c] = fmaf(o[cc], wr[c], acc[c]);
            }
        }
#pragma unroll
        for (int c = 0; c < DS; ++c) x[c] += acc[c];
        __syncthreads();   // K/V tile is rewritten by the next block

        // MLP: fc1 (dot) -> GELU(erf) -> fc2 (axpy)
        ln_row<DS>(x, W + LY::ln2_g, W + LY::ln2_b, 1e-5f, y);
#pragma unroll
        for (int c = 0; c < DS; ++c) acc[c] = W[LY::b2 + c];
#pragma unroll 2
        for (int j = 0; j < HS; ++j) {
            float hj = dot_row<DS>(y, W + LY::w1_t + j * DS) + W[LY::b1 + j];
            hj = 0.5f * hj * (1.0f + erff(hj * 0.70710678118654752440f));
            const float* __restrict__ wr = W + LY::w2 + j * DS;
#pragma unroll
            for (int c = 0; c < DS; ++c) acc[c] = fmaf(hj, wr[c], acc[c]);
        }
#pragma unroll
        for (int c = 0; c < DS; ++c) x[c] += acc[c];
    }

    float y[DS];
    ln_row<DS>(x, p.norm_g, p.norm_b, 1e-6f, y);
    if (valid) {
        float* o = out + ((size_t)frame * J + joint) * DS;
#pragma unroll
        for (int c = 0; c < DS; c += 4)
            *reinterpret_cast<float4*>(o + c) = make_float4(y[c], y[c + 1], y[c + 2], y[c + 3]);
    }
}

// =========================================================================================
// v2: MFMA spatial stack.  One WAVE (= one 64-thread workgroup) owns FR whole frames
// (FR * J <= 64 token rows), so nothing ever crosses a wave: no __syncthreads between
// workgroup waves, and co-resident waves overlap each other's MFMA, VALU and LDS phases.
//
//   home layout : thread = token row, x[d_s] in registers (LayerNorm, softmax, residual in-lane)
//   linear maps : v_mfma_f32_32x32x2_f32 on 2 row tiles of 32; the A operand is the wave's
//                 [64][36] LDS tile (ds_read_b128 fragments, k-slot order 8kk + 4h + s), the
//                 B operand is read from HBM/L2 in FRAGMENT ORDER ([kk][lane][4], packed by
//                 uu3d_commit_weights) - one coalesced 1 KiB load per k-slice, no LDS.
//   C/D tiles   : bias (+GELU) applied in the accumulator layout (column on the lane), then
//                 written [row][col] to LDS where the row-owning thread (or the next MFMA's
//                 A fragments) picks them up.
//   attention   : per thread, 8 heads x J keys, K/V rows read as 16-byte LDS fragments.
// LDS per wave: three [52][36] tiles = 22,464 B -> 7 waves per CU.  T0 stages y / q / o /
// projection / fc2 output; TK, TV hold K and V and are reused for the two 32-column halves of
// the fc1 output.  Row 51 is a dummy row: accumulator rows and lanes beyond the 51 tokens land there.
// =========================================================================================
template <int DS, int HS>
struct SpatialBlockLayoutV2 {
    static constexpr int ln1_g = 0, ln1_b = ln1_g + DS, ln2_g = ln1_b + DS, ln2_b = ln2_g + DS;
    static constexpr int bq = ln2_b + DS, bk = bq + DS, bv = bk + DS, bp = bv + DS, b1 = bp + DS, b2 = b1 + HS;
    static constexpr int fq = b2 + DS;                 // fragment-ordered weights, 16-byte aligned
    static constexpr int fk = fq + DS * DS, fv = fk + DS * DS, fp = fv + DS * DS;
    static constexpr int f1 = fp + DS * DS;            // [2 n-tiles][4 kk][64][4]
    static constexpr int f2 = f1 + DS * HS;            // [8 kk][64][4]
    static constexpr int size = f2 + HS * DS;
    static_assert(fq % 4 == 0, "fragment arrays must be 16-byte aligned");
};

__host__ __device__ inline constexpr size_t spatial_v2_lds_bytes() { return (size_t)3 * 52 * 36 * sizeof(float); }

namespace sv2 {
// exp via v_exp_f32 (2^x, ~1 ulp): softmax arguments are <= 0, relative error ~1e-6 at |x| ~ 10.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
// Exact-form GELU 0.5 x (1 + erf(x / sqrt 2)) with erf from Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7 + f32 rounding, measured 5e-7 abs on [-6, 6]; GELU abs error <= 4e-7),
// branch free: ocml's erff is ~45 instructions with a divergent range split, this is ~16.
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f); p = fmaf(p, t, -0.284496736f); p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-(z * z) * 1.44269504088896341f);
    const float erf_abs = fmaf(-(p * t), e, 1.0f);
    return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}
template <int KK>
__device__ __forceinline__ void load_afrags(const float* T, const int ld, const int lane, f32x4 (&a)[2][KK]) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
            a[mt][kk] = *reinterpret_cast<const f32x4*>(&T[min(32 * mt + (lane & 31), 51) * ld + 8 * kk + 4 * (lane >> 5)]);
}
template <int KK>
__device__ __forceinline__ void load_bfrags(const float* __restrict__ wfrag, const int lane, f32x4 (&b)[KK]) {
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) b[kk] = reinterpret_cast<const f32x4*>(wfrag)[kk * 64 + lane];
}
template <int KK>
__device__ __forceinline__ void mma(const f32x4 (&a)[2][KK], const f32x4 (&b)[KK], f32x16 (&acc)[2]) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][kk][s], b[kk][s], acc[mt], 0, 0, 0);
}
// accumulator tile (+bias) -> LDS [row][coloff + col]; rows above maxrow land on row maxrow (dummy)
__device__ __forceinline__ void store_ctile(float* T, const int ld, const int coloff, const int lane,
                                            const f32x16 (&acc)[2], const float bias, const int maxrow) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = min(32 * mt + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), maxrow);
            T[row * ld + coloff + (lane & 31)] = acc[mt][r] + bias;
        }
}
template <int DS>
__device__ __forceinline__ void write_row(float* T, const int lane, const float (&y)[DS]) {
#pragma unroll
    for (int c = 0; c < DS; c += 4)
        *reinterpret_cast<f32x4*>(&T[min(lane, 51) * 36 + c]) = (f32x4){y[c], y[c + 1], y[c + 2], y[c + 3]};
}
template <int DS>
__device__ __forceinline__ void read_row(const float* T, const int lane, float (&y)[DS]) {
#pragma unroll
    for (int c = 0; c < DS; c += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(&T[min(lane, 51) * 36 + c]);
        y[c] = v[0]; y[c + 1] = v[1]; y[c + 2] = v[2]; y[c + 3] = v[3];
    }
}
}  // namespace sv2

#ifndef UU3D_SPATIAL_WAVES
#define UU3D_SPATIAL_WAVES 2
#endif
template <int J, int FR>
__global__ void __launch_bounds__(64, UU3D_SPATIAL_WAVES)
spatial_stack_mfma_kernel(const float* __restrict__ kp2d, const SpatialParams p, float* __restrict__ out)
{
    constexpr int DS = 32, HS = 64, HEADS = 8, DH = 4, ROWS = FR * J, LD = 36;
    static_assert(ROWS <= 51, "FR * J must leave a dummy row in the 52-row LDS tiles");
    using LY = SpatialBlockLayoutV2<DS, HS>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* T0 = lds;                   // [52][36]
    float* TK = lds + 52 * LD;         // [52][36]  K, later fc1 output columns  0..31
    float* TV = TK + 52 * LD;          // [52][36]  V, later fc1 output columns 32..63

    const int lane = threadIdx.x;
    const int fl = min(lane / J, FR - 1);
    const int joint = lane - (lane / J) * J;
    int frame = blockIdx.x * FR + fl;
    int nframes = p.total_frames;
    if (p.frame_list != nullptr) {
        nframes = p.frame_list[p.total_frames];
        if ((int)blockIdx.x * FR >= nframes) return;           // workgroup-uniform: nothing left to do
        frame = p.frame_list[min(frame, nframes - 1)];
    }
    const bool valid = (lane < ROWS) && ((int)blockIdx.x * FR + fl < nframes);
    const int fbase = fl * J;
    const int col = lane & 31;

    float x[DS];
    {
        float kx = 0.f, ky = 0.f;
        if (valid) { const float2 k2 = *reinterpret_cast<const float2*>(kp2d + ((size_t)frame * J + joint) * 2); kx = k2.x; ky = k2.y; }
        const float* pe = p.pe + (lane < ROWS ? joint : 0) * DS;
#pragma unroll
        for (int c = 0; c < DS; ++c)
            x[c] = (fmaf(ky, p.embed_w[DS + c], kx * p.embed_w[c]) + p.embed_b[c]) + pe[c];
    }

    const float inv_sqrt_dh = 1.0f / sqrtf((float)DH);   // DH = 4: exactly 0.5
    for (int blk = 0; blk < p.depth; ++blk) {
        const float* __restrict__ W = p.blocks + (size_t)blk * LY::size;
        float y[DS];
        f32x4 a4[2][4];
        f32x4 b4[4];
        f32x16 acc[2];

        // ---- attention half ----
        ln_row<DS>(x, W + LY::ln1_g, W + LY::ln1_b, 1e-5f, y);
        sv2::write_row<DS>(T0, lane, y);
        __syncthreads();
        sv2::load_afrags<4>(T0, LD, lane, a4);
        __syncthreads();                                   // T0 is rewritten with q below
        sv2::load_bfrags<4>(W + LY::fq, lane, b4); sv2::mma<4>(a4, b4, acc);
        sv2::store_ctile(T0, LD, 0, lane, acc, W[LY::bq + col], 51);
        sv2::load_bfrags<4>(W + LY::fk, lane, b4); sv2::mma<4>(a4, b4, acc);
        sv2::store_ctile(TK, LD, 0, lane, acc, W[LY::bk + col], 51);
        sv2::load_bfrags<4>(W + LY::fv, lane, b4); sv2::mma<4>(a4, b4, acc);
        sv2::store_ctile(TV, LD, 0, lane, acc, W[LY::bv + col], 51);
        __syncthreads();

        float q[DS];
        sv2::read_row<DS>(T0, lane, q);
        float o[DS];
#pragma unroll
        for (int h = 0; h < HEADS; ++h) {
            float s[J];
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const f32x4 k4 = *reinterpret_cast<const f32x4*>(&TK[(fbase + j) * LD + h * DH]);
                float d = q[h * DH] * k4[0];
                d = fmaf(q[h * DH + 1], k4[1], d); d = fmaf(q[h * DH + 2], k4[2], d); d = fmaf(q[h * DH + 3], k4[3], d);
                s[j] = d * inv_sqrt_dh;
                mx = fmaxf(mx, s[j]);
            }
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < J; ++j) { s[j] = sv2::fast_exp(s[j] - mx); sum += s[j]; }
            float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
            const float rsum = 1.0f / sum;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const float pj = s[j] * rsum;
                const f32x4 v4 = *reinterpret_cast<const f32x4*>(&TV[(fbase + j) * LD + h * DH]);
                o0 = fmaf(pj, v4[0], o0); o1 = fmaf(pj, v4[1], o1); o2 = fmaf(pj, v4[2], o2); o3 = fmaf(pj, v4[3], o3);
            }
            o[h * DH] = o0; o[h * DH + 1] = o1; o[h * DH + 2] = o2; o[h * DH + 3] = o3;
        }
        __syncthreads();                                   // all q rows read before T0 is reused
        sv2::write_row<DS>(T0, lane, o);
        __syncthreads();
        sv2::load_afrags<4>(T0, LD, lane, a4);
        __syncthreads();
        sv2::load_bfrags<4>(W + LY::fp, lane, b4); sv2::mma<4>(a4, b4, acc);
        sv2::store_ctile(T0, LD, 0, lane, acc, W[LY::bp + col], 51);
        __syncthreads();
        sv2::read_row<DS>(T0, lane, y);
#pragma unroll
        for (int c = 0; c < DS; ++c) x[c] += y[c];
        __syncthreads();

        // ---- MLP half ----
        ln_row<DS>(x, W + LY::ln2_g, W + LY::ln2_b, 1e-5f, y);
        sv2::write_row<DS>(T0, lane, y);
        __syncthreads();
        sv2::load_afrags<4>(T0, LD, lane, a4);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            sv2::load_bfrags<4>(W + LY::f1 + nt * DS * DS, lane, b4); sv2::mma<4>(a4, b4, acc);
            const float bias = W[LY::b1 + 32 * nt + col];
            float* Th = nt == 0 ? TK : TV;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = min(32 * mt + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), 51);
                    Th[row * LD + col] = sv2::gelu_erf(acc[mt][r] + bias);
                }
        }
        __syncthreads();
        {
            f32x4 b8[8];
            sv2::load_bfrags<8>(W + LY::f2, lane, b8);
            f32x4 ah[2][4];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
#pragma unroll
            for (int half = 0; half < 2; ++half) {           // k = 0..31 from TK, 32..63 from TV
                sv2::load_afrags<4>(half == 0 ? TK : TV, LD, lane, ah);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
                            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ah[mt][kk][q], b8[4 * half + kk][q], acc[mt], 0, 0, 0);
            }
        }
        __syncthreads();                                   // T0 (LN2 output) fully consumed above
        sv2::store_ctile(T0, LD, 0, lane, acc, W[LY::b2 + col], 51);
        __syncthreads();
        sv2::read_row<DS>(T0, lane, y);
#pragma unroll
        for (int c = 0; c < DS; ++c) x[c] += y[c];
        __syncthreads();
    }

    float y[DS];
    ln_row<DS>(x, p.norm_g, p.norm_b, 1e-6f, y);
    if (valid) {
        float* o = out + ((size_t)frame * J + joint) * DS;
#pragma unroll
        for (int c = 0; c < DS; c += 4)
            *reinterpret_cast<float4*>(o + c) = make_float4(y[c], y[c + 1], y[c + 2], y[c + 3]);
    }
}

}  // namespace uu3d
