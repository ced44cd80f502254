// uu3d_spatial_h3.h -- the spatial (per-frame joint) transformer stack with f16x3 matrix products.
//
// Same function as spatial_stack_mfma_kernel (uu3d_spatial.h; reference u_u_t.py:313-330 and
// vision_transformer.py:71-195 at d = 32): keypoint embedding + PE, 4 pre-LN blocks of 8-head attention over the
// 17 joints of a frame and a GELU MLP 32 -> 64 -> 32, spatial_norm, one launch, one wave = 3 frames = 51 tokens.
// Two changes of structure:
//
// * Products are f16x3 (uu3d_gemm_h3.h): x ~= hi + lo / 2048, three v_mfma_f32_32x32x16_f16 per 16-deep k-step.
//   A K = 32 product of a 32 x 32 tile costs 6 MFMAs x 32 cycles instead of 16 x 64 cycles of 32x32x2_f32.
//
// * Every product is computed TRANSPOSED: C^T[n][m] = W^T[n][k] X^T[k][m].  The A operand is the weight (fragment
//   ordered f16 planes, one coalesced 1 KiB load per fragment straight from L2), the B operand is the activation
//   row of token m (16 bytes of a row-major LDS tile).  In the C/D map of the 32x32 MFMA a lane then holds, for
//   ITS token m = lane & 31 (+ 32 per m-tile), the 16 channels n = 8g + 4 (lane >> 5) + e (g, e = 0..3): groups of
//   four consecutive channels.  With d_h = 4 a group is exactly one head, so
//     - the residual stream lives in that layout for the whole kernel (2 tokens x 16 channels per lane; the two
//       lanes l and l + 32 share a token), GEMM results add to it in registers -- no C tile round trip through LDS;
//     - q stays in registers, lane (l, half) runs heads {half, 2 + half, 4 + half, 6 + half} of its two tokens;
//     - LayerNorm needs one cross-lane add (lane ^ 32) per moment;
//     - everything a lane writes to LDS (LayerNorm output, attention output, GELU output as f16 planes, K and V
//       as f32) is a group of 4 consecutive channels: one 8- or 16-byte store.
//   A workgroup is one wave, so LDS traffic is ordered by the wave itself and there is no barrier at all.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <math.h>
#include "uu3d_gemm_h3.h"
#include "uu3d_spatial.h"

namespace uu3d {

// f16 fragment planes of one block, in halfs: per matrix [n-tile][kk][plane hi/lo][lane][8],
// element = split(W[16 kk + 8 (lane >> 5) + e][32 nt + (lane & 31)]) of the Keras (in, out) kernel
struct SpatialFragLayoutH3 {
    static constexpr int frag = 64 * 8;                       // halfs per (n-tile, kk, plane)
    static constexpr int fq = 0, fk = fq + 1 * 2 * 2 * frag, fv = fk + 1 * 2 * 2 * frag, fp = fv + 1 * 2 * 2 * frag;
    static constexpr int f1 = fp + 1 * 2 * 2 * frag;          // 32 -> 64: 2 n-tiles x 2 kk
    static constexpr int f2 = f1 + 2 * 2 * 2 * frag;          // 64 -> 32: 1 n-tile  x 4 kk
    static constexpr int size = f2 + 1 * 4 * 2 * frag;        // 16384 halfs = 32 KiB per block
};

namespace sh3 {
constexpr int ROWS_T = 52;       // 51 tokens + one dummy row that out-of-range tokens read and write
constexpr int XLD = 40;          // halfs per row of the K = 32 operand tile (80 B: conflict-free 16-byte reads)
constexpr int HLD = 72;          // halfs per row of the K = 64 hidden tile (144 B)
constexpr int KLD = 36;          // floats per row of the K / V tiles
constexpr int NPARAM = 352;     // LayerNorm parameters and biases of one block (SpatialBlockLayoutV2 up to fq)
constexpr size_t lds_bytes() { return (size_t)2 * ROWS_T * XLD * 2 + (size_t)2 * ROWS_T * KLD * 4 + NPARAM * 4; }
static_assert(2 * ROWS_T * HLD * 2 == 2 * ROWS_T * KLD * 4, "the hidden planes reuse the K / V tiles byte for byte");

// The weight fragments of one product, loaded AHEAD of the product (hipcc otherwise issues each pair of fragment loads
// right in front of the MFMAs that use it: 16 exposed L2 round trips per block and wave).
template <int NT, int KK>
struct WFrag { h16x8 h[NT][KK], l[NT][KK]; };
// Issued by name (the compiler sinks a plain load back to its use) and waited for with a COUNTED s_waitcnt: vector loads
// return in order, so wait_w(w, n) lets the n loads issued after w's stay in flight.  The compiler's own vmcnt accounting
// does not see these loads; its waits can only become more conservative by that, never too weak.
template <int NT, int KK>
__device__ __forceinline__ void load_w(const _Float16* __restrict__ wf, const int lane, WFrag<NT, KK>& w) {
    const h16x8* base = reinterpret_cast<const h16x8*>(wf) + lane;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(w.h[nt][kk]) : "v"(base + ((nt * KK + kk) * 2 + 0) * 64) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(w.l[nt][kk]) : "v"(base + ((nt * KK + kk) * 2 + 1) * 64) : "memory");
        }
}
template <int NLATER, int NT, int KK>
__device__ __forceinline__ void wait_w(WFrag<NT, KK>& w) {
    static_assert(NT * KK == 2 || NT * KK == 4, "2 or 4 fragment pairs");
    if constexpr (NT * KK == 2)
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(w.h[0][0]), "+v"(w.l[0][0]), "+v"(w.h[0][1]), "+v"(w.l[0][1]) : "i"(NLATER));
    else
        asm volatile("s_waitcnt vmcnt(%8)" : "+v"(w.h[0][0]), "+v"(w.l[0][0]), "+v"(w.h[0][1]), "+v"(w.l[0][1]),
                     "+v"(w.h[NT - 1][KK - 2]), "+v"(w.l[NT - 1][KK - 2]), "+v"(w.h[NT - 1][KK - 1]), "+v"(w.l[NT - 1][KK - 1]) : "i"(NLATER));
}
template <int NT, int KK>
__device__ __forceinline__ void mm(const WFrag<NT, KK>& w, const _Float16* Bh, const _Float16* Bl, const int ldb,
                                   const int lane, float (&out)[NT][2][16]) {
    f32x16 acc0[NT][2], acc1[NT][2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[nt][mt][r] = 0.f; acc1[nt][mt][r] = 0.f; }
    const int tl = lane & 31, half = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        h16x8 bh[2], bl[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int row = min(32 * mt + tl, ROWS_T - 1);
            bh[mt] = *reinterpret_cast<const h16x8*>(Bh + row * ldb + 16 * kk + 8 * half);
            bl[mt] = *reinterpret_cast<const h16x8*>(Bl + row * ldb + 16 * kk + 8 * half);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                acc0[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.h[nt][kk], bh[mt], acc0[nt][mt], 0, 0, 0);
                acc1[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.h[nt][kk], bl[mt], acc1[nt][mt], 0, 0, 0);
                acc1[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.l[nt][kk], bh[mt], acc1[nt][mt], 0, 0, 0);
            }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[nt][mt][r] = acc0[nt][mt][r] + acc1[nt][mt][r] * (1.0f / H3_SCALE);
}

// C^T tiles of W^T X^T for NT output tiles (32 channels each) and both token tiles; K = 16 * KK.
// out[nt][mt][r]: token = 32 mt + (lane & 31), channel = 32 nt + 8 (r >> 2) + 4 (lane >> 5) + (r & 3).
template <int NT, int KK>
__device__ __forceinline__ void mm(const _Float16* __restrict__ wf, const _Float16* Bh, const _Float16* Bl, const int ldb,
                                   const int lane, float (&out)[NT][2][16]) {
    f32x16 acc0[NT][2], acc1[NT][2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[nt][mt][r] = 0.f; acc1[nt][mt][r] = 0.f; }
    const int tl = lane & 31, half = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        h16x8 bh[2], bl[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int row = min(32 * mt + tl, ROWS_T - 1);
            bh[mt] = *reinterpret_cast<const h16x8*>(Bh + row * ldb + 16 * kk + 8 * half);
            bl[mt] = *reinterpret_cast<const h16x8*>(Bl + row * ldb + 16 * kk + 8 * half);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const h16x8 ah = reinterpret_cast<const h16x8*>(wf)[((nt * KK + kk) * 2 + 0) * 64 + lane];
            const h16x8 al = reinterpret_cast<const h16x8*>(wf)[((nt * KK + kk) * 2 + 1) * 64 + lane];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                acc0[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[mt], acc0[nt][mt], 0, 0, 0);
                acc1[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[mt], acc1[nt][mt], 0, 0, 0);
                acc1[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[mt], acc1[nt][mt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[nt][mt][r] = acc0[nt][mt][r] + acc1[nt][mt][r] * (1.0f / H3_SCALE);
}

// LayerNormalization over the 32 channels of each of the lane's two tokens (16 here, 16 in lane ^ 32); same
// arithmetic as ln_row (non-fused Keras path: inv = rstd * gamma, y = x * inv + (beta - mean * inv))
__device__ __forceinline__ void ln_tokens(const float (&x)[2][16], const float* g, const float* b,
                                          const float eps, const int half, float (&y)[2][16]) {
#pragma clang fp contract(off)      // see the kernel: both unrolled token copies must round identically
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += x[mt][r];
        s += __shfl_xor(s, 32);
        const float mean = s * (1.0f / 32.0f);
        float q = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float d = x[mt][r] - mean; q = fmaf(d, d, q); }
        q += __shfl_xor(q, 32);
        const float rstd = 1.0f / sqrtf(q * (1.0f / 32.0f) + eps);
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(g + 8 * gq + 4 * half);
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(b + 8 * gq + 4 * half);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float inv = rstd * g4[e];
                y[mt][4 * gq + e] = fmaf(x[mt][4 * gq + e], inv, fmaf(-mean, inv, b4[e]));
            }
        }
    }
}

// the lane's 2 x 16 values -> hi / lo planes of a row-major tile (row = token, 4-channel groups of 8 bytes)
__device__ __forceinline__ void store_planes(_Float16* Th, _Float16* Tl, const int ld, const int coloff, const int lane,
                                             const float (&v)[2][16]) {
    const int tl = lane & 31, half = lane >> 5;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int row = min(32 * mt + tl, ROWS_T - 1);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 f = {v[mt][4 * g], v[mt][4 * g + 1], v[mt][4 * g + 2], v[mt][4 * g + 3]};
            h16x4 hi, lo;
            h3_split(f, hi, lo);
            *reinterpret_cast<h16x4*>(Th + row * ld + coloff + 8 * g + 4 * half) = hi;
            *reinterpret_cast<h16x4*>(Tl + row * ld + coloff + 8 * g + 4 * half) = lo;
        }
    }
}

// One (token, head): softmax(q k^T / sqrt(d_h)) v over the J keys of the token's frame; Kp / Vp point at the head's 4
// channels of the frame's first key row.
#ifndef UU3D_SPATIAL_ATTN_BYNAME
#define UU3D_SPATIAL_ATTN_BYNAME 1
#endif
template <int J>
__device__ __forceinline__ f32x4 head_attention(const f32x4 q4, const float* Kp, const float* Vp) {
    // softmax(x) with x = q.k / 2: exp(x - max) = exp2((q * log2e / 2).k - max'), so the scale and the base change are
    // folded into q once; the 1 / sum normalisation is applied to the 4 outputs instead of the 17 probabilities
    const f32x4 qs = {q4[0] * 0.72134752044448170368f, q4[1] * 0.72134752044448170368f,
                      q4[2] * 0.72134752044448170368f, q4[3] * 0.72134752044448170368f};
    float s[J];
    float mx = -INFINITY;
    if constexpr (J == 17 && UU3D_SPATIAL_ATTN_BYNAME) {
        // The 17 key rows and then the 17 value rows of this head by name, all in flight at once, with counted waits (LDS returns
        // in order; the counter starts from zero and the "memory" clobbers keep other memory operations out, see uu3d_attn.h).
        // hipcc had emitted read -> wait -> use for every row: 34 exposed LDS round trips per (token, head), 272 per block.
        f32x4 kv[17];
        const unsigned ka = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)Kp;
        const unsigned va = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)Vp;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 17; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kv[j]) : "v"(ka), "i"(j * KLD * 4) : "memory");
#define UU3D_SP_W9(x, o) "+v"(x[o]), "+v"(x[o + 1]), "+v"(x[o + 2]), "+v"(x[o + 3]), "+v"(x[o + 4]), "+v"(x[o + 5]), "+v"(x[o + 6]), "+v"(x[o + 7]), "+v"(x[o + 8])
        asm volatile("s_waitcnt lgkmcnt(8)" : UU3D_SP_W9(kv, 0) :: "memory");
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            float d = qs[0] * kv[j][0];
            d = fmaf(qs[1], kv[j][1], d); d = fmaf(qs[2], kv[j][2], d); d = fmaf(qs[3], kv[j][3], d);
            s[j] = d; mx = fmaxf(mx, d);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : UU3D_SP_W9(kv, 8) :: "memory");
#pragma unroll
        for (int j = 9; j < 17; ++j) {
            float d = qs[0] * kv[j][0];
            d = fmaf(qs[1], kv[j][1], d); d = fmaf(qs[2], kv[j][2], d); d = fmaf(qs[3], kv[j][3], d);
            s[j] = d; mx = fmaxf(mx, d);
        }
#pragma unroll
        for (int j = 0; j < 17; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kv[j]) : "v"(va), "i"(j * KLD * 4) : "memory");
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 17; ++j) { s[j] = __builtin_amdgcn_exp2f(s[j] - mx); sum += s[j]; }     // the value rows arrive meanwhile
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt lgkmcnt(8)" : UU3D_SP_W9(kv, 0) :: "memory");
#pragma unroll
        for (int j = 0; j < 9; ++j) { o[0] = fmaf(s[j], kv[j][0], o[0]); o[1] = fmaf(s[j], kv[j][1], o[1]); o[2] = fmaf(s[j], kv[j][2], o[2]); o[3] = fmaf(s[j], kv[j][3], o[3]); }
        asm volatile("s_waitcnt lgkmcnt(0)" : UU3D_SP_W9(kv, 8) :: "memory");
#pragma unroll
        for (int j = 9; j < 17; ++j) { o[0] = fmaf(s[j], kv[j][0], o[0]); o[1] = fmaf(s[j], kv[j][1], o[1]); o[2] = fmaf(s[j], kv[j][2], o[2]); o[3] = fmaf(s[j], kv[j][3], o[3]); }
#undef UU3D_SP_W9
        const float rsum = 1.0f / sum;
        o[0] *= rsum; o[1] *= rsum; o[2] *= rsum; o[3] *= rsum;
        return o;
    } else {
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const f32x4 k4 = *reinterpret_cast<const f32x4*>(Kp + j * KLD);
        float d = qs[0] * k4[0];
        d = fmaf(qs[1], k4[1], d); d = fmaf(qs[2], k4[2], d); d = fmaf(qs[3], k4[3], d);
        s[j] = d;
        mx = fmaxf(mx, d);
    }
    float sum = 0.f;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const float e = __builtin_amdgcn_exp2f(s[j] - mx);
        sum += e;
        const f32x4 v4 = *reinterpret_cast<const f32x4*>(Vp + j * KLD);
        o[0] = fmaf(e, v4[0], o[0]); o[1] = fmaf(e, v4[1], o[1]); o[2] = fmaf(e, v4[2], o[2]); o[3] = fmaf(e, v4[3], o[3]);
    }
    const float rsum = 1.0f / sum;
    o[0] *= rsum; o[1] *= rsum; o[2] *= rsum; o[3] *= rsum;
    return o;
    }
}
}  // namespace sh3

// out_lo == nullptr: out is the f32 (frames, J, 32) tensor; otherwise out / out_lo are its two f16 planes
#ifndef UU3D_SPATIAL_H3_WAVES
#define UU3D_SPATIAL_H3_WAVES 2     // 3 (168 VGPRs) spills into the block loop: 0.30 ms instead of 0.20
#endif
template <int J, int FR>
__global__ void __launch_bounds__(64, UU3D_SPATIAL_H3_WAVES)
spatial_stack_h3_kernel(const float* __restrict__ kp2d, const SpatialParams p, const _Float16* __restrict__ wfrag,
                        float* __restrict__ out, _Float16* __restrict__ out_hi, _Float16* __restrict__ out_lo)
{
    // A lane runs the same arithmetic twice, once per token (mt = 0, 1), fully unrolled; the two copies must round
    // identically or a frame's result depends on its slot in the wave (bitwise permutation test).  Contraction is off
    // and every intended FMA is an fmaf(); the f16 conversions go through h3_hi (one instruction form, see there).
#pragma clang fp contract(off)
    using namespace sh3;
    h3_flush_f16_denormals();
    constexpr int DS = 32, HS = 64, ROWS = FR * J;
    static_assert(ROWS <= ROWS_T - 1 && DS == 32 && HS == 64, "one wave = 3 frames of 17 joints, d = 32");
    using LY = SpatialBlockLayoutV2<DS, HS>;          // LayerNorm parameters and biases (f32) come from the V2 block
    using FL = SpatialFragLayoutH3;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    _Float16* Xh = reinterpret_cast<_Float16*>(lds_raw);               // [52][40] operand tile, hi
    _Float16* Xl = Xh + ROWS_T * XLD;                                   // lo
    float* TK = reinterpret_cast<float*>(Xl + ROWS_T * XLD);           // [52][36] K
    float* TV = TK + ROWS_T * KLD;                                      // [52][36] V
    _Float16* Hh = reinterpret_cast<_Float16*>(TK);                    // [52][72] GELU(fc1), hi (K / V are dead by then)
    _Float16* Hl = Hh + ROWS_T * HLD;
    float* P = TV + ROWS_T * KLD;                                       // [352] this block's LayerNorm parameters and biases

    const int lane = threadIdx.x, tl = lane & 31, half = lane >> 5;
    int nframes = p.total_frames;
    if (p.frame_list != nullptr) {
        nframes = p.frame_list[p.total_frames];
        if ((int)blockIdx.x * FR >= nframes) return;
    }
    int frame[2], joint[2], fbase[2];
    bool valid[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int tok = 32 * mt + tl;
        const int fl = min(tok / J, FR - 1);
        joint[mt] = min(tok - fl * J, J - 1);
        fbase[mt] = fl * J;
        int f = blockIdx.x * FR + fl;
        valid[mt] = (tok < ROWS) && (f < nframes);
        if (p.frame_list != nullptr) f = p.frame_list[min(f, nframes - 1)];
        frame[mt] = min(f, p.total_frames - 1);
    }

    // keypoint embedding + spatial PE (u_u_t.py:321-323), this lane's 16 channels of each token
    float x[2][16];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        float kx = 0.f, ky = 0.f;
        if (valid[mt]) { const float2 k2 = *reinterpret_cast<const float2*>(kp2d + ((size_t)frame[mt] * J + joint[mt]) * 2); kx = k2.x; ky = k2.y; }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = 8 * (r >> 2) + 4 * half + (r & 3);
            x[mt][r] = (fmaf(ky, p.embed_w[DS + c], kx * p.embed_w[c]) + p.embed_b[c]) + p.pe[joint[mt] * DS + c];
        }
    }

    for (int blk = 0; blk < p.depth; ++blk) {
        const _Float16* __restrict__ F = wfrag + (size_t)blk * FL::size;
        {   // one coalesced copy of the block's 352 parameters into LDS: the 16-byte group reads below then cost an LDS
            // round trip instead of a dependent global load each (133 of them per block before)
            const float* __restrict__ Wg = p.blocks + (size_t)blk * LY::size;
#pragma unroll
            for (int i = 0; i < (NPARAM + 63) / 64; ++i) { const int k = 64 * i + lane; if (k < NPARAM) P[k] = Wg[k]; }
        }
        const float* W = P;
        float y[2][16];

        // ---- attention half ----
        WFrag<1, 2> wq, wk, wv, wp;
        load_w<1, 2>(F + FL::fq, lane, wq); load_w<1, 2>(F + FL::fk, lane, wk); load_w<1, 2>(F + FL::fv, lane, wv);
        ln_tokens(x, W + LY::ln1_g, W + LY::ln1_b, 1e-5f, half, y);
        store_planes(Xh, Xl, XLD, 0, lane, y);
        float q[1][2][16];
        {
            float kv[1][2][16];
            wait_w<8>(wq);
            mm<1, 2>(wq, Xh, Xl, XLD, lane, q);
            wait_w<4>(wk);
            mm<1, 2>(wk, Xh, Xl, XLD, lane, kv);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int row = min(32 * mt + tl, ROWS_T - 1);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = 8 * g + 4 * half;
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(W + LY::bk + c);
                    *reinterpret_cast<f32x4*>(&TK[row * KLD + c]) =
                        (f32x4){kv[0][mt][4 * g] + b4[0], kv[0][mt][4 * g + 1] + b4[1], kv[0][mt][4 * g + 2] + b4[2], kv[0][mt][4 * g + 3] + b4[3]};
                }
            }
            load_w<1, 2>(F + FL::fp, lane, wp);            // in flight over the attention arithmetic
            wait_w<4>(wv);
            mm<1, 2>(wv, Xh, Xl, XLD, lane, kv);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int row = min(32 * mt + tl, ROWS_T - 1);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = 8 * g + 4 * half;
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(W + LY::bv + c);
                    *reinterpret_cast<f32x4*>(&TV[row * KLD + c]) =
                        (f32x4){kv[0][mt][4 * g] + b4[0], kv[0][mt][4 * g + 1] + b4[1], kv[0][mt][4 * g + 2] + b4[2], kv[0][mt][4 * g + 3] + b4[3]};
                }
            }
        }
        // scaled dot-product attention over the J joints of the token's frame; group g = head 2g + half
        float o[2][16];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 8 * g + 4 * half;
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(W + LY::bq + c);
                const f32x4 q4 = {q[0][mt][4 * g] + b4[0], q[0][mt][4 * g + 1] + b4[1], q[0][mt][4 * g + 2] + b4[2], q[0][mt][4 * g + 3] + b4[3]};
                const f32x4 o4 = head_attention<J>(q4, TK + fbase[mt] * KLD + c, TV + fbase[mt] * KLD + c);
                o[mt][4 * g] = o4[0]; o[mt][4 * g + 1] = o4[1]; o[mt][4 * g + 2] = o4[2]; o[mt][4 * g + 3] = o4[3];
            }
        store_planes(Xh, Xl, XLD, 0, lane, o);
        {
            float pr[1][2][16];
            wait_w<0>(wp);
            mm<1, 2>(wp, Xh, Xl, XLD, lane, pr);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(W + LY::bp + 8 * g + 4 * half);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[mt][4 * g + e] += pr[0][mt][4 * g + e] + b4[e];
            }
        }

        // ---- MLP half ----
        WFrag<2, 2> w1;
        load_w<2, 2>(F + FL::f1, lane, w1);
        ln_tokens(x, W + LY::ln2_g, W + LY::ln2_b, 1e-5f, half, y);
        store_planes(Xh, Xl, XLD, 0, lane, y);
        WFrag<1, 4> w2;
        {
            float hd[2][2][16];
            wait_w<0>(w1);
            mm<2, 2>(w1, Xh, Xl, XLD, lane, hd);
            load_w<1, 4>(F + FL::f2, lane, w2);             // in flight over the GELU
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(W + LY::b1 + 32 * nt + 8 * g + 4 * half);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int e = 0; e < 4; ++e) hd[nt][mt][4 * g + e] = sv2::gelu_erf(hd[nt][mt][4 * g + e] + b4[e]);
                }
                store_planes(Hh, Hl, HLD, 32 * nt, lane, hd[nt]);
            }
        }
        {
            float z[1][2][16];
            wait_w<0>(w2);
            mm<1, 4>(w2, Hh, Hl, HLD, lane, z);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(W + LY::b2 + 8 * g + 4 * half);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[mt][4 * g + e] += z[0][mt][4 * g + e] + b4[e];
            }
        }
    }

    float y[2][16];
    sh3::ln_tokens(x, p.norm_g, p.norm_b, 1e-6f, half, y);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        if (!valid[mt]) continue;
        const size_t at = ((size_t)frame[mt] * J + joint[mt]) * DS;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = 8 * g + 4 * half;
            const f32x4 f = {y[mt][4 * g], y[mt][4 * g + 1], y[mt][4 * g + 2], y[mt][4 * g + 3]};
            if (out_lo != nullptr) {
                h16x4 hi, lo;
                h3_split(f, hi, lo);
                *reinterpret_cast<h16x4*>(out_hi + at + c) = hi;
                *reinterpret_cast<h16x4*>(out_lo + at + c) = lo;
            } else {
                *reinterpret_cast<f32x4*>(out + at + c) = f;
            }
        }
    }
}

}  // namespace uu3d
