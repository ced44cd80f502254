// uu3d_attn.h -- temporal / strided self-attention for one (sequence, head) per workgroup.
//
// Replaces vit.MHA.scaled_dot_product_attention (vision_transformer.py:99-130) for the
// temporal stacks: logits = Q K^T / sqrt(d_h) (+ mask * -1e9), softmax over keys, P V.
// The probabilities are never written to memory (the reference materialises and returns them).
//
// Layout: qkv rows are [q(D) | k(D) | v(D)], head h owns channels [h*DH, (h+1)*DH).
// One workgroup = one (b, h); wave w owns query tile w (16 queries); NT = ceil(L / 16) waves.
// K and V of the head are staged once in LDS (zero padded to NT*16 rows, row stride DH+4).
//
// MFMA: v_mfma_f32_16x16x4_f32 (exact f32).  The wave computes the TRANSPOSED logit tile
// S^T = K Q^T (A = K rows, B = Q^T), so that in the C/D map (col = lane & 15 -> query,
// row = 4 * (lane >> 4) + reg -> key) every lane holds, for its query, the keys
// 16j + 4g + r.  The softmax over keys is then in-lane plus two cross-lane steps
// (xor 16, 32), and the probability registers are ALREADY the A operand of the P V product:
// A[i = query = lane & 15][k-slot g, step (j, s)] = P[query][key 16j + 4g + s] -- the
// k-slot order of an MFMA is free as long as B uses the same one, so V is read as
// V[16j + 4g + s][d].  No LDS round trip and no shuffle for P.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "uu3d_gemm.h"

namespace uu3d {

// SPLIT: the context rows are written as the two f16 planes (hi at out, lo at out + lo_off halfs; x ~= hi + lo / 2048,
// see uu3d_gemm_h3.h) that the f16x3 projection GEMM reads, instead of f32.  lo_off == ATTN_FRAG_ORDER: the planes go
// out in the row-panel GEMM's A-fragment order instead ([32-row panel][16-deep k-slice][plane][lane][8 halfs],
// uu3d_gemm_panel.h) for a contraction length of D.
static constexpr size_t ATTN_FRAG_ORDER = ~(size_t)0;
// A workgroup handles the items bh = blockIdx.x, blockIdx.x + gridDim.x, ... (n_items = B * H in all): with fewer
// workgroups than items the K / V / Q loads of the NEXT item are issued into registers before the current one is computed
// (one workgroup per item, all 1024 resident at once, ran load -> compute -> store in lockstep on every CU: 31 us of which
// ~9 is the memory floor of the 42 MB QKV tensor and ~14 the MFMA + softmax work).
template <int NT, int DH, bool SPLIT = false>
__global__ void __launch_bounds__(64 * NT)
attn_f32_kernel(const float* __restrict__ qkv, const int ld, const int D, const int L, const int H,
                const uint8_t* __restrict__ key_mask,   // (B, L) 1 = attend; nullptr = no mask
                float* __restrict__ out, const int ldo, const size_t lo_off = 0, const int n_items = 0)
{
    static_assert(DH % 16 == 0, "head dim must be a multiple of 16");
    constexpr int LD = DH + 4;
    constexpr int F4 = DH / 4;
    constexpr int KT = DH / 16;             // float4 k-groups per lane
    constexpr int NS = (16 * F4 + 63) / 64; // staging float4 per thread and matrix
    __shared__ __attribute__((aligned(16))) float Ks[NT * 16 * LD];
    __shared__ __attribute__((aligned(16))) float Vs[NT * 16 * LD];

    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int qi = lane & 15, g = lane >> 4;
    const int qrow = 16 * w + qi;
    const int items = n_items > 0 ? n_items : (int)gridDim.x;
    // Workgroups go round-robin over the 8 XCDs; remapped so that CONSECUTIVE items (the H heads of one sequence) run on the
    // SAME XCD: a head's 192-byte q / k / v slices straddle 128-byte lines that the neighbouring head also needs, and the heads'
    // partial-line output writes meet in one L2 instead of eight.
    const int first = ((int)gridDim.x & 7) == 0 ? ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;

    float4 kreg[NS], vreg[NS];
    f32x4 qnext[KT];
    auto issue = [&](const int bh) {
        const int b = bh / H, h = bh - b * H;
        const float* base = qkv + (size_t)b * L * ld + h * DH;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int idx = tid + s * 64 * NT;
            const int row = idx / F4, c4 = (idx - row * F4) * 4;
            kreg[s] = make_float4(0.f, 0.f, 0.f, 0.f); vreg[s] = kreg[s];
            if (idx < NT * 16 * F4 && row < L) {
                const float* p = base + (size_t)row * ld + c4;
                kreg[s] = *reinterpret_cast<const float4*>(p + D);
                vreg[s] = *reinterpret_cast<const float4*>(p + 2 * D);
            }
        }
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            qnext[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (qrow < L) qnext[t] = *reinterpret_cast<const f32x4*>(base + (size_t)qrow * ld + 16 * t + 4 * g);
        }
    };
    issue(first);
    for (int bh = first; ; ) {
    const int b = bh / H, h = bh - b * H;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int idx = tid + s * 64 * NT;
        const int row = idx / F4, c4 = (idx - row * F4) * 4;
        if (idx < NT * 16 * F4) {
            *reinterpret_cast<float4*>(&Ks[row * LD + c4]) = kreg[s];
            *reinterpret_cast<float4*>(&Vs[row * LD + c4]) = vreg[s];
        }
    }
    f32x4 qf[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) qf[t] = qnext[t];
    __syncthreads();
    const int bh_next = bh + (int)gridDim.x;
    if (bh_next < items) issue(bh_next);               // in flight while this item is computed

    // S^T tiles: st[j][r] = <Q[qrow], K[16j + 4g + r]>
    f32x4 st[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            const f32x4 kf = *reinterpret_cast<const f32x4*>(&Ks[(16 * j + qi) * LD + 16 * t + 4 * g]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s], qf[t][s], a, 0, 0, 0);
        }
        st[j] = a;
    }

    // logits / sqrt(d_h) (+ mask * -1e9), softmax over keys
    // 1 / sqrt(d_h) as a multiplication, exp as exp2 of a pre-scaled argument and one reciprocal of the sum: the division,
    // libm expf and per-probability division cost ~1000 VALU instructions per wave (as much SIMD time as the MFMAs);
    // the results differ from those forms by rounding only (parity bar 1e-4 on the outputs, checked by the tests)
    const float scale_mul = 1.0f / sqrtf((float)DH);
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = 16 * j + 4 * g + r;
            float v = st[j][r] * scale_mul;
            if (key < L) {
                if (key_mask != nullptr) v += (key_mask[(size_t)b * L + key] ? 0.0f : 1.0f) * -1e9f;
            } else {
                v = -INFINITY;
            }
            st[j][r] = v;
            mx = fmaxf(mx, v);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#ifndef UU3D_ATTN_LIBM_EXP
            const float e = __builtin_amdgcn_exp2f((st[j][r] - mx) * 1.44269504088896341f);
#else
            const float e = expf(st[j][r] - mx);
#endif
            st[j][r] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float rsum = 1.0f / sum;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) st[j][r] = st[j][r] * rsum;

    // O = P V : tile t covers head channels 16t .. 16t+15
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float vv = Vs[(16 * j + 4 * g + s) * LD + 16 * t + qi];
                o = __builtin_amdgcn_mfma_f32_16x16x4f32(st[j][s], vv, o, 0, 0, 0);
            }
        // C/D map: col = lane & 15 -> channel, row = 4g + r -> query
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = 16 * w + 4 * g + r;
            if (q < L) {
                const size_t at = ((size_t)b * L + q) * ldo + h * DH + 16 * t + qi;
                if (SPLIT) {
                    _Float16* oh = reinterpret_cast<_Float16*>(out);
                    const _Float16 hv = (fabsf(o[r]) < 6.103515625e-05f) ? (_Float16)0.f : (_Float16)o[r];   // = h3_hi (uu3d_gemm_h3.h), explicit: this kernel keeps f16 denormals on
                    const _Float16 lv = (_Float16)((o[r] - (float)hv) * 2048.0f);
                    if (lo_off == ATTN_FRAG_ORDER) {
                        const int row = b * L + q, k = h * DH + 16 * t + qi;
                        const size_t fi = ((((size_t)(row >> 5) * (D >> 4) + (k >> 4)) * 2) * 64 + ((k >> 3) & 1) * 32 + (row & 31)) * 8 + (k & 7);
                        oh[fi] = hv; oh[fi + 512] = lv;
                        continue;
                    }
                    oh[at] = hv;
                    oh[lo_off + at] = lv;
                } else {
                    out[at] = o[r];
                }
            }
        }
    }
    if (bh_next >= items) break;
    __syncthreads();                                    // every wave is done with this item's K / V tiles
    bh = bh_next;
    }
}

}  // namespace uu3d
