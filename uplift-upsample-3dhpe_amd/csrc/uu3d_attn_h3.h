// uu3d_attn_h3.h -- temporal self-attention with f16x3 products and an online softmax over key tiles: sequences of up
// to 384 tokens (SURVEY 8(d)'s "synthetic dense-351": 351 -> 117 -> 13 -> 1), and the shipped 71-token ones.
//
// Replaces vit.MHA.scaled_dot_product_attention (vision_transformer.py:99-130) like uu3d_attn.h; the exact-f32 kernels
// there hold all logits of a query tile in registers (<= 128 keys) and run on the f32 MFMA (1/16 of the f16 rate).
//
// One workgroup = one (sequence, head); wave w owns the 32-query tiles w, w + NW, ...
//   * K and V of the head are split ONCE into f16 hi / lo planes (x ~= hi + lo / 2048, uu3d_gemm_h3.h) while they are
//     staged into LDS, in the layouts the MFMA operand reads want:
//         K:  [plane][16-deep k-slice (3)][key][16 halfs]   a fragment read is one linear, conflict-free ds_read_b128
//         Vt: [plane][channel d (48) + a row of ones][key], row stride RS
//                                                           transposed; RS = 2 * odd dwords: the 32 lanes of a half read
//                                                           32 different channels at bank offsets 2 * (odd * d mod 32)
//   * logits TRANSPOSED, S^T = K Q^T with v_mfma_f32_32x32x16_f16 (K fragment = A, Q^T fragment = B, 3 k-slices x 3
//     passes): in the C/D map a lane holds ONE query (lane & 31) and 16 keys of the tile, so the softmax is in-lane plus
//     one lane ^ 32 exchange, and the probability registers are, converted to f16 pairs, already the B operand of
//         O^T = V^T P^T       (A = V^T fragment from Vt; MICROARCH guide: "an accumulator tile as the next MFMA's operand")
//     whose k order inside a 16-key step is 8 (j >> 2) + 4 h + (j & 3) -- the Vt reads use the same order.
//   * The softmax costs as many issue cycles as the MFMAs unless it is kept short (one wave: ~16 values x 20 VALU
//     instructions per tile at first), so the per-value work is folded away wherever the algebra allows:
//       - Q is multiplied by log2(e) / sqrt(d_h) before it is split: the accumulator is the exp2 argument;
//       - the accumulator STARTS at (key mask term - running maximum + 14): no subtraction per value;  [unmasked launches]
//       - probabilities are 2^14 times too large (p <= 16384 fits f16; the common factor cancels in O / l): no f16
//         denormals to flush, so hi comes from the packed conversion and lo = f16((p - hi) * 2048);
//       - the row sum l is row 48 of O^T: the padding rows of the second O^T tile read a row of ONES from Vt;
//       - O^T is rescaled only when some maximum of the wave grew (never after the first tiles of most rows).
//     Masked keys add -1e9 (finite, like the reference: an all-masked row stays uniform -- there the mask term is added in
//     f32 BEFORE the maximum is subtracted, as the reference does), keys past L add -inf.
//   * O^T has the query on the lane: 1 / l is a per-lane scalar; the two f16 planes the projection GEMM reads leave as
//     16-byte stores after a v_permlane32_swap pairs the 8-byte groups of lanes l and l ^ 32.
// The head dim 48 is 1.5 MFMA rows: the second 32-row tile of O^T is a third padding (25 % of the P V MFMAs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "uu3d_gemm_h3.h"

namespace uu3d {

static constexpr int ATTN_H3_MAX_L = 384;              // 12 key tiles: K + Vt planes of one head (148 KiB) fit the 160 KiB LDS; 13 tiles miss it by 16 bytes
__host__ __device__ inline constexpr int attn_h3_lpad(int L) { return (L + 31) / 32 * 32; }
// Vt row stride in halfs: >= Lpad, and 2 * odd as a dword count (see top): 4 * odd halfs
__host__ __device__ inline constexpr int attn_h3_vt_rs_halfs(int L) {
    int odd = (attn_h3_lpad(L) + 3) / 4;
    if ((odd & 1) == 0) odd += 1;
    return odd * 4;
}
__host__ __device__ inline constexpr size_t attn_h3_lds_bytes(int L, int DH) {
    return (size_t)2 * (DH / 16) * attn_h3_lpad(L) * 16 * sizeof(_Float16)          // K planes
         + (size_t)2 * (DH + 1) * attn_h3_vt_rs_halfs(L) * sizeof(_Float16)         // Vt planes + the row of ones
         + (size_t)attn_h3_lpad(L) * sizeof(float);                                 // additive key mask (x log2 e)
}

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

// MAXW = waves per workgroup the instantiation is compiled for; WPE = waves per SIMD the register budget must allow
// (short sequences: 3 waves per workgroup, four workgroups per CU = 3 per SIMD; long ones: 8 waves = 2 per SIMD).
// MASKED = a key mask is given (temporal block 1): the mask term is added before the running maximum is subtracted.
template <int DH, int MAXW, int WPE, bool MASKED>
__global__ void __launch_bounds__(64 * MAXW) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
attn_h3_kernel(const float* __restrict__ qkv, const int ld, const int D, const int L, const int H,
               const uint8_t* __restrict__ key_mask,   // (B, L) 1 = attend; nullptr = no mask
               _Float16* __restrict__ out, const size_t lo_off, const int ldo)
{
    static_assert(DH == 48, "operand layouts below are written for a head dim of 48 (3 k-slices, 1.5 output row tiles)");
    constexpr int KS = DH / 16;                        // k-slices of Q K^T
    constexpr int F4 = DH / 4;                         // float4 pieces per row
    constexpr float PSHIFT = 14.0f;                    // probabilities carry a factor 2^14 (see top)
    h3_flush_f16_denormals();                          // K / V / output planes: hi = 0 below the smallest normal half (uu3d_gemm_h3.h)
    extern __shared__ __attribute__((aligned(16))) unsigned char asm_[];
    const int Lpad = attn_h3_lpad(L), RS = attn_h3_vt_rs_halfs(L), NT = Lpad >> 5;
    _Float16* Kp = reinterpret_cast<_Float16*>(asm_);                               // [2][KS][Lpad][16]
    _Float16* Vt = Kp + (size_t)2 * KS * Lpad * 16;                                 // [2][DH + 1][RS]
    float* madd = reinterpret_cast<float*>(Vt + (size_t)2 * (DH + 1) * RS);         // [Lpad]
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, w = tid >> 6, NW = nthr >> 6;
    // consecutive workgroups (the heads of one sequence) on one XCD: they read neighbouring 192-byte column slices of the same rows
    const int bh = ((int)gridDim.x & 7) == 0 ? ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int b = bh / H, h = bh - b * H;
    const float* base = qkv + (size_t)b * L * ld + h * DH;
    constexpr float LOG2E = 1.44269504088896341f;

    // ---- K, V -> f16 planes in LDS.  ALL loads of the thread first (branch-free: clamped row, zeroed afterwards), one
    // memory round trip; the first query tile's rows go out with them ----
    constexpr int NPT = MAXW <= 3 ? 6 : 9;             // float4 pieces per thread and matrix: 12 Lpad / (64 waves), Lpad = 32 NT <= 384, waves = min(NT, 8)
    const int q31 = lane & 31, g = lane >> 5;
    f32x4 qx[KS][2];
    auto load_q = [&](int qt) {
        const float* qp = base + (size_t)min(32 * qt + q31, L - 1) * ld + g * 8;
#pragma unroll
        for (int s = 0; s < KS; ++s) { qx[s][0] = *reinterpret_cast<const f32x4*>(qp + 16 * s); qx[s][1] = *reinterpret_cast<const f32x4*>(qp + 16 * s + 4); }
    };
    {
        f32x4 kx[NPT], vx[NPT];
#pragma unroll
        for (int u = 0; u < NPT; ++u) {
            const int idx = u * nthr + tid, row = idx / F4, c4 = (idx - row * F4) * 4;
            const float* p = base + (size_t)min(row, L - 1) * ld + c4;
            kx[u] = *reinterpret_cast<const f32x4*>(p + D);
            vx[u] = *reinterpret_cast<const f32x4*>(p + 2 * D);
        }
        load_q(w);
#pragma unroll
        for (int u = 0; u < NPT; ++u) {
            const int idx = u * nthr + tid, row = idx / F4, c4 = (idx - row * F4) * 4;
            if (idx < Lpad * F4) {
                const float keep = row < L ? 1.0f : 0.0f;
                h16x4 hi, lo;
                h3_split(kx[u] * keep, hi, lo);
                _Float16* kd = Kp + ((size_t)(c4 >> 4) * Lpad + row) * 16 + (c4 & 15);
                *reinterpret_cast<h16x4*>(kd) = hi;
                *reinterpret_cast<h16x4*>(kd + (size_t)KS * Lpad * 16) = lo;
                h3_split(vx[u] * keep, hi, lo);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    Vt[(size_t)(c4 + e) * RS + row] = hi[e];
                    Vt[(size_t)((DH + 1) + c4 + e) * RS + row] = lo[e];
                }
            }
        }
    }
    for (int k = tid; k < Lpad; k += nthr) {
        const uint8_t mk = (MASKED && key_mask != nullptr) ? key_mask[(size_t)b * L + min(k, L - 1)] : (uint8_t)1;
        madd[k] = (k < L) ? (mk ? 0.0f : -1e9f * LOG2E) : -INFINITY;
        Vt[(size_t)DH * RS + k] = (_Float16)1.0f;                  // row DH of the hi plane: ones (the row sum l comes out as row DH of O^T)
        Vt[(size_t)((DH + 1) + DH) * RS + k] = (_Float16)0.0f;
    }
    __syncthreads();

    const float cscale = LOG2E / sqrtf((float)DH);                 // logits / sqrt(d_h), in log2 units
    const int dA = q31, dB = min(32 + q31, DH);                    // channel rows of the two O^T tiles; rows >= DH of tile 1 read the ones

    for (int qt = w; qt < NT; qt += NW) {                          // MAXW <= 3: exactly one pass
        // ---- Q^T fragments of this tile: lane = query, 8 consecutive k per slice, pre-scaled, split in registers; the next
        // tile's rows are requested right away ----
        h16x8 qh[KS], ql[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            h16x4 a, bq, c, d;
            h3_split(qx[s][0] * cscale, a, c); h3_split(qx[s][1] * cscale, bq, d);
            qh[s] = (h16x8){a[0], a[1], a[2], a[3], bq[0], bq[1], bq[2], bq[3]};
            ql[s] = (h16x8){c[0], c[1], c[2], c[3], d[0], d[1], d[2], d[3]};
        }
        if (MAXW > 3 && qt + NW < NT) load_q(qt + NW);          // (MAXW <= 3 instantiation: one wave per tile, launched with NW = NT)
        f32x16 oA0, oA1, oB0, oB1;                                 // O^T tiles (channels 0-31 / 32-47 + l), hi-hi and cross-term accumulators
#pragma unroll
        for (int r = 0; r < 16; ++r) { oA0[r] = 0.f; oA1[r] = 0.f; oB0[r] = 0.f; oB1[r] = 0.f; }
        float m_off = -PSHIFT;                                     // running maximum - PSHIFT (log2 units); the first tile sets it

        for (int kt = 0; kt < NT; ++kt) {
            // ---- S^T = K Q^T (32 keys x 32 queries); lane holds keys 32 kt + 8 j + 4 g + i of its query.  The hi-hi
            // accumulator starts at the key's mask term (minus the running maximum when nothing is masked) ----
            f32x16 s0, s1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 ma = *reinterpret_cast<const f32x4*>(madd + 32 * kt + 8 * j + 4 * g);
#pragma unroll
                for (int i = 0; i < 4; ++i) { s0[4 * j + i] = MASKED ? ma[i] : ma[i] - m_off; s1[4 * j + i] = 0.f; }
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const _Float16* kp = Kp + ((size_t)s * Lpad + 32 * kt + q31) * 16 + g * 8;
                const h16x8 kh = *reinterpret_cast<const h16x8*>(kp);
                const h16x8 kl = *reinterpret_cast<const h16x8*>(kp + (size_t)KS * Lpad * 16);
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[s], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[s], s1, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[s], s1, 0, 0, 0);
            }
            // ---- online softmax ----
            float t[16];
            float tmax = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) { t[r] = fmaf(s1[r], 1.0f / H3_SCALE, s0[r]); tmax = fmaxf(tmax, t[r]); }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
            if (MASKED) {
                // t = logit + mask term, rounded in f32 like the reference's sum.  The running maximum can jump by 1e9 (first
                // tiles masked, a later one not): every quantity is formed from the raw values, never through a difference of
                // two shifted ones.  An all-masked row has t == its maximum exactly: uniform probabilities.
                const float m_old = m_off + PSHIFT;
                const float m_new = kt == 0 ? tmax : fmaxf(m_old, tmax);
                const float alpha = kt == 0 ? 1.0f : __builtin_amdgcn_exp2f(m_old - m_new);
                m_off = m_new - PSHIFT;
#pragma unroll
                for (int r = 0; r < 16; ++r) t[r] -= m_off;
                if (__any(alpha != 1.0f)) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { oA0[r] *= alpha; oA1[r] *= alpha; oB0[r] *= alpha; oB1[r] *= alpha; }
                }
            } else {
                // t = logit - running maximum + PSHIFT already.  delta = how far the maximum moves (the first tile may move it down)
                float delta = tmax - PSHIFT;
                if (kt > 0) delta = fmaxf(delta, 0.f);
                if (__any(delta != 0.f)) {                         // wave-uniform; after the first tiles almost never taken
                    const float alpha = kt == 0 ? 1.0f : __builtin_amdgcn_exp2f(-delta);       // (O^T is still zero in the first tile)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { t[r] -= delta; oA0[r] *= alpha; oA1[r] *= alpha; oB0[r] *= alpha; oB1[r] *= alpha; }
                    m_off += delta;
                }
            }
            h16x8 ph[2], pl[2];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float p0 = __builtin_amdgcn_exp2f(t[r]), p1 = __builtin_amdgcn_exp2f(t[r + 1]);
                const h16x2 hv = __builtin_bit_cast(h16x2, __builtin_amdgcn_cvt_pkrtz(p0, p1));      // hi: any rounding will do, lo takes the rest
                const h16x2 lv = __builtin_bit_cast(h16x2, __builtin_amdgcn_cvt_pkrtz((p0 - (float)hv[0]) * H3_SCALE, (p1 - (float)hv[1]) * H3_SCALE));
                ph[r >> 3][r & 7] = hv[0]; ph[r >> 3][(r & 7) + 1] = hv[1];
                pl[r >> 3][r & 7] = lv[0]; pl[r >> 3][(r & 7) + 1] = lv[1];
            }
            // ---- O^T += V^T P^T : two 16-key steps, A = V^T fragment (keys 16 s + 8 (j >> 2) + 4 g + (j & 3)) ----
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (s == 1 && 32 * kt + 16 >= L) break;           // the step holds padding keys only (p = 0)
                const int key0 = 32 * kt + 16 * s + 4 * g;
                auto vfrag = [&](int d, int plane) {
                    const _Float16* vp = Vt + (size_t)(plane * (DH + 1) + d) * RS + key0;
                    const h16x4 a = *reinterpret_cast<const h16x4*>(vp), c = *reinterpret_cast<const h16x4*>(vp + 8);
                    return (h16x8){a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
                };
                const h16x8 vAh = vfrag(dA, 0), vAl = vfrag(dA, 1), vBh = vfrag(dB, 0), vBl = vfrag(dB, 1);
                oA0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vAh, ph[s], oA0, 0, 0, 0);
                oA1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vAh, pl[s], oA1, 0, 0, 0);
                oA1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vAl, ph[s], oA1, 0, 0, 0);
                oB0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vBh, ph[s], oB0, 0, 0, 0);
                oB1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vBh, pl[s], oB1, 0, 0, 0);
                oB1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vBl, ph[s], oB1, 0, 0, 0);
            }
        }
        // ---- normalise, split, store: lane (query, g) holds channels 32 t + 8 j + 4 g + (0..3); rows 16 + 4 g of the second
        // tile (register 8) are the ones-row product = the row sum l, with the same 2^14 factor as every other row ----
        const float rl = 1.0f / (oB0[8] + oB1[8] * (1.0f / H3_SCALE));
        const int q = 32 * qt + q31;
        _Float16* orow = out + (size_t)(b * L + min(q, L - 1)) * ldo + h * DH;
        auto pack4 = [&](const f32x16& a0, const f32x16& a1, int j, unsigned (&hi2)[2], unsigned (&lo2)[2]) {
            _Float16 hh[4], ll[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = (a0[4 * j + i] + a1[4 * j + i] * (1.0f / H3_SCALE)) * rl;
                hh[i] = h3_hi(v);
                ll[i] = (_Float16)((v - (float)hh[i]) * H3_SCALE);
            }
            hi2[0] = __builtin_bit_cast(unsigned, (h16x2){hh[0], hh[1]}); hi2[1] = __builtin_bit_cast(unsigned, (h16x2){hh[2], hh[3]});
            lo2[0] = __builtin_bit_cast(unsigned, (h16x2){ll[0], ll[1]}); lo2[1] = __builtin_bit_cast(unsigned, (h16x2){ll[2], ll[3]});
        };
        // blocks (j, j + 1) of 8 channels: after the swap lane g = 0 owns all 16 bytes of block j, lane g = 1 those of block j + 1
        auto store_pair = [&](const f32x16& a0, const f32x16& a1, int j, int ch0) {
            unsigned xh[2], xl[2], yh[2], yl[2];
            pack4(a0, a1, j, xh, xl); pack4(a0, a1, j + 1, yh, yl);
            unsigned oh[4], ol[4];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const u32x2 sh = __builtin_amdgcn_permlane32_swap(xh[e], yh[e], false, false);
                const u32x2 sl = __builtin_amdgcn_permlane32_swap(xl[e], yl[e], false, false);
                oh[e] = sh[0]; oh[2 + e] = sh[1]; ol[e] = sl[0]; ol[2 + e] = sl[1];
            }
            if (q < L) {
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                _Float16* d = orow + ch0 + 8 * (j + g);
                *reinterpret_cast<u32x4*>(d) = (u32x4){oh[0], oh[1], oh[2], oh[3]};
                *reinterpret_cast<u32x4*>(d + lo_off) = (u32x4){ol[0], ol[1], ol[2], ol[3]};
            }
        };
        store_pair(oA0, oA1, 0, 0); store_pair(oA0, oA1, 2, 0); store_pair(oB0, oB1, 0, 32);
    }
}

static_assert(attn_h3_lds_bytes(ATTN_H3_MAX_L, 48) <= 160 * 1024, "the largest sequence must fit the LDS");

}  // namespace uu3d
