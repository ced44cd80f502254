// uu3d_attn_h3.h -- temporal self-attention with f16x3 products and an online softmax over key tiles: sequences of up
// to 416 tokens (SURVEY 8(d)'s "synthetic dense-351": 351 -> 117 -> 13 -> 1), and the shipped 71-token ones.
//
// Replaces vit.MHA.scaled_dot_product_attention (vision_transformer.py:99-130) like uu3d_attn.h; the exact-f32 kernels
// there hold all logits of a query tile in registers (<= 128 keys) and run on the f32 MFMA (1/16 of the f16 rate).
//
// One workgroup = one (sequence, head); wave w owns the 32-query tiles w, w + NW, ...
//   * q, k, v arrive as f16 hi / lo PLANES (x ~= hi + lo / 2048, uu3d_gemm_h3.h), written by the QKV projection's epilogue
//     (q already multiplied by log2(e) / sqrt(d_h)): the kernel does no splitting, and K and V of the head go global -> LDS
//     by LDS-DMA (global_load_lds_dwordx4, 12 instructions per 32 keys, no registers, no VALU) in the layouts the MFMA
//     operand reads want:
//         K: [plane][16-deep k-slice (3)][key][16 halfs]    a fragment read is one linear, conflict-free ds_read_b128
//         V: [key][hi 48 halfs | lo 48 halfs]               row major, 192-byte rows; the V^T fragments come out of
//                                                           ds_read_b64_tr_b16 (4 keys x 16 channels per 16 lanes; the 4 rows
//                                                           of a 32-lane half fall on 4 x 64 bytes = all 64 banks)
//   * logits TRANSPOSED, S^T = K Q^T with v_mfma_f32_32x32x16_f16 (K fragment = A, Q^T fragment = B, 3 k-slices x 3
//     passes): in the C/D map a lane holds ONE query (lane & 31) and 16 keys of the tile, so the softmax is in-lane plus
//     one lane ^ 32 exchange, and the probability registers are, converted to f16 pairs, already the B operand of
//         O^T = V^T P^T       (A = V^T fragment; MICROARCH guide: "an accumulator tile as the next MFMA's operand")
//     whose k order inside a 16-key step is 8 (j >> 2) + 4 h + (j & 3) -- two transposed reads of 4 keys each per fragment.
//   * The softmax costs as many issue cycles as the MFMAs unless it is kept short (one wave: ~16 values x 20 VALU
//     instructions per tile at first), so the per-value work is folded away wherever the algebra allows:
//       - q arrives multiplied by log2(e) / sqrt(d_h): the accumulator is the exp2 argument;
//       - the accumulator STARTS at (key mask term - running maximum + 14): no subtraction per value;  [unmasked launches]
//       - probabilities are 2^14 times too large (p <= 16384 fits f16; the common factor cancels in O / l): no f16
//         denormals to flush, so hi comes from the packed conversion and lo = f16((p - hi) * 2048);
//       - the row sum l is row 48 of O^T: the padding rows of the second O^T tile read a block of ONES instead of V;
//       - O^T is rescaled only when some maximum of the wave grew (never after the first tiles of most rows).
//     Masked keys add -1e9 (finite, like the reference: an all-masked row stays uniform -- there the mask term is added in
//     f32 BEFORE the maximum is subtracted, as the reference does), keys past L add -inf.
//   * O^T has the query on the lane: 1 / l is a per-lane scalar; the two f16 planes the projection GEMM reads leave as
//     16-byte stores after a v_permlane32_swap pairs the 8-byte groups of lanes l and l ^ 32 -- row-major planes for the tiled
//     LDS-DMA GEMM, or (frag) the fragment order the row-panel GEMM reads: the same pieces at other addresses.
// The head dim 48 is 1.5 MFMA rows: the second 32-row tile of O^T is a third padding (25 % of the P V MFMAs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "uu3d_gemm_h3.h"

namespace uu3d {

static constexpr int ATTN_H3_MAX_L = 416;              // 13 key tiles: K + V planes of one head (158 KiB) fit the 160 KiB LDS
__host__ __device__ inline constexpr int attn_h3_lpad(int L) { return (L + 31) / 32 * 32; }
__host__ __device__ inline constexpr size_t attn_h3_lds_bytes(int L, int DH) {
    return (size_t)2 * DH * attn_h3_lpad(L) * sizeof(_Float16) * 2          // K planes + V rows
         + 64                                                                   // 16 ones, 16 zeros (halfs)
         + (size_t)attn_h3_lpad(L) * sizeof(float);                             // additive key mask (x log2 e)
}

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

// Timing builds only (tools/attn_loo_exp.hip; results are wrong by construction): UU3D_ATTN_LOO is a mask of what to leave out of the
// kernel -- 1 the K / V LDS-DMA, 2 the S^T MFMAs, 4 the exponentials and the hi / lo conversion of the probabilities, 8 the O^T MFMAs,
// 16 the output stores, 32 the V^T fragment reads, 64 the query fragment loads.  A left-out MFMA is replaced by one FMA on its
// operands, so that what feeds it stays alive.
#ifndef UU3D_ATTN_LOO
#define UU3D_ATTN_LOO 0
#endif
#define UU3D_ATTN_MFMA(bit, a, b, c) (((UU3D_ATTN_LOO) & (bit)) ? attn_loo_fma(a, b, c) : __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0))
__device__ __forceinline__ f32x16 attn_loo_fma(const h16x8& a, const h16x8& b, f32x16 c) { c[0] = fmaf((float)a[0], (float)b[0], c[0]); return c; }

// The end of a query tile: normalise O^T by the row sum, split into f16 hi / lo, store (used by attn_h3_kernel and attn_h3_pp_kernel).
template <int DH>
__device__ __forceinline__ void attn_h3_store_tile(const f32x16& oA0, const f32x16& oA1, const f32x16& oB0, const f32x16& oB1, const int qt, const int q31, const int g,
                                                   const int h, const int L, const size_t tok0, _Float16* __restrict__ out, const size_t lo_off, const int ldo, const int frag)
{
    // ---- normalise, split, store: lane (query, g) holds channels 32 t + 8 j + 4 g + (0..3); rows 16 + 4 g of the second
    // tile (register 8) are the ones-row product = the row sum l, with the same 2^14 factor as every other row ----
    const float rl = 1.0f / (oB0[8] + oB1[8] * (1.0f / H3_SCALE));
    const int q = 32 * qt + q31;
    // row-major planes: row * ldo + channel.  Fragment order: a 16-byte piece = 8 consecutive channels of one row, the pieces of 32
    // consecutive rows are contiguous (512 B): [32-row panel][16-channel slice][plane][channel half][row & 31][8]
    const size_t grow = tok0 + min(q, L - 1);
    _Float16* orow = frag ? out + (size_t)(grow >> 5) * (size_t)(ldo >> 4) * 1024 + (grow & 31) * 8
                          : out + grow * ldo + h * DH;
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
        if (((UU3D_ATTN_LOO) & 16) ? (oh[0] == 0x12345678u && ol[3] == 0x9abcdef0u) : q < L) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const int chn = h * DH + ch0 + 8 * (j + g);         // first of the piece's 8 channels
            _Float16* d = frag ? orow + (size_t)(chn >> 4) * 1024 + ((chn >> 3) & 1) * 256 : orow + ch0 + 8 * (j + g);
            *reinterpret_cast<u32x4*>(d) = (u32x4){oh[0], oh[1], oh[2], oh[3]};
            *reinterpret_cast<u32x4*>(d + lo_off) = (u32x4){ol[0], ol[1], ol[2], ol[3]};
        }
    };
    store_pair(oA0, oA1, 0, 0); store_pair(oA0, oA1, 2, 0); store_pair(oB0, oB1, 0, 32);
}

// MAXW = waves per workgroup the instantiation is compiled for; WPE = waves per SIMD the register budget must allow
// (short sequences: 3 waves per workgroup, four workgroups per CU = 3 per SIMD; long ones: 8 waves = 2 per SIMD).
// MASKED = a key mask is given (temporal block 1): the mask term is added before the running maximum is subtracted.
// PIPE (round 4): the LDS operand reads of a key tile are issued BY NAME ahead of their use -- the six K fragments of S^T = K Q^T at
// the top of the tile (hipcc emitted each read directly in front of its MFMA with a full lgkmcnt(0): five exposed LDS round trips
// per tile), the first 16-key step's V^T fragments behind the last S^T MFMA (they land during the softmax), the second step's
// into the K-fragment registers behind the first step's MFMAs; full key tiles skip the LDS read of the additive mask (it is 0).
template <int DH, int MAXW, int WPE, bool MASKED, bool PIPE = false>
__global__ void __launch_bounds__(64 * MAXW) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
attn_h3_kernel(const _Float16* __restrict__ qkv_h, const _Float16* __restrict__ qkv_l, const int ld, const int D, const int L, const int H,
               const uint8_t* __restrict__ key_mask,   // (B, L) 1 = attend; nullptr = no mask
               _Float16* __restrict__ out, const size_t lo_off, const int ldo,
               const int frag,                         // 1: the context rows leave in the row-panel GEMM's A-fragment order (uu3d_gemm_panel.h, K = ldo); lo_off = 512
               const int qfrag = 0)                    // 1 (round 5, the temporal chain): q | k | v arrive in FRAGMENT order (uu3d_tchain16.h, tchain_qf_index): qkv_h = the buffer, qkv_l / ld unused
{
    static_assert(DH == 48, "operand layouts below are written for a head dim of 48 (3 k-slices, 1.5 output row tiles)");
    constexpr int KS = DH / 16;                        // k-slices of Q K^T
    constexpr float PSHIFT = 14.0f;                    // probabilities carry a factor 2^14 (see top)
    constexpr int VROW = 2 * DH;                       // halfs per V row in LDS: hi | lo
    h3_flush_f16_denormals();                          // output planes: hi = 0 below the smallest normal half (uu3d_gemm_h3.h)
    extern __shared__ __attribute__((aligned(16))) unsigned char asm_[];
    const int Lpad = attn_h3_lpad(L), NT = Lpad >> 5;
    _Float16* Kp = reinterpret_cast<_Float16*>(asm_);                               // [2][KS][Lpad][16]
    _Float16* Vr = Kp + (size_t)2 * KS * Lpad * 16;                                 // [Lpad][hi DH | lo DH]
    _Float16* ones = Vr + (size_t)Lpad * VROW;                                      // 16 x 1.0, then 16 x 0.0
    float* madd = reinterpret_cast<float*>(ones + 32);                              // [Lpad]
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), NW = nthr >> 6;
    // consecutive workgroups (the heads of one sequence) on one XCD: they read neighbouring 96-byte column slices of the same rows
    const int bh = ((int)gridDim.x & 7) == 0 ? ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int b = bh / H, h = bh - b * H;
    const size_t tok0 = (size_t)b * L;
    constexpr float LOG2E = 1.44269504088896341f;
    const int q31 = lane & 31, g = lane >> 5;
    // fragment order: the 16-byte piece of (token, 16-channel group u, half gg, plane) -- [32-token panel][u (72)][plane][lane = token & 31 + 32 gg][8]
    auto qf_piece = [&](size_t token, int u, int gg, int plane) -> const _Float16* {
        return qkv_h + ((((token >> 5) * 72 + u) * 2 + plane) * 64 + (token & 31) + 32 * gg) * 8;
    };
    const int G = D >> 4;                                  // 16-channel groups per q / k / v block

    // ---- K, V of the head: global -> LDS by LDS-DMA, 12 NT instructions shared by the waves; rows past L: a copy of the last
    // row (finite; those keys get -inf / probability 0) ----
    if ((UU3D_ATTN_LOO) & 1) {
    } else if (NW == NT) {
        // one wave per key tile (every launch of up to 12 tiles): piece e = w + NT i, i = 0 .. 11.  K (i < 6): plane i / 3, slice i % 3, key
        // group w -- one key per lane pair for all six pieces.  V (i >= 6): 16-byte piece P = 64 (w + NT (i - 6)) + lane of the [key][12 pieces]
        // image.  The generic loop below divides by 3 NT and NT (run-time values) for every piece: ~60 VALU instructions per piece in front of
        // the last request; here the only division left is the one by 12.
        const int key = min(32 * w + (lane >> 1), L - 1);
        const size_t ko = (tok0 + key) * ld + D + h * DH + 8 * (lane & 1);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const _Float16* src = qfrag ? qf_piece(tok0 + key, G + 3 * h + i % 3, lane & 1, i / 3) : (i >= 3 ? qkv_l : qkv_h) + ko + 16 * (i % 3);
            __builtin_amdgcn_global_load_lds((h3_glb_void*)src, (h3_lds_void*)(Kp + ((size_t)i * Lpad + 32 * w) * 16), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int j = w + NT * i, P = 64 * j + lane, k0 = P / 12, wi = P - 12 * k0, w6 = wi >= 6 ? wi - 6 : wi;
            const _Float16* vp = (wi >= 6 ? qkv_l : qkv_h) + 2 * D + h * DH + 8 * w6;
            const _Float16* src = qfrag ? qf_piece(tok0 + min(k0, L - 1), 2 * G + 3 * h + (w6 >> 1), w6 & 1, wi >= 6) : vp + (tok0 + min(k0, L - 1)) * ld;
            __builtin_amdgcn_global_load_lds((h3_glb_void*)src, (h3_lds_void*)(Vr + (size_t)64 * j * 8), 16, 0, 0);
        }
    } else
    for (int e = w; e < 12 * NT; e += NW) {
        const _Float16* src; _Float16* dst;
        if (e < 6 * NT) {                                          // K: plane p, slice s, 32 keys kg; lane = (key, k-half)
            const int p = e / (3 * NT), r = e - p * 3 * NT, sl = r / NT, kg = r - sl * NT;
            const int key = min(32 * kg + (lane >> 1), L - 1);
            src = qfrag ? qf_piece(tok0 + key, G + 3 * h + sl, lane & 1, p) : (p ? qkv_l : qkv_h) + (tok0 + key) * ld + D + h * DH + 16 * sl + 8 * (lane & 1);
            dst = Kp + ((size_t)(p * KS + sl) * Lpad + 32 * kg) * 16;
        } else {                                                   // V: 64 consecutive 16-byte pieces of the [key][hi | lo] image
            const int i = e - 6 * NT, P = 64 * i + lane, key = min(P / 12, L - 1), wi = P % 12;
            src = qfrag ? qf_piece(tok0 + key, 2 * G + 3 * h + ((wi % 6) >> 1), (wi % 6) & 1, wi >= 6)
                        : (wi >= 6 ? qkv_l : qkv_h) + (tok0 + key) * ld + 2 * D + h * DH + 8 * (wi % 6);
            dst = Vr + (size_t)64 * i * 8;
        }
        __builtin_amdgcn_global_load_lds((h3_glb_void*)src, (h3_lds_void*)dst, 16, 0, 0);
    }
    // the first query tile's fragments: lane = query, 8 consecutive k per slice, as stored
    h16x8 qh[KS], ql[KS];
    auto load_q = [&](int qt) {
        const size_t qtok = tok0 + min(32 * qt + q31, L - 1);
        const size_t o = qtok * ld + h * DH + g * 8;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (qfrag) { qh[s] = *reinterpret_cast<const h16x8*>(qf_piece(qtok, 3 * h + s, g, 0)); ql[s] = *reinterpret_cast<const h16x8*>(qf_piece(qtok, 3 * h + s, g, 1)); }
            else { qh[s] = *reinterpret_cast<const h16x8*>(qkv_h + o + 16 * s); ql[s] = *reinterpret_cast<const h16x8*>(qkv_l + o + 16 * s); }
        }
    };
    if ((UU3D_ATTN_LOO) & 64) {
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) { qh[s][e] = (_Float16)(0.01f * (float)(lane + e)); ql[s][e] = (_Float16)(float)e; }
    } else
    load_q(w);
    for (int k = tid; k < Lpad; k += nthr) {
        const uint8_t mk = (MASKED && key_mask != nullptr) ? key_mask[tok0 + min(k, L - 1)] : (uint8_t)1;
        madd[k] = (k < L) ? (mk ? 0.0f : -1e9f * LOG2E) : -INFINITY;
    }
    if (tid < 32) ones[tid] = tid < 16 ? (_Float16)1.0f : (_Float16)0.0f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's DMAs have landed ...
    __syncthreads();                                               // ... and everybody else's

    // V^T fragment reads (ds_read_b64_tr_b16): the 16 lanes of a group (g = lane >> 5, channel base 16 ((lane >> 4) & 1)) address a
    // block of 4 keys x 16 channels, lane 4 q + p of the group the 8 bytes of key q, channels 4 p .. 4 p + 3, and lane i
    // receives channel i of the 4 keys.  Tile A = channels 0-31; tile B = channels 32-47 for the first group, the ones / zeros
    // block (plane hi / lo) for the second.
    const int grp = (lane >> 4) & 1, vq = (lane >> 2) & 3, vp = lane & 3;
    const unsigned vA = (unsigned)(uintptr_t)(h3_lds_void*)(Vr + (size_t)(4 * g + vq) * VROW + 16 * grp + 4 * vp);
    const unsigned vB_real = (unsigned)(uintptr_t)(h3_lds_void*)(Vr + (size_t)(4 * g + vq) * VROW + 32 + 4 * vp);
    const unsigned vB_ones = (unsigned)(uintptr_t)(h3_lds_void*)(ones + 4 * vp);


    for (int qt = w; qt < NT; qt += NW) {                          // MAXW <= 3: exactly one pass
        // (the next tile's query fragments are requested at the end of this one: MAXW > 3 only)
        f32x16 oA0, oA1, oB0, oB1;                                 // O^T tiles (channels 0-31 / 32-47 + l), hi-hi and cross-term accumulators
#pragma unroll
        for (int r = 0; r < 16; ++r) { oA0[r] = 0.f; oA1[r] = 0.f; oB0[r] = 0.f; oB1[r] = 0.f; }
        float m_off = -PSHIFT;                                     // running maximum - PSHIFT (log2 units); the first tile sets it

        for (int kt = 0; kt < NT; ++kt) {
            // ---- S^T = K Q^T (32 keys x 32 queries); lane holds keys 32 kt + 8 j + 4 g + i of its query.  The hi-hi
            // accumulator starts at the key's mask term (minus the running maximum when nothing is masked) ----
            f32x16 s0, s1;
            typedef _Float16 h16x4v __attribute__((ext_vector_type(4)));
            h16x4v va[8], vb[8];                                   // V^T fragments of the two 16-key steps (PIPE: requested by name)
            if constexpr (PIPE) {
                h16x8 kf[6];                                       // K fragments: hi slices 0..2, lo slices 0..2
                const unsigned kb = (unsigned)(uintptr_t)(h3_lds_void*)(Kp + ((size_t)32 * kt + q31) * 16 + g * 8);
                const unsigned sl = (unsigned)Lpad * 32u;          // bytes between two slices of a plane
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    asm volatile("ds_read_b128 %0, %1" : "=v"(kf[i]) : "v"(kb + (unsigned)i * sl) : "memory");
                if (!MASKED && 32 * kt + 32 <= L) {                // a full tile without a mask: the additive term is 0
#pragma unroll
                    for (int r = 0; r < 16; ++r) { s0[r] = -m_off; s1[r] = 0.f; }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 ma = *reinterpret_cast<const f32x4*>(madd + 32 * kt + 8 * j + 4 * g);
#pragma unroll
                        for (int i = 0; i < 4; ++i) { s0[4 * j + i] = MASKED ? ma[i] : ma[i] - m_off; s1[4 * j + i] = 0.f; }
                    }
                }
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    // LDS returns in order: the hi fragment of slice s has 5 - s younger reads behind it, the lo one 2 - s
                    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(kf[s]) : "i"(5 - s));
                    s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[s], qh[s], s0, 0, 0, 0);
                    s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[s], ql[s], s1, 0, 0, 0);
                    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(kf[3 + s]) : "i"(2 - s));
                    s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[3 + s], qh[s], s1, 0, 0, 0);
                }
            } else {
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
                s0 = UU3D_ATTN_MFMA(2, kh, qh[s], s0);
                s1 = UU3D_ATTN_MFMA(2, kh, ql[s], s1);
                s1 = UU3D_ATTN_MFMA(2, kl, qh[s], s1);
            }
            }
            // ---- online softmax ----
            float t[16];
            float tmax = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) { t[r] = fmaf(s1[r], 1.0f / H3_SCALE, s0[r]); tmax = fmaxf(tmax, t[r]); }
            if constexpr (PIPE) {
                // the first step's V^T fragments: requested here (the cross-term accumulator is dead: 16 registers), they land while the softmax runs
                {
                    const unsigned ko = (unsigned)((32 * kt) * VROW * 2);
                    const unsigned aA = vA + ko, aB = (grp ? vB_ones : vB_real) + (grp ? 0u : ko);
                    const unsigned aB8 = aB + (grp ? 0u : (unsigned)(8 * VROW * 2));
                    const unsigned lo_o = grp ? 32u : (unsigned)(DH * 2);
                    asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:%6\n\t"
                                 "ds_read_b64_tr_b16 %2, %4 offset:%5\n\tds_read_b64_tr_b16 %3, %4 offset:%7"
                                 : "=&v"(va[0]), "=&v"(va[1]), "=&v"(va[2]), "=&v"(va[3]) : "v"(aA), "i"(DH * 2), "i"(8 * VROW * 2), "i"(8 * VROW * 2 + DH * 2) : "memory");
                    asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %5\n\t"
                                 "ds_read_b64_tr_b16 %2, %6\n\tds_read_b64_tr_b16 %3, %7"
                                 : "=&v"(va[4]), "=&v"(va[5]), "=&v"(va[6]), "=&v"(va[7]) : "v"(aB), "v"(aB8), "v"(aB + lo_o), "v"(aB8 + lo_o) : "memory");
                }
            }
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
                if ((UU3D_ATTN_LOO) & 4) {
                    const h16x2 hv = __builtin_bit_cast(h16x2, __builtin_amdgcn_cvt_pkrtz(t[r], t[r + 1]));
                    ph[r >> 3][r & 7] = hv[0]; ph[r >> 3][(r & 7) + 1] = hv[1]; pl[r >> 3][r & 7] = hv[1]; pl[r >> 3][(r & 7) + 1] = hv[0];
                    continue;
                }
                const float p0 = __builtin_amdgcn_exp2f(t[r]), p1 = __builtin_amdgcn_exp2f(t[r + 1]);
                const h16x2 hv = __builtin_bit_cast(h16x2, __builtin_amdgcn_cvt_pkrtz(p0, p1));      // hi: any rounding will do, lo takes the rest
                const h16x2 lv = __builtin_bit_cast(h16x2, __builtin_amdgcn_cvt_pkrtz((p0 - (float)hv[0]) * H3_SCALE, (p1 - (float)hv[1]) * H3_SCALE));
                ph[r >> 3][r & 7] = hv[0]; ph[r >> 3][(r & 7) + 1] = hv[1];
                pl[r >> 3][r & 7] = lv[0]; pl[r >> 3][(r & 7) + 1] = lv[1];
            }
            // ---- O^T += V^T P^T : two 16-key steps, A = V^T fragment (keys 16 s + 8 (j >> 2) + 4 g + (j & 3)) ----
            if constexpr (PIPE) {
                const bool second = 32 * kt + 16 < L;              // (else the second step holds padding keys only: p = 0)
                if (second) {
                    const unsigned ko = (unsigned)((32 * kt + 16) * VROW * 2);
                    const unsigned aA = vA + ko, aB = (grp ? vB_ones : vB_real) + (grp ? 0u : ko);
                    const unsigned aB8 = aB + (grp ? 0u : (unsigned)(8 * VROW * 2));
                    const unsigned lo_o = grp ? 32u : (unsigned)(DH * 2);
                    asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:%6\n\t"
                                 "ds_read_b64_tr_b16 %2, %4 offset:%5\n\tds_read_b64_tr_b16 %3, %4 offset:%7"
                                 : "=&v"(vb[0]), "=&v"(vb[1]), "=&v"(vb[2]), "=&v"(vb[3]) : "v"(aA), "i"(DH * 2), "i"(8 * VROW * 2), "i"(8 * VROW * 2 + DH * 2) : "memory");
                    asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %5\n\t"
                                 "ds_read_b64_tr_b16 %2, %6\n\tds_read_b64_tr_b16 %3, %7"
                                 : "=&v"(vb[4]), "=&v"(vb[5]), "=&v"(vb[6]), "=&v"(vb[7]) : "v"(aB), "v"(aB8), "v"(aB + lo_o), "v"(aB8 + lo_o) : "memory");
                    asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(va[0]), "+v"(va[1]), "+v"(va[2]), "+v"(va[3]), "+v"(va[4]), "+v"(va[5]), "+v"(va[6]), "+v"(va[7]) :: "memory");
                } else
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(va[0]), "+v"(va[1]), "+v"(va[2]), "+v"(va[3]), "+v"(va[4]), "+v"(va[5]), "+v"(va[6]), "+v"(va[7]) :: "memory");
                auto step = [&](const h16x4v (&f)[8], int s) __attribute__((always_inline)) {
                    const h16x8 vAh = (h16x8){f[0][0], f[0][1], f[0][2], f[0][3], f[1][0], f[1][1], f[1][2], f[1][3]};
                    const h16x8 vAl = (h16x8){f[2][0], f[2][1], f[2][2], f[2][3], f[3][0], f[3][1], f[3][2], f[3][3]};
                    const h16x8 vBh = (h16x8){f[4][0], f[4][1], f[4][2], f[4][3], f[5][0], f[5][1], f[5][2], f[5][3]};
                    const h16x8 vBl = (h16x8){f[6][0], f[6][1], f[6][2], f[6][3], f[7][0], f[7][1], f[7][2], f[7][3]};
                    oA0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vAh, ph[s], oA0, 0, 0, 0);
                    oA1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vAh, pl[s], oA1, 0, 0, 0);
                    oA1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vAl, ph[s], oA1, 0, 0, 0);
                    oB0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vBh, ph[s], oB0, 0, 0, 0);
                    oB1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vBh, pl[s], oB1, 0, 0, 0);
                    oB1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vBl, ph[s], oB1, 0, 0, 0);
                };
                step(va, 0);
                if (second) {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vb[0]), "+v"(vb[1]), "+v"(vb[2]), "+v"(vb[3]), "+v"(vb[4]), "+v"(vb[5]), "+v"(vb[6]), "+v"(vb[7]) :: "memory");
                    step(vb, 1);
                }
            } else
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (s == 1 && 32 * kt + 16 >= L) break;           // the step holds padding keys only (p = 0)
                const unsigned ko = (unsigned)((32 * kt + 16 * s) * VROW * 2);     // byte offset of the step's first key row
                const unsigned aA = vA + ko, aB = (grp ? vB_ones : vB_real) + (grp ? 0u : ko);
                typedef _Float16 h16x4v __attribute__((ext_vector_type(4)));
                h16x4v a0, a1, a2, a3, b0, b1, b2, b3;             // tile A: hi keys 0-3 / 8-11, lo likewise; tile B the same
                if ((UU3D_ATTN_LOO) & 32) { a0 = a1 = a2 = a3 = b0 = b1 = b2 = b3 = (h16x4v){ph[s][0], ph[s][1], pl[s][2], pl[s][3]}; } else {
                asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:%6\n\t"
                             "ds_read_b64_tr_b16 %2, %4 offset:%5\n\tds_read_b64_tr_b16 %3, %4 offset:%7"
                             : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(aA), "i"(DH * 2), "i"(8 * VROW * 2), "i"(8 * VROW * 2 + DH * 2) : "memory");
                // tile B: the second group's block does not move with the key (stride 0), so the +8-keys read takes its own address
                const unsigned aB8 = aB + (grp ? 0u : (unsigned)(8 * VROW * 2));
                const unsigned lo_off = grp ? 32u : (unsigned)(DH * 2);             // ones -> zeros block / hi -> lo half of the row
                asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %5\n\t"
                             "ds_read_b64_tr_b16 %2, %6\n\tds_read_b64_tr_b16 %3, %7"
                             : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3) : "v"(aB), "v"(aB8), "v"(aB + lo_off), "v"(aB8 + lo_off) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) :: "memory");
                }
                const h16x8 vAh = (h16x8){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                const h16x8 vAl = (h16x8){a2[0], a2[1], a2[2], a2[3], a3[0], a3[1], a3[2], a3[3]};
                const h16x8 vBh = (h16x8){b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                const h16x8 vBl = (h16x8){b2[0], b2[1], b2[2], b2[3], b3[0], b3[1], b3[2], b3[3]};
                oA0 = UU3D_ATTN_MFMA(8, vAh, ph[s], oA0);
                oA1 = UU3D_ATTN_MFMA(8, vAh, pl[s], oA1);
                oA1 = UU3D_ATTN_MFMA(8, vAl, ph[s], oA1);
                oB0 = UU3D_ATTN_MFMA(8, vBh, ph[s], oB0);
                oB1 = UU3D_ATTN_MFMA(8, vBh, pl[s], oB1);
                oB1 = UU3D_ATTN_MFMA(8, vBl, ph[s], oB1);
            }
        }
        if (MAXW == 8 && qt + NW < NT) load_q(qt + NW);         // (the other instantiations are launched with one wave per tile)
        attn_h3_store_tile<DH>(oA0, oA1, oB0, oB1, qt, q31, g, h, L, tok0, out, lo_off, ldo, frag);
    }
}

static_assert(attn_h3_lds_bytes(ATTN_H3_MAX_L, 48) <= 160 * 1024, "the largest sequence must fit the LDS");

}  // namespace uu3d
