// tools/attn_pp_exp.h -- EXPERIMENT (round 4), not part of the library: temporal self-attention over long sequences (4 .. 13 key tiles of 32;
// SURVEY 8(d)'s "synthetic dense-351") with the softmax of one key tile running UNDER the MFMAs of its neighbours in ONE instruction stream.
// Same operands, same arithmetic in the same order as attn_h3_kernel (csrc/uu3d_attn_h3.h): the outputs are bit-identical
// (tools/attn_loo_exp.hip compares every half), only the schedule differs.
//
// Why it was tried (profiles/r04_attention_knockouts.txt): with one wave per query tile, three waves per SIMD, the launch costs the plain SUM of
// its MFMA time, its exponentials / conversions and its LDS round trips -- leave-one-out builds give each part back in full -- and
// tools/mfma_valu_overlap_exp.hip shows that MFMAs of one wave overlap another wave's v_exp_f32 or LDS round trips poorly (75.7 us against
// 58.2 + 26.7; 78.7 against 58.2 + 32.8), while VALU work issued BEHIND an MFMA in the same wave disappears (64.3 against 58.2 + 44.5).
// So: NW = 4 or 8 waves (one or two streams per SIMD), each walking its query tiles one after the other, and per key tile k the stream
//       phase 1:   S(k+1) = K(k+1) Q^T   [9 MFMAs]   with   B(k): probabilities -> f16 hi / lo pairs     [8 groups of ~8 VALU]
//       phase 2:   O += V(k)^T P(k)^T    [12 MFMAs]  with   A(k+1): combine, maximum, exponentials of tile k+1
// written group by group with scheduling barriers in between.  The LDS operand reads are by name and never cross the loop's back edge
// un-waited: V(k) is requested at the top of phase 1 and waited at its end, K(k+2) at the top of phase 2 and waited at its end.  The
// running-maximum rescale of O (rare) is a wave-uniform branch at the top of the iteration, where nothing is outstanding; the last key tile is
// peeled (a branch on "last" inside the loop cost 255 spilled registers).
//
// What it measured (MI355X, 351 tokens x 32 sequences x 8 heads; attn_h3_kernel 36.6 - 38.2 us): one stream per SIMD 41.6 - 42.8 us -- a single wave
// issues one instruction every ~4.6 cycles, and a key tile is ~175 VALU + 21 MFMA + 22 LDS + ~10 scalar instructions: the stream is ISSUE bound,
// the MFMAs still add 12 us to the 29.5 us the kernel takes without them; two streams per SIMD 37.7 us = a tie.  Three different schedules of
// the same instructions land within 10 % of each other: the launch is bounded by what it executes (staging 14.5 k cycles per workgroup, ~700
// MFMAs and ~5.8 k VALU instructions per SIMD, 17 exponentials per tile), not by how it is ordered.  Not shipped.
#pragma once
#include <type_traits>
#include "../uplift-upsample-3dhpe_amd/csrc/uu3d_attn_h3.h"

namespace uu3d {

#ifdef UU3D_PP_STAMP
// timing builds (tools/attn_loo_exp.hip): shader cycles (s_memtime) per workgroup, wave 0: [0] prologue of a pass, [1] phase 1, [2] phase 2, [3] last tile +
// store, [4] passes, [5] loop iterations, [6] staging (entry -> barrier)
__device__ unsigned long long pp_stamps[8];
#define UU3D_PP_T(var) const long long var = __builtin_amdgcn_s_memtime()
#define UU3D_PP_ACC(i, a, b) pp_acc[i] += (long long)((b) - (a))
#else
#define UU3D_PP_T(var)
#define UU3D_PP_ACC(i, a, b)
#endif

template <int DH, bool MASKED, int NW = 8>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2, 2)))
attn_h3_pp_kernel(const _Float16* __restrict__ qkv_h, const _Float16* __restrict__ qkv_l, const int ld, const int D, const int L, const int H,
                  const uint8_t* __restrict__ key_mask,   // (B, L) 1 = attend; nullptr = no mask
                  _Float16* __restrict__ out, const size_t lo_off, const int ldo,
                  const int frag)                         // as attn_h3_kernel
{
    static_assert(DH == 48, "operand layouts are those of attn_h3_kernel: head dim 48");
    constexpr int KS = DH / 16;
    constexpr float PSHIFT = 14.0f;
    constexpr int VROW = 2 * DH;
    static_assert(NW == 4 || NW == 8, "one or two instruction streams per SIMD");
    typedef _Float16 h16x4v __attribute__((ext_vector_type(4)));
    h3_flush_f16_denormals();
    extern __shared__ __attribute__((aligned(16))) unsigned char asm_[];
    const int Lpad = attn_h3_lpad(L), NT = Lpad >> 5;
    _Float16* Kp = reinterpret_cast<_Float16*>(asm_);                               // [2][KS][Lpad][16]
    _Float16* Vr = Kp + (size_t)2 * KS * Lpad * 16;                                 // [Lpad][hi DH | lo DH]
    _Float16* ones = Vr + (size_t)Lpad * VROW;                                      // 16 x 1.0, then 16 x 0.0
    float* madd = reinterpret_cast<float*>(ones + 32);                              // [Lpad]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bh = ((int)gridDim.x & 7) == 0 ? ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int b = bh / H, h = bh - b * H;
    const size_t tok0 = (size_t)b * L;
    constexpr float LOG2E = 1.44269504088896341f;
    const int q31 = lane & 31, g = lane >> 5;
#ifdef UU3D_PP_STAMP
    long long pp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};            // summed in registers, written once at the end
#endif
    UU3D_PP_T(t_entry);

    // ---- K, V of the head: global -> LDS by LDS-DMA, tile by tile.  Piece i < 6 of a tile's K = (plane i / 3, k-slice i % 3), one key per lane
    // pair; piece j < 6 of its 6 KB V image = 64 consecutive 16-byte pieces of [key][hi | lo].  Wave w < 6 requests piece w of both (four waves: w and w + 4). ----
    if (NW == 4 || w < 6) {
        // piece w; with four waves also piece w + 4 (waves 0, 1)
        const int vc0 = (64 * w + lane) / 12, vw0 = (64 * w + lane) - 12 * vc0;                    // piece w of a V tile: key offset, 16-byte column
        const int vc1 = (64 * (w + 4) + lane) / 12, vw1 = (64 * (w + 4) + lane) - 12 * vc1;
        const _Float16* kp0 = (w >= 3 ? qkv_l : qkv_h) + D + h * DH + 16 * (w >= 3 ? w - 3 : w) + 8 * (lane & 1);
        const _Float16* kp1 = qkv_l + D + h * DH + 16 * (w + 1) + 8 * (lane & 1);                  // piece w + 4 = lo plane, slice w + 1
        const _Float16* vp0 = (vw0 >= 6 ? qkv_l : qkv_h) + 2 * D + h * DH + 8 * (vw0 >= 6 ? vw0 - 6 : vw0);
        const _Float16* vp1 = (vw1 >= 6 ? qkv_l : qkv_h) + 2 * D + h * DH + 8 * (vw1 >= 6 ? vw1 - 6 : vw1);
        for (int t = 0; t < NT; ++t) {
            const size_t krow = (tok0 + min(32 * t + (lane >> 1), L - 1)) * ld;
            __builtin_amdgcn_global_load_lds((h3_glb_void*)(kp0 + krow), (h3_lds_void*)(Kp + ((size_t)w * Lpad + 32 * t) * 16), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((h3_glb_void*)(vp0 + (tok0 + min(32 * t + vc0, L - 1)) * ld), (h3_lds_void*)(Vr + (size_t)64 * (6 * t + w) * 8), 16, 0, 0);
            if (NW == 4 && w < 2) {
                __builtin_amdgcn_global_load_lds((h3_glb_void*)(kp1 + krow), (h3_lds_void*)(Kp + ((size_t)(w + 4) * Lpad + 32 * t) * 16), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((h3_glb_void*)(vp1 + (tok0 + min(32 * t + vc1, L - 1)) * ld), (h3_lds_void*)(Vr + (size_t)64 * (6 * t + w + 4) * 8), 16, 0, 0);
            }
        }
    }
    h16x8 qh[KS], ql[KS];
    auto load_q = [&](int qt, h16x8 (&dh)[KS], h16x8 (&dl)[KS]) {
        const size_t o = (tok0 + min(32 * qt + q31, L - 1)) * ld + h * DH + g * 8;
#pragma unroll
        for (int s = 0; s < KS; ++s) { dh[s] = *reinterpret_cast<const h16x8*>(qkv_h + o + 16 * s); dl[s] = *reinterpret_cast<const h16x8*>(qkv_l + o + 16 * s); }
    };
    load_q(w, qh, ql);
    for (int k = tid; k < Lpad; k += 64 * NW) {
        const uint8_t mk = (MASKED && key_mask != nullptr) ? key_mask[tok0 + min(k, L - 1)] : (uint8_t)1;
        madd[k] = (k < L) ? (mk ? 0.0f : -1e9f * LOG2E) : -INFINITY;
    }
    if (tid < 32) ones[tid] = tid < 16 ? (_Float16)1.0f : (_Float16)0.0f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    UU3D_PP_T(t_staged); UU3D_PP_ACC(6, t_entry, t_staged);

    // LDS addresses (bytes).  K fragment of tile kt, plane / slice i: kb + 1024 kt + i sl.  V^T blocks as in attn_h3_kernel.
    const unsigned kb = (unsigned)(uintptr_t)(h3_lds_void*)(Kp + (size_t)q31 * 16 + g * 8);
    const unsigned sl = (unsigned)Lpad * 32u;
    const unsigned mb = (unsigned)(uintptr_t)(h3_lds_void*)(madd + 4 * g);
    const int grp = (lane >> 4) & 1, vq = (lane >> 2) & 3, vp = lane & 3;
    const unsigned vA = (unsigned)(uintptr_t)(h3_lds_void*)(Vr + (size_t)(4 * g + vq) * VROW + 16 * grp + 4 * vp);
    const unsigned vBb = grp ? (unsigned)(uintptr_t)(h3_lds_void*)(ones + 4 * vp) : (unsigned)(uintptr_t)(h3_lds_void*)(Vr + (size_t)(4 * g + vq) * VROW + 32 + 4 * vp);
    const unsigned vBs = grp ? 0u : 1u;                            // the ones / zeros block does not move with the key
    const unsigned lo_o = grp ? 32u : (unsigned)(DH * 2);

    for (int qt = w; qt < NT; qt += NW) {
        const bool more = qt + NW < NT;
        f32x16 oA0, oA1, oB0, oB1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { oA0[r] = 0.f; oA1[r] = 0.f; oB0[r] = 0.f; oB1[r] = 0.f; }
        float m_off = -PSHIFT, alpha = 1.0f;
        f32x16 s0, s1;                                             // S^T accumulators of the tile AHEAD
        float p[16];                                               // probabilities (x 2^14) of the current tile, f32
        h16x8 ph[2], pl[2];
        h16x8 kf[6]; h16x4v vf[16];

        auto request_k = [&](int kt) __attribute__((always_inline)) {
            const unsigned a = kb + 1024u * (unsigned)kt;
#pragma unroll
            for (int i = 0; i < 6; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(kf[i]) : "v"(a + (unsigned)i * sl) : "memory");
        };
        auto wait_k = [&]() __attribute__((always_inline)) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]), "+v"(kf[4]), "+v"(kf[5]) :: "memory");
        };
        auto request_v = [&](int kt) __attribute__((always_inline)) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const unsigned ko = (unsigned)((32 * kt + 16 * s) * VROW * 2);
                const unsigned aA = vA + ko, aB = vBb + vBs * ko, aB8 = aB + vBs * (unsigned)(8 * VROW * 2);
                asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:%6\n\t"
                             "ds_read_b64_tr_b16 %2, %4 offset:%5\n\tds_read_b64_tr_b16 %3, %4 offset:%7"
                             : "=&v"(vf[8 * s + 0]), "=&v"(vf[8 * s + 1]), "=&v"(vf[8 * s + 2]), "=&v"(vf[8 * s + 3]) : "v"(aA), "i"(DH * 2), "i"(8 * VROW * 2), "i"(8 * VROW * 2 + DH * 2) : "memory");
                asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %5\n\t"
                             "ds_read_b64_tr_b16 %2, %6\n\tds_read_b64_tr_b16 %3, %7"
                             : "=&v"(vf[8 * s + 4]), "=&v"(vf[8 * s + 5]), "=&v"(vf[8 * s + 6]), "=&v"(vf[8 * s + 7]) : "v"(aB), "v"(aB8), "v"(aB + lo_o), "v"(aB8 + lo_o) : "memory");
            }
        };
        auto wait_v = [&]() __attribute__((always_inline)) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vf[0]), "+v"(vf[1]), "+v"(vf[2]), "+v"(vf[3]), "+v"(vf[4]), "+v"(vf[5]), "+v"(vf[6]), "+v"(vf[7]),
                         "+v"(vf[8]), "+v"(vf[9]), "+v"(vf[10]), "+v"(vf[11]), "+v"(vf[12]), "+v"(vf[13]), "+v"(vf[14]), "+v"(vf[15]) :: "memory");
        };
        // S^T of key tile kt: accumulator start (mask term, minus the running maximum when nothing is masked; a full tile of an unmasked launch
        // has mask term 0 and reads nothing -- called with no LDS request outstanding), then MFMA number i of 9 in the order of attn_h3_kernel
        // (per k-slice: s0 += kh qh, s1 += kh ql, s1 += kl qh)
        auto s_init = [&](int kt) __attribute__((always_inline)) {
            if (!MASKED && 32 * kt + 32 <= L) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { s0[r] = 0.0f - m_off; s1[r] = 0.f; }
            } else {
                f32x4 mk[4];
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:32\n\tds_read_b128 %2, %4 offset:64\n\tds_read_b128 %3, %4 offset:96\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(mk[0]), "=&v"(mk[1]), "=&v"(mk[2]), "=&v"(mk[3]) : "v"(mb + 128u * (unsigned)kt) : "memory");
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) { s0[4 * j + i] = MASKED ? mk[j][i] : mk[j][i] - m_off; s1[4 * j + i] = 0.f; }
            }
        };
        auto s_mfma = [&](auto itag) __attribute__((always_inline)) {
            constexpr int i = decltype(itag)::value, s = i / 3;
            if constexpr (i % 3 == 0) s0 = UU3D_ATTN_MFMA(2, kf[s], qh[s], s0);
            else if constexpr (i % 3 == 1) s1 = UU3D_ATTN_MFMA(2, kf[s], ql[s], s1);
            else s1 = UU3D_ATTN_MFMA(2, kf[3 + s], qh[s], s1);
        };
        // O^T MFMA number i of 12: step i / 6 (16 keys), within a step the order of attn_h3_kernel
        auto o_mfma = [&](auto itag) __attribute__((always_inline)) {
            constexpr int i = decltype(itag)::value, s = i / 6, e = i % 6;
            constexpr int o = 8 * s;
            const h16x8 a = e < 2  ? (h16x8){vf[o][0], vf[o][1], vf[o][2], vf[o][3], vf[o + 1][0], vf[o + 1][1], vf[o + 1][2], vf[o + 1][3]}
                          : e == 2 ? (h16x8){vf[o + 2][0], vf[o + 2][1], vf[o + 2][2], vf[o + 2][3], vf[o + 3][0], vf[o + 3][1], vf[o + 3][2], vf[o + 3][3]}
                          : e < 5  ? (h16x8){vf[o + 4][0], vf[o + 4][1], vf[o + 4][2], vf[o + 4][3], vf[o + 5][0], vf[o + 5][1], vf[o + 5][2], vf[o + 5][3]}
                                   : (h16x8){vf[o + 6][0], vf[o + 6][1], vf[o + 6][2], vf[o + 6][3], vf[o + 7][0], vf[o + 7][1], vf[o + 7][2], vf[o + 7][3]};
            if constexpr (e == 0) oA0 = UU3D_ATTN_MFMA(8, a, ph[s], oA0);
            else if constexpr (e == 1) oA1 = UU3D_ATTN_MFMA(8, a, pl[s], oA1);
            else if constexpr (e == 2) oA1 = UU3D_ATTN_MFMA(8, a, ph[s], oA1);
            else if constexpr (e == 3) oB0 = UU3D_ATTN_MFMA(8, a, ph[s], oB0);
            else if constexpr (e == 4) oB1 = UU3D_ATTN_MFMA(8, a, pl[s], oB1);
            else oB1 = UU3D_ATTN_MFMA(8, a, ph[s], oB1);
        };
        // B: probabilities 2 r, 2 r + 1 -> f16 hi / lo (hi: any rounding will do, lo takes the rest)
        auto b_pair = [&](auto rtag) __attribute__((always_inline)) {
            constexpr int r = 2 * decltype(rtag)::value;
            if ((UU3D_ATTN_LOO) & 4) { const h16x2 hv = __builtin_bit_cast(h16x2, __builtin_amdgcn_cvt_pkrtz(p[r], p[r + 1]));
                ph[r >> 3][r & 7] = hv[0]; ph[r >> 3][(r & 7) + 1] = hv[1]; pl[r >> 3][r & 7] = hv[1]; pl[r >> 3][(r & 7) + 1] = hv[0]; return; }
            const h16x2 hv = __builtin_bit_cast(h16x2, __builtin_amdgcn_cvt_pkrtz(p[r], p[r + 1]));
            const h16x2 lv = __builtin_bit_cast(h16x2, __builtin_amdgcn_cvt_pkrtz((p[r] - (float)hv[0]) * H3_SCALE, (p[r + 1] - (float)hv[1]) * H3_SCALE));
            ph[r >> 3][r & 7] = hv[0]; ph[r >> 3][(r & 7) + 1] = hv[1];
            pl[r >> 3][r & 7] = lv[0]; pl[r >> 3][(r & 7) + 1] = lv[1];
        };
        // A, in four parts: combine (two halves), maximum + running-maximum update, exponentials (four quarters)
        float tmax;
        auto a_combine = [&](auto htag) __attribute__((always_inline)) {
            constexpr int h8 = 8 * decltype(htag)::value;
#pragma unroll
            for (int r = h8; r < h8 + 8; ++r) p[r] = fmaf(s1[r], 1.0f / H3_SCALE, s0[r]);
        };
        auto a_max = [&](bool first) __attribute__((always_inline)) {
            tmax = fmaxf(fmaxf(fmaxf(p[0], p[1]), fmaxf(p[2], p[3])), fmaxf(fmaxf(p[4], p[5]), fmaxf(p[6], p[7])));
            tmax = fmaxf(tmax, fmaxf(fmaxf(fmaxf(p[8], p[9]), fmaxf(p[10], p[11])), fmaxf(fmaxf(p[12], p[13]), fmaxf(p[14], p[15]))));
            // the other half's maximum without an LDS round trip (attn_h3_kernel's __shfl_xor is a ds_bpermute).  The swap is written in asm:
            // through __builtin_amdgcn_permlane32_swap hipcc (ROCm 7.2) drops the maximum of the two results and keeps the first one
            // (tools/attn_loo_exp.hip caught it: rows whose maximum sits in the other half came out wrong).  s_nop 1: VALU write -> permlane read.
            {
                float ta = tmax, tb = tmax;
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ta), "+v"(tb));
                tmax = fmaxf(ta, tb);
            }
            float sub;                                             // what every logit of the tile loses before the exponential
            if (MASKED) {
                const float m_old = m_off + PSHIFT;
                const float m_new = first ? tmax : fmaxf(m_old, tmax);
                alpha = first ? 1.0f : __builtin_amdgcn_exp2f(m_old - m_new);
                m_off = m_new - PSHIFT;
                sub = m_off;
            } else {
                float delta = tmax - PSHIFT;
                if (!first) delta = fmaxf(delta, 0.f);
                alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);
                m_off += delta;
                sub = delta;
            }
            tmax = sub;
        };
        auto a_exp = [&](auto qtag) __attribute__((always_inline)) {
            constexpr int q4 = 4 * decltype(qtag)::value;
#pragma unroll
            for (int r = q4; r < q4 + 4; ++r) p[r] = ((UU3D_ATTN_LOO) & 4) ? p[r] - tmax : __builtin_amdgcn_exp2f(p[r] - tmax);
        };
#define UU3D_PP_I(n) std::integral_constant<int, n>{}
#ifdef UU3D_PP_NOSB
#define UU3D_PP_SB()
#else
#define UU3D_PP_SB() __builtin_amdgcn_sched_barrier(0)
#endif

        // ---- prologue: S(0), A(0); K(1) requested and landed ----
        UU3D_PP_T(t_p0);
        s_init(0);
        request_k(0); wait_k();
        s_mfma(UU3D_PP_I(0)); s_mfma(UU3D_PP_I(1)); s_mfma(UU3D_PP_I(2)); s_mfma(UU3D_PP_I(3)); s_mfma(UU3D_PP_I(4));
        s_mfma(UU3D_PP_I(5)); s_mfma(UU3D_PP_I(6)); s_mfma(UU3D_PP_I(7)); s_mfma(UU3D_PP_I(8));
        UU3D_PP_SB();
        request_k(min(1, NT - 1));
        a_combine(UU3D_PP_I(0)); a_combine(UU3D_PP_I(1)); a_max(true);
        a_exp(UU3D_PP_I(0)); a_exp(UU3D_PP_I(1)); a_exp(UU3D_PP_I(2)); a_exp(UU3D_PP_I(3));
        wait_k();

        // (rare) the maximum moved in the A that ran last: O^T of the tiles before it shrinks.  alpha == 1 exactly where it did not.
        auto rescale = [&]() __attribute__((always_inline)) {
            if (__any(alpha != 1.0f)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { oA0[r] *= alpha; oA1[r] *= alpha; oB0[r] *= alpha; oB1[r] *= alpha; }
            }
        };
        UU3D_PP_T(t_p1); UU3D_PP_ACC(0, t_p0, t_p1); UU3D_PP_ACC(4, 0, 1);
        for (int kt = 0; kt + 1 < NT; ++kt) {
            UU3D_PP_T(t_a);
            rescale();
            // ---- phase 1: S(kt + 1) under B(kt); V(kt) travels ----
            s_init(kt + 1);
            request_v(kt);
            UU3D_PP_SB();
            s_mfma(UU3D_PP_I(0)); UU3D_PP_SB(); b_pair(UU3D_PP_I(0)); UU3D_PP_SB();
            s_mfma(UU3D_PP_I(1)); UU3D_PP_SB(); b_pair(UU3D_PP_I(1)); UU3D_PP_SB();
            s_mfma(UU3D_PP_I(2)); UU3D_PP_SB(); b_pair(UU3D_PP_I(2)); UU3D_PP_SB();
            s_mfma(UU3D_PP_I(3)); UU3D_PP_SB(); b_pair(UU3D_PP_I(3)); UU3D_PP_SB();
            s_mfma(UU3D_PP_I(4)); UU3D_PP_SB(); b_pair(UU3D_PP_I(4)); UU3D_PP_SB();
            s_mfma(UU3D_PP_I(5)); UU3D_PP_SB(); b_pair(UU3D_PP_I(5)); UU3D_PP_SB();
            s_mfma(UU3D_PP_I(6)); UU3D_PP_SB(); b_pair(UU3D_PP_I(6)); UU3D_PP_SB();
            s_mfma(UU3D_PP_I(7)); UU3D_PP_SB(); b_pair(UU3D_PP_I(7)); UU3D_PP_SB();
            s_mfma(UU3D_PP_I(8)); UU3D_PP_SB();
            wait_v();
            UU3D_PP_T(t_b); UU3D_PP_ACC(1, t_a, t_b);
            // ---- phase 2: O += V(kt)^T P(kt)^T under A(kt + 1); K(kt + 2) travels ----
            request_k(min(kt + 2, NT - 1));
            o_mfma(UU3D_PP_I(0)); UU3D_PP_SB(); a_combine(UU3D_PP_I(0)); UU3D_PP_SB();
            o_mfma(UU3D_PP_I(1)); UU3D_PP_SB(); a_combine(UU3D_PP_I(1)); UU3D_PP_SB();
            o_mfma(UU3D_PP_I(2)); UU3D_PP_SB(); a_max(false); UU3D_PP_SB();
            o_mfma(UU3D_PP_I(3)); UU3D_PP_SB(); a_exp(UU3D_PP_I(0)); UU3D_PP_SB();
            o_mfma(UU3D_PP_I(4)); UU3D_PP_SB(); a_exp(UU3D_PP_I(1)); UU3D_PP_SB();
            o_mfma(UU3D_PP_I(5)); UU3D_PP_SB(); a_exp(UU3D_PP_I(2)); UU3D_PP_SB();
            o_mfma(UU3D_PP_I(6)); UU3D_PP_SB(); a_exp(UU3D_PP_I(3)); UU3D_PP_SB();
            o_mfma(UU3D_PP_I(7)); o_mfma(UU3D_PP_I(8)); o_mfma(UU3D_PP_I(9)); o_mfma(UU3D_PP_I(10)); o_mfma(UU3D_PP_I(11));
            UU3D_PP_SB();
            wait_k();
            UU3D_PP_T(t_c); UU3D_PP_ACC(2, t_b, t_c); UU3D_PP_ACC(5, 0, 1);
        }
        UU3D_PP_T(t_l0);
        // ---- the last key tile: B, O (a second 16-key step of padding keys only carries p = 0: the products add nothing) ----
        rescale();
        if (more) load_q(qt + NW, qh, ql);                         // the pass's last S^T is done: the next pass's query fragments travel under the rest
        request_v(NT - 1);
        b_pair(UU3D_PP_I(0)); b_pair(UU3D_PP_I(1)); b_pair(UU3D_PP_I(2)); b_pair(UU3D_PP_I(3));
        b_pair(UU3D_PP_I(4)); b_pair(UU3D_PP_I(5)); b_pair(UU3D_PP_I(6)); b_pair(UU3D_PP_I(7));
        wait_v();
        o_mfma(UU3D_PP_I(0)); o_mfma(UU3D_PP_I(1)); o_mfma(UU3D_PP_I(2)); o_mfma(UU3D_PP_I(3)); o_mfma(UU3D_PP_I(4)); o_mfma(UU3D_PP_I(5));
        o_mfma(UU3D_PP_I(6)); o_mfma(UU3D_PP_I(7)); o_mfma(UU3D_PP_I(8)); o_mfma(UU3D_PP_I(9)); o_mfma(UU3D_PP_I(10)); o_mfma(UU3D_PP_I(11));
#undef UU3D_PP_I
#undef UU3D_PP_SB
        attn_h3_store_tile<DH>(oA0, oA1, oB0, oB1, qt, q31, g, h, L, tok0, out, lo_off, ldo, frag);
        UU3D_PP_T(t_l1); UU3D_PP_ACC(3, t_l0, t_l1);
    }
#ifdef UU3D_PP_STAMP
    if (w == 0 && lane == 0) for (int i = 0; i < 7; ++i) atomicAdd(&pp_stamps[i], (unsigned long long)pp_acc[i]);
#endif
}

}  // namespace uu3d
