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
// see uu3d_gemm_h3.h) that the f16x3 projection GEMM reads, instead of f32.
typedef _Float16 h16x8v __attribute__((ext_vector_type(8)));
#ifdef UU3D_ATTN_STAMP
__device__ unsigned long long attn_clk[8];     // tools/attn_stamp_exp: s_memtime ticks per phase, summed over workgroups (wave 0)
#define ATTN_STAMP(...) __VA_ARGS__
#else
#define ATTN_STAMP(...)
#endif
template <int NT, int DH, bool SPLIT = false>
__global__ void __launch_bounds__(64 * NT)
attn_f32_kernel(const float* __restrict__ qkv, const int ld, const int D, const int L, const int H,
                const uint8_t* __restrict__ key_mask,   // (B, L) 1 = attend; nullptr = no mask
                float* __restrict__ out, const int ldo, const size_t lo_off = 0)
{
    static_assert(DH % 16 == 0, "head dim must be a multiple of 16");
    constexpr int LD = DH + 4;
    constexpr int F4 = DH / 4;
    constexpr int KT = DH / 16;             // float4 k-groups per lane
    constexpr int NS = (16 * F4 + 63) / 64; // staging float4 per thread and matrix
    __shared__ __attribute__((aligned(16))) float Ks[NT * 16 * LD];
    __shared__ __attribute__((aligned(16))) float Vs[NT * 16 * LD];
    __shared__ __attribute__((aligned(16))) _Float16 Os[SPLIT ? NT : 1][SPLIT ? 2 * 16 * DH : 8];     // per-wave output tile, f16 planes

    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int qi = lane & 15, g = lane >> 4;
    const int qrow = 16 * w + qi;
    // Workgroups go round-robin over the 8 XCDs; remapped so that CONSECUTIVE items (the H heads of one sequence) run on the
    // SAME XCD: a head's 192-byte q / k / v slices straddle 128-byte lines that the neighbouring head also needs, and the heads'
    // partial-line output writes meet in one L2 instead of eight.
    const int bh = ((int)gridDim.x & 7) == 0 ? ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;

    const int b = bh / H, h = bh - b * H;
    float4 kreg[NS], vreg[NS];
    f32x4 qf[KT];
    {
        const float* base = qkv + (size_t)b * L * ld + h * DH;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int idx = tid + s * 64 * NT;
            const int row = idx / F4, c4 = (idx - row * F4) * 4;
            // branch-free: clamped row, zeroed afterwards (a test around a load = a divergent branch with the wait inside it)
            const float* p = base + (size_t)min(row, L - 1) * ld + c4;
            const float keep = (row < L) ? 1.0f : 0.0f;
            const float4 kv = *reinterpret_cast<const float4*>(p + D), vv = *reinterpret_cast<const float4*>(p + 2 * D);
            kreg[s] = make_float4(kv.x * keep, kv.y * keep, kv.z * keep, kv.w * keep);
            vreg[s] = make_float4(vv.x * keep, vv.y * keep, vv.z * keep, vv.w * keep);
        }
#pragma unroll
        for (int t = 0; t < KT; ++t)      // query rows above L: a copy of the last row, never stored
            qf[t] = *reinterpret_cast<const f32x4*>(base + (size_t)min(qrow, L - 1) * ld + 16 * t + 4 * g);
    }
    ATTN_STAMP(const long long c0 = clock64();)
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int idx = tid + s * 64 * NT;
        const int row = idx / F4, c4 = (idx - row * F4) * 4;
        if ((16 * F4) % 64 == 0 || idx < NT * 16 * F4) {          // whole staging passes (d_h = 48): no guard for hipcc to sink the loads into
            *reinterpret_cast<float4*>(&Ks[row * LD + c4]) = kreg[s];
            *reinterpret_cast<float4*>(&Vs[row * LD + c4]) = vreg[s];
        }
    }
    __syncthreads();
    ATTN_STAMP(const long long c1 = clock64();)

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

    ATTN_STAMP(asm volatile("s_nop 0" :: "v"(st[NT - 1][3])); const long long c2 = clock64();)
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
            // clamped, branch-free byte load (a test around it costs one branch + wait, i.e. a memory round trip, per key)
            const uint8_t mk = (key_mask != nullptr) ? key_mask[(size_t)b * L + min(key, L - 1)] : (uint8_t)1;
            const float v = (key < L) ? st[j][r] * scale_mul + (mk ? 0.0f : 1.0f) * -1e9f : -INFINITY;
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

    ATTN_STAMP(asm volatile("s_nop 0" :: "v"(st[NT - 1][3])); const long long c3 = clock64();)
    // O = P V : tile t covers head channels 16t .. 16t+15
    f32x4 ot[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) ot[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float vv = Vs[(16 * j + 4 * g + s) * LD + 16 * t + qi];
                    ot[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(st[j][s], vv, ot[t], 0, 0, 0);
                }
    // C/D map: col = lane & 15 -> channel, row = 4g + r -> query
    if constexpr (SPLIT) {
        // the wave's 16 x DH tile as two f16 planes through LDS, out as 16-byte pieces (see attn_head_wave_kernel)
        _Float16* Ow = Os[w];
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const _Float16 hv = (fabsf(ot[t][r]) < 6.103515625e-05f) ? (_Float16)0.f : (_Float16)ot[t][r];   // = h3_hi (uu3d_gemm_h3.h), explicit: this kernel keeps f16 denormals on
                const _Float16 lv = (_Float16)((ot[t][r] - (float)hv) * 2048.0f);
                Ow[(4 * g + r) * DH + 16 * t + qi] = hv;
                Ow[(16 + 4 * g + r) * DH + 16 * t + qi] = lv;
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        constexpr int PPR = DH / 8, PIECES = 2 * 16 * PPR;
        _Float16* oh = reinterpret_cast<_Float16*>(out);
#pragma unroll
        for (int i = 0; i < (PIECES + 63) / 64; ++i) {
            const int pc = lane + 64 * i;
            const int plane = pc / (16 * PPR), rem = pc - plane * 16 * PPR, row = rem / PPR, c8 = rem - row * PPR;
            const int q = 16 * w + row;
            if (pc < PIECES && q < L) {
                const h16x8v piece = *reinterpret_cast<const h16x8v*>(&Ow[(plane * 16 + row) * DH + 8 * c8]);
                const int grow = b * L + q, k = h * DH + 8 * c8;
                // lo_off == 512 (no row-major plane pair is that small): the row-panel GEMM's A-fragment order (uu3d_gemm_panel.h, K = ldo) --
                // the same 16-byte pieces at [32-row panel][16-channel slice][plane][channel half][row & 31][8]
                _Float16* dst = lo_off == 512 ? oh + ((size_t)(grow >> 5) * (size_t)(ldo >> 4) + (size_t)(k >> 4)) * 1024 + (size_t)plane * 512 + ((k >> 3) & 1) * 256 + (grow & 31) * 8
                                              : oh + (size_t)plane * lo_off + (size_t)grow * ldo + k;
                *reinterpret_cast<h16x8v*>(dst) = piece;
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = 16 * w + 4 * g + r;
                if (q < L) out[((size_t)b * L + q) * ldo + h * DH + 16 * t + qi] = ot[t][r];
            }
    }
    ATTN_STAMP(asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (tid == 0) { const long long c4 = clock64();
        atomicAdd(&attn_clk[0], (unsigned long long)(c1 - c0)); atomicAdd(&attn_clk[1], (unsigned long long)(c2 - c1));
        atomicAdd(&attn_clk[2], (unsigned long long)(c3 - c2)); atomicAdd(&attn_clk[3], (unsigned long long)(c4 - c3)); atomicAdd(&attn_clk[4], 1ull); })
}

// ---------------------------------------------------------------------------------------------------------------------
// The same attention with ONE WAVE PER (sequence, head): a workgroup is 4 waves = 4 consecutive heads of one sequence, every
// wave stages the K and V of its own head in its own LDS region and walks over the NT query tiles itself.
// Why: with one workgroup of NT = 5 waves per item, a CU holds 4 workgroups = 20 waves, the kernel's phases (load, Q K^T,
// softmax, P V, store) run in lockstep on all of them, and 5 waves do not divide over 4 SIMDs (s_memtime stamps,
// tools/attn_stamp_exp: 20 k cycles per workgroup of which the MFMAs are 3.8 k).  Here a SIMD runs exactly one wave = one
// item from start to end: no barrier (LDS traffic is ordered by the wave itself), 1024 items = 1024 SIMDs in one round.
// LDS: 4 x 2 x NT*16 x 52 floats (NT = 5: 130 KiB) -> one workgroup per CU.
template <int NT, int DH, bool SPLIT = false>
__global__ void __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1)))
attn_head_wave_kernel(const float* __restrict__ qkv, const int ld, const int D, const int L, const int H,
                      const uint8_t* __restrict__ key_mask, float* __restrict__ out, const int ldo, const size_t lo_off, const int items)
{
    static_assert(DH % 16 == 0, "head dim must be a multiple of 16");
    constexpr int LD = DH + 4;
    constexpr int F4 = DH / 4;
    constexpr int KT = DH / 16;
    constexpr int NS = (NT * 16 * F4 + 63) / 64;          // staging float4 per lane and matrix
    constexpr int OS_HALFS = 2 * 16 * DH;                  // per-wave output tile [plane][16 rows][DH] (SPLIT)
    extern __shared__ __attribute__((aligned(16))) float attn_lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int qi = lane & 15, g = lane >> 4;
    float* Ks = attn_lds + (size_t)w * 2 * NT * 16 * LD;
    float* Vs = Ks + NT * 16 * LD;
    _Float16* Os = reinterpret_cast<_Float16*>(attn_lds + (size_t)4 * 2 * NT * 16 * LD) + w * OS_HALFS;
    // the two workgroups of a sequence on one XCD (see attn_f32_kernel)
    const int wg = ((int)gridDim.x & 7) == 0 ? ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int bh = wg * 4 + w;
    if (bh >= items) return;                               // whole waves only: no barrier anywhere below
    // kernel arguments of the store path, loaded NOW: a lazily placed s_load between a tile's by-name V reads and their counted
    // lgkmcnt waits would break the count (scalar loads return out of order; tests/test_isa_cpu.py checks the placement)
    size_t lo_off_s = lo_off; int ldo_s = ldo;
    asm volatile("" : "+s"(lo_off_s), "+s"(ldo_s));
    const int b = bh / H, h = bh - b * H;
    const float* base = qkv + (size_t)b * L * ld + h * DH;
    ATTN_STAMP(const long long c0 = clock64(); long long tqk = 0, tsm = 0, tpv = 0;)

    // ---- K, V of this head -> LDS (zero rows above L) ----
    static_assert((NT * 16 * F4) % 64 == 0, "whole staging passes");
    {
        // branch-free (row clamped, zeroed afterwards): a test around a load costs a divergent branch with the wait for the data
        // inside it, i.e. one memory round trip per load instead of one for all of them
        // All 2 NS loads in flight at once, by name: K rows then V rows (scalar base of this wave's head + one 32-bit lane
        // offset per pass, shared by K and V); hipcc kept ~16 in flight and paid the memory round trip twice.
        static_assert(NS <= 15, "the counted waits list at most 15 registers");
        f32x4 kx[15], vx[15];
        auto scalar_ptr = [](const float* q) {                   // wave-uniform by construction (one head per wave): into SGPRs
            const unsigned long long a = (unsigned long long)q;
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
            return (const float*)(((unsigned long long)hi << 32) | lo);
        };
        const float* kbase = scalar_ptr(base + D);
        const float* vbase = scalar_ptr(base + 2 * D);
        // gfx9 hazard: a VMEM instruction must not read an SGPR within 5 wait states of the VALU (v_readfirstlane) that wrote
        // it.  The compiler pads this for its own instructions but cannot see into inline asm: without the nops the 4-tile
        // instantiation issued the first load directly behind the readfirstlane pair, used a stale base and faulted (the 5-tile
        // one happened to have other instructions in between).  The "+s" operands order the nops behind the readfirstlanes.
        asm volatile("s_nop 4" : "+s"(kbase), "+s"(vbase));
        unsigned off[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) { const int idx = lane + s * 64; const int row = idx / F4; off[s] = (unsigned)((min(row, L - 1) * ld + (idx - row * F4) * 4) * 4); }
#pragma unroll
        for (int s = 0; s < NS; ++s) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(kx[s]) : "v"(off[s]), "s"(kbase) : "memory");
#pragma unroll
        for (int s = 0; s < NS; ++s) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(vx[s]) : "v"(off[s]), "s"(vbase) : "memory");
#pragma unroll
        for (int s = NS; s < 15; ++s) { kx[s] = (f32x4){0.f, 0.f, 0.f, 0.f}; vx[s] = kx[s]; }
        auto put = [&](float* T, const int s, const f32x4 x4) {
            const int idx = lane + s * 64;
            const int row = idx / F4, c4 = (idx - row * F4) * 4;
            *reinterpret_cast<f32x4*>(&T[row * LD + c4]) = x4 * (row < L ? 1.0f : 0.0f);      // finite inputs: x * 0 = 0
        };
#define UU3D_ATTN_W15(x) "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14])
        asm volatile("s_waitcnt vmcnt(%15)" : UU3D_ATTN_W15(kx) : "i"(NS));       // the K rows are here, the V rows still in flight
#pragma unroll
        for (int s = 0; s < NS; ++s) put(Ks, s, kx[s]);
        asm volatile("s_waitcnt vmcnt(0)" : UU3D_ATTN_W15(vx));
#pragma unroll
        for (int s = 0; s < NS; ++s) put(Vs, s, vx[s]);
#undef UU3D_ATTN_W15
    }
    f32x4 qnext[KT];
    auto issue_q = [&](const int tile) {
        const int qrow = min(16 * tile + qi, L - 1);        // rows above L: a copy of the last row, never stored
#pragma unroll
        for (int t = 0; t < KT; ++t) qnext[t] = *reinterpret_cast<const f32x4*>(base + (size_t)qrow * ld + 16 * t + 4 * g);
    };
    issue_q(0);
    // key mask of this sequence as additive terms, once (the same for every query tile)
    float madd[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = 16 * j + 4 * g + r;
            // branch-free (clamped) byte load: with the test around it hipcc emits one branch + wait per key, 4 NT serial round trips
            const uint8_t mk = (key_mask != nullptr) ? key_mask[(size_t)b * L + min(key, L - 1)] : (uint8_t)1;
            madd[j][r] = (key < L) ? (mk ? 0.0f : 1.0f) * -1e9f : -INFINITY;
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const float scale_mul = 1.0f / sqrtf((float)DH);
    ATTN_STAMP(asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const long long c1 = clock64();)

#pragma unroll
    for (int tile = 0; tile < NT; ++tile) {
        ATTN_STAMP(const long long t0 = clock64();)
        f32x4 qf[KT];
#pragma unroll
        for (int t = 0; t < KT; ++t) qf[t] = qnext[t];
        if (tile + 1 < NT) issue_q(tile + 1);              // in flight while this tile is computed
        // S^T tiles: st[j][r] = <Q[16 tile + qi], K[16j + 4g + r]>; the NT chains are independent, interleaved
        f32x4 st[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) st[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            f32x4 kf[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) kf[j] = *reinterpret_cast<const f32x4*>(&Ks[(16 * j + qi) * LD + 16 * t + 4 * g]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    st[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j][s], qf[t][s], st[j], 0, 0, 0);
        }
        ATTN_STAMP(asm volatile("s_nop 0" :: "v"(st[NT - 1][3])); const long long t1 = clock64();)
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = st[j][r] * scale_mul + madd[j][r];
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
                const float e = __builtin_amdgcn_exp2f((st[j][r] - mx) * 1.44269504088896341f);
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

        ATTN_STAMP(asm volatile("s_nop 0" :: "v"(st[NT - 1][3])); const long long t2 = clock64();)
        // O = P V : the KT channel tiles are independent chains, interleaved
        f32x4 o[KT];
#pragma unroll
        for (int t = 0; t < KT; ++t) o[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            // V values by name, two key tiles ahead of the MFMAs that use them, with counted waits (LDS returns in order):
            // hipcc placed each ds_read directly in front of its two MFMAs -- an LDS round trip per 64 MFMA cycles
            float vv[3][4 * KT];
            const unsigned vb = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(Vs + (4 * g) * LD + qi);
#define UU3D_ATTN_VREAD(buf, jj) \
            _Pragma("unroll") for (int s = 0; s < 4; ++s) \
            _Pragma("unroll") for (int t = 0; t < KT; ++t) \
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(vv[buf][s * KT + t]) : "v"(vb), "i"(((16 * (jj) + s) * LD + 16 * t) * 4) : "memory");
            // The counted waits below are only right while nothing but these reads is outstanding on lgkmcnt: scalar loads
            // return out of order.  Start from zero; the "memory" clobbers keep the compiler's own memory operations (a lazily
            // placed kernel-argument s_load included) from being scheduled in between.  (tests/test_isa_cpu.py checks.)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            UU3D_ATTN_VREAD(0, 0)
            if (NT > 1) { UU3D_ATTN_VREAD(1, 1) }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (j + 2 < NT) { UU3D_ATTN_VREAD((j + 2) % 3, j + 2) }
                static_assert(KT == 3, "the counted wait below lists 4 * KT = 12 registers");
                float (&c)[4 * KT] = vv[j % 3];
                if (j + 2 < NT)      asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8]), "+v"(c[9]), "+v"(c[10]), "+v"(c[11]) :: "memory");
                else if (j + 1 < NT) asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8]), "+v"(c[9]), "+v"(c[10]), "+v"(c[11]) :: "memory");
                else                 asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8]), "+v"(c[9]), "+v"(c[10]), "+v"(c[11]) :: "memory");
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int t = 0; t < KT; ++t)
                        o[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(st[j][s], c[s * KT + t], o[t], 0, 0, 0);
            }
#undef UU3D_ATTN_VREAD
        }
        if constexpr (SPLIT) {
            // The two f16 planes of the 16 x DH tile go through a per-wave LDS image and leave as 16-byte pieces (direct from
            // the C/D registers they were 8 * KT two-byte stores per lane, each a 64-address scatter: the longest phase of a tile).
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const _Float16 hv = (fabsf(o[t][r]) < 6.103515625e-05f) ? (_Float16)0.f : (_Float16)o[t][r];   // = h3_hi, f16 denormals stay on in this kernel
                    const _Float16 lv = (_Float16)((o[t][r] - (float)hv) * 2048.0f);
                    Os[(4 * g + r) * DH + 16 * t + qi] = hv;
                    Os[(16 + 4 * g + r) * DH + 16 * t + qi] = lv;
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            constexpr int PPR = DH / 8, PIECES = 2 * 16 * PPR;             // 16-byte pieces per row / per tile
            _Float16* oh = reinterpret_cast<_Float16*>(out);
#pragma unroll
            for (int i = 0; i < (PIECES + 63) / 64; ++i) {
                const int pc = lane + 64 * i;
                const int plane = pc / (16 * PPR), rem = pc - plane * 16 * PPR, row = rem / PPR, c8 = rem - row * PPR;
                const int q = 16 * tile + row;
                if (pc < PIECES && q < L) {
                    const h16x8v piece = *reinterpret_cast<const h16x8v*>(&Os[(plane * 16 + row) * DH + 8 * c8]);
                    const int grow = b * L + q, k = h * DH + 8 * c8;
                    _Float16* dst = lo_off_s == 512 ? oh + ((size_t)(grow >> 5) * (size_t)(ldo_s >> 4) + (size_t)(k >> 4)) * 1024 + (size_t)plane * 512 + ((k >> 3) & 1) * 256 + (grow & 31) * 8
                                                    : oh + (size_t)plane * lo_off_s + (size_t)grow * ldo_s + k;      // (lo_off == 512: A-fragment order, see attn_f32_kernel)
                    *reinterpret_cast<h16x8v*>(dst) = piece;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else {
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = 16 * tile + 4 * g + r;
                    if (q < L) out[((size_t)b * L + q) * ldo_s + h * DH + 16 * t + qi] = o[t][r];
                }
        }
        ATTN_STAMP(asm volatile("s_nop 0" :: "v"(o[KT - 1])); const long long t3 = clock64(); tqk += t1 - t0; tsm += t2 - t1; tpv += t3 - t2;)
    }
    ATTN_STAMP(if (tid == 0) { atomicAdd(&attn_clk[0], (unsigned long long)(c1 - c0)); atomicAdd(&attn_clk[1], (unsigned long long)tqk);
        atomicAdd(&attn_clk[2], (unsigned long long)tsm); atomicAdd(&attn_clk[3], (unsigned long long)tpv); atomicAdd(&attn_clk[4], 1ull); })
}
template <int NT, int DH>
constexpr size_t attn_head_wave_lds_bytes() { return (size_t)4 * 2 * NT * 16 * (DH + 4) * sizeof(float) + (size_t)4 * 2 * 16 * DH * sizeof(_Float16); }

}  // namespace uu3d
