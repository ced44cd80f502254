// tools/uu3d_tchain.h -- RECORD: the round-5 form of the temporal chain (128-row tiles, residual adds as float atomics), replaced by
// csrc/uu3d_tchain16.h in round 6 and kept for tools/tchain_exp.hip (profiles/r05_tchain_ab.txt, r06_ab_tchain16.txt).  Not part of the library.
//
// uu3d_tchain.h -- every ROW-LOCAL stage of a temporal block in ONE launch (round 5; throughput schedule).
//
// Reference: vit.TransformerBlock.call (common/net/vision_transformer.py:176-195) minus the attention products (:117-129):
//     x += projection(context) ; y = LayerNorm2(x) ; x += fc2(relu(fc1(y))) ; [next block:] q | k | v = wqkv(LayerNorm1(x))
// Round 4 ran this as five chip-wide launches per block (row-panel projection, LayerNorm pass, fused MLP with three partial-sum
// slabs, combine + LayerNorm pass, row-panel QKV); each paid ~13 us of fixed costs and moved its activations through L2 two to
// three times.  Here one workgroup OWNS 128 token rows for the whole chain and walks a CONCATENATED weight stream
//     Wp (12 chunks) | W1 (24) | W2[hidden 0..383] (12) | W2[hidden 384..767] (12) | Wqkv of the next block (36)      = 96 x 48 KiB
// through the 8-wave chunk loop of uu3d_gemm_panel8.h (contraction split over wave pairs, two waves per SIMD, 3 x 48 KiB LDS ring
// refilled in half-chunks by LDS-DMA, counted waits) -- the ring never drains at a stage boundary.
//
// What makes the chain cheap is the TRANSPOSED product and a k order chosen for it:
//   * every product is C^T = W^T A^T: the weight fragment is the MFMA's A operand, the token fragment its B operand.  In the 32 x 32
//     C/D map a lane then holds ONE token (lane & 31) and, per 32-channel chunk, the channels 8 (r >> 2) + 4 g + (r & 3) (g = lane >> 5);
//   * wave (q, hh) of a pair finishes the 16 channels 16 hh .. 16 hh + 15 of every chunk (its partner sends the other half of its
//     partial sums through LDS, as in the 8-wave kernel; wave group 1 reads its weight fragments with the row index flipped by 16 so that
//     "registers 0..7" are the kept ones in both groups);
//   * the 8 finished values of chunk c ARE the token fragment of k-slice 2 c + hh of the next stage, if that stage's weights are packed
//     in the k order 16 s + 8 (j >> 2) + 4 g + (j & 3) and the pair splits the contraction by slice PARITY: LayerNorm output, ReLU
//     output and residual stream never change lanes between stages.  What does not fit the 256 registers of a wave (the 96 values of the
//     residual stream per lane next to 96 operand registers and 64 accumulators) makes a LANE-PRIVATE round trip through an L2-hot scratch
//     slab (16-byte pieces, lane-linear: no barrier, no visibility question -- a lane reads back what it stored itself);
//   * LayerNorm statistics are the only thing tokens need from other lanes: two sums over (lane, lane ^ 32, partner wave) through LDS;
//   * biases that an epilogue needs inside the chunk loop (fc1, QKV) come through the SCALAR cache (s_load_dwordx16 in front of the
//     chunk's first barrier): the loop holds no vector-memory load at all, so nothing by name crosses its back edge and the chunk loop is
//     a real loop (peeled four chunks for the exact counts of the first waits).
//
// Row tiles are always whole: tokens past M read row M - 1 and store into a trash page (lane-local arithmetic: a padded lane cannot
// disturb a live one).
#pragma once
#include "uu3d_tchain16.h"      // flags, TChainArgs, parameter table, tchain_qf_index (csrc/: the product kernel's header holds what the forms share)

namespace uu3d {

static constexpr size_t TC_H_HALFS_PER_TILE = 24 * 8 * 2 * 64 * 8;           // relu(fc1) of a tile as fc2's token fragments, lane-linear
static constexpr size_t TC_X_FLOATS_PER_TILE = 128 * 384;                    // a tile of the residual stream, lane-linear (tchain_xs_index)

__host__ __device__ inline constexpr size_t tchain_scratch_bytes(int m_tiles) {
    return (size_t)m_tiles * (TC_H_HALFS_PER_TILE * 2 + 2 * TC_X_FLOATS_PER_TILE * 4) + TC_TRASH_BYTES;
}

// Lane-linear order of the residual stream INSIDE the chain (row-major only at its boundaries): element (row, channel) of a 128-row tile sits where
// the lane that owns it in the transposed accumulator map stores 256 contiguous bytes per wave instruction -- [chunk c][wave = 4 hh + q][i][e][lane = t + 32 g]
// with channel = 32 c + 16 hh + 8 i + 4 g + e and row = 32 q + t.  An atomic instruction of a wave then touches 2 cache lines instead of the 32 of a
// row-major tile (measured: 21 k cycles per chunk with row-major atomics, the whole gain gone).
__host__ __device__ inline size_t tchain_xs_index(int row, int ch) {
    const int tile = row >> 7, q = (row >> 5) & 3, t = row & 31;
    const int c = ch >> 5, hh = (ch >> 4) & 1, i = (ch >> 3) & 1, g = (ch >> 2) & 1, e = ch & 3;
    return (size_t)tile * TC_X_FLOATS_PER_TILE + ((((size_t)(c * 8 + 4 * hh + q) * 2 + i) * 4 + e) * 64 + t + 32 * g);
}

// ---- host side: one stage's chunks of the weight stream from the transposed, padded planes Bt[n][Kp] (k contiguous; lo pre-scaled) ----
// chunk c (output channels 32 c ..), wave group hh, position kk (0..11), plane p, lane l, element j:
//     attention output (context rows in A-fragment order, channels inside a 16-slice in the order v arrived in: lane order, see tchain_qf_index):
//                                                           k = 16 (12 hh + kk) + 8 (j >> 2) + 4 (l >> 5) + (j & 3)
//     lane order (everything this kernel produced itself):  k = kofs + 16 (2 kk + hh) + 8 (j >> 2) + 4 (l >> 5) + (j & 3)
inline void tchain_pack_stage(const _Float16* Bh, const _Float16* Bl, int N, int Kp, int kofs, bool natural, _Float16* out) {
    for (int c = 0; c < N / 32; ++c)
        for (int hh = 0; hh < 2; ++hh)
            for (int kk = 0; kk < 12; ++kk)
                for (int p = 0; p < 2; ++p)
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 8; ++j) {
                            const int n = 32 * c + (l & 31);
                            const int k = natural ? 16 * (12 * hh + kk) + 8 * (j >> 2) + 4 * (l >> 5) + (j & 3)
                                                  : kofs + 16 * (2 * kk + hh) + 8 * (j >> 2) + 4 * (l >> 5) + (j & 3);
                            out[(((((size_t)c * 2 + hh) * 12 + kk) * 2 + p) * 64 + l) * 8 + j] = (p ? Bl : Bh)[(size_t)n * Kp + k];
                        }
}

typedef float f32x16s __attribute__((ext_vector_type(16)));

#ifndef UU3D_TC_LOO
#define UU3D_TC_LOO 0          // tools/tchain_exp: leave-one-out timing builds (results wrong): 1 no refill DMA, 2 no finish, 4 no exchange, 8 no mid barrier, 16 no fragment reads, 32 no MFMA, 64 plane stores coalesced, 128 no bias load, 256 transitions without memory traffic, 512 projection and fc2's first half store into a slab instead of adding atomically, 1024 the waves of token panels 2 and 3 only refill the ring and keep the barriers (a 64-row tile's arithmetic and traffic on the 128-row skeleton)
#endif
#ifdef UU3D_TC_STAMP
// tools/tchain_exp: per workgroup 16 pairs (s_memtime = shader clock ticks, s_memrealtime = 100 MHz) at the chain's stage boundaries
__device__ unsigned long long tchain_stamps[256 * 32];
__device__ unsigned long long tchain_acc[32 * 4];      // per stage set (FLAGS): sum of workgroup cycles, sum of 100 MHz ticks, workgroups, -
#define TC_STAMP(i) do { if (tid == 0 && bm < 256) { tchain_stamps[bm * 32 + 2 * (i)] = __builtin_amdgcn_s_memtime(); tchain_stamps[bm * 32 + 2 * (i) + 1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define TC_STAMP(i)
#endif

// ---- epilogues of a stage: what happens to the 8 finished values v[j] (channels 32 c + 16 hh + 8 (j >> 2) + 4 g + (j & 3)) of a lane ----
// x[token][channel] += v (+ bias): the residual Dense layers (projection, the two K halves of fc2) add into the residual stream IN MEMORY with
// no-return float atomics -- fire-and-forget like stores (nothing comes back into a register, so nothing by name crosses the loop), executed at
// L2; every element is touched by exactly ONE lane once per stage, so the sum has a fixed order.  Round 5's first form stored the partial result to a
// scratch slab and added it at the stage transition: 0.6-1.2 MB of exposed memory traffic per transition and workgroup, 20-27 % of the kernel
// (timing build without that traffic: 171 -> 182 k sequences/s at batch 128, 184 -> 208 k at batch 512).
template <bool BIAS>
struct TcEpResidual {
    static constexpr int kStores = 8; static constexpr bool kBias = BIAS;
    float* x;                  // the tile of the residual stream, lane-linear (tchain_xs_index), + wave * 512 + lane floats: (c, i, e) at + c * 4096 + i * 256 + e * 64
    const float* bias;
};
template <bool BIAS>
struct TcEpSlabT {             // TIMING BUILDS ONLY (UU3D_TC_LOO & 512): v (+ bias) as two 16-byte stores into a slab instead of the atomics (the result is lost)
    static constexpr int kStores = 2; static constexpr bool kBias = BIAS;
    float* s; const float* bias;
};
struct TcEpHidden {            // relu(v + b1) split into hi / lo = the token fragment of fc2's k-slice 2 c + hh -> lane-linear scratch
    static constexpr int kStores = 2; static constexpr bool kBias = true;
    h16x8* __restrict__ hs;
    const float* bias;
};
struct TcEpQkvFrag {           // (v + bias) [* qscale for chunks < 12] split into hi / lo = ONE 16-byte piece per plane of the fragment-ordered q | k | v (tchain_qf_index)
    static constexpr int kStores = 2; static constexpr bool kBias = true;
    h16x8* __restrict__ qf;    // this wave's panel + lane: group u = 2 c + hh at + u * 128 (hi), + 64 (lo)
    const float* bias; float qscale;
};
struct TcEpPlanes {            // (v + bias) [* qscale for chunks < qchunks] [relu] split into row-major hi / lo planes of leading dimension ld
    static constexpr int kStores = 4; static constexpr bool kBias = true;
    unsigned char* __restrict__ ph; unsigned char* __restrict__ pl;     // this lane's row in the planes (or the trash page) + (16 hh + 4 g) halfs
    const float* bias; int qchunks; float qscale; int relu;
};

template <int FLAGS>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
tchain_kernel(const TChainArgs a)
{
    constexpr int HS = 12;
    constexpr int GT = tchain_chunks(FLAGS);
    static_assert(!((FLAGS & TC_MLP) && (FLAGS & TC_FC1_PLANES)), "one MLP form per launch");
    static_assert(GT >= 12, "at least one stage");
    h3_flush_f16_denormals();
    extern __shared__ __attribute__((aligned(16))) unsigned char psm[];

    const int bm = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = wave >> 2, q = wave & 3, g = lane >> 5;
    const int tok = bm * 128 + q * 32 + (lane & 31);
    const bool live = tok < a.M;
    const int tokc = min(tok, a.M - 1);
    const int chl = 16 * hh + 4 * g;                       // this lane's first channel inside a 32-channel chunk (second group: + 8)
    const bool dead = (UU3D_TC_LOO & 1024) != 0 && q >= 2; // (timing builds)

    // ---- weight stream -> ring: as gemm_h3_panel8_kernel (half-chunk g2 = 2 G + j: k positions [6 j, 6 j + 6) of both wave groups) ----
    const unsigned wofs = (unsigned)(hh * HS * 2048 + q * 3072);
    const unsigned char* const wsrc = reinterpret_cast<const unsigned char*>(a.W) + wofs;
    const unsigned lane16 = (unsigned)lane * 16u;
    auto dma1 = [&](int Gc, int j, int slot, int i) __attribute__((always_inline)) {
        const unsigned char* s = wsrc + (size_t)min(Gc, GT - 1) * P8_CHUNK_BYTES + j * (6 * 2048);
        unsigned char* d = psm + slot * P8_CHUNK_BYTES + wofs + j * (6 * 2048);
        switch (i) {
            case 0: __builtin_amdgcn_global_load_lds((h3_glb_void*)(s + lane16), (h3_lds_void*)d, 16, 0, 0); break;
            case 1: __builtin_amdgcn_global_load_lds((h3_glb_void*)(s + lane16), (h3_lds_void*)d, 16, 1024, 0); break;
            default: __builtin_amdgcn_global_load_lds((h3_glb_void*)(s + lane16), (h3_lds_void*)d, 16, 2048, 0); break;
        }
    };
#pragma unroll
    for (int g2 = 0; g2 < 5; ++g2)
#pragma unroll
        for (int i = 0; i < 3; ++i) dma1(g2 >> 1, g2 & 1, g2 >> 1, i);
    int G = 0, slot = 0;                                   // next chunk of the stream to be consumed and its ring slot (G % 3)

    unsigned char* const xmine = psm + P8_RING_BYTES + wave * 2048 + lane16;
    unsigned char* const xpart = psm + P8_RING_BYTES + (wave ^ 4) * 2048 + lane16;
    float* const stat = reinterpret_cast<float*>(psm + P8_RING_BYTES);
    const unsigned rd0 = (unsigned)(uintptr_t)(h3_lds_void*)(psm + hh * HS * 2048 + (unsigned)(lane ^ (hh << 4)) * 16u);

    h16x8 ah[HS], al[HS];                                  // the token fragments of the running stage (this wave's 12 k positions)
    h16x8 bh[3] = {}, bl[3] = {};
#define UU3D_TC_READ(i, sb, kk) \
    if (!(UU3D_TC_LOO & 16) && !dead) asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" \
                 : "=&v"(bh[i]), "=&v"(bl[i]) : "v"(sb), "i"((kk) * 2048), "i"((kk) * 2048 + 1024))

    // ---- the finished values of one chunk, half i (j = 4 i .. 4 i + 3): own partial sum + the partner's ----
    auto combine = [&](const f32x16& p0, const f32x16& p1, const float (&rv)[8], int i) __attribute__((always_inline)) -> f32x4 {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (p0[4 * i + e] + p1[4 * i + e] * (1.0f / H3_SCALE)) + rv[4 * i + e];
        return v;
    };
    auto bias4 = [&](const f32x16s& sb, int i) __attribute__((always_inline)) -> f32x4 {       // scalar registers 8 i + 4 g + e
        // (the two candidates are made opaque first: hipcc otherwise folds the select into a DYNAMIC index of the 16-vector, a chain of
        // 16 v_cmp / v_cndmask pairs per value whose compare masks it then spilled with v_writelane -- 170 of them per chunk pair.  Opaque as
        // VECTOR registers: behind an asm that DEFINES scalar registers hipcc assumes a scalar load in flight and puts an s_waitcnt lgkmcnt(0)
        // in front of the next scalar operand -- four per chunk in the middle of the fragment prefetch, pinned by tests/test_isa_cpu.py)
        f32x4 b;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float b0 = sb[8 * i + e], b1 = sb[8 * i + 4 + e];
            asm("" : "+v"(b0), "+v"(b1));
            b[e] = g ? b1 : b0;
        }
        return b;
    };
    unsigned char* const psm_dummy = a.scratch + (size_t)bm * (TC_H_HALFS_PER_TILE * 2);
    // finish half i of chunk cp; the 16-byte fragments of TcEpHidden need both halves: `keep` carries half 0 to half 1
    struct Keep { h16x4 h0, l0; };
    auto finish = [&](auto ep, int cp, int i, const f32x4 v, const f32x16s& sb, Keep& keep) __attribute__((always_inline)) {
        using EP = decltype(ep);
        if constexpr (std::is_same<EP, TcEpResidual<false>>::value || std::is_same<EP, TcEpResidual<true>>::value) {
            f32x4 y = v;
            if constexpr (EP::kBias) y = y + bias4(sb, i);
            float* d = ep.x + cp * 4096 + i * 256;
            asm volatile("global_atomic_add_f32 %0, %1, off\n\tglobal_atomic_add_f32 %0, %2, off offset:256\n\t"
                         "global_atomic_add_f32 %0, %3, off offset:512\n\tglobal_atomic_add_f32 %0, %4, off offset:768"
                         :: "v"(d), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]) : "memory");
        } else if constexpr (std::is_same<EP, TcEpSlabT<false>>::value || std::is_same<EP, TcEpSlabT<true>>::value) {
            f32x4 y = v;
            if constexpr (EP::kBias) y = y + bias4(sb, i);
            *reinterpret_cast<f32x4*>(ep.s + cp * 4096 + i * 256) = y;
        } else if constexpr (std::is_same<EP, TcEpQkvFrag>::value) {
            f32x4 y = v + bias4(sb, i);
            if (cp < 12) y = y * ep.qscale;
            h16x4 hi, lo;
            h3_split(y, hi, lo);
            if (i == 0) { keep.h0 = hi; keep.l0 = lo; }
            else {
                h16x8 fh, fl;
#pragma unroll
                for (int e = 0; e < 4; ++e) { fh[e] = keep.h0[e]; fh[4 + e] = hi[e]; fl[e] = keep.l0[e]; fl[4 + e] = lo[e]; }
                h16x8* d = ep.qf + (size_t)(2 * cp + hh) * 128;
                d[0] = fh; d[64] = fl;
            }
        } else if constexpr (std::is_same<EP, TcEpHidden>::value) {
            f32x4 y = v + bias4(sb, i);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], 0.f);
            h16x4 hi, lo;
            h3_split(y, hi, lo);
            if (i == 0) { keep.h0 = hi; keep.l0 = lo; }
            else {
                h16x8 fh, fl;
#pragma unroll
                for (int e = 0; e < 4; ++e) { fh[e] = keep.h0[e]; fh[4 + e] = hi[e]; fl[e] = keep.l0[e]; fl[4 + e] = lo[e]; }
                h16x8* d = ep.hs + ((size_t)(cp * 8 + wave) * 2) * 64;
                d[0] = fh; d[64] = fl;
            }
        } else {
            f32x4 y = v + bias4(sb, i);
            if (cp < ep.qchunks) y = y * ep.qscale;
            if (ep.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], 0.f);
            }
            h16x4 hi, lo;
            h3_split(y, hi, lo);
            const unsigned o = (unsigned)(32 * cp + 8 * i) * 2u;
            if (UU3D_TC_LOO & 64) {                        // (timing: the same bytes as coalesced 512-byte runs)
                unsigned char* d = psm_dummy + (size_t)((cp * 8 + wave) * 4 + 2 * i) * 512 + lane * 8;
                *reinterpret_cast<h16x4*>(d) = hi; *reinterpret_cast<h16x4*>(d + 512) = lo;
            } else {
            *reinterpret_cast<h16x4*>(ep.ph + o) = hi;
            *reinterpret_cast<h16x4*>(ep.pl + o) = lo;
            }
        }
    };

    // ---- one chunk of a stage.  CL = min(local chunk index, 3) decides the exact counts of the waits: vector-memory operations per
    // half-interval in issue order are [first half] 3 pieces, [second half] kStores stores of chunk c - 1 (c > 0), 3 pieces; the barrier
    // that opens a half-interval needs the pieces issued four half-intervals earlier (uu3d_gemm_panel8.h) ----
    // XS (first two chunks of a stage only): vector-memory operations YOUNGER than the pieces those chunks' barriers need that may still be in flight when the
    // stage starts -- the stores of the transition in front of it (store_xs / store_rows).  The counted waits let that many more operations stay outstanding
    // (6-bit counter: at most 63); without it the first barrier of the stage waited for the transition's stores to COMPLETE (flags 23: 178 us against 167).
    // A stage behind another stage starts drained (the epilogue's vmcnt(0)), so any XS is safe there; the kernel's FIRST stage needs XS <= the real count.
    auto chunk = [&](auto cl_tag, auto pre_tag, auto xs_tag, auto ep, const int c, f32x16& x0, f32x16& x1, const f32x16& p0, const f32x16& p1) __attribute__((always_inline)) {
        constexpr int CL = decltype(cl_tag)::value;
        constexpr int XS = CL <= 1 ? decltype(xs_tag)::value : 0;
        constexpr bool PRE_IN = (decltype(pre_tag)::value & 1) != 0, PRE_OUT = (decltype(pre_tag)::value & 2) != 0;
        using EP = decltype(ep);
        constexpr int NST = EP::kStores;
        const int pslot = slot == 0 ? 2 : slot - 1;
        const unsigned sb = rd0 + (unsigned)slot * P8_CHUNK_BYTES;
        // ---- barrier B_c: half-chunk 2 c + 1 landed (own pieces); every LDS read of the previous chunk returned ----
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(9 + NST * ((CL >= 2) + (CL >= 3)) + XS > 63 ? 63 : 9 + NST * ((CL >= 2) + (CL >= 3)) + XS) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!PRE_IN) {                           // (else: issued by the previous chunk of this body, in front of the barrier)
            UU3D_TC_READ(0, sb, 0);
            UU3D_TC_READ(1, sb, 1);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { x0[r] = 0.f; x1[r] = 0.f; }
#pragma unroll
        for (int kk = 0; kk < 6; ++kk) {
            UU3D_TC_READ((kk + 2) % 3, sb, kk + 2);
            asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(bh[kk % 3]), "+v"(bl[kk % 3]) : "i"((UU3D_TC_LOO & 20) ? 0 : CL > 0 && (kk == 2 || kk == 3) ? 6 : 4));
            if ((UU3D_TC_LOO & 32) || dead) asm volatile("" : "+v"(x0), "+v"(x1) : "v"(bh[kk % 3]), "v"(bl[kk % 3]), "v"(ah[kk]), "v"(al[kk]));
            else {
            x0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[kk % 3], ah[kk], x0, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[kk % 3], al[kk], x1, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[kk % 3], ah[kk], x1, 0, 0, 0);
            }
            if (CL > 0 && kk == 1 && !(UU3D_TC_LOO & 4) && !dead) {                       // send the other wave's half of chunk c - 1 (its MFMAs have drained by now)
                f32x4 s0, s1;
#pragma unroll
                for (int e = 0; e < 4; ++e) { s0[e] = p0[8 + e] + p1[8 + e] * (1.0f / H3_SCALE); s1[e] = p0[12 + e] + p1[12 + e] * (1.0f / H3_SCALE); }
                asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:1024" :: "v"((unsigned)(uintptr_t)(h3_lds_void*)xmine), "v"(s0), "v"(s1) : "memory");
            }
            if (kk >= 3 && !(UU3D_TC_LOO & 1)) dma1(G + 2, 1, pslot, kk - 3);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- barrier B'_c: half-chunk 2 c + 2 landed; everybody read the first halves of chunk c and wrote the exchange area ----
        // The bias of chunk c - 1 (16 channels of this wave group) is requested through the scalar cache HERE and becomes a value at the
        // lgkmcnt(0) of k position 7: a scalar load in flight only makes the counted LDS waits in between stricter by one operation.
        f32x16s sbias = {};
        if constexpr (CL > 0 && EP::kBias && !(UU3D_TC_LOO & 128)) {
            const float* bp = ep.bias + 32 * (c - 1);          // (ep.bias already points at this wave group's 16 channels)
            asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(sbias) : "s"(bp) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(4)" :: "i"(9 + NST * (CL >= 2) + XS > 63 ? 63 : 9 + NST * (CL >= 2) + XS) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (!(UU3D_TC_LOO & 8)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        f32x4 r0, r1;
        float rv[8];
        Keep keep;
        if (CL > 0 && !(UU3D_TC_LOO & 4) && !dead)
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "=&v"(r0), "=&v"(r1) : "v"((unsigned)(uintptr_t)(h3_lds_void*)xpart) : "memory");
        else { r0 = f32x4{0.f, 0.f, 0.f, 0.f}; r1 = r0; }
        // The two waves of a SIMD (hh = 0 / 1 of a pair) finish the previous chunk at DIFFERENT k positions -- 7, 8 and 10, 11 -- and
        // issue their refill pieces at the others (9 .. 11 and 6 .. 8): finished in lockstep, the epilogue's vector instructions of both
        // waves left the matrix pipe idle (leave-one-out builds: MFMA time, epilogue time and barrier skeleton simply added up).  The
        // counts of the barriers' waits are the smaller ones of the two orders (stores in front of / behind the pieces).
#pragma unroll
        for (int kk = 6; kk < HS; ++kk) {
            if (kk + 2 < HS) UU3D_TC_READ((kk + 2) % 3, sb, kk + 2);
            const int young = (kk + 1 < HS ? 2 : 0) + (kk + 2 < HS ? 2 : 0);
            if (UU3D_TC_LOO & 20) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bh[kk % 3]), "+v"(bl[kk % 3]), "+v"(r0), "+v"(r1));
            else if (CL > 0 && kk == 6) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(bh[kk % 3]), "+v"(bl[kk % 3]));
            else if (CL > 0 && kk == 7) {
                if constexpr (EP::kBias) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bh[kk % 3]), "+v"(bl[kk % 3]), "+v"(r0), "+v"(r1), "+s"(sbias));
                else asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bh[kk % 3]), "+v"(bl[kk % 3]), "+v"(r0), "+v"(r1));
            }
            else asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(bh[kk % 3]), "+v"(bl[kk % 3]) : "i"(young));
            if ((UU3D_TC_LOO & 32) || dead) asm volatile("" : "+v"(x0), "+v"(x1) : "v"(bh[kk % 3]), "v"(bl[kk % 3]), "v"(ah[kk]), "v"(al[kk]));
            else {
            x0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[kk % 3], ah[kk], x0, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[kk % 3], al[kk], x1, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[kk % 3], ah[kk], x1, 0, 0, 0);
            }
            if constexpr (CL > 0) {
                if (kk == 7) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { rv[e] = r0[e]; rv[4 + e] = r1[e]; }
                }
                if (!(UU3D_TC_LOO & 2) && !dead) {
                    if (hh == 0) {
                        if (kk == 7) finish(ep, c - 1, 0, combine(p0, p1, rv, 0), sbias, keep);
                        if (kk == 8) finish(ep, c - 1, 1, combine(p0, p1, rv, 1), sbias, keep);
                    } else {
                        if (kk == 10) finish(ep, c - 1, 0, combine(p0, p1, rv, 0), sbias, keep);
                        if (kk == 11) finish(ep, c - 1, 1, combine(p0, p1, rv, 1), sbias, keep);
                    }
                }
            }
            if (!(UU3D_TC_LOO & 1)) {
                if (hh == 0) { if (kk >= 9) dma1(G + 3, 0, slot, kk - 9); }
                else { if (kk <= 8) dma1(G + 3, 0, slot, kk - 6); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        G += 1;
        slot = slot == 2 ? 0 : slot + 1;
        if constexpr (PRE_OUT) {                           // the next chunk's first fragments: its first half landed one barrier ago
            const unsigned nb = rd0 + (unsigned)slot * P8_CHUNK_BYTES;
            UU3D_TC_READ(0, nb, 0);
            UU3D_TC_READ(1, nb, 1);
        }
    };

    // ---- a stage of NCH chunks over the token fragments in ah / al; leaves every result finished and stored ----
    auto stage = [&](auto nch_tag, auto ep, auto xs_tag) __attribute__((always_inline)) {
        constexpr int NCH = decltype(nch_tag)::value;
        using EP = decltype(ep);
        static_assert(NCH % 4 == 0 && NCH >= 4, "bodies of four chunks");
        f32x16 a0, a1, b0 = {}, b1 = {};
        // (pre tag: 1 = the chunk's first two fragment reads were issued by its predecessor, 2 = it issues its successor's.  Nothing requested
        // by name crosses the loop's back edge: the first chunk of a body reads behind its barrier.)
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        chunk(I0{}, I2{}, xs_tag, ep, 0, a0, a1, b0, b1);
        chunk(I1{}, I3{}, xs_tag, ep, 1, b0, b1, a0, a1);
        chunk(I2{}, I3{}, xs_tag, ep, 2, a0, a1, b0, b1);
        chunk(I3{}, I1{}, xs_tag, ep, 3, b0, b1, a0, a1);
#pragma unroll 1
        for (int c = 4; c < NCH; c += 4) {
            chunk(I3{}, I2{}, xs_tag, ep, c, a0, a1, b0, b1);
            chunk(I3{}, I3{}, xs_tag, ep, c + 1, b0, b1, a0, a1);
            chunk(I3{}, I3{}, xs_tag, ep, c + 2, a0, a1, b0, b1);
            chunk(I3{}, I1{}, xs_tag, ep, c + 3, b0, b1, a0, a1);
        }
        // ---- the last chunk (in b0 / b1): send, barrier, receive, finish ----
        constexpr int c = NCH - 1;
        f32x16s sbias = {};
        if constexpr (EP::kBias) {
            const float* bp = ep.bias + 32 * c;
            asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(sbias) : "s"(bp) : "memory");
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+s"(sbias) :: "memory");
        } else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // everybody's reads of the exchange area (chunk c - 1) returned
        f32x4 s0, s1, r0, r1;
#pragma unroll
        for (int e = 0; e < 4; ++e) { s0[e] = b0[8 + e] + b1[8 + e] * (1.0f / H3_SCALE); s1[e] = b0[12 + e] + b1[12 + e] * (1.0f / H3_SCALE); }
        asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)" :: "v"((unsigned)(uintptr_t)(h3_lds_void*)xmine), "v"(s0), "v"(s1) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r0), "=&v"(r1) : "v"((unsigned)(uintptr_t)(h3_lds_void*)xpart) : "memory");
        float rv[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { rv[e] = r0[e]; rv[4 + e] = r1[e]; }
        Keep keep;
        if (!dead) {
        finish(ep, c, 0, combine(b0, b1, rv, 0), sbias, keep);
        finish(ep, c, 1, combine(b0, b1, rv, 1), sbias, keep);
        }
        __builtin_amdgcn_s_barrier();                      // the exchange area is free again (the transitions keep their statistics there)
    };

    // ================= transitions: everything below works on the lane's own 96 values of the residual stream =================
    f32x4 xr[12][2];                                       // x[token][32 c + 16 hh + 8 i + 4 g + (0..3)]
    unsigned char* const trash = a.scratch + (size_t)a.m_tiles * (TC_H_HALFS_PER_TILE * 2 + 2 * TC_X_FLOATS_PER_TILE * 4);
    h16x8* const hsl = reinterpret_cast<h16x8*>(a.scratch) + (size_t)bm * (TC_H_HALFS_PER_TILE / 8) + lane;
    // the tile of the residual stream in lane-linear order (tchain_xs_index): xs = the temporal stack's, xas = the first strided block's (x + pe)
    float* const xs = reinterpret_cast<float*>(a.scratch + (size_t)a.m_tiles * (TC_H_HALFS_PER_TILE * 2)) + (size_t)bm * TC_X_FLOATS_PER_TILE + wave * 512 + lane;
    float* const xas = xs + (size_t)a.m_tiles * TC_X_FLOATS_PER_TILE;
    auto row_ptr = [&](float* base) __attribute__((always_inline)) -> float* {      // this lane's 16 hh + 4 g slot of its row in a ROW-MAJOR stream (dead lanes: the trash page)
        return live ? base + (size_t)tok * 384 + chl : reinterpret_cast<float*>(trash) + chl;
    };

    // This lane's 96 values out of a lane-linear tile AFTER the stage's atomics: every atomic of this wave has completed (vmcnt(0): atomics count like
    // stores), and the loads bypass the CU's vector L1 (sc0 sc1), which the atomics -- executed at L2 -- never updated.
    auto load_xs = [&](const float* t) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (dead) return;
#pragma unroll
        for (int c = 0; c < 12; ++c)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float* p = t + c * 4096 + i * 256;
                asm volatile("global_load_dword %0, %4, off sc0 sc1\n\tglobal_load_dword %1, %4, off offset:256 sc0 sc1\n\t"
                             "global_load_dword %2, %4, off offset:512 sc0 sc1\n\tglobal_load_dword %3, %4, off offset:768 sc0 sc1"
                             : "=&v"(xr[c][i][0]), "=&v"(xr[c][i][1]), "=&v"(xr[c][i][2]), "=&v"(xr[c][i][3]) : "v"(p) : "memory");
            }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(xr[0][0]), "+v"(xr[0][1]), "+v"(xr[1][0]), "+v"(xr[1][1]), "+v"(xr[2][0]), "+v"(xr[2][1]), "+v"(xr[3][0]), "+v"(xr[3][1]),
                                             "+v"(xr[4][0]), "+v"(xr[4][1]), "+v"(xr[5][0]), "+v"(xr[5][1]), "+v"(xr[6][0]), "+v"(xr[6][1]), "+v"(xr[7][0]), "+v"(xr[7][1]),
                                             "+v"(xr[8][0]), "+v"(xr[8][1]), "+v"(xr[9][0]), "+v"(xr[9][1]), "+v"(xr[10][0]), "+v"(xr[10][1]), "+v"(xr[11][0]), "+v"(xr[11][1]) :: "memory");
    };
    auto store_xs = [&](float* t) __attribute__((always_inline)) {
        if (dead) return;
#pragma unroll
        for (int c = 0; c < 12; ++c)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) t[c * 4096 + i * 256 + e * 64] = xr[c][i][e];
    };
    auto load_rows = [&](const float* base) __attribute__((always_inline)) {          // (rows past M: row M - 1)
        if (dead) return;
        const float* p = base + (size_t)tokc * 384 + chl;
#pragma unroll
        for (int c = 0; c < 12; ++c)
#pragma unroll
            for (int i = 0; i < 2; ++i) xr[c][i] = *reinterpret_cast<const f32x4*>(p + 32 * c + 8 * i);
    };
    auto store_rows = [&](float* base) __attribute__((always_inline)) {
        if (dead) return;
        float* p = row_ptr(base);
#pragma unroll
        for (int c = 0; c < 12; ++c)
#pragma unroll
            for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(p + 32 * c + 8 * i) = xr[c][i];
    };

    // sum over the token's 384 channels: this lane's 96 + lane ^ 32 + the partner wave (both waves add the same two numbers)
    auto token_sum = [&](float s, int phase) __attribute__((always_inline)) -> float {
        s += __shfl_xor(s, 32);
        if (g == 0) stat[phase * 256 + wave * 32 + (lane & 31)] = s;
        // (not __syncthreads(): its fence waits for vmcnt(0), i.e. for a transition's stores still in flight; the LDS write and the barrier are all this needs)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        return s + stat[phase * 256 + (wave ^ 4) * 32 + (lane & 31)];
    };
    // LayerNorm (two-pass, eps inside the root) of xr WITHOUT its affine part -> the next stage's token fragments.  gamma and beta are
    // folded into the Dense layer behind it when the stream is packed (W' = diag(gamma) W, b' = b + beta W): 48 broadcast loads of 16
    // bytes per lane and ~200 vector instructions less per transition.
    auto layer_norm = [&]() __attribute__((always_inline)) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 12; ++c)
#pragma unroll
            for (int i = 0; i < 2; ++i) s += (xr[c][i][0] + xr[c][i][1]) + (xr[c][i][2] + xr[c][i][3]);
        const float mean = token_sum(s, 0) * (1.0f / 384.0f);
        float v = 0.f;
#pragma unroll
        for (int c = 0; c < 12; ++c)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f32x4 d = xr[c][i] - mean;
                v += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
            }
        const float rstd = 1.0f / sqrtf(token_sum(v, 1) * (1.0f / 384.0f) + 1e-5f);
#pragma unroll
        for (int c = 0; c < 12; ++c) {
            h16x4 hi[2], lo[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) h3_split((xr[c][i] - mean) * rstd, hi[i], lo[i]);
#pragma unroll
            for (int e = 0; e < 4; ++e) { ah[c][e] = hi[0][e]; ah[c][4 + e] = hi[1][e]; al[c][e] = lo[0][e]; al[c][4 + e] = lo[1][e]; }
        }
    };

    using XS0 = std::integral_constant<int, 0>; using XSD = std::integral_constant<int, 54>;      // chunk(): XS
    // ================= the chain =================
    // Where the residual stream lives: ROW-MAJOR (a.X / a.XA) at the chain's boundaries -- the first launch reads what spatial_to_temporal_fc wrote,
    // the last temporal launch writes x for the full-sequence head and xa = x + pe for the first strided block, the strided block's launch writes
    // xa += projection for its convolution's residual rows -- and LANE-LINEAR (xs / xas in the scratch) in between, where the epilogues add into it.
    TC_STAMP(0);
    constexpr bool kStrided1 = (FLAGS & TC_FC1_PLANES) != 0;                   // the launch of the first strided block: its stream is xa
    if constexpr ((FLAGS & TC_PROJ) != 0) {
        const int panel = min(bm * 128 + q * 32, a.M - 1) >> 5;
        const h16x8* ap = reinterpret_cast<const h16x8*>(a.Of) + (size_t)panel * 24 * 2 * 64 + lane;
        if (!dead) {
#pragma unroll
        for (int s = 0; s < HS; ++s) { ah[s] = ap[((HS * hh + s) * 2 + 0) * 64]; al[s] = ap[((HS * hh + s) * 2 + 1) * 64]; }
        }
        TC_STAMP(1);
        if constexpr ((UU3D_TC_LOO & 512) != 0 && !kStrided1) stage(std::integral_constant<int, 12>{}, TcEpSlabT<true>{xas - lane + lane * 4, a.P + TCP_BP + 16 * hh}, XS0{});
        else
        stage(std::integral_constant<int, 12>{}, TcEpResidual<true>{kStrided1 ? xas : xs, a.P + TCP_BP + 16 * hh}, XS0{});      // (the kernel's first stage; the fragment loads in front of it are its own operands)
        TC_STAMP(2);
        load_xs(kStrided1 ? xas : xs);
        if constexpr (kStrided1) store_rows(a.XA);                            // (the strided convolution's residual rows, EpConvResidual)
    } else {
        load_rows(a.X);
        store_xs(xs);
    }
    if constexpr ((FLAGS & (TC_MLP | TC_FC1_PLANES)) != 0) {
        TC_STAMP(3);
        layer_norm();
        TC_STAMP(4);
        if constexpr (kStrided1) {
            unsigned char* ph = live ? reinterpret_cast<unsigned char*>(a.H + (size_t)tok * 768 + chl) : trash;
            unsigned char* pl = live ? reinterpret_cast<unsigned char*>(a.H + ((size_t)a.M + tok) * 768 + chl) : trash + 4096;
            stage(std::integral_constant<int, 24>{}, TcEpPlanes{ph, pl, a.P + TCP_B1 + 16 * hh, 0, 1.0f, 1}, XSD{});                 // (behind store_rows(a.XA))
        } else {
            stage(std::integral_constant<int, 24>{}, TcEpHidden{hsl, a.P + TCP_B1 + 16 * hh}, XSD{});
            TC_STAMP(5);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
                if (!dead) {
#pragma unroll
                for (int s = 0; s < HS; ++s) {
                    const h16x8* d = hsl + ((size_t)((12 * half + s) * 8 + wave) * 2) * 64;
                    ah[s] = d[0]; al[s] = d[64];
                }
                }
                if (half == 0) {
                    if constexpr ((UU3D_TC_LOO & 512) != 0) stage(std::integral_constant<int, 12>{}, TcEpSlabT<false>{xas - lane + lane * 4, nullptr}, XSD{});
                    else stage(std::integral_constant<int, 12>{}, TcEpResidual<false>{xs, nullptr}, XSD{});
                }
                else stage(std::integral_constant<int, 12>{}, TcEpResidual<true>{xs, a.P + TCP_B2 + 16 * hh}, XSD{});
            }
            TC_STAMP(6);
            load_xs(xs);
            TC_STAMP(7);
            if constexpr ((FLAGS & TC_QKV) == 0 || (FLAGS & TC_PE) != 0) store_rows(a.X);        // the temporal stack's result: head1 (and head2 without strided blocks) read it
        }
    }
    if constexpr ((FLAGS & TC_QKV) != 0) {
        if constexpr ((FLAGS & TC_PE) != 0) {
            const float* pr = a.pe + (size_t)(tokc % a.period) * 384 + chl;
#pragma unroll
            for (int c = 0; c < 12; ++c)
#pragma unroll
                for (int i = 0; i < 2; ++i) xr[c][i] = xr[c][i] + *reinterpret_cast<const f32x4*>(pr + 32 * c + 8 * i);
            store_xs(xas);                                                     // the stream of the first strided block's launch
        }
        layer_norm();
        TC_STAMP(8);
        // (q's scale in a VECTOR register: as a scalar kernel argument hipcc re-loaded it inside the chunk loop -- an s_load and an s_waitcnt lgkmcnt(0)
        // per use, four per chunk, each of which also waits for the prefetched weight fragments)
        float qs = a.qscale;
        asm("" : "+v"(qs));
        h16x8* const qf = reinterpret_cast<h16x8*>(a.Q) + (size_t)(bm * 4 + q) * (72 * 2 * 64) + lane;     // (whole tiles: the buffer holds m_tiles * 128 rows)
        // (flags = TC_QKV alone: the kernel's first stage, behind store_xs' 96 stores -- 54 of them may stay in flight; else a drained stage behind store_rows / store_xs)
        stage(std::integral_constant<int, 36>{}, TcEpQkvFrag{qf, a.P + TCP_BQKV + 16 * hh, qs}, XSD{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the clamped tail pieces must not outlive the LDS allocation)
    TC_STAMP(9);
#ifdef UU3D_TC_STAMP
    if (tid == 0 && bm < 256) {
        atomicAdd(&tchain_acc[(FLAGS & 31) * 4 + 0], tchain_stamps[bm * 32 + 18] - tchain_stamps[bm * 32 + 0]);
        atomicAdd(&tchain_acc[(FLAGS & 31) * 4 + 1], tchain_stamps[bm * 32 + 19] - tchain_stamps[bm * 32 + 1]);
        atomicAdd(&tchain_acc[(FLAGS & 31) * 4 + 2], 1ull);
    }
#endif
#undef UU3D_TC_READ
}

}  // namespace uu3d
