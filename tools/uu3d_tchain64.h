// tools/uu3d_tchain64.h -- RECORD: the first 64-row form of the temporal chain (four waves of 512 registers), replaced by csrc/uu3d_tchain16.h
// and kept for tools/tchain64_exp.hip (profiles/r06_tchain64.txt, r06_ab_tchain16.txt).  Not part of the library.
//
// uu3d_tchain64.h -- the temporal chain on 64-ROW tiles with the residual stream and the hidden activations ON CHIP (round 6).
//
// Reference: vit.TransformerBlock.call (common/net/vision_transformer.py:176-195) minus the attention products (:117-129):
//     x += projection(context) ; y = LayerNorm2(x) ; x += fc2(relu(fc1(y))) ; [next block:] q | k | v = wqkv(LayerNorm1(x))
//
// Round 5's chain (uu3d_tchain.h) gives a workgroup 128 token rows on 8 waves of 256 registers.  What those rows need between two
// stages -- 192 KiB of residual stream, 192 KiB of token fragments, 384 KiB of relu(fc1) -- does not fit a CU (512 KiB of registers +
// 160 KiB of LDS, 144 KiB of it the weight ring), so the residual adds went to memory as float atomics (executed at the MEMORY side,
// ~1.3 TB/s chip-wide: 1.8 x the cycles of the other stages when every CU runs the chain) and relu(fc1) made a round trip through L2.
// Here a workgroup owns 64 rows on FOUR waves of 512 registers (one per SIMD): per lane 96 registers of residual stream, 96 of the
// running stage's token fragments, 96 of HALF of relu(fc1) -- and the MLP is walked as
//     fc1[hidden 0..383] -> fc2[K half 0] -> fc1[hidden 384..767] -> fc2[K half 1]
// so that only one half of the hidden activations is alive at a time.  Nothing but the launch's input (attention output fragments,
// the residual tile) and output (residual tile, q | k | v) touches memory: no atomics, no lane-private round trips.  The price is
// the weight stream per token row (4.6 MB per 64 rows instead of per 128): L2 -> LDS by LDS-DMA delivers it at ~100 GB/s per CU with
// every CU streaming (tools/wstream_exp.hip: 1.1-1.2 k cycles per 48 KiB chunk against 1.15 k cycles of MFMA work per chunk and SIMD).
//
// Everything else is the round-5 scheme: transposed products (C^T = W^T A^T, a lane holds ONE token), the wave pair (q, hh = 0 / 1)
// splits the contraction by k-slice parity and exchanges half of its partial sums through LDS, the 8 values a lane finishes for
// chunk c ARE its token fragment of k-slice 2 c + hh of the next stage, LayerNorm's affine part is folded into the Dense layer
// behind it, biases come through the scalar cache, the 3 x 48 KiB ring is refilled in half-chunks with counted waits.
// Every stage is the rolled loop of four-chunk bodies.  The stages whose results stay in registers need STATIC register indices: a body always works on
// the first four chunk slots of its register array, and the array is ROTATED by four slots behind every body (96 register moves per four chunks;
// three rotations per 12-chunk stage are the identity) -- fully unrolled stages (60 bodies) sent hipcc's allocator into thousands of spills.
#pragma once
#include <utility>
#include "uu3d_tchain.h"

namespace uu3d {

static constexpr size_t T64_XCHG_BYTES = 4 * 2048;                        // 4 waves x 64 lanes x 8 floats
static constexpr size_t T64_LDS_TOTAL = P8_RING_BYTES + T64_XCHG_BYTES;   // 155648
static constexpr size_t T64_X_FLOATS_PER_TILE = T16_X_FLOATS_PER_TILE;

// Lane-linear order of a 64-row tile of the residual stream between two launches: [chunk c][wave = 2 hh + q][i][lane = t + 32 g][e] with
// channel = 32 c + 16 hh + 8 i + 4 g + e and row = 32 q + t: one coalesced 1 KiB store / load per (chunk, i) and wave.
__host__ __device__ inline size_t tchain64_xs_index(int row, int ch) {
    const int tile = row >> 6, q = (row >> 5) & 1, t = row & 31;
    const int c = ch >> 5, hh = (ch >> 4) & 1, i = (ch >> 3) & 1, g = (ch >> 2) & 1, e = ch & 3;
    return (size_t)tile * T64_X_FLOATS_PER_TILE + ((((size_t)(c * 4 + 2 * hh + q) * 2 + i) * 64 + t + 32 * g) * 4 + e);
}
__host__ __device__ inline constexpr size_t tchain64_scratch_bytes(int m_tiles64) { return tchain16_scratch_bytes(m_tiles64); }
inline void tchain64_reorder_mlp(const _Float16* in, _Float16* out) { tchain16_reorder_mlp(in, out); }

template <bool BIAS> struct T64EpResidual { static constexpr int kStores = 0; static constexpr bool kBias = BIAS; static constexpr int kReg = 1; const float* bias; };
struct T64EpQkvFrag { static constexpr int kStores = 2; static constexpr bool kBias = true; h16x8* __restrict__ qf; const float* bias; };   // (TcEpQkvFrag without q's scale)
struct T64EpHidden { static constexpr int kStores = 0; static constexpr bool kBias = true; static constexpr int kReg = 2; const float* bias; };


// A token fragment lives in ACCUMULATION registers from the moment it is complete: the kernel holds ~380 registers of state per lane, the vector ALU
// addresses 256 of the wave's 512; the MFMA reads its B operand from either file.  (An empty asm whose output is tied to its input: hipcc copies the
// value into an AGPR tuple once, v_accvgpr_write_b32 x 4, and every later use is the MFMA's.)
__device__ __forceinline__ h16x8 t64_park(h16x8 v) { h16x8 o; asm("" : "=a"(o) : "0"(v)); return o; }

#ifndef UU3D_T64_DEPTH
#define UU3D_T64_DEPTH 2       // weight fragments are requested this many k positions ahead of the MFMAs that use them (a ring of DEPTH + 1 register pairs)
#endif
#ifndef UU3D_T64_LOO
#define UU3D_T64_LOO 0         // tools/tchain64_exp: leave-one-out timing builds (results wrong): 1 no refill DMA, 2 no finish, 8 no mid barrier, 16 no chunk barrier, 32 no MFMA, 64 no fragment reads
#endif
#ifdef UU3D_TC_STAMP
#define T64_STAMP(i) do { if (tid == 0 && bm < 256) { tchain_stamps[bm * 32 + 2 * (i)] = __builtin_amdgcn_s_memtime(); tchain_stamps[bm * 32 + 2 * (i) + 1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
// (tools/tchain64_exp -DUU3D_T64_CHUNK_STAMPS: six more stamps inside chunk 9 of the QKV stage -- in front of / behind B_c, behind k position 5, in front of / behind B'_c, behind k position 11)
#ifdef UU3D_T64_CHUNK_STAMPS
#define T64_CSTAMP(i) do { if (std::is_same<EP, T64EpQkvFrag>::value && c == 9) T64_STAMP(10 + (i)); } while (0)
#else
#define T64_CSTAMP(i)
#endif
#else
#define T64_STAMP(i)
#define T64_CSTAMP(i)
#endif

template <int FLAGS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
tchain64_kernel(const TChainArgs a)
{
    constexpr int HS = 12;
    constexpr int GT = tchain_chunks(FLAGS);
    static_assert(!((FLAGS & TC_MLP) && (FLAGS & TC_FC1_PLANES)), "one MLP form per launch");
    static_assert(GT >= 12, "at least one stage");
    h3_flush_f16_denormals();
    extern __shared__ __attribute__((aligned(16))) unsigned char psm[];

    const int bm = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = wave >> 1, q = wave & 1, g = lane >> 5;
    const int tok = bm * 64 + q * 32 + (lane & 31);
    const bool live = tok < a.M;
    const int tokc = min(tok, a.M - 1);
    const int chl = 16 * hh + 4 * g;                       // this lane's first channel inside a 32-channel chunk (second group: + 8)

    // ---- weight stream -> ring: half-chunk g2 = 2 G + j holds the k positions [6 j, 6 j + 6) of both wave groups; this wave moves 6 KiB of
    // the 12 KiB of its group (six 1 KiB pieces) ----
    const unsigned wofs = (unsigned)(hh * HS * 2048 + q * 6144);
    const unsigned char* const wsrc = reinterpret_cast<const unsigned char*>(a.W) + wofs;
    const unsigned lane16 = (unsigned)lane * 16u;
    auto dma1 = [&](int Gc, int j, int slot, int i) __attribute__((always_inline)) {
        if (UU3D_T64_LOO & 1) return;
        const unsigned char* s = wsrc + (size_t)min(Gc, GT - 1) * P8_CHUNK_BYTES + j * (6 * 2048) + (i >= 4 ? 4096 : 0);
        unsigned char* d = psm + slot * P8_CHUNK_BYTES + wofs + j * (6 * 2048) + (i >= 4 ? 4096 : 0);
        switch (i & 3) {
            case 0: __builtin_amdgcn_global_load_lds((h3_glb_void*)(s + lane16), (h3_lds_void*)d, 16, 0, 0); break;
            case 1: __builtin_amdgcn_global_load_lds((h3_glb_void*)(s + lane16), (h3_lds_void*)d, 16, 1024, 0); break;
            case 2: __builtin_amdgcn_global_load_lds((h3_glb_void*)(s + lane16), (h3_lds_void*)d, 16, 2048, 0); break;
            default: __builtin_amdgcn_global_load_lds((h3_glb_void*)(s + lane16), (h3_lds_void*)d, 16, 3072, 0); break;
        }
    };

    // ================= the lane's state =================
    f32x4 xr[12][2];                                       // residual stream: x[token][32 c + 16 hh + 8 i + 4 g + (0..3)]
    h16x8 ah[HS], al[HS];                                  // token fragments of the running stage (LayerNorm output / attention output)
    h16x8 fh[HS], fl[HS];                                  // one half of relu(fc1) as fc2's token fragments
    constexpr int DP = UU3D_T64_DEPTH, RB = DP + 1;
    h16x8 bh[RB] = {}, bl[RB] = {};

    // Addresses used at the END of the kernel (the residual tile's store, the rows of x / xa, the positional encoding) are computed from a workgroup index that
    // passes through an empty asm right there: hipcc otherwise forms all of them in the prologue and carries them -- through scratch -- across the whole chain.
    auto late = [&](int v) __attribute__((always_inline)) -> int { asm volatile("" : "+s"(v)); return v; };
    const unsigned xoff = (unsigned)(wave * 512 + lane * 4);
    auto xs_tile = [&](int b, bool strided1) __attribute__((always_inline)) -> float* {      // residual tiles: temporal stack | first strided block (x + pe)
        return reinterpret_cast<float*>(a.scratch) + ((size_t)b + (strided1 ? (size_t)a.m_tiles : 0)) * T64_X_FLOATS_PER_TILE;
    };
    auto load_xs = [&](const float* t) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < 12; ++c)
#pragma unroll
            for (int i = 0; i < 2; ++i) xr[c][i] = *reinterpret_cast<const f32x4*>(t + (xoff + (unsigned)(c * 2048 + i * 256)));
    };
    auto store_xs = [&](float* t) __attribute__((always_inline)) {
        unsigned xo = xoff;
        asm volatile("" : "+v"(xo));                       // (not the 24 zero-extended offsets load_xs formed two hundred microseconds ago)
#pragma unroll
        for (int c = 0; c < 12; ++c)
#pragma unroll
            for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(t + (xo + (unsigned)(c * 2048 + i * 256))) = xr[c][i];
    };
    auto load_rows = [&](const float* base) __attribute__((always_inline)) {          // (rows past M: row M - 1)
        const float* p = base + (size_t)tokc * 384 + chl;
#pragma unroll
        for (int c = 0; c < 12; ++c)
#pragma unroll
            for (int i = 0; i < 2; ++i) xr[c][i] = *reinterpret_cast<const f32x4*>(p + 32 * c + 8 * i);
    };
    auto store_rows = [&](float* base) __attribute__((always_inline)) {               // (dead lanes: the trash page)
        const int tk = late(bm) * 64 + q * 32 + (lane & 31);
        unsigned char* const tr = a.scratch + (size_t)a.m_tiles * (2 * T64_X_FLOATS_PER_TILE * 4);
        float* p = tk < a.M ? base + (size_t)tk * 384 + chl : reinterpret_cast<float*>(tr) + chl;
#pragma unroll
        for (int c = 0; c < 12; ++c)
#pragma unroll
            for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(p + 32 * c + 8 * i) = xr[c][i];
    };

    // ---- what the launch reads by name, in FRONT of the ring's first pieces (vector memory returns in order) ----
    constexpr bool kStrided1 = (FLAGS & TC_FC1_PLANES) != 0;                   // the launch of the first strided block: its stream is xa
    T64_STAMP(0);
    if constexpr ((FLAGS & TC_PROJ) != 0) {
        load_xs(xs_tile(bm, kStrided1));
        const int panel = min(bm * 64 + q * 32, a.M - 1) >> 5;
        const h16x8* ap = reinterpret_cast<const h16x8*>(a.Of) + (size_t)panel * 24 * 2 * 64 + lane;
#pragma unroll
        for (int s = 0; s < HS; ++s) { ah[s] = t64_park(ap[((HS * hh + s) * 2 + 0) * 64]); al[s] = t64_park(ap[((HS * hh + s) * 2 + 1) * 64]); }
    } else {
        load_rows(a.X);
    }
#pragma unroll
    for (int g2 = 0; g2 < 5; ++g2)
#pragma unroll
        for (int i = 0; i < 6; ++i) dma1(g2 >> 1, g2 & 1, g2 >> 1, i);
    int G = 0, slot = 0;                                   // next chunk of the stream to be consumed and its ring slot (G % 3)

    unsigned char* const xmine = psm + P8_RING_BYTES + wave * 2048 + lane16;
    unsigned char* const xpart = psm + P8_RING_BYTES + (wave ^ 2) * 2048 + lane16;
    float* const stat = reinterpret_cast<float*>(psm + P8_RING_BYTES);
    const unsigned rd0 = (unsigned)(uintptr_t)(h3_lds_void*)(psm + hh * HS * 2048 + (unsigned)(lane ^ (hh << 4)) * 16u);

#define UU3D_T64_READ(i, sb, kk) \
    if (!(UU3D_T64_LOO & 64)) asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" \
                 : "=&v"(bh[i]), "=&v"(bl[i]) : "v"(sb), "i"((kk) * 2048), "i"((kk) * 2048 + 1024))

    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    // ---- the epilogue of a chunk, value by value.  A wave is ALONE on its SIMD: whatever it issues between two MFMAs beyond ~24 cycles leaves the matrix
    // pipe idle, and a block of vector instructions behind a group of MFMAs waits for the whole group to issue first.  So the epilogue of chunk c - 1 is cut
    // into slices of a few instructions and every slice sits in ONE gap behind one MFMA of chunk c (gap g = 3 kk + m behind MFMA m of k position kk):
    //     g 3 .. 6    the partner's half of the partial sums (accumulator registers 8 .. 15), two values per gap;      g 7: send (two ds_write_b128)
    //     g 9 .. 16   the own half u[j] = x0[j] + x1[j] / 2048, one value per gap
    //     g 21 .. 29  value j = g - 21: + the partner's sum + bias (+ residual add / ReLU), hi part; its lo part one gap later beside the next value's first step
    //     g 31        the epilogue's stores / the fragment's move into its register slot
    struct Fin { float u[8], y[8]; f32x4 s0, s1, r0, r1; _Float16 vh[8], vl[8]; };
    auto bias1 = [&](const f32x16s& sb, const int j) __attribute__((always_inline)) -> float {       // scalar registers 8 i + 4 g + e of value j = 4 i + e (see uu3d_tchain.h)
        float b0 = sb[8 * (j >> 2) + (j & 3)], b1 = sb[8 * (j >> 2) + 4 + (j & 3)];
        asm("" : "+v"(b0), "+v"(b1));
        return g ? b1 : b0;
    };
    // value j in two steps that sit in CONSECUTIVE gaps (g and g + 1), so that a gap holds step B of value j - 1 beside step A of value j: two independent
    // dependency chains per gap (a chain of dependent vector instructions issues one every ~8 cycles, two interleaved chains one every ~4)
    auto fin_a = [&](auto ep, auto fin_tag, const int j, Fin& f, const f32x16s& sb) __attribute__((always_inline)) {
        using EP = decltype(ep);
        constexpr int fs = decltype(fin_tag)::value;        // register slot of the finished chunk (the register epilogues; see stage())
        float y = f.u[j] + (j < 4 ? f.r0[j & 3] : f.r1[j & 3]);
        if constexpr (EP::kBias) y += bias1(sb, j);
        if constexpr (std::is_same<EP, T64EpResidual<false>>::value || std::is_same<EP, T64EpResidual<true>>::value) {
            xr[fs][j >> 2][j & 3] += y;
        } else {
            if constexpr (std::is_same<EP, T64EpHidden>::value) y = fmaxf(y, 0.f);
            if constexpr (std::is_same<EP, TcEpPlanes>::value) { if (ep.relu) y = fmaxf(y, 0.f); }
            // hi = f16(y) (v_cvt_f16_f32 by name: flushes denormal results, see h3_hi) and its float value, one asm statement (no s_nop between the two)
            _Float16 h; float hf;
            asm("v_cvt_f16_f32 %0, %2\n\tv_cvt_f32_f16 %1, %0" : "=&v"(h), "=v"(hf) : "v"(y));
            f.y[j] = y - hf;
            f.vh[j] = h;
        }
    };
    auto fin_b = [&](auto ep, const int j, Fin& f) __attribute__((always_inline)) {
        using EP = decltype(ep);
        if constexpr (!(std::is_same<EP, T64EpResidual<false>>::value || std::is_same<EP, T64EpResidual<true>>::value))
            f.vl[j] = (_Float16)(f.y[j] * H3_SCALE);
    };
    auto fin_store = [&](auto ep, const int cp, auto fin_tag, Fin& f) __attribute__((always_inline)) {
        using EP = decltype(ep);
        constexpr int fs = decltype(fin_tag)::value;
        if constexpr (std::is_same<EP, T64EpHidden>::value || std::is_same<EP, T64EpQkvFrag>::value) {
            h16x8 ph, pl;
#pragma unroll
            for (int j = 0; j < 8; ++j) { ph[j] = f.vh[j]; pl[j] = f.vl[j]; }
            if constexpr (std::is_same<EP, T64EpHidden>::value) { fh[fs] = t64_park(ph); fl[fs] = t64_park(pl); }
            else { h16x8* d = ep.qf + (size_t)(2 * cp + hh) * 128; d[0] = ph; d[64] = pl; }
        } else if constexpr (std::is_same<EP, TcEpPlanes>::value) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                h16x4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) { hi[e] = f.vh[4 * i + e]; lo[e] = f.vl[4 * i + e]; }
                const unsigned o = (unsigned)(32 * cp + 8 * i) * 2u;
                *reinterpret_cast<h16x4*>(ep.ph + o) = hi;
                *reinterpret_cast<h16x4*>(ep.pl + o) = lo;
            }
        }
    };

    // ---- one chunk of a stage over the token fragments Ah / Al (uu3d_tchain.h: chunk()).  Vector-memory operations per half-interval in issue
    // order: [first half] 6 pieces, [second half] 6 pieces with the kStores stores of chunk c - 1 between them; the barrier that opens a
    // half-interval needs the pieces issued four half-intervals earlier.  Every stage starts drained (the tail's vmcnt(0)). ----
    auto chunk = [&](auto cl_tag, auto pre_tag, const int c, auto fin_tag, auto ep, const h16x8 (&Ah)[HS], const h16x8 (&Al)[HS],
                     f32x16& x0, f32x16& x1, const f32x16& p0, const f32x16& p1) __attribute__((always_inline)) {
        constexpr int CL = decltype(cl_tag)::value;
        constexpr bool PRE_IN = (decltype(pre_tag)::value & 1) != 0, PRE_OUT = (decltype(pre_tag)::value & 2) != 0;
        using EP = decltype(ep);
        constexpr int NST = EP::kStores;
        const int pslot = slot == 0 ? 2 : slot - 1;
        const unsigned sb = rd0 + (unsigned)slot * P8_CHUNK_BYTES;
        Fin f;
        f32x16s sbias = {};
        // one finished value of the previous chunk: x0[k] + x1[k] / 2048, its two accumulator registers read HERE (by name: hipcc otherwise copies all 32 registers
        // of the finished accumulators into vector registers in front of the chunk's first barrier and keeps them there for half a chunk)
        auto acc = [&](const int k) __attribute__((always_inline)) -> float {
            float a, b;                                    // (the multiply-add inside the asm: hipcc puts an s_nop behind every asm statement whose result the next instruction reads)
            asm volatile("v_accvgpr_read_b32 %0, %2\n\tv_accvgpr_read_b32 %1, %3\n\tv_fmac_f32 %0, 0x3a000000, %1" : "=&v"(a), "=&v"(b) : "a"(p0[k]), "a"(p1[k]));
            static_assert(H3_SCALE == 2048.0f, "0x3a000000 = 1 / 2048");
            return a;
        };
        auto gapwork = [&](const int gp) __attribute__((always_inline)) {
            if constexpr (CL > 0 && (UU3D_T64_LOO & 2) != 0) { if (gp == 3) asm volatile("" :: "a"(p0), "a"(p1)); }      // (timing builds: the previous chunk's MFMAs stay alive)
            if constexpr (CL > 0 && !(UU3D_T64_LOO & 2)) {
                if (gp >= 3 && gp <= 6) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int k = 2 * (gp - 3) + t;
                        const float v = acc(8 + k);
                        if (k < 4) f.s0[k & 3] = v; else f.s1[k & 3] = v;
                    }
                }
                if (gp == 7) asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:1024" :: "v"((unsigned)(uintptr_t)(h3_lds_void*)xmine), "v"(f.s0), "v"(f.s1) : "memory");
                if (gp >= 9 && gp <= 16) f.u[gp - 9] = acc(gp - 9);
                if (gp >= 22 && gp <= 29) fin_b(ep, gp - 22, f);
                if (gp >= 21 && gp <= 28) fin_a(ep, fin_tag, gp - 21, f, sbias);
                if (gp == 31) fin_store(ep, c - 1, fin_tag, f);
            }
        };
        // ---- barrier B_c: half-chunk 2 c + 1 landed (own pieces) ----
        // (LDS: every read of the previous chunk has been waited for by the MFMA that used it, the exchange's by the finish; the reads in flight
        // here are this chunk's first two fragment pairs (PRE_IN), out of a slot that is not refilled before B'_c -- they stay in flight across the barrier)
        T64_CSTAMP(0);
        if constexpr (PRE_IN) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(18 + NST * ((CL >= 2) + (CL >= 3))) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(18 + NST * ((CL >= 2) + (CL >= 3))) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (!(UU3D_T64_LOO & 16)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        T64_CSTAMP(1);
        if constexpr (!PRE_IN) {
#pragma unroll
            for (int d = 0; d < DP; ++d) UU3D_T64_READ(d, sb, d);
        }
        // (Tried: the MFMAs as inline asm with their accumulators in vector registers -- no v_accvgpr_read_b32 per finished value, 32 instructions per chunk
        // less.  Same time within 2 %, and hipcc, which no longer saw MFMAs, spilled registers of fragment reads IN FLIGHT into AGPRs: results wrong and
        // different from run to run in two of the five stage sets.  The builtins stay.)
#pragma unroll
        for (int r = 0; r < 16; ++r) { x0[r] = 0.f; x1[r] = 0.f; }
#define UU3D_T64_KK(kk) \
            if (UU3D_T64_LOO & 32) { asm volatile("" : "+v"(x0), "+v"(x1) : "v"(bh[(kk) % RB]), "v"(bl[(kk) % RB]), "v"(Ah[kk]), "v"(Al[kk])); gapwork(3 * (kk)); gapwork(3 * (kk) + 1); gapwork(3 * (kk) + 2); } \
            else { \
            x0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[(kk) % RB], Ah[kk], x0, 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0); gapwork(3 * (kk)); __builtin_amdgcn_sched_barrier(0); \
            x1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[(kk) % RB], Al[kk], x1, 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0); gapwork(3 * (kk) + 1); __builtin_amdgcn_sched_barrier(0); \
            x1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[(kk) % RB], Ah[kk], x1, 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0); gapwork(3 * (kk) + 2); __builtin_amdgcn_sched_barrier(0); \
            }
#pragma unroll
        for (int kk = 0; kk < 6; ++kk) {
            UU3D_T64_READ((kk + DP) % RB, sb, kk + DP);
            // (LDS operations in issue order: ... reads kk + DP - 1 | [the send's two writes, behind k position 2] | reads kk + DP: the writes are younger than reads kk for kk <= 2 + DP)
            asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(bh[kk % RB]), "+v"(bl[kk % RB]) : "i"(2 * DP + (CL > 0 && !(UU3D_T64_LOO & 2) && kk >= 3 && kk <= 2 + DP ? 2 : 0)));
            UU3D_T64_KK(kk)
            dma1(G + 2, 1, pslot, kk);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- barrier B'_c: half-chunk 2 c + 2 landed; everybody read the first halves of chunk c and wrote the exchange area ----
        if constexpr (CL > 0 && EP::kBias) {
            const float* bp = ep.bias + 32 * (c - 1);          // (ep.bias already points at this wave group's 16 channels)
            asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(sbias) : "s"(bp) : "memory");
        }
        T64_CSTAMP(2);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(%1)" :: "i"(18 + NST * (CL >= 2)), "i"(2 * DP) : "memory");      // (the fragment pairs of k positions 6 .. 5 + DP stay in flight; the send's writes are older)
        __builtin_amdgcn_sched_barrier(0);
        if (!(UU3D_T64_LOO & 8)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        T64_CSTAMP(3);
        if constexpr (CL > 0 && !(UU3D_T64_LOO & 2))
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "=&v"(f.r0), "=&v"(f.r1) : "v"((unsigned)(uintptr_t)(h3_lds_void*)xpart) : "memory");
        else { f.r0 = f32x4{0.f, 0.f, 0.f, 0.f}; f.r1 = f.r0; }
#pragma unroll
        for (int kk = 6; kk < HS; ++kk) {
            if (kk + DP < HS) UU3D_T64_READ((kk + DP) % RB, sb, kk + DP);
            // younger than reads kk: the fragment pairs kk + 1 .. min(kk + DP, 11), and the receive's two reads while reads kk were issued in front of B'_c (kk <= 5 + DP)
            const int ahead = 2 * ((kk + DP < HS ? kk + DP : HS - 1) - kk);
            const bool fin_on = CL > 0 && !(UU3D_T64_LOO & 2);
            if (fin_on && kk == 7) {                       // the finish starts behind this wait: the receive (and the bias) must be there
                const int behind_recv = 6 + DP < HS ? 2 * ((7 + DP < HS ? 7 + DP : HS - 1) - (6 + DP) + 1) : 0;
                if constexpr (EP::kBias) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bh[kk % RB]), "+v"(bl[kk % RB]), "+v"(f.r0), "+v"(f.r1), "+s"(sbias));
                else asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(bh[kk % RB]), "+v"(bl[kk % RB]), "+v"(f.r0), "+v"(f.r1) : "i"(behind_recv < ahead ? behind_recv : ahead));
            }
            else asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(bh[kk % RB]), "+v"(bl[kk % RB]) : "i"(ahead + (fin_on && kk == 6 ? 2 : 0)));
            UU3D_T64_KK(kk)
            dma1(G + 3, 0, slot, kk - 6);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef UU3D_T64_KK
        T64_CSTAMP(4);
        G += 1;
        slot = slot == 2 ? 0 : slot + 1;
        if constexpr (PRE_OUT) {                           // the next chunk's first fragments: its first half landed one barrier ago
            const unsigned nb = rd0 + (unsigned)slot * P8_CHUNK_BYTES;
#pragma unroll
            for (int d = 0; d < DP; ++d) UU3D_T64_READ(d, nb, d);
        }
    };

    // ---- the last chunk of a stage (in b0 / b1): send, barrier, receive, finish; leaves the stage drained ----
    auto stage_tail = [&](auto ep, const int c, auto fin_tag, const f32x16& b0, const f32x16& b1) __attribute__((always_inline)) {
        using EP = decltype(ep);
        f32x16s sbias = {};
        if constexpr (EP::kBias) {
            const float* bp = ep.bias + 32 * c;
            asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(sbias) : "s"(bp) : "memory");
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+s"(sbias) :: "memory");
        } else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // everybody's reads of the exchange area (chunk c - 1) returned
        const f32x16& t0 = b0; const f32x16& t1 = b1;
        Fin f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { f.s0[e] = t0[8 + e] + t1[8 + e] * (1.0f / H3_SCALE); f.s1[e] = t0[12 + e] + t1[12 + e] * (1.0f / H3_SCALE); }
        asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)" :: "v"((unsigned)(uintptr_t)(h3_lds_void*)xmine), "v"(f.s0), "v"(f.s1) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)" : "=&v"(f.r0), "=&v"(f.r1) : "v"((unsigned)(uintptr_t)(h3_lds_void*)xpart) : "memory");
#pragma unroll
        for (int j = 0; j < 8; ++j) f.u[j] = t0[j] + t1[j] * (1.0f / H3_SCALE);
#pragma unroll
        for (int j = 0; j < 8; ++j) { fin_a(ep, fin_tag, j, f, sbias); fin_b(ep, j, f); }
        fin_store(ep, c, fin_tag, f);
        __builtin_amdgcn_s_barrier();                      // the exchange area is free again (the transitions keep their statistics there)
    };

    // ---- a stage of NCH chunks: bodies of four chunks (nothing requested by name crosses the back edge).  REG = 0: the results go to memory;
    // 1 / 2: they stay in xr / fh, fl, whose slots rotate by four behind every body (NCH = 12: three rotations = the identity).  The chunk finished
    // inside body chunk j sits in slot j - 1, the previous body's last chunk in slot 11 (it has rotated once). ----
    using F0 = std::integral_constant<int, 0>; using F1 = std::integral_constant<int, 1>; using F2 = std::integral_constant<int, 2>; using F11 = std::integral_constant<int, 11>;
    auto stage = [&](auto nch_tag, auto ep, const h16x8 (&Ah)[HS], const h16x8 (&Al)[HS]) __attribute__((always_inline)) {
        constexpr int NCH = decltype(nch_tag)::value;
        using EP = decltype(ep);
        static_assert(NCH % 4 == 0 && NCH >= 8, "bodies of four chunks");
        auto rotate = [&]() __attribute__((always_inline)) {
            if constexpr (std::is_same<EP, T64EpResidual<false>>::value || std::is_same<EP, T64EpResidual<true>>::value) {
                static_assert(NCH == 12, "three rotations");
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < 2; ++i) { const f32x4 t = xr[s][i]; xr[s][i] = xr[s + 4][i]; xr[s + 4][i] = xr[s + 8][i]; xr[s + 8][i] = t; __builtin_amdgcn_sched_barrier(0); }      // (cycle by cycle: four temporaries, not 96)
            } else if constexpr (std::is_same<EP, T64EpHidden>::value) {
                static_assert(NCH == 12, "three rotations");
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const h16x8 th = fh[s], tl = fl[s];
                    fh[s] = fh[s + 4]; fl[s] = fl[s + 4]; fh[s + 4] = fh[s + 8]; fl[s + 4] = fl[s + 8]; fh[s + 8] = th; fl[s + 8] = tl;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        f32x16 a0, a1, b0 = {}, b1 = {};
        chunk(I0{}, I2{}, 0, F0{}, ep, Ah, Al, a0, a1, b0, b1);
        chunk(I1{}, I3{}, 1, F0{}, ep, Ah, Al, b0, b1, a0, a1);
        chunk(I2{}, I3{}, 2, F1{}, ep, Ah, Al, a0, a1, b0, b1);
        chunk(I3{}, I1{}, 3, F2{}, ep, Ah, Al, b0, b1, a0, a1);
        rotate();
#pragma unroll 1
        for (int c = 4; c < NCH; c += 4) {
            chunk(I3{}, I2{}, c, F11{}, ep, Ah, Al, a0, a1, b0, b1);
            chunk(I3{}, I3{}, c + 1, F0{}, ep, Ah, Al, b0, b1, a0, a1);
            chunk(I3{}, I3{}, c + 2, F1{}, ep, Ah, Al, a0, a1, b0, b1);
            chunk(I3{}, I1{}, c + 3, F2{}, ep, Ah, Al, b0, b1, a0, a1);
            rotate();
        }
        stage_tail(ep, (int)(NCH - 1), F11{}, b0, b1);
    };
    using N12 = std::integral_constant<int, 12>;

    // sum over the token's 384 channels: this lane's 96 + lane ^ 32 + the partner wave (both waves add the same two numbers)
    auto token_sum = [&](float s, int phase) __attribute__((always_inline)) -> float {
        s += __shfl_xor(s, 32);
        if (g == 0) stat[phase * 128 + wave * 32 + (lane & 31)] = s;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        return s + stat[phase * 128 + (wave ^ 2) * 32 + (lane & 31)];
    };
    // LayerNorm (two-pass, eps inside the root) of xr WITHOUT its affine part (folded into the Dense layer behind it) -> ah / al
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
            h16x8 ph, pl;
#pragma unroll
            for (int e = 0; e < 4; ++e) { ph[e] = hi[0][e]; ph[4 + e] = hi[1][e]; pl[e] = lo[0][e]; pl[4 + e] = lo[1][e]; }
            ah[c] = t64_park(ph); al[c] = t64_park(pl);
        }
        // (the statistics sit in the exchange area: a wave sends again in chunk 1 of the next stage, two workgroup barriers after every wave has read them)
    };

    // ================= the chain =================
    T64_STAMP(1);
    if constexpr ((FLAGS & TC_PROJ) != 0) {
        stage(N12{}, T64EpResidual<true>{a.P + TCP_BP + 16 * hh}, ah, al);
        T64_STAMP(2);
        if constexpr (kStrided1) store_rows(a.XA);                            // (the strided convolution's residual rows, EpConvResidual)
    } else {
        store_xs(xs_tile(bm, false));                                         // (the first launch: the residual stream enters the chain's order)
    }
    if constexpr ((FLAGS & (TC_MLP | TC_FC1_PLANES)) != 0) {
        T64_STAMP(3);
        layer_norm();
        T64_STAMP(4);
        if constexpr (kStrided1) {
            unsigned char* const trash = a.scratch + (size_t)a.m_tiles * (2 * T64_X_FLOATS_PER_TILE * 4);
            unsigned char* ph = live ? reinterpret_cast<unsigned char*>(a.H + (size_t)tok * 768 + chl) : trash;
            unsigned char* pl = live ? reinterpret_cast<unsigned char*>(a.H + ((size_t)a.M + tok) * 768 + chl) : trash + 4096;
            stage(std::integral_constant<int, 24>{}, TcEpPlanes{ph, pl, a.P + TCP_B1 + 16 * hh, 0, 1.0f, 1}, ah, al);
        } else {
            stage(N12{}, T64EpHidden{a.P + TCP_B1 + 16 * hh}, ah, al);                       // relu(fc1)[0..383]
            stage(N12{}, T64EpResidual<false>{nullptr}, fh, fl);                            // x += it . W2[0..383]
            T64_STAMP(5);
            stage(N12{}, T64EpHidden{a.P + TCP_B1 + 384 + 16 * hh}, ah, al);                 // relu(fc1)[384..767]
            stage(N12{}, T64EpResidual<true>{a.P + TCP_B2 + 16 * hh}, fh, fl);              // x += it . W2[384..767] + b2
            T64_STAMP(6);
            if constexpr ((FLAGS & TC_QKV) == 0 || (FLAGS & TC_PE) != 0) store_rows(a.X);        // the temporal stack's result: head1 (and head2 without strided blocks) read it
        }
    }
    if constexpr ((FLAGS & TC_QKV) != 0) {
        T64_STAMP(7);
        if constexpr ((FLAGS & TC_PE) != 0) {
            const int tkc = min(late(bm) * 64 + q * 32 + (lane & 31), a.M - 1);
            const float* pr = a.pe + (size_t)(tkc % a.period) * 384 + chl;
#pragma unroll
            for (int c = 0; c < 12; ++c)
#pragma unroll
                for (int i = 0; i < 2; ++i) xr[c][i] = xr[c][i] + *reinterpret_cast<const f32x4*>(pr + 32 * c + 8 * i);
            store_xs(xs_tile(late(bm), true));                                 // the stream of the first strided block's launch
        } else if constexpr ((FLAGS & (TC_PROJ | TC_MLP)) != 0) {
            store_xs(xs_tile(late(bm), false));                                // the next temporal launch's residual tile
        }
        layer_norm();
        T64_STAMP(8);
        h16x8* const qf = reinterpret_cast<h16x8*>(a.Q) + (size_t)(late(bm) * 2 + q) * (72 * 2 * 64) + lane;     // (whole tiles: the buffer holds m_tiles * 64 rows)
        stage(std::integral_constant<int, 36>{}, T64EpQkvFrag{qf, a.P + TCP_BQKV + 16 * hh}, ah, al);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the clamped tail pieces must not outlive the LDS allocation)
    T64_STAMP(9);
#ifdef UU3D_TC_STAMP
    if (tid == 0 && bm < 256) {
        atomicAdd(&tchain_acc[(FLAGS & 31) * 4 + 0], tchain_stamps[bm * 32 + 18] - tchain_stamps[bm * 32 + 0]);
        atomicAdd(&tchain_acc[(FLAGS & 31) * 4 + 1], tchain_stamps[bm * 32 + 19] - tchain_stamps[bm * 32 + 1]);
        atomicAdd(&tchain_acc[(FLAGS & 31) * 4 + 2], 1ull);
    }
#endif
#undef UU3D_T64_READ
}

}  // namespace uu3d
