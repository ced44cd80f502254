// uu3d_tchain16.h -- every ROW-LOCAL stage of a temporal block in ONE launch: the temporal chain (throughput schedule).
//
// Reference: vit.TransformerBlock.call (common/net/vision_transformer.py:176-195) minus the attention products (:117-129):
//     x += projection(context) ; y = LayerNorm2(x) ; x += fc2(relu(fc1(y))) ; [next block:] q | k | v = wqkv(LayerNorm1(x))
//
// One workgroup OWNS 64 token rows for the whole chain and walks a CONCATENATED weight stream
//     Wp (12 chunks) | W1[0..11] | W2[hidden 0..383] (12) | W1[12..23] | W2[hidden 384..767] (12) | Wqkv of the next block (36)   = 96 x 48 KiB
// through a 3 x 48 KiB LDS ring refilled in half-chunks by LDS-DMA with counted waits; the ring never drains at a stage boundary.
// The residual stream and relu(fc1) of the 64 rows never leave the CU: a CU has 512 KiB of registers and 160 KiB of LDS (144 KiB of it the
// ring); 64 rows need 96 KiB of residual stream, 96 KiB of token fragments and -- with the MLP walked fc1[0:384] -> fc2 half 0 ->
// fc1[384:768] -> fc2 half 1 -- 96 KiB of HALF of relu(fc1).  Nothing but the launch's input (attention output fragments, the residual
// tile) and output (residual tile, q | k | v) touches memory: no atomics, no scratch round trips.  The price is the weight stream per
// token row (4.6 MB per 64 rows): L2 -> LDS by LDS-DMA delivers ~100 GB/s per CU with every CU streaming (tools/wstream_exp.hip).
//
// Eight waves on 16-TOKEN panels (4 panels x 2 k-halves), two per SIMD: a single in-order wave per SIMD is issue bound (what it issues
// besides its MFMAs adds to the MFMA time), and 8 waves x 256 registers cannot hold 64 rows on 32-token panels without every token
// fragment twice; with v_mfma_f32_16x16x32_f16 they can: per lane 48 registers of residual stream, 48 of token fragments, 48 of half of
// relu(fc1), 32 of accumulators.
//
// Layout of the 16 x 16 x 32 MFMA (tools/mfma16_layout.hip): A (16 x 32): lane l = row l % 16, k = 8 (l / 16) + j; B (32 x 16): lane l =
// column l % 16, k = 8 (l / 16) + j; D (16 x 16): lane l = column l % 16, rows 4 (l / 16) + r.  Transposed product C^T = W^T A^T: A = a
// weight fragment (16 output channels x 32 k), B = a token fragment, so lane (t = l % 16, g = l / 16) holds token t and the output
// channels 4 g + r of the 16-channel tile.  A chunk = 32 output channels = two tiles; a wave walks its K half (192) in 6 k-steps of 32:
// 12 (k-step, tile) positions kk = 2 S + mm per chunk, 3 MFMAs each (hi.hi, hi.lo, lo.hi) = 36 per chunk.  The wave pair (q, hh = 0 / 1)
// splits the contraction: wave hh finishes tile hh of every chunk (its stream holds the tiles in the order own | partner's, so that
// accumulator 0 is always the kept one) and sends the partner's tile through LDS (4 floats per lane and chunk).  The 4 values a lane
// finishes for chunk c are elements 4 (c & 1) .. + 3 of its token fragment of k-step c >> 1 of the next stage: k order
//     k (hh; S, g, j) = 32 (2 S + (j >> 2)) + 16 hh + 4 g + (j & 3)
// so LayerNorm output, ReLU output and residual stream never change lanes.  LayerNorm's affine part is folded into the Dense layer behind
// it; its statistics are the only thing tokens need from other lanes (two shuffles + LDS).  Biases come from a table in LDS (one
// ds_read_b128 per chunk: the lane's 4 channels), copied there when the stage starts.
// Every stage is the rolled loop of four-chunk bodies.  Results that stay in registers need STATIC register indices: a body always works
// on the first four chunk slots of its register array and the array is ROTATED by four slots behind every body (three rotations per
// 12-chunk stage are the identity) -- fully unrolled stages (60 bodies) sent hipcc's allocator into thousands of spills.  The epilogue of
// chunk c - 1 is sliced into the gaps behind single MFMAs of chunk c.
// Row tiles are always whole: tokens past M read row M - 1 and store into a trash page (lane-local arithmetic: a padded lane cannot
// disturb a live one).
//
// The two earlier forms of this kernel (round 5: 128 rows, residual adds as float atomics; round 6 first form: 64 rows on four waves of
// 512 registers) are kept with their harnesses under tools/ (uu3d_tchain.h, uu3d_tchain64.h); profiles/r06_ab_tchain16.txt is the A/B.
#pragma once
#include <algorithm>
#include "uu3d_gemm_panel8.h"

namespace uu3d {

enum : int {
    TC_PROJ = 1,        // starts with x += context Wp + bp (context = attention output, A-fragment order, natural k)
    TC_MLP = 2,         // LayerNorm 2, fc1, ReLU, fc2, residual
    TC_QKV = 4,         // ends with LayerNorm 1 + QKV of the NEXT block (f16 planes for attn_h3_kernel, q pre-scaled)
    TC_FC1_PLANES = 8,  // (instead of TC_MLP) LayerNorm 2, fc1 (Conv1D k = 1), ReLU -> row-major f16 planes: the first strided block, whose convolution is another kernel
    TC_PE = 16,         // in front of the final LayerNorm 1: xa = x + pe[token % period] is stored and normalised (temporal stack -> strided block 1, u_u_t.py:126-128)
};

static constexpr int TC_CHUNK_HALFS = P8_CHUNK_BYTES / 2;
__host__ __device__ inline constexpr int tchain_chunks(int flags) {
    return ((flags & TC_PROJ) ? 12 : 0) + ((flags & TC_MLP) ? 48 : 0) + ((flags & TC_FC1_PLANES) ? 24 : 0) + ((flags & TC_QKV) ? 36 : 0);
}
static constexpr size_t TC_TRASH_BYTES = 16384;

// Parameter table of one launch (floats, packed at commit time: ONE pointer instead of eight)
// (b1 and bqkv are the FOLDED biases b + beta W of the LayerNorm in front, and their W is diag(gamma) W; q's scale is folded into wq and bq)
enum : int { TCP_BP = 0, TCP_B1 = 384, TCP_B2 = 1152, TCP_BQKV = 1536, TCP_FLOATS = 2688 };

struct TChainArgs {
    int M, m_tiles, period; float qscale;
    const _Float16* Of;          // TC_PROJ: attention output [panel][24 slices][plane][lane][8]
    float* X;                    // residual stream [M][384] (row-major: read by the first launch of a forward, written by the last ones)
    float* XA; const float* pe;  // TC_PE: xa = x + pe[token % period]
    const _Float16* W;           // the launch's weight stream (tchain16_pack_stage per stage), tchain_chunks(flags) x 48 KiB
    const float* P;              // parameter table (TCP_*)
    _Float16* Q;                 // TC_QKV: q | k | v in FRAGMENT order (tchain_qf_index): [32-token panel][16-channel group 72][plane][lane][8], whole 128-row tiles
    _Float16* H;                 // TC_FC1_PLANES: hi plane [M][768], lo plane M * 768 halfs further
    unsigned char* scratch;      // tchain16_scratch_bytes(m_tiles): residual tiles (lane-linear) | the same of the first strided block (x + pe) | trash page
};

// q | k | v for attn_h3_kernel in FRAGMENT order: the 8 values of a 16-channel group (channels 16 u + 8 (j >> 2) + 4 g + (j & 3)) that belong to one
// (token, g) are ONE 16-byte piece, the pieces of a 32-token panel contiguous: the attention kernel's Q / K / V loads are contiguous pieces.  The
// channel permutation inside a group is the same for q and k (their dot product does not see it) and reaches the attention OUTPUT through v: the
// projection's weights are packed in that order.   halfs: element (token, channel) of plane p
__host__ __device__ inline size_t tchain_qf_index(size_t token, int ch, int plane) {
    const int u = ch >> 4, w = ch & 15, j = ((w >> 3) << 2) | (w & 3), g = (w >> 2) & 1;
    return ((((token >> 5) * 72 + u) * 2 + plane) * 64 + (token & 31) + 32 * g) * 8 + j;
}
__host__ __device__ inline constexpr size_t tchain_qf_halfs(int m_tiles) { return (size_t)m_tiles * 4 * 72 * 2 * 512; }      // m_tiles: 128-row tiles

static constexpr size_t T16_X_FLOATS_PER_TILE = 64 * 384;
// scratch: residual tiles (temporal stack) | residual tiles of the first strided block (x + pe) | trash page
__host__ __device__ inline constexpr size_t tchain16_scratch_bytes(int m_tiles64) {
    return (size_t)m_tiles64 * (2 * T16_X_FLOATS_PER_TILE * 4) + TC_TRASH_BYTES;
}
// The MLP's chunks in the order the kernel walks them.  (host) `in`: W1 (24 chunks) | W2 half 0 | W2 half 1, `out`: W1[0..11] | W2 half 0 | W1[12..23] | W2 half 1
inline void tchain16_reorder_mlp(const _Float16* in, _Float16* out) {
    const size_t C = TC_CHUNK_HALFS;
    std::copy(in, in + 12 * C, out);                                  // W1[0..11]
    std::copy(in + 24 * C, in + 36 * C, out + 12 * C);                // W2 half 0
    std::copy(in + 12 * C, in + 24 * C, out + 24 * C);                // W1[12..23]
    std::copy(in + 36 * C, in + 48 * C, out + 36 * C);                // W2 half 1
}

static constexpr size_t T16_XCHG_BYTES = 8 * 1024;                        // 8 waves x 64 lanes x 4 floats
static constexpr size_t T16_BIAS_BYTES = 1152 * 4;                        // the running stage's bias vector (QKV: 1152 floats)
static constexpr size_t T16_LDS_TOTAL = P8_RING_BYTES + T16_XCHG_BYTES + T16_BIAS_BYTES;   // 160256 <= 163840

// Lane-linear order of a 64-row tile of the residual stream between two launches: [chunk 12][wave = 4 hh + q][lane = t + 16 g][r 4] with
// channel = 32 c + 16 hh + 4 g + r and row = 16 q + t.
__host__ __device__ inline size_t tchain16_xs_index(int row, int ch) {
    const int tile = row >> 6, q = (row >> 4) & 3, t = row & 15;
    const int c = ch >> 5, hh = (ch >> 4) & 1, g = (ch >> 2) & 3, r = ch & 3;
    return (size_t)tile * T16_X_FLOATS_PER_TILE + (((size_t)(c * 8 + 4 * hh + q) * 64 + t + 16 * g) * 4 + r);
}

// ---- host side: one stage's chunks of the weight stream from the transposed planes Bt[n][Kp] (k contiguous; lo pre-scaled) ----
// chunk c, wave group hh, position kk = 2 S + mm, plane p, lane l = t + 16 g, element j: row n = 32 c + 16 (mm ^ hh) + t, column
//     attention output (natural): k = 16 (12 hh + 2 S + (g >> 1)) + 8 (j >> 2) + 4 (g & 1) + (j & 3)      (the 16-byte pieces of the A-fragment-ordered O)
//     lane order:                 k = kofs + 32 (2 S + (j >> 2)) + 16 hh + 4 g + (j & 3)
inline void tchain16_pack_stage(const _Float16* Bh, const _Float16* Bl, int N, int Kp, int kofs, bool natural, _Float16* out) {
    for (int c = 0; c < N / 32; ++c)
        for (int hh = 0; hh < 2; ++hh)
            for (int kk = 0; kk < 12; ++kk)
                for (int p = 0; p < 2; ++p)
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 8; ++j) {
                            const int S = kk >> 1, mm = kk & 1, t = l & 15, g = l >> 4;
                            const int n = 32 * c + 16 * (mm ^ hh) + t;
                            const int k = natural ? 16 * (12 * hh + 2 * S + (g >> 1)) + 8 * (j >> 2) + 4 * (g & 1) + (j & 3)
                                                  : kofs + 32 * (2 * S + (j >> 2)) + 16 * hh + 4 * g + (j & 3);
                            out[(((((size_t)c * 2 + hh) * 12 + kk) * 2 + p) * 64 + l) * 8 + j] = (p ? Bl : Bh)[(size_t)n * Kp + k];
                        }
}

template <bool BIAS> struct T16EpResidual { static constexpr int kStores = 0; static constexpr bool kBias = BIAS; const float* bias; int nbias; };
struct T16EpHidden { static constexpr int kStores = 0; static constexpr bool kBias = true; const float* bias; int nbias; };      // (bias: this half's 384 values)
struct T16EpQkv { static constexpr int kStores = 2; static constexpr bool kBias = true; h16x4* __restrict__ qf; const float* bias; int nbias; };
struct T16EpPlanes { static constexpr int kStores = 2; static constexpr bool kBias = true; unsigned char* __restrict__ ph; unsigned char* __restrict__ pl; const float* bias; int nbias; };

#ifndef UU3D_T16_DEPTH
#define UU3D_T16_DEPTH 2       // weight fragments are requested this many positions kk ahead of the MFMAs that use them (a ring of DEPTH + 1 register pairs)
#endif
#ifndef UU3D_T16_LOO
#define UU3D_T16_LOO 0         // tools/tchain16_exp: leave-one-out timing builds (results wrong): 1 no refill DMA, 2 no finish
#endif

template <int FLAGS>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
tchain16_kernel(const TChainArgs a)
{
    constexpr int HS = 12, KS = 6;
    constexpr int GT = tchain_chunks(FLAGS);
    static_assert(!((FLAGS & TC_MLP) && (FLAGS & TC_FC1_PLANES)), "one MLP form per launch");
    static_assert(GT >= 12, "at least one stage");
    h3_flush_f16_denormals();
    extern __shared__ __attribute__((aligned(16))) unsigned char psm[];

    const int bm = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = wave >> 2, q = wave & 3, g = lane >> 4, t = lane & 15;
    const int tok = bm * 64 + q * 16 + t;
    const bool live = tok < a.M;
    const int tokc = min(tok, a.M - 1);
    const int chl = 16 * hh + 4 * g;                       // this lane's first channel inside a 32-channel chunk

    // ---- weight stream -> ring (as the 8-wave kernels): half-chunk g2 = 2 G + j holds the positions [6 j, 6 j + 6) of both wave groups; this wave moves 3 KiB of it ----
    const unsigned wofs = (unsigned)(hh * HS * 2048 + q * 3072);
    const unsigned char* const wsrc = reinterpret_cast<const unsigned char*>(a.W) + wofs;
    const unsigned lane16 = (unsigned)lane * 16u;
    auto dma1 = [&](int Gc, int j, int slot, int i) __attribute__((always_inline)) {
        if (UU3D_T16_LOO & 1) return;
        const unsigned char* s = wsrc + (size_t)min(Gc, GT - 1) * P8_CHUNK_BYTES + j * (6 * 2048);
        unsigned char* d = psm + slot * P8_CHUNK_BYTES + wofs + j * (6 * 2048);
        switch (i) {
            case 0: __builtin_amdgcn_global_load_lds((h3_glb_void*)(s + lane16), (h3_lds_void*)d, 16, 0, 0); break;
            case 1: __builtin_amdgcn_global_load_lds((h3_glb_void*)(s + lane16), (h3_lds_void*)d, 16, 1024, 0); break;
            default: __builtin_amdgcn_global_load_lds((h3_glb_void*)(s + lane16), (h3_lds_void*)d, 16, 2048, 0); break;
        }
    };

    // ================= the lane's state =================
    f32x4 xr[12];                                          // residual stream: x[token][32 c + 16 hh + 4 g + (0..3)]
    h16x8 ah[KS], al[KS];                                  // token fragments of the running stage, one per k-step
    h16x8 fh[KS], fl[KS];                                  // one half of relu(fc1) as fc2's token fragments
    constexpr int DP = UU3D_T16_DEPTH, RB = DP + 1;
    h16x8 bh[RB] = {}, bl[RB] = {};                        // weight fragments (hi / lo plane) of DP + 1 consecutive positions kk

    auto late = [&](int v) __attribute__((always_inline)) -> int { asm volatile("" : "+s"(v)); return v; };
    const unsigned xoff = (unsigned)(wave * 256 + lane * 4);
    auto xs_tile = [&](int b, bool strided1) __attribute__((always_inline)) -> float* {      // residual tiles: temporal stack | first strided block (x + pe)
        return reinterpret_cast<float*>(a.scratch) + ((size_t)b + (strided1 ? (size_t)a.m_tiles : 0)) * T16_X_FLOATS_PER_TILE;
    };
    auto load_xs = [&](const float* tl) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < 12; ++c) xr[c] = *reinterpret_cast<const f32x4*>(tl + (xoff + (unsigned)(c * 2048)));
    };
    auto store_xs = [&](float* tl) __attribute__((always_inline)) {
        unsigned xo = xoff;
        asm volatile("" : "+v"(xo));
#pragma unroll
        for (int c = 0; c < 12; ++c) *reinterpret_cast<f32x4*>(tl + (xo + (unsigned)(c * 2048))) = xr[c];
    };
    auto load_rows = [&](const float* base) __attribute__((always_inline)) {          // (rows past M: row M - 1)
        const float* p = base + (size_t)tokc * 384 + chl;
#pragma unroll
        for (int c = 0; c < 12; ++c) xr[c] = *reinterpret_cast<const f32x4*>(p + 32 * c);
    };
    auto store_rows = [&](float* base) __attribute__((always_inline)) {               // (dead lanes: the trash page)
        const int tk = late(bm) * 64 + q * 16 + t;
        unsigned char* const tr = a.scratch + (size_t)a.m_tiles * (2 * T16_X_FLOATS_PER_TILE * 4);
        float* p = tk < a.M ? base + (size_t)tk * 384 + chl : reinterpret_cast<float*>(tr) + chl;
#pragma unroll
        for (int c = 0; c < 12; ++c) *reinterpret_cast<f32x4*>(p + 32 * c) = xr[c];
    };

    // ---- what the launch reads by name, in FRONT of the ring's first pieces ----
    constexpr bool kStrided1 = (FLAGS & TC_FC1_PLANES) != 0;
    if constexpr ((FLAGS & TC_PROJ) != 0) {
        load_xs(xs_tile(bm, kStrided1));
        // the attention output: A-fragment order [32-token panel][16-deep slice 24][plane][lane = token % 32 + 32 gk][8]; this lane's k-step S is the piece
        // (slice 12 hh + 2 S + (g >> 1), gk = g & 1) of its token
        const int row0 = min(bm * 64 + q * 16, a.M - 1);
        const h16x8* ap = reinterpret_cast<const h16x8*>(a.Of) + (size_t)(row0 >> 5) * 24 * 2 * 64 + (row0 & 16) + t + 32 * (g & 1);
#pragma unroll
        for (int S = 0; S < KS; ++S) {
            const int sl = 12 * hh + 2 * S + (g >> 1);
            ah[S] = ap[(sl * 2 + 0) * 64]; al[S] = ap[(sl * 2 + 1) * 64];
        }
    } else {
        load_rows(a.X);
    }
#pragma unroll
    for (int g2 = 0; g2 < 5; ++g2)
#pragma unroll
        for (int i = 0; i < 3; ++i) dma1(g2 >> 1, g2 & 1, g2 >> 1, i);
    int G = 0, slot = 0;                                   // next chunk of the stream to be consumed and its ring slot (G % 3)

    unsigned char* const xmine = psm + P8_RING_BYTES + wave * 1024 + lane16;
    unsigned char* const xpart = psm + P8_RING_BYTES + (wave ^ 4) * 1024 + lane16;
    float* const stat = reinterpret_cast<float*>(psm + P8_RING_BYTES);
    float* const bias_lds = reinterpret_cast<float*>(psm + P8_RING_BYTES + T16_XCHG_BYTES);
    const unsigned rd0 = (unsigned)(uintptr_t)(h3_lds_void*)(psm + hh * HS * 2048 + lane16);
    const unsigned bias_rd = (unsigned)(uintptr_t)(h3_lds_void*)(psm + P8_RING_BYTES + T16_XCHG_BYTES) + (unsigned)chl * 4u;

#define UU3D_T16_READ(i, sb, kk) \
    asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" \
                 : "=&v"(bh[i]), "=&v"(bl[i]) : "v"(sb), "i"((kk) * 2048), "i"((kk) * 2048 + 1024))

    struct Acc { f32x4 a0[2], a1[2]; };                    // [0]: this wave's tile of the chunk (kept), [1]: the partner's (sent)
    struct Fin { f32x4 u, s, r, b, y; h16x4 vh, vl; };
    struct Keep { h16x4 h, l; };                           // the even chunk's half of a hidden fragment until the odd chunk's arrives
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

    // value j (0..3) of the finished chunk: + the partner's sum + bias, the epilogue's arithmetic
    auto fin_a = [&](auto ep, auto fin_tag, const int j, Fin& f) __attribute__((always_inline)) {
        using EP = decltype(ep);
        constexpr int fs = decltype(fin_tag)::value;        // chunk slot (0..3 of the body, 11 = the previous body's last chunk, rotated once)
        float y = f.u[j] + f.r[j];
        if constexpr (EP::kBias) y += f.b[j];
        if constexpr (std::is_same<EP, T16EpResidual<false>>::value || std::is_same<EP, T16EpResidual<true>>::value) {
            xr[fs][j] += y;
        } else {
            if constexpr (std::is_same<EP, T16EpHidden>::value || std::is_same<EP, T16EpPlanes>::value) y = fmaxf(y, 0.f);
            _Float16 h; float hf;
            asm("v_cvt_f16_f32 %0, %2\n\tv_cvt_f32_f16 %1, %0" : "=&v"(h), "=v"(hf) : "v"(y));
            f.y[j] = y - hf;
            f.vh[j] = h;
        }
    };
    auto fin_b = [&](auto ep, const int j, Fin& f) __attribute__((always_inline)) {
        using EP = decltype(ep);
        if constexpr (!(std::is_same<EP, T16EpResidual<false>>::value || std::is_same<EP, T16EpResidual<true>>::value))
            f.vl[j] = (_Float16)(f.y[j] * H3_SCALE);
    };
    auto fin_store = [&](auto ep, const int cp, auto fin_tag, Fin& f, Keep& hk) __attribute__((always_inline)) {
        using EP = decltype(ep);
        constexpr int fs = decltype(fin_tag)::value;
        if constexpr (std::is_same<EP, T16EpHidden>::value) {
            // hidden chunk cp -> elements 4 (cp & 1) .. + 3 of fc2's fragment of k-step cp >> 1: the even chunk waits in `hk` for the odd one
            if constexpr ((fs & 1) == 0) { hk.h = f.vh; hk.l = f.vl; }
            else {
                h16x8 ph, pl;
#pragma unroll
                for (int e = 0; e < 4; ++e) { ph[e] = hk.h[e]; ph[4 + e] = f.vh[e]; pl[e] = hk.l[e]; pl[4 + e] = f.vl[e]; }
                constexpr int ks = fs == 11 ? 5 : fs >> 1;      // (slot 11 = the previous body's chunk 3: k-step 1 of that body, rotated by two k-steps)
                fh[ks] = ph; fl[ks] = pl;
            }
        } else if constexpr (std::is_same<EP, T16EpQkv>::value) {
            h16x4* d = ep.qf + (size_t)(2 * cp + hh) * 256;          // 16-channel group u = 2 cp + hh: 2 planes x 64 lanes x 2 pieces of 4 halfs
            d[0] = f.vh; d[128] = f.vl;
        } else if constexpr (std::is_same<EP, T16EpPlanes>::value) {
            const unsigned o = (unsigned)(32 * cp) * 2u;
            *reinterpret_cast<h16x4*>(ep.ph + o) = f.vh;
            *reinterpret_cast<h16x4*>(ep.pl + o) = f.vl;
        }
    };

    // ---- one chunk of a stage over the token fragments Ah / Al.  Vector-memory operations per half-interval in issue order: [first half] 3 pieces,
    // [second half] 3 pieces with the kStores stores of chunk c - 1 between them; the barrier that opens a half-interval needs the pieces issued four
    // half-intervals earlier.  LDS operations in issue order (fragment pairs two positions ahead): ... pair kk + 1 | [send, behind position 1] | pair kk + 2;
    // behind B'_c: receive, bias, pair 8 ... ----
    auto chunk = [&](auto cl_tag, auto pre_tag, const int c, auto fin_tag, auto ep, const h16x8 (&Ah)[KS], const h16x8 (&Al)[KS],
                     Acc& x, const Acc& p, Keep& hk, const int bias_c0) __attribute__((always_inline)) {
        constexpr int CL = decltype(cl_tag)::value;
        constexpr bool PRE_IN = (decltype(pre_tag)::value & 1) != 0, PRE_OUT = (decltype(pre_tag)::value & 2) != 0;
        using EP = decltype(ep);
        constexpr int NST = EP::kStores;
        constexpr bool FIN = CL > 0 && !(UU3D_T16_LOO & 2);
        const int pslot = slot == 0 ? 2 : slot - 1;
        const unsigned sb = rd0 + (unsigned)slot * P8_CHUNK_BYTES;
        Fin f;
        auto gapwork = [&](const int gp) __attribute__((always_inline)) {
            if constexpr (CL > 0 && (UU3D_T16_LOO & 2) != 0) { if (gp == 3) asm volatile("" :: "v"(p.a0[0]), "v"(p.a0[1]), "v"(p.a1[0]), "v"(p.a1[1])); }      // (timing builds: the previous chunk's MFMAs stay alive)
            if constexpr (FIN) {
                if (gp == 3 || gp == 4) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) { const int k = 2 * (gp - 3) + e; f.s[k] = p.a0[1][k] + p.a1[1][k] * (1.0f / H3_SCALE); }
                }
                if (gp == 5) asm volatile("ds_write_b128 %0, %1" :: "v"((unsigned)(uintptr_t)(h3_lds_void*)xmine), "v"(f.s) : "memory");
                if (gp >= 9 && gp <= 12) f.u[gp - 9] = p.a0[0][gp - 9] + p.a1[0][gp - 9] * (1.0f / H3_SCALE);
                if (gp >= 22 && gp <= 25) fin_b(ep, gp - 22, f);
                if (gp >= 21 && gp <= 24) fin_a(ep, fin_tag, gp - 21, f);
                if (gp == 27) fin_store(ep, c - 1, fin_tag, f, hk);
            }
        };
        // ---- barrier B_c: half-chunk 2 c + 1 landed (own pieces) ----
        if constexpr (PRE_IN) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(9 + NST * ((CL >= 2) + (CL >= 3))) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(9 + NST * ((CL >= 2) + (CL >= 3))) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!PRE_IN) {
#pragma unroll
            for (int d = 0; d < DP; ++d) UU3D_T16_READ(d, sb, d);
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) { x.a0[m] = f32x4{0.f, 0.f, 0.f, 0.f}; x.a1[m] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#define UU3D_T16_KK(kk) \
            x.a0[(kk) & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[(kk) % RB], Ah[(kk) >> 1], x.a0[(kk) & 1], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0); gapwork(3 * (kk)); __builtin_amdgcn_sched_barrier(0); \
            x.a1[(kk) & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[(kk) % RB], Al[(kk) >> 1], x.a1[(kk) & 1], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0); gapwork(3 * (kk) + 1); __builtin_amdgcn_sched_barrier(0); \
            x.a1[(kk) & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[(kk) % RB], Ah[(kk) >> 1], x.a1[(kk) & 1], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0); gapwork(3 * (kk) + 2); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 6; ++kk) {
            UU3D_T16_READ((kk + DP) % RB, sb, kk + DP);
            // younger than pair kk: pairs kk + 1 .. kk + DP, and the send (one write, behind position 1) while pair kk was requested in front of it (2 <= kk <= 1 + DP)
            asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(bh[kk % RB]), "+v"(bl[kk % RB]) : "i"(2 * DP + (FIN && kk >= 2 && kk <= 1 + DP ? 1 : 0)));
            UU3D_T16_KK(kk)
            if ((kk & 1) == 0) dma1(G + 2, 1, pslot, kk >> 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- barrier B'_c: half-chunk 2 c + 2 landed; everybody read the first halves of chunk c and wrote the exchange area ----
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(%1)" :: "i"(9 + NST * (CL >= 2)), "i"(2 * DP) : "memory");      // (pairs 6 .. 5 + DP stay in flight; the send is older)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        constexpr int NRB = FIN ? (EP::kBias ? 2 : 1) : 0;   // LDS reads behind B'_c in front of pair 8: the receive (+ the bias)
        if constexpr (FIN) {
            asm volatile("ds_read_b128 %0, %1" : "=&v"(f.r) : "v"((unsigned)(uintptr_t)(h3_lds_void*)xpart) : "memory");
            if constexpr (EP::kBias) asm volatile("ds_read_b128 %0, %1" : "=&v"(f.b) : "v"(bias_rd + (unsigned)(bias_c0 + c - 1) * 128u) : "memory");
        }
#pragma unroll
        for (int kk = 6; kk < HS; ++kk) {
            if (kk + DP < HS) UU3D_T16_READ((kk + DP) % RB, sb, kk + DP);
            // younger than pair kk: the pairs kk + 1 .. min(kk + DP, 11), and the NRB reads behind B'_c while pair kk was requested in front of it (kk <= 5 + DP)
            const int ahead = 2 * ((kk + DP < HS ? kk + DP : HS - 1) - kk);
            if (FIN && kk == 7) {                          // the finish starts behind this wait: the receive and the bias must be there; younger than them: the pairs requested behind B'_c
                const int behind = 6 + DP < HS ? 2 * ((7 + DP < HS ? 7 + DP : HS - 1) - (6 + DP) + 1) : 0;
                if constexpr (EP::kBias) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(bh[kk % RB]), "+v"(bl[kk % RB]), "+v"(f.r), "+v"(f.b) : "i"(behind < ahead ? behind : ahead));
                else asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(bh[kk % RB]), "+v"(bl[kk % RB]), "+v"(f.r) : "i"(behind < ahead ? behind : ahead));
            }
            else asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(bh[kk % RB]), "+v"(bl[kk % RB]) : "i"(ahead + (FIN && kk == 6 ? NRB : 0)));
            UU3D_T16_KK(kk)
            if ((kk & 1) == 0) dma1(G + 3, 0, slot, (kk - 6) >> 1);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef UU3D_T16_KK
        G += 1;
        slot = slot == 2 ? 0 : slot + 1;
        if constexpr (PRE_OUT) {                           // the next chunk's first fragments: its first half landed one barrier ago
            const unsigned nb = rd0 + (unsigned)slot * P8_CHUNK_BYTES;
#pragma unroll
            for (int d = 0; d < DP; ++d) UU3D_T16_READ(d, nb, d);
        }
    };

    // ---- the last chunk of a stage: send, barrier, receive, finish; leaves the stage drained ----
    auto stage_tail = [&](auto ep, const int c, auto fin_tag, const Acc& p, Keep& hk, const int bias_c0) __attribute__((always_inline)) {
        using EP = decltype(ep);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // everybody's reads of the exchange area (chunk c - 1) returned
        Fin f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { f.s[e] = p.a0[1][e] + p.a1[1][e] * (1.0f / H3_SCALE); f.u[e] = p.a0[0][e] + p.a1[0][e] * (1.0f / H3_SCALE); }
        asm volatile("ds_write_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"((unsigned)(uintptr_t)(h3_lds_void*)xmine), "v"(f.s) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("ds_read_b128 %0, %1" : "=&v"(f.r) : "v"((unsigned)(uintptr_t)(h3_lds_void*)xpart) : "memory");
        if constexpr (EP::kBias) asm volatile("ds_read_b128 %0, %1" : "=&v"(f.b) : "v"(bias_rd + (unsigned)(bias_c0 + c) * 128u) : "memory");
        else f.b = f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.r), "+v"(f.b));
#pragma unroll
        for (int j = 0; j < 4; ++j) { fin_a(ep, fin_tag, j, f); fin_b(ep, j, f); }
        fin_store(ep, c, fin_tag, f, hk);
        __builtin_amdgcn_s_barrier();                      // the exchange area and the bias table are free again
    };

    // ---- a stage of NCH chunks: bodies of four chunks; the register array the epilogue writes rotates behind every body (NCH = 12: three rotations = the
    // identity): xr by four chunk slots, the hidden fragments by two k-steps.  The chunk finished inside body chunk j sits in slot j - 1, the previous
    // body's last chunk in slot 11. ----
    using F0 = std::integral_constant<int, 0>; using F1 = std::integral_constant<int, 1>; using F2 = std::integral_constant<int, 2>; using F11 = std::integral_constant<int, 11>;
    auto stage = [&](auto nch_tag, auto ep, const h16x8 (&Ah)[KS], const h16x8 (&Al)[KS], const int bias_c0) __attribute__((always_inline)) {
        constexpr int NCH = decltype(nch_tag)::value;
        using EP = decltype(ep);
        static_assert(NCH % 4 == 0 && NCH >= 8, "bodies of four chunks");
        // the stage's bias vector -> LDS (the previous stage's tail ended with a barrier; the first read is three barriers away)
        if constexpr (EP::kBias) {
            for (int i = tid; i < ep.nbias / 4; i += 512) reinterpret_cast<f32x4*>(bias_lds)[i] = reinterpret_cast<const f32x4*>(ep.bias)[i];
        }
        auto rotate = [&]() __attribute__((always_inline)) {
            if constexpr (std::is_same<EP, T16EpResidual<false>>::value || std::is_same<EP, T16EpResidual<true>>::value) {
                static_assert(NCH == 12, "three rotations");
#pragma unroll
                for (int s = 0; s < 4; ++s) { const f32x4 tt = xr[s]; xr[s] = xr[s + 4]; xr[s + 4] = xr[s + 8]; xr[s + 8] = tt; __builtin_amdgcn_sched_barrier(0); }
            } else if constexpr (std::is_same<EP, T16EpHidden>::value) {
                static_assert(NCH == 12, "three rotations");
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const h16x8 th = fh[s], tl = fl[s];
                    fh[s] = fh[s + 2]; fl[s] = fl[s + 2]; fh[s + 2] = fh[s + 4]; fl[s + 2] = fl[s + 4]; fh[s + 4] = th; fl[s + 4] = tl;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        Acc xa, xb;
#pragma unroll
        for (int m = 0; m < 2; ++m) { xb.a0[m] = f32x4{0.f, 0.f, 0.f, 0.f}; xb.a1[m] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        Keep hk{};
        chunk(I0{}, I2{}, 0, F0{}, ep, Ah, Al, xa, xb, hk, bias_c0);
        chunk(I1{}, I3{}, 1, F0{}, ep, Ah, Al, xb, xa, hk, bias_c0);
        chunk(I2{}, I3{}, 2, F1{}, ep, Ah, Al, xa, xb, hk, bias_c0);
        chunk(I3{}, I1{}, 3, F2{}, ep, Ah, Al, xb, xa, hk, bias_c0);
        rotate();
#pragma unroll 1
        for (int c = 4; c < NCH; c += 4) {
            chunk(I3{}, I2{}, c, F11{}, ep, Ah, Al, xa, xb, hk, bias_c0);
            chunk(I3{}, I3{}, c + 1, F0{}, ep, Ah, Al, xb, xa, hk, bias_c0);
            chunk(I3{}, I3{}, c + 2, F1{}, ep, Ah, Al, xa, xb, hk, bias_c0);
            chunk(I3{}, I1{}, c + 3, F2{}, ep, Ah, Al, xb, xa, hk, bias_c0);
            rotate();
        }
        stage_tail(ep, (int)(NCH - 1), F11{}, xb, hk, bias_c0);
    };
    using N12 = std::integral_constant<int, 12>;

    // sum over the token's 384 channels: this lane's 48 + the lanes g = 1 .. 3 of the token + the partner wave
    auto token_sum = [&](float s, int phase) __attribute__((always_inline)) -> float {
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (g == 0) stat[phase * 128 + wave * 16 + t] = s;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        return s + stat[phase * 128 + (wave ^ 4) * 16 + t];
    };
    // LayerNorm (two-pass, eps inside the root) of xr WITHOUT its affine part (folded into the Dense layer behind it) -> ah / al
    auto layer_norm = [&]() __attribute__((always_inline)) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 12; ++c) s += (xr[c][0] + xr[c][1]) + (xr[c][2] + xr[c][3]);
        const float mean = token_sum(s, 0) * (1.0f / 384.0f);
        float v = 0.f;
#pragma unroll
        for (int c = 0; c < 12; ++c) {
            const f32x4 d = xr[c] - mean;
            v += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
        const float rstd = 1.0f / sqrtf(token_sum(v, 1) * (1.0f / 384.0f) + 1e-5f);
#pragma unroll
        for (int S = 0; S < KS; ++S) {
            h16x4 hi[2], lo[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) h3_split((xr[2 * S + i] - mean) * rstd, hi[i], lo[i]);
#pragma unroll
            for (int e = 0; e < 4; ++e) { ah[S][e] = hi[0][e]; ah[S][4 + e] = hi[1][e]; al[S][e] = lo[0][e]; al[S][4 + e] = lo[1][e]; }
        }
        // (the statistics sit in the exchange area: the next send is in chunk 1 of the next stage, three workgroup barriers after every wave has read them)
    };

    // ================= the chain =================
    if constexpr ((FLAGS & TC_PROJ) != 0) {
        stage(N12{}, T16EpResidual<true>{a.P + TCP_BP, 384}, ah, al, 0);
        if constexpr (kStrided1) store_rows(a.XA);                            // (the strided convolution's residual rows, EpConvResidual)
    } else {
        store_xs(xs_tile(bm, false));                                         // (the first launch: the residual stream enters the chain's order)
    }
    if constexpr ((FLAGS & (TC_MLP | TC_FC1_PLANES)) != 0) {
        layer_norm();
        if constexpr (kStrided1) {
            unsigned char* const trash = a.scratch + (size_t)a.m_tiles * (2 * T16_X_FLOATS_PER_TILE * 4);
            unsigned char* ph = live ? reinterpret_cast<unsigned char*>(a.H + (size_t)tok * 768 + chl) : trash;
            unsigned char* pl = live ? reinterpret_cast<unsigned char*>(a.H + ((size_t)a.M + tok) * 768 + chl) : trash + 4096;
            stage(std::integral_constant<int, 24>{}, T16EpPlanes{ph, pl, a.P + TCP_B1, 768}, ah, al, 0);
        } else {
            stage(N12{}, T16EpHidden{a.P + TCP_B1, 384}, ah, al, 0);                    // relu(fc1)[0..383]
            stage(N12{}, T16EpResidual<false>{nullptr, 0}, fh, fl, 0);                 // x += it . W2[0..383]
            stage(N12{}, T16EpHidden{a.P + TCP_B1 + 384, 384}, ah, al, 0);              // relu(fc1)[384..767]
            stage(N12{}, T16EpResidual<true>{a.P + TCP_B2, 384}, fh, fl, 0);           // x += it . W2[384..767] + b2
            if constexpr ((FLAGS & TC_QKV) == 0 || (FLAGS & TC_PE) != 0) store_rows(a.X);        // the temporal stack's result: head1 (and head2 without strided blocks) read it
        }
    }
    if constexpr ((FLAGS & TC_QKV) != 0) {
        if constexpr ((FLAGS & TC_PE) != 0) {
            const int tkc = min(late(bm) * 64 + q * 16 + t, a.M - 1);
            const float* pr = a.pe + (size_t)(tkc % a.period) * 384 + chl;
#pragma unroll
            for (int c = 0; c < 12; ++c) xr[c] = xr[c] + *reinterpret_cast<const f32x4*>(pr + 32 * c);
            store_xs(xs_tile(late(bm), true));                                 // the stream of the first strided block's launch
        } else if constexpr ((FLAGS & (TC_PROJ | TC_MLP)) != 0) {
            store_xs(xs_tile(late(bm), false));                                // the next temporal launch's residual tile
        }
        layer_norm();
        // q | k | v in fragment order (tchain_qf_index): this lane's 4 channels of 16-channel group u = 2 c + hh are halfs 4 (g >> 1) .. + 3 of the 16-byte piece
        // of lane (token % 32) + 32 (g & 1): one 8-byte store per plane and chunk
        const int lb = late(bm);
        h16x4* const qf = reinterpret_cast<h16x4*>(a.Q) + ((size_t)(lb * 2 + (q >> 1)) * (72 * 2 * 64) + (q & 1) * 16 + t + 32 * (g & 1)) * 2 + (g >> 1);
        stage(std::integral_constant<int, 36>{}, T16EpQkv{qf, a.P + TCP_BQKV, 1152}, ah, al, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the clamped tail pieces must not outlive the LDS allocation)
#undef UU3D_T16_READ
}

}  // namespace uu3d
