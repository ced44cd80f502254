// uu3d_spatial_h3.h -- the spatial (per-frame joint) transformer stack with f16x3 matrix products.
//
// Same function as spatial_stack_mfma_kernel (uu3d_spatial.h; reference u_u_t.py:313-330 and
// vision_transformer.py:71-195 at d = 32): keypoint embedding + PE, 4 pre-LN blocks of 8-head attention over the
// 17 joints of a frame and a GELU MLP 32 -> 64 -> 32, spatial_norm, one launch, one workgroup = 3 frames = 51 tokens.
// Four changes of structure:
//
// * Products are f16x3 (uu3d_gemm_h3.h): x ~= hi + lo / 2048, three v_mfma_f32_32x32x16_f16 per 16-deep k-step.
//   A K = 32 product of a 32 x 32 tile costs 6 MFMAs x 32 cycles instead of 16 x 64 cycles of 32x32x2_f32.
//
// * Every product is computed TRANSPOSED: C^T[n][m] = W^T[n][k] X^T[k][m].  The A operand is the weight (fragment
//   ordered f16 planes, one coalesced 1 KiB load per fragment straight from L2), the B operand is the activation
//   row of token m (16 bytes of a row-major LDS tile).  In the C/D map of the 32x32 MFMA a lane then holds, for
//   ITS token m = lane & 31 (+ 32 per m-tile), the 16 channels n = 8g + 4 (lane >> 5) + e (g, e = 0..3): groups of
//   four consecutive channels.  With d_h = 4 a group is exactly one head, so
//     - the residual stream lives in that layout for the whole kernel (16 channels of a token per lane; the two
//       lanes l and l + 32 share a token), GEMM results add to it in registers -- no C tile round trip through LDS;
//     - q stays in registers, lane (l, half) runs heads {half, 2 + half, 4 + half, 6 + half} of its token(s);
//     - LayerNorm needs one cross-lane add (lane ^ 32) per moment;
//     - everything a lane writes to LDS (LayerNorm output, attention output, GELU output as f16 planes, K and V
//       as f32) is a group of 4 consecutive channels: one 8- or 16-byte store.
//
// * Round 2, occupancy: the kernel waits for LATENCY (one wave alone on a SIMD needs 62 us for 3 frames, two sharing it 81 us
//   each), so a workgroup is TWO waves that share the 3 frames and the LDS tiles, wave w owning token tile w (template
//   parameter MT = 1; MT = 2 is the one-wave form with two tokens per lane): half the registers (152: three waves per SIMD
//   instead of 1.5), half the dependent chain per wave, three s_barriers per block where K / V cross the waves.  164 -> 135 us.
//
// * Round 2: the kernel is VALU bound (19 k VALU instructions per wave against 384 MFMAs; SQ counters in
//   profiles/r02_final_sq_summary.csv), so its elementwise work runs on PACKED f32 instructions (v_pk_fma_f32 /
//   v_pk_mul_f32 / v_pk_add_f32: two floats per lane and instruction) written by name:
//     - a lane's 16 channels of a token live as 8 register PAIRS of consecutive channels -- the layout the MFMA result
//       registers already have -- so LayerNorm, bias / residual adds, the hi / lo split, GELU and the acc0 + acc1 / 2048
//       combine are pair operations; per-token scalars (mean, rstd, maximum) are splatted into a pair once;
//     - attention pairs KEYS instead: K and V sit in LDS as [frame][key pair][channel][2], so (logit j, logit j + 1) =
//       sum_c (q_c, q_c) * (k_jc, k_j+1,c) and the P V sums run over even / odd keys in the two halves;
//     - none of these instructions carries op_sel / op_sel_hi: the form that loses an operand next to a busy matrix pipe
//       (docs/HISTORY.md E.12) reads the OTHER half of a register pair, which only those modifiers do.  hipcc itself
//       still never emits packed f32 (target feature off); tests/test_isa_cpu.py pins both facts.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <math.h>
#include <type_traits>
#include "uu3d_gemm_h3.h"
#include "uu3d_spatial.h"
#include "uu3d_pk.h"

#ifndef UU3D_SP_SKIP
#define UU3D_SP_SKIP 0      // tools/spatial_stamp_exp: leave a phase out (1 attention, 2 GELU, 3 LayerNorms, 4 parameter copy after block 0, 5 hi/lo splits) to time it by difference
#endif

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
constexpr int TOK = 27;          // token slots per 32-lane tile: 3 frames x 9 joints (tile 0: joints 0..8, tile 1: joints 9..16 + one spare)
constexpr int ROWS_T = 2 * TOK + 1;   // rows of the LDS operand tiles: 54 token slots + one dummy row for lanes 27..31
__device__ __forceinline__ int tile_row(const int mt, const int tl) { return tl < TOK ? TOK * mt + tl : ROWS_T - 1; }
constexpr int XLD = 40;          // halfs per row of the K = 32 operand tile (80 B: conflict-free 16-byte reads)
constexpr int HLD = 72;          // halfs per row of the K = 64 hidden tile (144 B)
constexpr int KLD = 36;          // floats per token row the K / V region is sized by (the hidden planes reuse it)
constexpr int KPLD = 68;         // floats per key pair of a frame: [32 channels][2 keys] + 4 (272 B: 16-byte aligned, pairs 4 banks apart)
#ifndef UU3D_KFPAD
#define UU3D_KFPAD 8
#endif
constexpr int KFPAD = UU3D_KFPAD; // floats between the key-pair images of two frames: frame stride 9 * 68 + 8 = 620 = 44 banks -- the 9 lanes of a frame
                                 // store one channel to banks 4 jp + parity; the three frames of a wave then land on disjoint banks (at 612 = 36 banks
                                 // frames 0 and 2 shared five of nine: two-way conflicts on every K / V store)
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
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

// C^T tiles of W^T X^T for NT output tiles (32 channels each) and the wave's MT token tiles (tiles mt0 .. mt0 + MT - 1);
// K = 16 * KK.  out[nt][mt][i] = the pair of registers (2 i, 2 i + 1): token slot (lane & 31) of tile mt0 + mt, channels
// 32 nt + 8 (i >> 1) + 4 (lane >> 5) + 2 (i & 1) and the next one.
template <int MT, int NT, int KK>
__device__ __forceinline__ void mm(const WFrag<NT, KK>& w, const _Float16* Bh, const _Float16* Bl, const int ldb,
                                   const int lane, const int mt0, f32x2 (&out)[NT][MT][8]) {
    f32x16 acc0[NT][MT], acc1[NT][MT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[nt][mt][r] = 0.f; acc1[nt][mt][r] = 0.f; }
    const int tl = lane & 31, half = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        h16x8 bh[MT], bl[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int row = tile_row(mt0 + mt, tl);
            bh[mt] = *reinterpret_cast<const h16x8*>(Bh + row * ldb + 16 * kk + 8 * half);
            bl[mt] = *reinterpret_cast<const h16x8*>(Bl + row * ldb + 16 * kk + 8 * half);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                acc0[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.h[nt][kk], bh[mt], acc0[nt][mt], 0, 0, 0);
                acc1[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.h[nt][kk], bl[mt], acc1[nt][mt], 0, 0, 0);
                acc1[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.l[nt][kk], bh[mt], acc1[nt][mt], 0, 0, 0);
            }
    }
    // The MFMA -> VALU read hazard is software's to cover (8-pass MFMA: 11 wait states, a 16-pass one 19), and hipcc does not
    // cover it for inline asm readers: without these 20 wait states the packed ops below read the accumulators too early (NaN
    // in every output).  The operands tie the asm between the MFMAs and every reader: volatile asm statements keep their order,
    // so the empty ones (and with them the readers of their operands) stay behind the wait.
    pk::mfma_fence(acc0[NT - 1][MT - 1], acc1[NT - 1][MT - 1]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            if (nt != NT - 1 || mt != MT - 1) pk::behind_fence(acc0[nt][mt], acc1[nt][mt]);
    const f32x2 inv = pk::splat(1.0f / H3_SCALE);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                out[nt][mt][i] = pk::fma((f32x2){acc1[nt][mt][2 * i], acc1[nt][mt][2 * i + 1]}, inv,
                                         (f32x2){acc0[nt][mt][2 * i], acc0[nt][mt][2 * i + 1]});
}

// v[l] + v[l ^ 32] with v = p.x + p.y, in every lane: one v_permlane32_swap (lanes 32..63 of the first operand <-> lanes 0..31 of the second)
// instead of a ds_bpermute round trip through LDS
// The pair's own two halves are added by name: left to hipcc, "p.x + p.y, twice" (the swap consumes two copies) becomes ONE
// v_pk_add_f32 with op_sel:[0,1] op_sel_hi:[1,0] -- the cross-half form this kernel must not contain (docs/HISTORY.md E.12).
__device__ __forceinline__ float sum_halves(const f32x2 p) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    float v;
    asm("v_add_f32 %0, %1, %2\n\ts_nop 1" : "=v"(v) : "v"(p[0]), "v"(p[1]));      // + wait states hipcc would put between a VALU write and the lane swap reading it
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// LayerNormalization over the 32 channels of each of the lane's two tokens (16 here, 16 in lane ^ 32); the arithmetic of
// ln_row (non-fused Keras path: inv = rstd * gamma, y = x * inv + (beta - mean * inv)) on channel pairs, the two moments summed
// over even / odd channels first
template <int MT>
__device__ __forceinline__ void ln_tokens(const f32x2 (&x)[MT][8], const float* g, const float* b,
                                          const float eps, const int half, f32x2 (&y)[MT][8], float2* stats = nullptr) {
    f32x2 gp[8], bp[8];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(g + 8 * gq + 4 * half);
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(b + 8 * gq + 4 * half);
        gp[2 * gq] = (f32x2){g4[0], g4[1]}; gp[2 * gq + 1] = (f32x2){g4[2], g4[3]};
        bp[2 * gq] = (f32x2){b4[0], b4[1]}; bp[2 * gq + 1] = (f32x2){b4[2], b4[3]};
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const f32x2 s2 = pk::add(pk::add(pk::add(x[mt][0], x[mt][1]), pk::add(x[mt][2], x[mt][3])),
                                 pk::add(pk::add(x[mt][4], x[mt][5]), pk::add(x[mt][6], x[mt][7])));
        const float s = sum_halves(s2);
        const float mean = s * (1.0f / 32.0f);
        const f32x2 m2 = pk::splat(mean);
        f32x2 q2;
        { const f32x2 d = pk::sub(x[mt][0], m2); q2 = pk::mul(d, d); }
#pragma unroll
        for (int i = 1; i < 8; ++i) { const f32x2 d = pk::sub(x[mt][i], m2); q2 = pk::fma(d, d, q2); }
        const float q = sum_halves(q2);
        const float rstd = 1.0f / sqrtf(q * (1.0f / 32.0f) + eps);
        if (stats != nullptr) stats[mt] = make_float2(mean, rstd);     // (training: the backward pass reads the row statistics)
        const f32x2 r2 = pk::splat(rstd);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x2 inv = pk::mul(r2, gp[i]);
            y[mt][i] = pk::fma(x[mt][i], inv, pk::fnma(m2, inv, bp[i]));
        }
    }
}

// two consecutive channel pairs -> 4 hi and 4 lo halfs (h3_split's arithmetic; the lo conversion may keep f16 denormals,
// which the MFMA reads as zero either way)
__device__ __forceinline__ void split_pairs(const f32x2 a, const f32x2 b, h16x4& hi, h16x4& lo) {
    if constexpr (UU3D_SP_SKIP == 5) { hi = (h16x4){(_Float16)a[0], (_Float16)a[1], (_Float16)b[0], (_Float16)b[1]}; lo = hi; return; }
    const _Float16 h0 = h3_hi(a[0]), h1 = h3_hi(a[1]), h2 = h3_hi(b[0]), h3 = h3_hi(b[1]);
    const f32x2 sc = pk::splat(H3_SCALE);
    const f32x2 ra = pk::mul(pk::sub(a, (f32x2){(float)h0, (float)h1}), sc);
    const f32x2 rb = pk::mul(pk::sub(b, (f32x2){(float)h2, (float)h3}), sc);
    h16x2 la, lb;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(la) : "v"(ra[0]), "v"(ra[1]));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lb) : "v"(rb[0]), "v"(rb[1]));
    hi = (h16x4){h0, h1, h2, h3};
    lo = (h16x4){la[0], la[1], lb[0], lb[1]};
}

// the lane's 2 x 16 values -> hi / lo planes of a row-major tile (row = token, 4-channel groups of 8 bytes)
template <int MT>
__device__ __forceinline__ void store_planes(_Float16* Th, _Float16* Tl, const int ld, const int coloff, const int lane, const int mt0,
                                             const f32x2 (&v)[MT][8]) {
    const int tl = lane & 31, half = lane >> 5;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int row = tile_row(mt0 + mt, tl);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            h16x4 hi, lo;
            split_pairs(v[mt][2 * g], v[mt][2 * g + 1], hi, lo);
            *reinterpret_cast<h16x4*>(Th + row * ld + coloff + 8 * g + 4 * half) = hi;
            *reinterpret_cast<h16x4*>(Tl + row * ld + coloff + 8 * g + 4 * half) = lo;
        }
    }
}

// One head of the lane's NTOK tokens (two: same frame, see the kernel): softmax(q k^T / sqrt(d_h)) v over the J keys of the frame.
// ka / va = LDS byte addresses of the head's 4 channels in key pair 0 of the frame; key pair jp is KPLD floats further and
// holds, per channel c, (key 2 jp, key 2 jp + 1): two ds_read_b128 = channels (c0, c0 + 1) and (c0 + 2, c0 + 3).  Every K / V
// register feeds both tokens.  J odd: the last pair's second key is a finite dummy whose probability is forced to zero.
template <int J, int NTOK>
__device__ __forceinline__ void head_attention(const f32x2 (&qa)[NTOK], const f32x2 (&qb)[NTOK], const unsigned ka, const unsigned va,
                                               f32x2 (&oa)[NTOK], f32x2 (&ob)[NTOK]) {
    static_assert(J == 17, "9 key pairs, the last one half empty");
    constexpr int NP = 9;
    // softmax(x) with x = q.k / 2: exp(x - max) = exp2((q * log2e / 2).k - max'), so the scale and the base change are
    // folded into q once; the 1 / sum normalisation is applied to the 4 outputs instead of the 17 probabilities
    const f32x2 c = pk::splat(0.72134752044448170368f);
    f32x2 q0[NTOK], q1[NTOK], q2[NTOK], q3[NTOK];
#pragma unroll
    for (int t = 0; t < NTOK; ++t) {
        const f32x2 sa = pk::mul(qa[t], c), sb = pk::mul(qb[t], c);
        q0[t] = pk::splat(sa[0]); q1[t] = pk::splat(sa[1]); q2[t] = pk::splat(sb[0]); q3[t] = pk::splat(sb[1]);
    }
    // Key and value pairs by name in two batches each (pairs 0..4: 10 reads, pairs 5..8: 8 reads) through the same 10
    // registers, each batch in flight at once and waited for as a whole (the "memory" clobbers keep other LDS operations out of
    // the counter, see uu3d_attn.h); hipcc emits read -> wait -> use for every row otherwise.  The second batch goes out when
    // the first is used up: its latency is what the other waves of the SIMD are for (all 18 in flight cost 32 more registers --
    // with one token tile per wave, the difference between three waves per SIMD and spilling).
    f32x4 kv[10];
#define UU3D_SP_W10(x) "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9])
#define UU3D_SP_W8(x) "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
    auto read_pairs = [&](auto first_tag, const unsigned base) __attribute__((always_inline)) {
        constexpr int P0 = decltype(first_tag)::value, N = P0 == 0 ? 10 : 8;
#pragma unroll
        for (int j = 0; j < N; ++j)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kv[j]) : "v"(base), "i"((P0 + (j >> 1)) * KPLD * 4 + (j & 1) * 16) : "memory");
        if constexpr (P0 == 0) asm volatile("s_waitcnt lgkmcnt(0)" : UU3D_SP_W10(kv) :: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" : UU3D_SP_W8(kv) :: "memory");
    };
    f32x2 d[NTOK][NP];
    auto logits = [&](int jp, int r) __attribute__((always_inline)) {       // the scalar kernel's order: q0 k0, then fma over channels 1..3
        const f32x2 k0 = {kv[2 * r][0], kv[2 * r][1]}, k1 = {kv[2 * r][2], kv[2 * r][3]};
        const f32x2 k2 = {kv[2 * r + 1][0], kv[2 * r + 1][1]}, k3 = {kv[2 * r + 1][2], kv[2 * r + 1][3]};
#pragma unroll
        for (int t = 0; t < NTOK; ++t) d[t][jp] = pk::fma(q3[t], k3, pk::fma(q2[t], k2, pk::fma(q1[t], k1, pk::mul(q0[t], k0))));
    };
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    read_pairs(std::integral_constant<int, 0>{}, ka);
#pragma unroll
    for (int jp = 0; jp < 5; ++jp) logits(jp, jp);
    read_pairs(std::integral_constant<int, 5>{}, ka);
#pragma unroll
    for (int jp = 5; jp < NP; ++jp) logits(jp, jp - 5);
    // first value batch: issued here, waited for after the softmax arithmetic
#pragma unroll
    for (int j = 0; j < 10; ++j)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kv[j]) : "v"(va), "i"((j >> 1) * KPLD * 4 + (j & 1) * 16) : "memory");
    f32x2 sum2[NTOK];
#pragma unroll
    for (int t = 0; t < NTOK; ++t) {
        float mx = d[t][NP - 1][0];                    // v_max3 by name: fmaxf() first canonicalises every asm result (17 extra v_max per head)
#pragma unroll
        for (int jp = 0; jp < NP - 1; ++jp) asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(d[t][jp][0]), "v"(d[t][jp][1]));
        const f32x2 m2 = pk::splat(mx);
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) {
            const f32x2 u = pk::sub(d[t][jp], m2);
            d[t][jp][0] = __builtin_amdgcn_exp2f(u[0]);
            d[t][jp][1] = jp < NP - 1 ? __builtin_amdgcn_exp2f(u[1]) : 0.f;
        }
        asm volatile("s_nop 0" : "+v"(d[t][0]), "+v"(d[t][1]), "+v"(d[t][2]), "+v"(d[t][3]), "+v"(d[t][4]), "+v"(d[t][5]), "+v"(d[t][6]), "+v"(d[t][7]), "+v"(d[t][8]));   // pk::fence
        sum2[t] = pk::add(pk::add(pk::add(d[t][0], d[t][1]), pk::add(d[t][2], d[t][3])), pk::add(pk::add(pk::add(d[t][4], d[t][5]), pk::add(d[t][6], d[t][7])), d[t][8]));
    }
    f32x2 o[NTOK][4];
    auto pv = [&](int jp, int r) __attribute__((always_inline)) {
        const f32x2 v0 = {kv[2 * r][0], kv[2 * r][1]}, v1 = {kv[2 * r][2], kv[2 * r][3]};
        const f32x2 v2 = {kv[2 * r + 1][0], kv[2 * r + 1][1]}, v3 = {kv[2 * r + 1][2], kv[2 * r + 1][3]};
#pragma unroll
        for (int t = 0; t < NTOK; ++t) {
            if (jp == 0) { o[t][0] = pk::mul(d[t][0], v0); o[t][1] = pk::mul(d[t][0], v1); o[t][2] = pk::mul(d[t][0], v2); o[t][3] = pk::mul(d[t][0], v3); }
            else { o[t][0] = pk::fma(d[t][jp], v0, o[t][0]); o[t][1] = pk::fma(d[t][jp], v1, o[t][1]); o[t][2] = pk::fma(d[t][jp], v2, o[t][2]); o[t][3] = pk::fma(d[t][jp], v3, o[t][3]); }
        }
    };
    asm volatile("s_waitcnt lgkmcnt(0)" : UU3D_SP_W10(kv) :: "memory");
#pragma unroll
    for (int jp = 0; jp < 5; ++jp) pv(jp, jp);
    read_pairs(std::integral_constant<int, 5>{}, va);
#pragma unroll
    for (int jp = 5; jp < NP; ++jp) pv(jp, jp - 5);
#undef UU3D_SP_W10
#undef UU3D_SP_W8
#pragma unroll
    for (int t = 0; t < NTOK; ++t) {
        f32x2 rs = pk::splat(__builtin_amdgcn_rcpf(sum2[t][0] + sum2[t][1]));     // 1 ulp; a correctly rounded quotient costs 10 instructions per head
        pk::fence(rs);
        oa[t] = pk::mul((f32x2){o[t][0][0] + o[t][0][1], o[t][1][0] + o[t][1][1]}, rs);
        ob[t] = pk::mul((f32x2){o[t][2][0] + o[t][2][1], o[t][3][0] + o[t][3][1]}, rs);
    }
}

// GELU 0.5 x (1 + erf(x / sqrt 2)) of a channel pair, erf from Abramowitz-Stegun 7.1.28: erf(z) = 1 - (1 + a1 z + ... + a6 z^6)^-16,
// |error| <= 3e-7; measured on [-8, 8] in f32: GELU abs error <= 8.8e-7 (7.1.26 as in sv2::gelu_erf: 4.7e-7).  One transcendental
// (v_rcp) per element instead of two (v_rcp + v_exp, a quarter of the VALU rate each); everything else is packed.  The powers of
// 1 / sqrt 2 are folded into the coefficients; 0.5 x (1 + sign(x) E) is evaluated as 0.5 (x + |x| E).
__device__ __forceinline__ f32x2 gelu_pair(const f32x2 x) {
    const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
    f32x2 P = pk::fma(pk::splat(5.3829750000e-06f), ax, pk::splat(4.8890635643e-05f));
    P = pk::fma(P, ax, pk::splat(3.8003575000e-05f)); P = pk::fma(P, ax, pk::splat(3.2776263241e-03f));
    P = pk::fma(P, ax, pk::splat(2.1141006150e-02f)); P = pk::fma(P, ax, pk::splat(4.9867346967e-02f));
    P = pk::fma(P, ax, pk::splat(1.0f));
    f32x2 r = {__builtin_amdgcn_rcpf(P[0]), __builtin_amdgcn_rcpf(P[1])};
    pk::fence(r);
    r = pk::mul(r, r); r = pk::mul(r, r); r = pk::mul(r, r); r = pk::mul(r, r);
    const f32x2 E = pk::sub(pk::splat(1.0f), r);                                          // erf(|x| / sqrt 2)
    return pk::mul(pk::fma(ax, E, x), pk::splat(0.5f));
}
}  // namespace sh3

#ifdef UU3D_SPATIAL_STAMP
__device__ unsigned long long spatial_clk[12];   // tools/spatial_stamp_exp: s_memtime ticks per phase, summed over waves and blocks; [11] = waves
#define SP_STAMP(i) { const long long t_ = clock64(); sp_t[i] += (unsigned)(t_ - sp_last); sp_last = t_; }
#else
#define SP_STAMP(i)
#endif
// out_lo == nullptr: out is the f32 (frames, J, 32) tensor; otherwise out / out_lo are its two f16 planes
#ifndef UU3D_SPATIAL_H3_WAVES
#define UU3D_SPATIAL_H3_WAVES 2     // 3 (168 VGPRs) spills into the block loop: 0.30 ms instead of 0.20
#endif
// (UU3D_PK_TARGET: uu3d_pk.h -- this kernel switches the packed-fp32-ops target feature back on for itself)
// Training-mode forward (TRAIN = true; uu3d_train_step.inc): the same kernel also writes what the backward pass reads -- per block
// its input, both LayerNorms' row statistics (mean, 1 / sqrt(var + eps)), q | k | v with bias, the attention output, the stream
// after the attention residual and the pre-GELU hidden activations; at the end the stack's output before spatial_norm and that
// LayerNorm's statistics -- and applies the DropPath gates of vision_transformer.py:16-43 (per frame, scaled by 1 / keep) to the
// two residual branches.  One launch instead of the 31 of the unfused chain (generic GEMMs with K = 32 / 64 at 77 k rows).
struct SpatialTrainIO {
    float* X[9]; float2* St1[8]; float* QKV[8]; float* O[8]; float* Xmid[8]; float2* St2[8]; float* Hpre[8]; float2* StF;
    const float* gate1[8]; const float* gate2[8]; float inv_keep[8];      // gate == nullptr: no DropPath in that block
};

// MT = token tiles per wave.  MT = 2: one wave per workgroup runs both tiles of its 3 frames (two tokens per lane).  MT = 1: a
// workgroup of TWO waves shares the frames and the LDS tiles, wave w owns tile w (one token per lane): half the registers per
// wave (three waves per SIMD instead of 1.5), half the dependent instruction chain, and s_barriers where the waves exchange
// K / V through LDS.
template <int J, int FR, int MT, bool TRAIN = false>
__global__ void __launch_bounds__(64 * (2 / MT), MT == 2 ? UU3D_SPATIAL_H3_WAVES : 3) UU3D_PK_TARGET
spatial_stack_h3_kernel(const float* __restrict__ kp2d, const SpatialParams p, const _Float16* __restrict__ wfrag,
                        float* __restrict__ out, _Float16* __restrict__ out_hi, _Float16* __restrict__ out_lo, const SpatialTrainIO tio = SpatialTrainIO{})
{
    // The two token tiles run the same arithmetic, fully unrolled (MT = 2) or in two waves; the copies must round
    // identically or a frame's result depends on its slot in the wave (bitwise permutation test).  Contraction is off
    // and every intended FMA is an fmaf() / a packed fma by name; the f16 conversions go through h3_hi (one instruction form).
#pragma clang fp contract(off)
    using namespace sh3;
    static_assert(MT == 1 || MT == 2, "one or two token tiles per wave");
    h3_flush_f16_denormals();
    constexpr int DS = 32, HS = 64;
    static_assert(FR * ((J + 1) / 2) == TOK && DS == 32 && HS == 64, "one workgroup = 3 frames of 17 joints, d = 32");
    using LY = SpatialBlockLayoutV2<DS, HS>;          // LayerNorm parameters and biases (f32) come from the V2 block
    using FL = SpatialFragLayoutH3;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    _Float16* Xh = reinterpret_cast<_Float16*>(lds_raw);               // [55][40] operand tile, hi
    _Float16* Xl = Xh + ROWS_T * XLD;                                   // lo
    float* TK = reinterpret_cast<float*>(Xl + ROWS_T * XLD);           // K: [3 frames][9 key pairs][68] floats inside a [55][36] region
    float* TV = TK + ROWS_T * KLD;                                      // V likewise
    _Float16* Hh = reinterpret_cast<_Float16*>(TK);                    // [55][72] GELU(fc1), hi (K / V are dead by then)
    _Float16* Hl = Hh + ROWS_T * HLD;
    float* P = TV + ROWS_T * KLD;                                       // [352] this block's LayerNorm parameters and biases

    const int lane = threadIdx.x & 63, tl = lane & 31, half = lane >> 5;
    const int mt0 = MT == 2 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);    // first (only) token tile of this wave
    // every wave of the workgroup is done with the LDS tiles the next phase overwrites (MT = 1; own LDS operations retired first)
    auto wg_sync = [&]() __attribute__((always_inline)) {
        if constexpr (MT == 1) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    };
    int nframes = p.total_frames;
    if (p.frame_list != nullptr) {
        nframes = p.frame_list[p.total_frames];
        if ((int)blockIdx.x * FR >= nframes) return;
    }
    // Token -> lane: lane tl of token tile t holds joint (tl % JH) + JH * t of frame tl / JH of this workgroup (JH = 9 joints
    // per frame and tile; lanes 27..31 and the tenth joint of tile 1 are padding).  With MT = 2 a lane's two tokens then belong
    // to the SAME frame: the attention below loads every K / V register once for both.
    constexpr int JH = (J + 1) / 2;
    static_assert(FR * JH == TOK, "the frames of a workgroup side by side in one token tile");
    int frame[MT], joint[MT];
    bool valid[MT], real[MT];
    const int fl = min(tl / JH, FR - 1);                // frame of this lane inside the workgroup
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int jn = (tl - fl * JH) + JH * (mt0 + mt);
        real[mt] = (tl < TOK) && (jn < J);
        joint[mt] = min(jn, J - 1);
        int f = blockIdx.x * FR + fl;
        valid[mt] = real[mt] && (f < nframes);
        if (p.frame_list != nullptr) f = p.frame_list[min(f, nframes - 1)];
        frame[mt] = min(f, p.total_frames - 1);
    }

    // K / V slots of this lane's two tokens in the key-pair layout (floats from TK / TV): [frame][key pair][channel][2]; padding
    // tokens go to the spare half of the last pair of their frame (never read with a non-zero probability)
    constexpr int NP = (J + 1) / 2;
    constexpr int KFLD = NP * KPLD + KFPAD;               // floats per frame
    static_assert((J & 1) == 1 && FR * KFLD <= ROWS_T * KLD && KFLD % 4 == 0, "odd J: one spare key slot per frame; fits the K / V region; 16-byte aligned frames");
    int kslot[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int jp = real[mt] ? (joint[mt] >> 1) : NP - 1, par = real[mt] ? (joint[mt] & 1) : 1;
        kslot[mt] = fl * KFLD + jp * KPLD + par;
    }
    const unsigned kfr = (unsigned)(fl * KFLD * 4);        // LDS byte offset of the lane's frame inside TK / TV
    const unsigned tk_a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)TK;
    const unsigned tv_a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)TV;

    // keypoint embedding + spatial PE (u_u_t.py:321-323), this lane's 16 channels of each token as 8 pairs.  Packed by name
    // like everything else: left as scalar code, hipcc's SLP pass pairs it up itself -- with op_sel broadcasts of (kx, ky).
    f32x2 x[MT][8];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        float kx = 0.f, ky = 0.f;
        if (valid[mt]) { const float2 k2 = *reinterpret_cast<const float2*>(kp2d + ((size_t)frame[mt] * J + joint[mt]) * 2); kx = k2.x; ky = k2.y; }
        const f32x2 kx2 = pk::splat(kx), ky2 = pk::splat(ky);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = 8 * (i >> 1) + 4 * half + 2 * (i & 1);
            const f32x2 w0 = *reinterpret_cast<const f32x2*>(p.embed_w + c), w1 = *reinterpret_cast<const f32x2*>(p.embed_w + DS + c);
            const f32x2 eb = *reinterpret_cast<const f32x2*>(p.embed_b + c), pe = *reinterpret_cast<const f32x2*>(p.pe + joint[mt] * DS + c);
            x[mt][i] = pk::add(pk::add(pk::fma(ky2, w1, pk::mul(kx2, w0)), eb), pe);
        }
    }
    // bias pairs of this lane's channels: param[8 g + 4 half + 2 e .. + 2]
    auto bias_pairs = [&](const float* b, f32x2 (&bp)[8]) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(b + 8 * g + 4 * half);
            bp[2 * g] = (f32x2){b4[0], b4[1]}; bp[2 * g + 1] = (f32x2){b4[2], b4[3]};
        }
    };

    // TRAIN: this lane's 16 channels of its token(s) to row (frame * J + joint) of a (rows, ldt) tensor, 4 channels per store
    size_t trow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) trow[mt] = (size_t)frame[mt] * J + joint[mt];
    auto save16 = [&](float* T, const int ldt, const int coloff, const f32x2 (&v)[MT][8]) __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (!valid[mt]) continue;
            float* d = T + trow[mt] * ldt + coloff + 4 * half;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f32x4*>(d + 8 * g) = (f32x4){v[mt][2 * g][0], v[mt][2 * g][1], v[mt][2 * g + 1][0], v[mt][2 * g + 1][1]};
        }
    };
    auto save_stats = [&](float2* T, const float2 (&st)[MT]) __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) if (valid[mt] && half == 0) T[trow[mt]] = st[mt];
    };
    // DropPath scale of a residual branch: gate[frame] / keep (1 without the layer)
    auto gate_pairs = [&](const float* gate, const float inv_keep, f32x2 (&sc)[MT]) __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) sc[mt] = pk::splat(gate != nullptr ? gate[frame[mt]] * inv_keep : 1.0f);
    };
#ifdef UU3D_SPATIAL_STAMP
    unsigned sp_t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long sp_last = clock64();
    const long long sp_first = sp_last;
#endif
    for (int blk = 0; blk < p.depth; ++blk) {
        const _Float16* __restrict__ F = wfrag + (size_t)blk * FL::size;
        wg_sync();                                         // the other wave may still read the previous block's parameters and hidden planes
        {   // one coalesced copy (per wave: each reads back what it wrote itself) of the block's 352 parameters into LDS: the 16-byte group reads below then cost an LDS
            // round trip instead of a dependent global load each (133 of them per block before)
            const float* __restrict__ Wg = p.blocks + (size_t)blk * LY::size;
            if (UU3D_SP_SKIP != 4 || blk == 0)
#pragma unroll
            for (int i = 0; i < (NPARAM + 63) / 64; ++i) { const int k = 64 * i + lane; if (k < NPARAM) P[k] = Wg[k]; }
        }
        const float* W = P;
        f32x2 y[MT][8];

        // ---- attention half ----
        WFrag<1, 2> wq, wk, wv, wp;
        load_w<1, 2>(F + FL::fq, lane, wq); load_w<1, 2>(F + FL::fk, lane, wk); load_w<1, 2>(F + FL::fv, lane, wv);
        float2 lnst[MT];
        if constexpr (TRAIN) save16(tio.X[blk], DS, 0, x);
        if constexpr (UU3D_SP_SKIP == 3) { for (int mt = 0; mt < MT; ++mt) for (int i = 0; i < 8; ++i) y[mt][i] = x[mt][i]; } else
        ln_tokens(x, W + LY::ln1_g, W + LY::ln1_b, 1e-5f, half, y, TRAIN ? lnst : nullptr);
        if constexpr (TRAIN) save_stats(tio.St1[blk], lnst);
        store_planes<MT>(Xh, Xl, XLD, 0, lane, mt0, y);
        SP_STAMP(0)
        f32x2 q[1][MT][8];
        {
            f32x2 kv[1][MT][8], bp[8];
            wait_w<8>(wq);
            mm<MT, 1, 2>(wq, Xh, Xl, XLD, lane, mt0, q);
            bias_pairs(W + LY::bq, bp);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 8; ++i) q[0][mt][i] = pk::add(q[0][mt][i], bp[i]);
            if constexpr (TRAIN) save16(tio.QKV[blk], 3 * DS, 0, q[0]);
            wait_w<4>(wk);
            mm<MT, 1, 2>(wk, Xh, Xl, XLD, lane, mt0, kv);
            // the spare key slot of every frame: finite (zero) whatever the hidden planes of the previous block left there
            if (lane < 2 * DS) { const int c = lane & 31; float* T = lane < DS ? TK : TV;
#pragma unroll
                for (int f = 0; f < FR; ++f) T[f * KFLD + (NP - 1) * KPLD + 2 * c + 1] = 0.f; }
            bias_pairs(W + LY::bk, bp);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const f32x2 v = pk::add(kv[0][mt][i], bp[i]);
                    const int c = 8 * (i >> 1) + 4 * half + 2 * (i & 1);
                    TK[kslot[mt] + 2 * c] = v[0]; TK[kslot[mt] + 2 * c + 2] = v[1];
                    if constexpr (TRAIN) kv[0][mt][i] = v;
                }
            if constexpr (TRAIN) save16(tio.QKV[blk], 3 * DS, DS, kv[0]);
            load_w<1, 2>(F + FL::fp, lane, wp);            // in flight over the attention arithmetic
            wait_w<4>(wv);
            mm<MT, 1, 2>(wv, Xh, Xl, XLD, lane, mt0, kv);
            bias_pairs(W + LY::bv, bp);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const f32x2 v = pk::add(kv[0][mt][i], bp[i]);
                    const int c = 8 * (i >> 1) + 4 * half + 2 * (i & 1);
                    TV[kslot[mt] + 2 * c] = v[0]; TV[kslot[mt] + 2 * c + 2] = v[1];
                    if constexpr (TRAIN) kv[0][mt][i] = v;
                }
            if constexpr (TRAIN) save16(tio.QKV[blk], 3 * DS, 2 * DS, kv[0]);
        }
        SP_STAMP(1)
        wg_sync();                                         // K / V of every token of the frames are in LDS
        // scaled dot-product attention over the J joints of the lane's frame, the lane's tokens at once; group g = head 2g + half
        f32x2 o[MT][8];
        {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const unsigned off = kfr + (unsigned)((8 * g + 4 * half) * 2 * 4);
                f32x2 qa[MT], qb[MT], oa[MT], ob[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) { qa[mt] = q[0][mt][2 * g]; qb[mt] = q[0][mt][2 * g + 1]; }
                if constexpr (UU3D_SP_SKIP == 1) { for (int mt = 0; mt < MT; ++mt) { oa[mt] = qa[mt]; ob[mt] = qb[mt]; } (void)off; (void)tk_a; (void)tv_a; }
                else head_attention<J, MT>(qa, qb, tk_a + off, tv_a + off, oa, ob);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) { o[mt][2 * g] = oa[mt]; o[mt][2 * g + 1] = ob[mt]; }
            }
        }
        wg_sync();                                         // nobody reads K / V any more: the hidden planes may overwrite them
        SP_STAMP(2)
        if constexpr (TRAIN) save16(tio.O[blk], DS, 0, o);
        store_planes<MT>(Xh, Xl, XLD, 0, lane, mt0, o);
        {
            f32x2 pr[1][MT][8], bp[8], sc[MT];
            wait_w<0>(wp);
            mm<MT, 1, 2>(wp, Xh, Xl, XLD, lane, mt0, pr);
            bias_pairs(W + LY::bp, bp);
            if constexpr (TRAIN) gate_pairs(tio.gate1[blk], tio.inv_keep[blk], sc);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if constexpr (TRAIN) x[mt][i] = pk::fma(pk::add(pr[0][mt][i], bp[i]), sc[mt], x[mt][i]);
                    else x[mt][i] = pk::add(x[mt][i], pk::add(pr[0][mt][i], bp[i]));
                }
        }
        if constexpr (TRAIN) save16(tio.Xmid[blk], DS, 0, x);

        SP_STAMP(3)
        // ---- MLP half ----
        WFrag<2, 2> w1;
        load_w<2, 2>(F + FL::f1, lane, w1);
        if constexpr (UU3D_SP_SKIP == 3) { for (int mt = 0; mt < MT; ++mt) for (int i = 0; i < 8; ++i) y[mt][i] = x[mt][i]; } else
        ln_tokens(x, W + LY::ln2_g, W + LY::ln2_b, 1e-5f, half, y, TRAIN ? lnst : nullptr);
        if constexpr (TRAIN) save_stats(tio.St2[blk], lnst);
        store_planes<MT>(Xh, Xl, XLD, 0, lane, mt0, y);
        SP_STAMP(4)
        WFrag<1, 4> w2;
        {
            f32x2 hd[2][MT][8], bp[8];
            wait_w<0>(w1);
            mm<MT, 2, 2>(w1, Xh, Xl, XLD, lane, mt0, hd);
            load_w<1, 4>(F + FL::f2, lane, w2);             // in flight over the GELU
            SP_STAMP(5)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                bias_pairs(W + LY::b1 + 32 * nt, bp);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int i = 0; i < 8; ++i) hd[nt][mt][i] = pk::add(hd[nt][mt][i], bp[i]);
                if constexpr (TRAIN) save16(tio.Hpre[blk], HS, 32 * nt, hd[nt]);      // pre-GELU: the backward pass differentiates the exact GELU
                if constexpr (UU3D_SP_SKIP != 2) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int i = 0; i < 8; ++i) hd[nt][mt][i] = gelu_pair(hd[nt][mt][i]);
                }
                store_planes<MT>(Hh, Hl, HLD, 32 * nt, lane, mt0, hd[nt]);
            }
        }
        SP_STAMP(6)
        {
            f32x2 z[1][MT][8], bp[8], sc[MT];
            wait_w<0>(w2);
            mm<MT, 1, 4>(w2, Hh, Hl, HLD, lane, mt0, z);
            bias_pairs(W + LY::b2, bp);
            if constexpr (TRAIN) gate_pairs(tio.gate2[blk], tio.inv_keep[blk], sc);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if constexpr (TRAIN) x[mt][i] = pk::fma(pk::add(z[0][mt][i], bp[i]), sc[mt], x[mt][i]);
                    else x[mt][i] = pk::add(x[mt][i], pk::add(z[0][mt][i], bp[i]));
                }
        }
        SP_STAMP(7)
    }
#ifdef UU3D_SPATIAL_STAMP
    if (lane == 0) {
        for (int i = 0; i < 8; ++i) atomicAdd(&spatial_clk[i], (unsigned long long)sp_t[i]);
        atomicAdd(&spatial_clk[10], (unsigned long long)(clock64() - sp_first));
        atomicAdd(&spatial_clk[11], 1ull);
    }
#endif

    f32x2 y[MT][8];
    float2 lnstf[MT];
    if constexpr (TRAIN) save16(tio.X[p.depth], DS, 0, x);
    sh3::ln_tokens(x, p.norm_g, p.norm_b, 1e-6f, half, y, TRAIN ? lnstf : nullptr);
    if constexpr (TRAIN) save_stats(tio.StF, lnstf);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (!valid[mt]) continue;
        const size_t at = ((size_t)frame[mt] * J + joint[mt]) * DS;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = 8 * g + 4 * half;
            if (out_lo != nullptr) {
                h16x4 hi, lo;
                split_pairs(y[mt][2 * g], y[mt][2 * g + 1], hi, lo);
                *reinterpret_cast<h16x4*>(out_hi + at + c) = hi;
                *reinterpret_cast<h16x4*>(out_lo + at + c) = lo;
            } else {
                *reinterpret_cast<f32x4*>(out + at + c) = (f32x4){y[mt][2 * g][0], y[mt][2 * g][1], y[mt][2 * g + 1][0], y[mt][2 * g + 1][1]};
            }
        }
    }
}

}  // namespace uu3d
