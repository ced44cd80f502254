// uu3d_gemm_h3.h -- "f16x3" GEMM: f32-grade products on the f16-rate MFMA pipe.
//
// Every f32 operand x is split into two halves  x ~= hi + lo / 2048  with
//     hi = f16(x)                (round to nearest, 11 significant bits)
//     lo = f16((x - hi) * 2048)  (x - hi is exact in f32; the 2^11 scale keeps lo in the normal range)
// and a product is taken as     a*b ~= ah*bh + (ah*bl + al*bh) / 2048      (al*bl ~ 2^-22 is dropped)
// with three v_mfma_f32_32x32x16_f16 per 16-deep k-step, accumulating in f32: acc0 += ah*bh,
// acc1 += ah*bl + al*bh, result = acc0 + acc1 / 2048.  Each operand is represented to ~2^-22 relative,
// f16 x f16 products are exact in f32, so the error is that of an f32 GEMM (measured on the whole
// model: 1.0-1.4e-5 max-abs against float64, the exact-f32 path gives 0.9-1.2e-5) while the matrix pipe
// works at 3 * (2 cycles per k) instead of 32 cycles per k of v_mfma_f32_32x32x2_f32.
//
// Weights are split once at commit time into two f16 planes Bh / Bl ([Np][Kp], lo pre-scaled);
// activations are split while they are staged into LDS (after the LayerNorm / gather transform of
// the A loader).  LDS holds four f16 planes per stage (A hi/lo, B hi/lo), rows of 32 halfs padded to
// 40 (80 B): the 16 lanes of every ds_read_b128 group fall on 16 distinct 16-byte slots.
// Workgroup = 256 threads = 4 waves as 2 x 2, wave tile (32 TM) x (32 TN).
#pragma once
#include <cstdlib>
#include <type_traits>
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "uu3d_gemm.h"

namespace uu3d {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

static constexpr int H3_LD = 40;          // halfs per LDS row (32 + 8 pad)
static constexpr float H3_SCALE = 2048.0f;

__host__ __device__ inline constexpr size_t gemm_h3_lds_bytes(int BM, int BN) {
    return (size_t)2 /*stages*/ * 2 /*planes*/ * (BM + BN) * H3_LD * sizeof(_Float16);
}

// v_mfma_*_f16 reads f16 DENORMAL inputs as zero (measured: with a plain hi = f16(x) the spatial stack, whose GELU
// outputs are full of values below the smallest normal half 2^-14 = 6.1e-5, was off by up to 3e-4 on single tokens).
// A value below that threshold must therefore have hi = 0 and go entirely into lo = x * 2048 (normal down to 3e-8).
// On the device that costs nothing: every kernel that splits first switches the wave's f16 denormal mode to
// flush (h3_flush_f16_denormals), so v_cvt_f16_f32 itself returns 0 there.  The host (weights) compares.
__device__ __forceinline__ void h3_flush_f16_denormals() {
    // s_setreg_imm32_b32 hwreg(HW_REG_MODE, 6, 2), 0 : MODE.FP_DENORM[3:2] (f16 / f64 denormals) = flush in and out
    __builtin_amdgcn_s_setreg((1 /*MODE*/) | (6 << 6) | ((2 - 1) << 11), 0);
}
__host__ __device__ inline _Float16 h3_hi(const float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    // v_cvt_f16_f32 by name: it flushes denormal results under h3_flush_f16_denormals(); hipcc otherwise mixes it
    // with v_cvt_pk_f16_f32, which keeps them -- two unrolled copies of the same code then disagree by an ulp
    _Float16 h;
    asm("v_cvt_f16_f32 %0, %1" : "=v"(h) : "v"(x));
    return h;
#else
    return ((x < 0.f ? -x : x) < 6.103515625e-05f) ? (_Float16)0.f : (_Float16)x;
#endif
}
__device__ __forceinline__ void h3_split(const f32x4 x, h16x4& hi, h16x4& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const _Float16 h = h3_hi(x[e]);
        hi[e] = h;
        lo[e] = (_Float16)((x[e] - (float)h) * H3_SCALE);   // a denormal lo is read as 0 by the MFMA either way: any cvt form will do
    }
}

inline bool gemm_h3_deep(int workgroups) {
    static const int limit = getenv("UU3D_GEMM_DEEP_WGS") ? atoi(getenv("UU3D_GEMM_DEEP_WGS")) : 640;      // 0: never
    return workgroups <= limit;
}

template <int TM, int TN, class AL, class EP, int DEEP = 0>
__global__ void __launch_bounds__(256)
gemm_h3_kernel(const AL al, const _Float16* __restrict__ Bh, const _Float16* __restrict__ Bl, const int M, const int N,
               const int Kp, const int m_tiles, const int n_tiles, const int kt_per_split, const EP ep)
{
    h3_flush_f16_denormals();
    constexpr int BM = 64 * TM, BN = 64 * TN, LD = H3_LD;
    constexpr int AI = BM / 32;                 // f32x4 staging loads per thread per k-tile (A)
    constexpr int BI = BN / 64;                 // 16-byte staging loads per thread per plane per k-tile (B)
    extern __shared__ __attribute__((aligned(16))) _Float16 hsm[];
    constexpr int STAGE = 2 * (BM + BN) * LD;   // halfs per stage
    // stage layout: Ah [BM][LD] | Al [BM][LD] | Bh [BN][LD] | Bl [BN][LD]

    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int bn = slot % n_tiles;
    const int bm = (slot / n_tiles) * 8 + xcd;
    if (bm >= m_tiles) return;
    const int bm0 = bm * BM, bn0 = bn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int arow = tid >> 3, acol = (tid & 7) * 4;          // A staging: 32 rows x 8 float4 per pass
    const int brow = tid >> 2, bcol = (tid & 3) * 8;          // B staging: 64 rows x 4 x (8 halfs) per pass

    typename AL::Ctx actx[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) actx[i] = al.prep(bm0 + arow + 32 * i);
    const _Float16* bhp[BI]; const _Float16* blp[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const size_t o = (size_t)(bn0 + brow + 64 * i) * Kp + bcol;
        bhp[i] = Bh + o; blp[i] = Bl + o;
    }

    // Global loads run NS - 1 k-tiles ahead of their use, in NS register sets.  DEEP (NS = 4) is for launches of at most ~2
    // workgroups per CU: a k-tile is 12 MFMAs = 0.2 us of work per wave and such a workgroup's whole life is its k-loop, which one
    // tile ahead ran at the latency of a global load per k-tile (~1 us; 27.7 -> 24.5 us for the 432-workgroup input-gradient GEMMs
    // of the training step).  With more workgroups per CU the other workgroups hide that latency and the 48 extra registers
    // only cost occupancy (1208 workgroups: 12.8 -> 14.1 us), hence the launcher's choice (gemm_h3_deep).
    constexpr int NS = (DEEP && sizeof(typename AL::Raw) <= 20) ? 4 : 2;
    typename AL::Raw ra[NS][AI];
    h16x8 rbh[NS][BI], rbl[NS][BI];
    f32x16 acc0[TM][TN], acc1[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }

    const int kt_lo = blockIdx.y * kt_per_split;
    const int KT = min(Kp / GEMM_BK, kt_lo + kt_per_split);

    auto issue = [&](auto set, int kt) {                // kt is clamped by the caller: a tile past the end re-reads the last one
        constexpr int Q = decltype(set)::value;
        const int k0 = kt * GEMM_BK;
#pragma unroll
        for (int i = 0; i < AI; ++i) ra[Q][i] = al.issue(actx[i], k0 + acol);
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            rbh[Q][i] = *reinterpret_cast<const h16x8*>(bhp[i] + k0);
            rbl[Q][i] = *reinterpret_cast<const h16x8*>(blp[i] + k0);
        }
    };
    auto stage = [&](auto set, int kt, int buf) {
        constexpr int Q = decltype(set)::value;
        _Float16* S = hsm + buf * STAGE;
        const int k0 = kt * GEMM_BK;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const f32x4 x = al.finish(actx[i], k0 + acol, ra[Q][i]);
            h16x4 hi, lo;
            h3_split(x, hi, lo);
            *reinterpret_cast<h16x4*>(&S[(arow + 32 * i) * LD + acol]) = hi;
            *reinterpret_cast<h16x4*>(&S[BM * LD + (arow + 32 * i) * LD + acol]) = lo;
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            *reinterpret_cast<h16x8*>(&S[2 * BM * LD + (brow + 64 * i) * LD + bcol]) = rbh[Q][i];
            *reinterpret_cast<h16x8*>(&S[2 * BM * LD + BN * LD + (brow + 64 * i) * LD + bcol]) = rbl[Q][i];
        }
    };
    auto set_c = [](auto q) { return std::integral_constant<int, decltype(q)::value % NS>{}; };

    // tiles kt_lo .. kt_lo + NS - 2 go out first (tile t lives in set t % NS), the first one is staged
    issue(std::integral_constant<int, 0>{}, kt_lo);
    if constexpr (NS > 2) {
        issue(std::integral_constant<int, 1>{}, min(kt_lo + 1, KT - 1));
        issue(std::integral_constant<int, 2>{}, min(kt_lo + 2, KT - 1));
    }
    stage(std::integral_constant<int, 0>{}, kt_lo, kt_lo & 1);
    __syncthreads();

    const int fr = lane & 31, fk = (lane >> 5) * 8;
    // one k-tile: tile kt (set Q) is in LDS buffer kt & 1; the set of tile kt - 1 is free and receives tile kt + NS - 1
    auto body = [&](auto q, int kt) {
        const int cur = kt & 1;
        issue(set_c(std::integral_constant<int, decltype(q)::value + NS - 1>{}), min(kt + NS - 1, KT - 1));
        const _Float16* S = hsm + cur * STAGE;
        const _Float16* Ahp = S + (wm * (BM / 2) + fr) * LD + fk;
        const _Float16* Alp = Ahp + BM * LD;
        const _Float16* Bhp = S + 2 * BM * LD + (wn * (BN / 2) + fr) * LD + fk;
        const _Float16* Blp = Bhp + BN * LD;
#pragma unroll
        for (int kk = 0; kk < GEMM_BK / 16; ++kk) {
            h16x8 ah[TM], alo[TM], bh[TN], blo[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = *reinterpret_cast<const h16x8*>(Ahp + i * 32 * LD + kk * 16);
                alo[i] = *reinterpret_cast<const h16x8*>(Alp + i * 32 * LD + kk * 16);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = *reinterpret_cast<const h16x8*>(Bhp + j * 32 * LD + kk * 16);
                blo[j] = *reinterpret_cast<const h16x8*>(Blp + j * 32 * LD + kk * 16);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc0[i][j], 0, 0, 0);
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], blo[j], acc1[i][j], 0, 0, 0);
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[i], bh[j], acc1[i][j], 0, 0, 0);
                }
        }
        stage(set_c(std::integral_constant<int, decltype(q)::value + 1>{}), min(kt + 1, KT - 1), cur ^ 1);
        __syncthreads();
    };
    for (int kt = kt_lo; kt < KT; kt += NS) {
        body(std::integral_constant<int, 0>{}, kt);
        if (kt + 1 < KT) body(std::integral_constant<int, 1>{}, kt + 1);
        if constexpr (NS > 2) {
            if (kt + 2 < KT) body(std::integral_constant<int, 2>{}, kt + 2);
            if (kt + 3 < KT) body(std::integral_constant<int, 3>{}, kt + 3);
        }
    }

    const int crow0 = bm0 + wm * (BM / 2) + 4 * (lane >> 5);
    const int ccol0 = bn0 + wn * (BN / 2) + (lane & 31);
    const bool interior = (bm0 + BM <= M) && (bn0 + BN <= N);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = ccol0 + j * 32;
            if (interior) {
                const float2 cv = ep.colv(col);
                float2 pr[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) pr[r] = ep.pre(crow0 + i * 32 + (r & 3) + 8 * (r >> 2), col);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ep.store(crow0 + i * 32 + (r & 3) + 8 * (r >> 2), col,
                             acc0[i][j][r] + acc1[i][j][r] * (1.0f / H3_SCALE), cv, pr[r]);
            } else if (col < N) {
                const float2 cv = ep.colv(col);
                float2 pr[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) pr[r] = ep.pre(min(crow0 + i * 32 + (r & 3) + 8 * (r >> 2), M - 1), col);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = crow0 + i * 32 + (r & 3) + 8 * (r >> 2);
                    if (row < M) ep.store(row, col, acc0[i][j][r] + acc1[i][j][r] * (1.0f / H3_SCALE), cv, pr[r]);
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------
// Pre-split A operand.  The on-the-fly kernel above repeats the hi/lo split of one A tile in every workgroup
// along N and carries that VALU work in its main loop.  Here the producer of an activation (attention, the ReLU
// epilogue, optionally a LayerNorm pass) writes the two f16 planes once -- the same 4 bytes per element as the
// f32 tensor they replace -- and the GEMM main loop is loads -> LDS -> MFMA with no arithmetic at all.
// Measured per launch at M = 4544 rows: fc2 (K = 768) 23.7 -> 16.8 us, projection 13.0 -> 11.0 us, strided
// conv (K = 2304, M = 1472) 53.2 -> 24.6 us; LayerNorm-fed GEMMs stay on the on-the-fly loader by default
// because a separate LayerNorm+split pass costs what the faster GEMM saves.

// y = bias + acc, ReLU, stored as the two planes the next GEMM reads (fc1 -> fc2 / conv).
struct EpBiasReluSplit {
    _Float16* __restrict__ Oh; _Float16* __restrict__ Ol; const float* __restrict__ bias; int ldo;
    __device__ __forceinline__ float2 colv(int col) const { return make_float2(bias[col], 0.f); }
    __device__ __forceinline__ float2 pre(int, int) const { return make_float2(0.f, 0.f); }
    __device__ __forceinline__ void store(int row, int col, float acc, float2 cv, float2) const {
        const float v = fmaxf(acc + cv.x, 0.f);
        const _Float16 h = h3_hi(v);
        Oh[(size_t)row * ldo + col] = h;
        Ol[(size_t)row * ldo + col] = (_Float16)((v - (float)h) * H3_SCALE);
    }
};

// y = acc + bias, the first `qcols` columns (q of a [q | k | v] projection) times qscale = log2(e) / sqrt(d_h), stored as the two
// planes attn_h3_kernel reads (uu3d_attn_h3.h: no splitting and no scaling left for the attention kernel to do).
struct EpBiasSplitQ {
    _Float16* __restrict__ Oh; _Float16* __restrict__ Ol; const float* __restrict__ bias; int ldo, qcols; float qscale;
    __device__ __forceinline__ float2 colv(int col) const { return make_float2(bias[col], col < qcols ? qscale : 1.0f); }
    __device__ __forceinline__ float2 pre(int, int) const { return make_float2(0.f, 0.f); }
    __device__ __forceinline__ void store(int row, int col, float acc, float2 cv, float2) const {
        const float v = (acc + cv.x) * cv.y;
        const _Float16 h = h3_hi(v);
        Oh[(size_t)row * ldo + col] = h;
        Ol[(size_t)row * ldo + col] = (_Float16)((v - (float)h) * H3_SCALE);
    }
};

// ------------------------------------------------------------------------------------------------
// The pre-split GEMM, staged by LDS-DMA: the four planes of a k-tile go global -> LDS with
// global_load_lds_dwordx4 (no VGPR staging, no ds_write pass), three LDS buffers, tile kt+2 in flight
// across the per-k-tile barrier (counted s_waitcnt vmcnt, raw s_barrier).  The same loop with register
// staging (tools/gemm_planes_regstage_exp.h) measured 3-10 % slower.
// LDS image per stage: [A hi BM rows | A lo | B hi BN rows | B lo], rows of 32 halfs = 64 B with NO padding;
// the 16-byte chunk c of row r lives at chunk c ^ ((r >> 2) & 3).  One DMA instruction writes
// wave-base + lane * 16, i.e. 16 rows x 4 chunks, so the swizzle is applied to each lane's SOURCE chunk and
// again on the fragment read; both the DMA image (4 rows x 4 chunks per 16 lanes) and the fragment reads
// (16 rows, one chunk) then touch 16 distinct 16-byte slots of the 256-byte bank row.
typedef __attribute__((address_space(3))) void h3_lds_void;
typedef __attribute__((address_space(1))) const void h3_glb_void;

struct GLoadPlain {      // A[M][lda] as two planes
    const _Float16* __restrict__ Ah; const _Float16* __restrict__ Al; int lda, M;
    struct Ctx { size_t off; };
    __device__ __forceinline__ Ctx prep(int row) const { Ctx c; c.off = (size_t)min(row, M - 1) * lda; return c; }
    // address of the 8 halfs [k, k+8) of this lane's row in plane 0 (hi) / 1 (lo)
    __device__ __forceinline__ const _Float16* src(const Ctx& c, int k, int plane) const { return (plane ? Al : Ah) + c.off + k; }
};

// ZeroPadding1D + strided Conv1D(k=3) as a 3-tap row gather (see ALoadConv3): output row (b, t) contracts over
// k = j*C + c with source row t*stride + j - pad_left of sequence b; a tap outside [0, L_in) reads `zero`
// (>= 16 bytes of zeros: an LDS-DMA cannot synthesise them).  C % 32 == 0, so a k-tile never straddles two taps.
struct GLoadConv3 {
    const _Float16* __restrict__ Hh; const _Float16* __restrict__ Hl;   // (B * L_in, C) each
    const _Float16* __restrict__ zero;
    int C, L_in, L_out, stride, pad_left, M;
    struct Ctx { int base_row; int t0; };
    __device__ __forceinline__ Ctx prep(int row) const {
        const int rc = min(row, M - 1);
        Ctx c; const int b = rc / L_out; const int t = rc - b * L_out;
        c.base_row = b * L_in; c.t0 = t * stride - pad_left;
        return c;
    }
    // Branch free on purpose: a divergent return would make hipcc issue one LDS-DMA per path, and the kernel's
    // counted s_waitcnt vmcnt(N) is only right when every DMA call is exactly one instruction.
    __device__ __forceinline__ const _Float16* src(const Ctx& c, int k, int plane) const {
        const int j = k / C; const int ch = k - j * C;
        const int s = c.t0 + j;
        const bool ok = (s >= 0) && (s < L_in);
        const uintptr_t in = reinterpret_cast<uintptr_t>((plane ? Hl : Hh) + (size_t)(c.base_row + (ok ? s : 0)) * C + ch);
        const uintptr_t z = reinterpret_cast<uintptr_t>(zero);
        return reinterpret_cast<const _Float16*>(ok ? in : z);
    }
};

template <int TM, int TN, class GL, class EP, int NBUF = 3>
__global__ void __launch_bounds__(256)
gemm_h3g_kernel(const GL gl, const _Float16* __restrict__ Bh, const _Float16* __restrict__ Bl, const int M, const int N,
                const int Kp, const int m_tiles, const int n_tiles, const int kt_per_split, const EP ep)
{
    h3_flush_f16_denormals();                     // the epilogue may split its result (EpBiasReluSplit)
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int STAGE = 2 * (BM + BN) * 32;     // halfs per stage
    constexpr int NPA = BM / 64, NPB = BN / 64;   // DMA passes (64 rows each) per plane
    constexpr int NP = 2 * (NPA + NPB);           // DMA instructions per thread per k-tile
    extern __shared__ __attribute__((aligned(16))) _Float16 hsm[];

    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int bn = slot % n_tiles;
    const int bm = (slot / n_tiles) * 8 + xcd;
    if (bm >= m_tiles) return;
    const int bm0 = bm * BM, bn0 = bn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
    const int drow = 16 * wave + (lane >> 2);                       // row inside a 64-row DMA pass
    const int dk = ((lane & 3) ^ ((lane >> 4) & 3)) * 8;            // logical chunk this lane fetches (halfs)

    typename GL::Ctx actx[NPA];
#pragma unroll
    for (int p = 0; p < NPA; ++p) actx[p] = gl.prep(bm0 + 64 * p + drow);
    const _Float16* bsrc[2 * NPB];
#pragma unroll
    for (int p = 0; p < NPB; ++p) {
        const size_t o = (size_t)min(bn0 + 64 * p + drow, N - 1) * Kp;
        bsrc[p] = Bh + o; bsrc[NPB + p] = Bl + o;
    }

    const int kt_lo = blockIdx.y * kt_per_split;
    const int KT = min(Kp / GEMM_BK, kt_lo + kt_per_split);

    auto dma = [&](int kt, int buf) {
        const int k = min(kt, KT - 1) * GEMM_BK + dk;
        _Float16* d = hsm + buf * STAGE + 16 * wave * 32;           // wave-uniform destination of pass 0
#pragma unroll
        for (int p = 0; p < 2 * NPA; ++p)      // hi passes first, then lo
            __builtin_amdgcn_global_load_lds((h3_glb_void*)gl.src(actx[p % NPA], k, p / NPA), (h3_lds_void*)(d + p * 64 * 32), 16, 0, 0);
#pragma unroll
        for (int p = 0; p < 2 * NPB; ++p)
            __builtin_amdgcn_global_load_lds((h3_glb_void*)(bsrc[p] + k), (h3_lds_void*)(d + (2 * NPA + p) * 64 * 32), 16, 0, 0);
    };
    f32x16 acc0[TM][TN], acc1[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }

    dma(kt_lo, 0);
    if (NBUF == 3) {
        dma(kt_lo + 1, 1);
        asm volatile("s_waitcnt vmcnt(%0)" :: "i"(NP) : "memory");  // tile kt_lo landed (this wave's part)
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                                    // ... and everybody else's

    const int fr = lane & 31, fkc = lane >> 5;
    int cur = 0;
    for (int kt = kt_lo; kt < KT; ++kt) {
        // buffer (cur + 2) % 3 was last read in iteration kt-1; every wave is past that barrier
        if (NBUF == 3) dma(kt + 2, cur >= 1 ? cur - 1 : 2); else dma(kt + 1, cur ^ 1);
        const _Float16* S = hsm + cur * STAGE;
#pragma unroll
        for (int kk = 0; kk < GEMM_BK / 16; ++kk) {
            h16x8 ah[TM], alo[TM], bh[TN], blo[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = wm * (BM / 2) + 32 * i + fr, c = (2 * kk + fkc) ^ ((r >> 2) & 3);
                ah[i] = *reinterpret_cast<const h16x8*>(S + r * 32 + c * 8);
                alo[i] = *reinterpret_cast<const h16x8*>(S + (BM + r) * 32 + c * 8);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = wn * (BN / 2) + 32 * j + fr, c = (2 * kk + fkc) ^ ((r >> 2) & 3);
                bh[j] = *reinterpret_cast<const h16x8*>(S + (2 * BM + r) * 32 + c * 8);
                blo[j] = *reinterpret_cast<const h16x8*>(S + (2 * BM + BN + r) * 32 + c * 8);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc0[i][j], 0, 0, 0);
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], blo[j], acc1[i][j], 0, 0, 0);
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[i], bh[j], acc1[i][j], 0, 0, 0);
                }
        }
        // tile kt+1 landed (tile kt+2 stays in flight), and this wave's fragment reads of buffer `cur` have returned:
        // the next iteration's DMA overwrites the buffer read one iteration earlier, so no read may still be
        // pending when a wave passes the barrier (hipcc otherwise sinks the lgkmcnt wait below it)
        if (NBUF == 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(NP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        cur = (NBUF == 3) ? (cur == 2 ? 0 : cur + 1) : (cur ^ 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the clamped tail DMAs must not outlive the LDS allocation

    const int crow0 = bm0 + wm * (BM / 2) + 4 * (lane >> 5);
    const int ccol0 = bn0 + wn * (BN / 2) + (lane & 31);
    const bool interior = (bm0 + BM <= M) && (bn0 + BN <= N);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = ccol0 + j * 32;
            if (interior) {
                const float2 cv = ep.colv(col);
                float2 pr[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) pr[r] = ep.pre(crow0 + i * 32 + (r & 3) + 8 * (r >> 2), col);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ep.store(crow0 + i * 32 + (r & 3) + 8 * (r >> 2), col,
                             acc0[i][j][r] + acc1[i][j][r] * (1.0f / H3_SCALE), cv, pr[r]);
            } else if (col < N) {
                const float2 cv = ep.colv(col);
                float2 pr[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) pr[r] = ep.pre(min(crow0 + i * 32 + (r & 3) + 8 * (r >> 2), M - 1), col);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = crow0 + i * 32 + (r & 3) + 8 * (r >> 2);
                    if (row < M) ep.store(row, col, acc0[i][j][r] + acc1[i][j][r] * (1.0f / H3_SCALE), cv, pr[r]);
                }
            }
        }
}
__host__ __device__ inline constexpr size_t gemm_h3g_lds_bytes(int BM, int BN, int nbuf = 3) { return (size_t)nbuf * 2 * (BM + BN) * 32 * sizeof(_Float16); }

// Plain f32 -> planes conversion (operands that have no producer kernel of their own).
static __global__ void __launch_bounds__(256)
split_rows_kernel(const float* __restrict__ x, const int ld, const int D, const int M,
                  _Float16* __restrict__ Ph, _Float16* __restrict__ Pl, const int ldp)
{
    h3_flush_f16_denormals();
    const int per_row = D / 4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)M * per_row) return;
    const int row = (int)(idx / per_row), c = (int)(idx - (long)row * per_row) * 4;
    const f32x4 y = *reinterpret_cast<const f32x4*>(x + (size_t)row * ld + c);
    h16x4 hi, lo;
    h3_split(y, hi, lo);
    *reinterpret_cast<h16x4*>(Ph + (size_t)row * ldp + c) = hi;
    *reinterpret_cast<h16x4*>(Pl + (size_t)row * ldp + c) = lo;
}

}  // namespace uu3d
