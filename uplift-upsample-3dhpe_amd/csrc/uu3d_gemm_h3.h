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

__device__ __forceinline__ void h3_split(const f32x4 x, h16x4& hi, h16x4& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const _Float16 h = (_Float16)x[e];
        hi[e] = h;
        lo[e] = (_Float16)((x[e] - (float)h) * H3_SCALE);
    }
}

template <int TM, int TN, class AL, class EP>
__global__ void __launch_bounds__(256)
gemm_h3_kernel(const AL al, const _Float16* __restrict__ Bh, const _Float16* __restrict__ Bl, const int M, const int N,
               const int Kp, const int m_tiles, const int n_tiles, const int kt_per_split, const EP ep)
{
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

    typename AL::Raw ra[AI];
    h16x8 rbh[BI], rbl[BI];
    f32x16 acc0[TM][TN], acc1[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }

    const int kt_lo = blockIdx.y * kt_per_split;
    const int KT = min(Kp / GEMM_BK, kt_lo + kt_per_split);

    auto issue = [&](int kt) {
        const int k0 = kt * GEMM_BK;
#pragma unroll
        for (int i = 0; i < AI; ++i) ra[i] = al.issue(actx[i], k0 + acol);
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            rbh[i] = *reinterpret_cast<const h16x8*>(bhp[i] + k0);
            rbl[i] = *reinterpret_cast<const h16x8*>(blp[i] + k0);
        }
    };
    auto stage = [&](int kt, int buf) {
        _Float16* S = hsm + buf * STAGE;
        const int k0 = kt * GEMM_BK;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const f32x4 x = al.finish(actx[i], k0 + acol, ra[i]);
            h16x4 hi, lo;
            h3_split(x, hi, lo);
            *reinterpret_cast<h16x4*>(&S[(arow + 32 * i) * LD + acol]) = hi;
            *reinterpret_cast<h16x4*>(&S[BM * LD + (arow + 32 * i) * LD + acol]) = lo;
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            *reinterpret_cast<h16x8*>(&S[2 * BM * LD + (brow + 64 * i) * LD + bcol]) = rbh[i];
            *reinterpret_cast<h16x8*>(&S[2 * BM * LD + BN * LD + (brow + 64 * i) * LD + bcol]) = rbl[i];
        }
    };

    issue(kt_lo);
    stage(kt_lo, kt_lo & 1);
    __syncthreads();

    const int fr = lane & 31, fk = (lane >> 5) * 8;
    for (int kt = kt_lo; kt < KT; ++kt) {
        const int cur = kt & 1;
        const int ktn = min(kt + 1, KT - 1);
        issue(ktn);
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
        stage(ktn, cur ^ 1);
        __syncthreads();
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

}  // namespace uu3d
