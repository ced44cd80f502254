// uu3d_gemm_wt.h -- f16x3 GEMM for FEW ROWS: one workgroup per 32 x 32 output tile, the contraction split over its waves.
//
// The strided blocks 2-3 and the two heads multiply 128 ... 384 (head1: 9088 x 51) rows: a handful of 64 x 64 tiles.  The
// tiled kernels cut such a product along K into slabs and need a second launch to add them up (splitk_reduce_kernel), and a
// LayerNorm in front costs a third one (row_stats_kernel): 3 launches of 5-10 us each for < 1 GFLOP, 36 launches = a fifth
// of the forward.  Here the split-K lives INSIDE a workgroup:
//   * a workgroup owns one 32 x 32 tile of C; wave w of its KW waves owns the k-slices [w SPW, (w + 1) SPW) (16 deep each);
//   * no LDS staging at all: a lane's MFMA fragment is 16 contiguous bytes of a row (A: activation row lane & 31, B: row
//     lane & 31 of the transposed weight planes Bt[N][Kp]), loaded straight into the registers the MFMA reads -- every load
//     of a wave is issued before the first is used: ONE memory round trip per GEMM;
//   * LayerNorm in front (two-pass, the arithmetic of row_stats_kernel + ALoadLayerNorm): the waves exchange their partial
//     row sums through LDS (two barriers), then normalise and split their own slices in registers;
//   * the KW partial tiles are added in wave order by wave 0 (deterministic), which then runs the usual epilogue.
// Both operands are re-read by every tile that shares a row / column range (32 x 32 tiles: 16 flop per byte), so this only
// pays where the alternative is three launches: the LayerNorm-fed Dense layers of a block with <= 512 rows (measured,
// h36m_351 batch 128, strided block 3: 16.6 vs 20.5 us and 16.8 vs 19.1 us).  For the projection, the strided convolution
// (K = 2304) and the heads the same structure was SLOWER than split-K tiles + reduce (conv 29 vs 20 us, head1 19 vs 12 us,
// projection 11.5 vs 9.6 us) and is not used; the 2944-row layers of strided block 2 stay on the row-panel kernel.
#pragma once
#include "uu3d_gemm_h3.h"

namespace uu3d {

// ---- A operand loader: per lane, row = lane & 31 of the tile, slice q -> k = 16 q + 8 (lane >> 5) .. + 8 ----
struct WtLoadF32 {               // f32 rows [M][lda], split on the fly; ln != 0: LayerNorm (gamma, beta, eps) in front, K = the full row
    const float* __restrict__ A; int lda, M, K;
    const float* __restrict__ gamma; const float* __restrict__ beta; float eps; int ln;
    struct Ctx { const float* p; };
    struct Raw { f32x4 a, b; };
    __device__ __forceinline__ Ctx prep(int row) const { Ctx c; c.p = A + (size_t)min(row, M - 1) * lda; return c; }
    __device__ __forceinline__ Raw load(const Ctx& c, int k) const {
        const int kc = min(k, K - 8);
        Raw r; r.a = *reinterpret_cast<const f32x4*>(c.p + kc); r.b = *reinterpret_cast<const f32x4*>(c.p + kc + 4); return r;
    }
};
static constexpr int WT_MAX_WAVES = 8;                 // 512 threads: two waves per SIMD, 256 registers each (a wave keeps SPW slices of both operands in flight)
__host__ __device__ inline constexpr size_t gemm_wt_lds_bytes(int waves) { return (size_t)waves * 16 * 64 * sizeof(float); }

// C[M][N] = A[M][K] Bt^T (+ epilogue); Kp = 16 * slices.  Wave w owns the slices [w SPW, (w + 1) SPW): one batch of loads, one
// memory round trip; blockDim = 64 * ceil(slices / SPW) <= 512.
template <class AL, class EP, int SPW>
__global__ void __launch_bounds__(64 * WT_MAX_WAVES)
gemm_h3_wt_kernel(const AL al, const _Float16* __restrict__ Bh, const _Float16* __restrict__ Bl, const int M, const int N,
                  const int Kp, const int n_tiles, const EP ep)
{
    h3_flush_f16_denormals();
    extern __shared__ __attribute__((aligned(16))) float wt_red[];      // partial tiles [wave][register][lane] (4 KiB per wave); first also the LayerNorm partial sums
    float (*red)[16][64] = reinterpret_cast<float (*)[16][64]>(wt_red);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, KW = blockDim.x >> 6;
    const int r = lane & 31, g = lane >> 5;
    const int bm = blockIdx.x / n_tiles, bn = blockIdx.x - bm * n_tiles;
    const int slices = Kp >> 4;
    const typename AL::Ctx ctx = al.prep(bm * 32 + r);
    const size_t brow = (size_t)(bn * 32 + r) * Kp + g * 8;          // Bt is padded to a multiple of 128 rows: always in range
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }

    const int s_lo = w * SPW;
    // ---- every load of this wave, then one wait ----
    typename AL::Raw ra[SPW];
    h16x8 bh[SPW], bl[SPW];
#pragma unroll
    for (int q = 0; q < SPW; ++q) {
        const int k = min(s_lo + q, slices - 1) * 16 + g * 8;
        ra[q] = al.load(ctx, k);
        bh[q] = *reinterpret_cast<const h16x8*>(Bh + brow + min(s_lo + q, slices - 1) * 16);
        bl[q] = *reinterpret_cast<const h16x8*>(Bl + brow + min(s_lo + q, slices - 1) * 16);
    }
    h16x8 ah[SPW], alo[SPW];
    {
        float mean = 0.f, rstd = 1.f;
        if (al.ln) {
            // two-pass statistics over the whole row: partial sums of this wave's slices (both k-halves: lanes l, l ^ 32), exchanged
            // through LDS -- row_stats_kernel's arithmetic with a different summation tree
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < SPW; ++q)
                if (s_lo + q < slices) s += ((ra[q].a[0] + ra[q].a[1]) + (ra[q].a[2] + ra[q].a[3])) + ((ra[q].b[0] + ra[q].b[1]) + (ra[q].b[2] + ra[q].b[3]));
            s += __shfl_xor(s, 32);
            if (g == 0) red[w][0][r] = s;
            __syncthreads();
            float tot = 0.f;
            for (int i = 0; i < KW; ++i) tot += red[i][0][r];
            mean = tot / (float)al.K;
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < SPW; ++q)
                if (s_lo + q < slices) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float a = ra[q].a[e] - mean, b = ra[q].b[e] - mean; v += a * a + b * b; }
                }
            v += __shfl_xor(v, 32);
            if (g == 0) red[w][1][r] = v;
            __syncthreads();
            float vt = 0.f;
            for (int i = 0; i < KW; ++i) vt += red[i][1][r];
            rstd = 1.0f / sqrtf(vt / (float)al.K + al.eps);
            __syncthreads();                               // red is reused for the partial tiles below
        }
#pragma unroll
        for (int q = 0; q < SPW; ++q) {
            f32x4 xa = ra[q].a, xb = ra[q].b;
            if (al.ln) {
                const int k = min(s_lo + q, slices - 1) * 16 + g * 8;
                const f32x4 ga = *reinterpret_cast<const f32x4*>(al.gamma + k), gb = *reinterpret_cast<const f32x4*>(al.gamma + k + 4);
                const f32x4 ba = *reinterpret_cast<const f32x4*>(al.beta + k), bb = *reinterpret_cast<const f32x4*>(al.beta + k + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ia = rstd * ga[e], ib = rstd * gb[e];
                    xa[e] = xa[e] * ia + (ba[e] - mean * ia);
                    xb[e] = xb[e] * ib + (bb[e] - mean * ib);
                }
            }
            h16x4 h0, l0, h1, l1;
            h3_split(xa, h0, l0); h3_split(xb, h1, l1);
            ah[q] = (h16x8){h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
            alo[q] = (h16x8){l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
        }
    }

#pragma unroll
    for (int q = 0; q < SPW; ++q)
        if (s_lo + q < slices) {                           // wave-uniform
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[q], bh[q], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[q], bl[q], acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[q], bh[q], acc1, 0, 0, 0);
        }
    // ---- combine the KW partial tiles in wave order (deterministic), epilogue by wave 0 ----
    if (KW > 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) red[w][i][lane] = acc0[i] + acc1[i] * (1.0f / H3_SCALE);
        __syncthreads();
        if (w != 0) return;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float v = red[0][i][lane];
            for (int u = 1; u < KW; ++u) v += red[u][i][lane];
            acc0[i] = v;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc0[i] = acc0[i] + acc1[i] * (1.0f / H3_SCALE);
    }
    const int col = bn * 32 + r, row0 = bm * 32 + 4 * g;
    if (col < N) {
        const float2 cv = ep.colv(col);
        float2 pr[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) pr[i] = ep.pre(min(row0 + (i & 3) + 8 * (i >> 2), M - 1), col);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = row0 + (i & 3) + 8 * (i >> 2);
            if (row < M) ep.store(row, col, acc0[i], cv, pr[i]);
        }
    }
}

}  // namespace uu3d
