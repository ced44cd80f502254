// Experiment record, included only by tools/gemm_bench.hip.
#pragma once
#include "../uplift-upsample-3dhpe_amd/csrc/uu3d_gemm.h"
namespace uu3d {

// ------------------------------------------------------------------------------------
// Persistent GEMM EXPERIMENT (not used by the product: measured slower than the plain 64x64 kernel,
// see DESIGN.md "Experiments that did not pay").  Measured on MI355X (tools/): the hardware dispatcher
// PACKS workgroups onto CUs up to the occupancy limit instead of spreading them, so a grid
// of 2556 tiles runs as three full "rounds" of 1024 (83 % of the work/throughput bound) and
// an 852-tile grid leaves 43 CUs idle.  This kernel therefore launches exactly
// 256 CUs x 4 resident workgroups and lets them PULL tiles from a ticket counter:
//   * first tile = blockIdx.x, later tiles = gridDim.x + atomicAdd(ticket) (zeroed per launch)
//   * the next ticket is drawn at the start of a tile and parked in LDS behind the k-loop's
//     barriers; during the LAST k-iteration of a tile the first k-tile of the next one is
//     already in flight, and the epilogue's stores overlap the next tile's MFMAs.
//   * MFMA is v_mfma_f32_16x16x4_f32 (exact f32): it holds a ~8 % higher clock than the
//     32x32x2 form under load on this chip, and allows 32-row tiles (finer work units for the
//     N = 384 GEMMs: 1704 tiles instead of 852).
// Wave grid 2 x 2, wave tile (16 TM) x (16 TN), workgroup tile (32 TM) x (32 TN).
// LDS rows are 40 floats apart: the 16 lanes of each ds_read_b128 group (row = lane & 15,
// k-quad = lane >> 4) then fall on 16 distinct 16-byte slots.
// ------------------------------------------------------------------------------------
static constexpr int GEMMP_LD = 40;
__host__ __device__ inline constexpr size_t gemmp_lds_bytes(int TM, int TN) {
    return (size_t)2 * 32 * (TM + TN) * GEMMP_LD * sizeof(float) + 16;
}

template <int TM, int TN, class AL, class EP>
__global__ void __launch_bounds__(256)
gemm_f32p_kernel(const AL al, const float* __restrict__ Bt, const int M, const int N, const int Kp,
                 const int n_tiles, const int total_tiles, int* __restrict__ ticket, const EP ep)
{
    constexpr int BM = 32 * TM, BN = 32 * TN, LD = GEMMP_LD;
    constexpr int AI = TM, BI = TN;                // 16-byte staging loads per thread per k-tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                              // [2][BM][LD]
    float* Bs = smem + 2 * BM * LD;                // [2][BN][LD]
    int* slot = reinterpret_cast<int*>(smem + 2 * (BM + BN) * LD);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int srow = tid >> 3, scol = (tid & 7) * 4;
    const int fr = lane & 15, fg = lane >> 4;
    const int KT = Kp / GEMM_BK;

    int tile = blockIdx.x;
    if (tile >= total_tiles) return;
    int bm0 = (tile / n_tiles) * BM, bn0 = (tile % n_tiles) * BN;

    typename AL::Ctx actx[AI], nctx[AI];
    const float* bptr[BI]; const float* nbptr[BI];
#pragma unroll
    for (int i = 0; i < AI; ++i) actx[i] = al.prep(bm0 + srow + 32 * i);
#pragma unroll
    for (int i = 0; i < BI; ++i) bptr[i] = Bt + (size_t)(bn0 + srow + 32 * i) * Kp + scol;

    typename AL::Raw ra[AI];
    f32x4 rb[BI];
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // prologue: k-tile 0 of the first tile -> LDS buffer 0
#pragma unroll
    for (int i = 0; i < AI; ++i) ra[i] = al.issue(actx[i], scol);
#pragma unroll
    for (int i = 0; i < BI; ++i) rb[i] = *reinterpret_cast<const f32x4*>(bptr[i]);
#pragma unroll
    for (int i = 0; i < AI; ++i) *reinterpret_cast<f32x4*>(&As[(srow + 32 * i) * LD + scol]) = al.finish(actx[i], scol, ra[i]);
#pragma unroll
    for (int i = 0; i < BI; ++i) *reinterpret_cast<f32x4*>(&Bs[(srow + 32 * i) * LD + scol]) = rb[i];
    __syncthreads();

    int it = 0;
    for (;;) {
        int next = total_tiles;
        int nbm0 = 0, nbn0 = 0;
        int drawn = 0;
        for (int kt = 0; kt < KT; ++kt, ++it) {
            const int buf = it & 1;
            const bool last = (kt == KT - 1);
            if (kt == 0 && tid == 0) drawn = (int)gridDim.x + atomicAdd(ticket, 1);
            if (!last) {
                const int k0 = (kt + 1) * GEMM_BK;
#pragma unroll
                for (int i = 0; i < AI; ++i) ra[i] = al.issue(actx[i], k0 + scol);
#pragma unroll
                for (int i = 0; i < BI; ++i) rb[i] = *reinterpret_cast<const f32x4*>(bptr[i] + k0);
            } else {
                // next tile id was parked in LDS at least one barrier ago (KT >= 2)
                next = __builtin_amdgcn_readfirstlane(*slot);
                const int nt = min(next, total_tiles - 1);           // clamp: loads stay in bounds, results unused
                nbm0 = (nt / n_tiles) * BM; nbn0 = (nt % n_tiles) * BN;
#pragma unroll
                for (int i = 0; i < AI; ++i) nctx[i] = al.prep(nbm0 + srow + 32 * i);
#pragma unroll
                for (int i = 0; i < BI; ++i) nbptr[i] = Bt + (size_t)(nbn0 + srow + 32 * i) * Kp + scol;
#pragma unroll
                for (int i = 0; i < AI; ++i) ra[i] = al.issue(nctx[i], scol);
#pragma unroll
                for (int i = 0; i < BI; ++i) rb[i] = *reinterpret_cast<const f32x4*>(nbptr[i]);
            }

            const float* Ac = As + buf * BM * LD + (wm * (BM / 2) + fr) * LD + 4 * fg;
            const float* Bc = Bs + buf * BN * LD + (wn * (BN / 2) + fr) * LD + 4 * fg;
#pragma unroll
            for (int kk = 0; kk < GEMM_BK / 16; ++kk) {
                f32x4 af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(Ac + i * 16 * LD + kk * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bc + j * 16 * LD + kk * 16);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
            }

            const int nxt = buf ^ 1;
            if (!last) {
                const int k0 = (kt + 1) * GEMM_BK;
#pragma unroll
                for (int i = 0; i < AI; ++i)
                    *reinterpret_cast<f32x4*>(&As[nxt * BM * LD + (srow + 32 * i) * LD + scol]) = al.finish(actx[i], k0 + scol, ra[i]);
            } else {
#pragma unroll
                for (int i = 0; i < AI; ++i)
                    *reinterpret_cast<f32x4*>(&As[nxt * BM * LD + (srow + 32 * i) * LD + scol]) = al.finish(nctx[i], scol, ra[i]);
            }
#pragma unroll
            for (int i = 0; i < BI; ++i)
                *reinterpret_cast<f32x4*>(&Bs[nxt * BN * LD + (srow + 32 * i) * LD + scol]) = rb[i];
            if (kt == 0 && tid == 0) *slot = drawn;
            __syncthreads();
        }

        // epilogue of `tile`; C/D map of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + r
        {
            const int crow0 = bm0 + wm * (BM / 2) + 4 * fg;
            const int ccol0 = bn0 + wn * (BN / 2) + fr;
            if (bm0 + BM <= M && bn0 + BN <= N) {
                float2 cv[TN];
                float2 pr[TM][TN][4];
#pragma unroll
                for (int j = 0; j < TN; ++j) cv[j] = ep.colv(ccol0 + 16 * j);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) pr[i][j][r] = ep.pre(crow0 + 16 * i + r, ccol0 + 16 * j);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            ep.store(crow0 + 16 * i + r, ccol0 + 16 * j, acc[i][j][r], cv[j], pr[i][j][r]);
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int col = ccol0 + 16 * j;
                        if (col < N) {
                            const float2 cv = ep.colv(col);
                            float2 pr[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) pr[r] = ep.pre(min(crow0 + 16 * i + r, M - 1), col);
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = crow0 + 16 * i + r;
                                if (row < M) ep.store(row, col, acc[i][j][r], cv, pr[r]);
                            }
                        }
                    }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (next >= total_tiles) break;
        tile = next; bm0 = nbm0; bn0 = nbn0;
#pragma unroll
        for (int i = 0; i < AI; ++i) actx[i] = nctx[i];
#pragma unroll
        for (int i = 0; i < BI; ++i) bptr[i] = nbptr[i];
    }
}

}  // namespace uu3d
