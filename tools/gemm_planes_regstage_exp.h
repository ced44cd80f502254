// Experiment kept for tools/gemm_bench.hip: the pre-split (planes) f16x3 GEMM with REGISTER staging
// (global_load_dwordx4 -> VGPR -> ds_write_b128, padded LDS rows), prefetch depth 1 or 2.  The product uses the
// LDS-DMA kernel gemm_h3g_kernel (uu3d_gemm_h3.h); this one measured 3-10 % slower on the model's shapes.
#pragma once
#include "../uplift-upsample-3dhpe_amd/csrc/uu3d_gemm_h3.h"
namespace uu3d {
#ifdef H3P_CLOCK
__device__ unsigned long long h3p_clk[3];
#endif
struct PLoadPlain {
    const _Float16* __restrict__ Ah; const _Float16* __restrict__ Al;   // [M][lda] each, lda % 8 == 0
    int lda, M;
    struct Ctx { size_t off; };
    __device__ __forceinline__ Ctx prep(int row) const { Ctx c; c.off = (size_t)min(row, M - 1) * lda; return c; }
    __device__ __forceinline__ void issue(const Ctx& c, int k, h16x8& hi, h16x8& lo) const {
        hi = *reinterpret_cast<const h16x8*>(Ah + c.off + k);
        lo = *reinterpret_cast<const h16x8*>(Al + c.off + k);
    }
};

// ZeroPadding1D + strided Conv1D(k=3) as a 3-tap gather over pre-split rows (see ALoadConv3): output row
// (b, t) contracts over k = j*C + c with source row t*stride + j - pad_left of sequence b.  C % 32 == 0,
// so one k-tile never straddles two taps.
struct PLoadConv3 {
    const _Float16* __restrict__ Hh; const _Float16* __restrict__ Hl;   // (B * L_in, C) each
    int C, L_in, L_out, stride, pad_left, M;
    struct Ctx { int base_row; int t0; };
    __device__ __forceinline__ Ctx prep(int row) const {
        const int rc = min(row, M - 1);
        Ctx c; const int b = rc / L_out; const int t = rc - b * L_out;
        c.base_row = b * L_in; c.t0 = t * stride - pad_left;
        return c;
    }
    __device__ __forceinline__ void issue(const Ctx& c, int k, h16x8& hi, h16x8& lo) const {
        const int j = k / C; const int ch = k - j * C;
        const int src = c.t0 + j;
        const bool ok = (src >= 0) && (src < L_in);
        const size_t off = (size_t)(c.base_row + (ok ? src : 0)) * C + ch;
        hi = *reinterpret_cast<const h16x8*>(Hh + off);
        lo = *reinterpret_cast<const h16x8*>(Hl + off);
        if (!ok) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { hi[e] = (_Float16)0.f; lo[e] = (_Float16)0.f; }
        }
    }
};

// DEPTH = how many k-tiles ahead the global loads run: 1 = loads of tile kt+1 are issued at the top of
// iteration kt and written to LDS at its end (one tile of compute, ~0.3-0.5 us, to cover an L2/HBM round trip
// that takes longer); 2 = a second register set keeps the loads of tile kt+2 in flight as well.
template <int TM, int TN, class PL, class EP, int DEPTH = 2>
__global__ void __launch_bounds__(256)
gemm_h3p_kernel(const PL pl, const _Float16* __restrict__ Bh, const _Float16* __restrict__ Bl, const int M, const int N,
                const int Kp, const int m_tiles, const int n_tiles, const int kt_per_split, const EP ep)
{
    constexpr int BM = 64 * TM, BN = 64 * TN, LD = H3_LD;
    constexpr int AI = BM / 64, BI = BN / 64;   // 16-byte staging loads per thread per plane per k-tile
    extern __shared__ __attribute__((aligned(16))) _Float16 hsm[];
    constexpr int STAGE = 2 * (BM + BN) * LD;   // Ah | Al | Bh | Bl

    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int bn = slot % n_tiles;
    const int bm = (slot / n_tiles) * 8 + xcd;
    if (bm >= m_tiles) return;
    const int bm0 = bm * BM, bn0 = bn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int srow = tid >> 2, scol = (tid & 3) * 8;          // staging: 64 rows x 4 x (8 halfs) per pass

    typename PL::Ctx actx[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) actx[i] = pl.prep(bm0 + srow + 64 * i);
    const _Float16* bhp[BI]; const _Float16* blp[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const size_t o = (size_t)(bn0 + srow + 64 * i) * Kp + scol;
        bhp[i] = Bh + o; blp[i] = Bl + o;
    }

    struct Regs { h16x8 ah[AI], al[AI], bh[BI], bl[BI]; };
    Regs R0, R1;
    f32x16 acc0[TM][TN], acc1[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }

    const int kt_lo = blockIdx.y * kt_per_split;
    const int KT = min(Kp / GEMM_BK, kt_lo + kt_per_split);

    auto issue = [&](int kt, Regs& R) {
        const int k0 = kt * GEMM_BK;
#pragma unroll
        for (int i = 0; i < AI; ++i) pl.issue(actx[i], k0 + scol, R.ah[i], R.al[i]);
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            R.bh[i] = *reinterpret_cast<const h16x8*>(bhp[i] + k0);
            R.bl[i] = *reinterpret_cast<const h16x8*>(blp[i] + k0);
        }
    };
    auto stage = [&](int buf, const Regs& R) {
        _Float16* S = hsm + buf * STAGE;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            *reinterpret_cast<h16x8*>(&S[(srow + 64 * i) * LD + scol]) = R.ah[i];
            *reinterpret_cast<h16x8*>(&S[BM * LD + (srow + 64 * i) * LD + scol]) = R.al[i];
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            *reinterpret_cast<h16x8*>(&S[2 * BM * LD + (srow + 64 * i) * LD + scol]) = R.bh[i];
            *reinterpret_cast<h16x8*>(&S[2 * BM * LD + BN * LD + (srow + 64 * i) * LD + scol]) = R.bl[i];
        }
    };
    const int fr = lane & 31, fk = (lane >> 5) * 8;
    auto compute = [&](int cur) {
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
    };

#ifdef H3P_CLOCK
    const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    issue(kt_lo, R0);
    stage(kt_lo & 1, R0);
    if (DEPTH == 2) issue(min(kt_lo + 1, KT - 1), R1);
    __syncthreads();

    if (DEPTH == 1) {
        for (int kt = kt_lo; kt < KT; ++kt) {
            issue(min(kt + 1, KT - 1), R0);
            compute(kt & 1);
            stage((kt & 1) ^ 1, R0);
            __syncthreads();
        }
    } else {
        // iteration kt: LDS[kt & 1] holds tile kt, RS holds (in flight) tile kt+1, RI takes tile kt+2
        auto body = [&](int kt, Regs& RI, const Regs& RS) {
            issue(min(kt + 2, KT - 1), RI);
            compute(kt & 1);
            stage((kt & 1) ^ 1, RS);
            __syncthreads();
        };
        for (int kt = kt_lo; kt < KT;) {
            body(kt, R0, R1); ++kt;
            if (kt >= KT) break;
            body(kt, R1, R0); ++kt;
        }
    }
#ifdef H3P_CLOCK
    if (tid == 0 && (blockIdx.x % 97) == 5) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        atomicAdd(&h3p_clk[0], t1 - clk_t0); atomicAdd(&h3p_clk[1], r1 - clk_r0); atomicAdd(&h3p_clk[2], 1ull);
    }
#endif

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
