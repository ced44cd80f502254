// tools/gemm_rows_ln_exp.h (round 3, measured and NOT kept: 31.9-33.9 us per launch at 9088 rows against 20.8 + 9.9 us for the
// projection on 64 x 128 tiles + ln_split_frag, h36m_351 batch 128, profiles/r03_rows_ab.txt; a workgroup that owns full rows has to
// stream the whole weight operand through its own CU, and with two 56 KiB LDS buffers only one k-tile is in flight: the loop runs at the
// round trip of a 57 KB LDS-DMA burst per k-tile) -- the attention projection of a transformer block with everything that follows it up to the MLP's first
// Dense layer in ONE launch:   x += o Wp + bp ;  A2 = split(LayerNorm2(x))   (vision_transformer.py:186-188 and the norm2 of :189;
// strided blocks: uplift_upsample_transformer.py:129-134).
//
// Before: gemm_h3g_kernel<1, 2> on 64 x 128 tiles (18.9 us at 9088 rows: three workgroups per 64 rows, the residual stream written
// by the GEMM's epilogue) + ln_split_frag_kernel (7.8 us: the same 14 MB read again, normalised, split, stored in fragment order).
// A LayerNorm needs whole rows, so here a workgroup OWNS 64 full rows: tile 64 x 384 = the LDS-DMA kernel's structure
// (uu3d_gemm_h3.h: four f16 planes of a k-tile global -> LDS by global_load_lds_dwordx4, XOR-swizzled 64-byte rows) with
// TN = 6: each of the 2 x 2 waves accumulates 32 rows x 192 columns (12 accumulators = 192 registers), two LDS buffers of
// 56 KiB.  The epilogue turns the accumulator tile through LDS (64 rows x 400 floats, the freed stage buffers): acc + bias ->
// LDS; then 16 threads per row read it back next to the residual row (16-byte pieces), store the new residual stream, and run
// ln_split_frag's arithmetic (two-pass statistics, eps inside the root, hi / lo split, fragment-ordered 8-byte stores) on the
// registers they hold.  142 workgroups at 9088 rows -- one per CU, 55 % of the chip -- each streaming the whole 590 KB weight
// operand: the launch is bound by that stream (~11 us at the ~55 GB/s a CU takes from L2), not by its MFMAs (6.6 us per SIMD).
#pragma once
#include "uu3d_gemm_h3.h"
#include "uu3d_gemm_panel.h"

namespace uu3d {

static constexpr int ROWS_BM = 64, ROWS_BN = 384, ROWS_TLD = 400;      // tile; floats per row of the epilogue's LDS image (conflict-free 16-byte reads)
static constexpr size_t ROWS_LDS_BYTES = (size_t)2 * 2 * (ROWS_BM + ROWS_BN) * 32 * sizeof(_Float16);   // 114 688 >= 64 * 400 * 4
static_assert(ROWS_LDS_BYTES >= (size_t)ROWS_BM * ROWS_TLD * 4, "epilogue image fits the stage buffers");

// Ah / Al: A planes [M][K] (K = 32 KT), Bh / Bl: weight planes Bt[384][K]; x: residual stream [M][384] (updated in place);
// Af: fragment-ordered LayerNorm output (panel_a_index), allocated in whole 32-row panels.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
gemm_h3g_rows_ln_kernel(const _Float16* __restrict__ Ah, const _Float16* __restrict__ Al, const _Float16* __restrict__ Bh,
                        const _Float16* __restrict__ Bl, const int M, const int Kp, float* __restrict__ x, const float* __restrict__ bias,
                        const float* __restrict__ gamma, const float* __restrict__ beta, const float eps, _Float16* __restrict__ Af)
{
    h3_flush_f16_denormals();
    constexpr int BM = ROWS_BM, BN = ROWS_BN, TN = 6;
    constexpr int STAGE = 2 * (BM + BN) * 32;     // halfs per stage
    constexpr int NPB = BN / 64;                  // DMA passes (64 rows each) per B plane
    extern __shared__ __attribute__((aligned(16))) _Float16 hsm[];

    const int bm0 = blockIdx.x * BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
    const int drow = 16 * wave + (lane >> 2);                       // row inside a 64-row DMA pass
    const int dk = ((lane & 3) ^ ((lane >> 4) & 3)) * 8;            // logical chunk this lane fetches (halfs)
    const size_t aoff = (size_t)min(bm0 + drow, M - 1) * Kp;
    const _Float16* bsrc[2 * NPB];
#pragma unroll
    for (int p = 0; p < NPB; ++p) {
        const size_t o = (size_t)(64 * p + drow) * Kp;
        bsrc[p] = Bh + o; bsrc[NPB + p] = Bl + o;
    }
    const int KT = Kp / 32;
    auto dma = [&](int kt, int buf) {
        const int k = min(kt, KT - 1) * 32 + dk;
        _Float16* d = hsm + buf * STAGE + 16 * wave * 32;           // wave-uniform destination of pass 0
        __builtin_amdgcn_global_load_lds((h3_glb_void*)(Ah + aoff + k), (h3_lds_void*)d, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((h3_glb_void*)(Al + aoff + k), (h3_lds_void*)(d + 64 * 32), 16, 0, 0);
#pragma unroll
        for (int p = 0; p < 2 * NPB; ++p)
            __builtin_amdgcn_global_load_lds((h3_glb_void*)(bsrc[p] + k), (h3_lds_void*)(d + (2 + p) * 64 * 32), 16, 0, 0);
    };
    f32x16 acc0[TN], acc1[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[j][r] = 0.f; acc1[j][r] = 0.f; }

    dma(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int fr = lane & 31, fkc = lane >> 5;
    int cur = 0;
    for (int kt = 0; kt < KT; ++kt) {
        dma(kt + 1, cur ^ 1);                     // buffer cur ^ 1 was last read in iteration kt - 1; every wave is past that barrier
        const _Float16* S = hsm + cur * STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            h16x8 ah, alo, bh[TN], blo[TN];
            {
                const int r = wm * 32 + fr, c = (2 * kk + fkc) ^ ((r >> 2) & 3);
                ah = *reinterpret_cast<const h16x8*>(S + r * 32 + c * 8);
                alo = *reinterpret_cast<const h16x8*>(S + (BM + r) * 32 + c * 8);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = wn * (BN / 2) + 32 * j + fr, c = (2 * kk + fkc) ^ ((r >> 2) & 3);
                bh[j] = *reinterpret_cast<const h16x8*>(S + (2 * BM + r) * 32 + c * 8);
                blo[j] = *reinterpret_cast<const h16x8*>(S + (2 * BM + BN + r) * 32 + c * 8);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc0[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[j], acc0[j], 0, 0, 0);
                acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, blo[j], acc1[j], 0, 0, 0);
                acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bh[j], acc1[j], 0, 0, 0);
            }
        }
        // tile kt + 1 landed and this wave's fragment reads of buffer `cur` have returned (the next iteration's DMA overwrites it)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        cur ^= 1;
    }
    // (the last iteration's clamped DMA has landed: vmcnt(0) above; nobody reads the stage buffers any more)

    // ---- epilogue 1: acc + bias -> LDS image T[64][ROWS_TLD] ----
    float* const T = reinterpret_cast<float*>(hsm);
    {
        const int trow0 = wm * 32 + 4 * (lane >> 5), tcol0 = wn * (BN / 2) + (lane & 31);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = tcol0 + 32 * j;
            const float bv = bias[col];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                T[(trow0 + (r & 3) + 8 * (r >> 2)) * ROWS_TLD + col] = acc0[j][r] + acc1[j][r] * (1.0f / H3_SCALE) + bv;
        }
    }
    __syncthreads();
    // ---- epilogue 2: x += T; LayerNorm; split; fragment-ordered store.  16 threads per row, 16 rows per pass (ln_split_frag_body's
    // mapping: 4-float pieces interleaved across the 16 threads, so a 16-lane group reads 256 contiguous bytes of LDS) ----
    constexpr int D = 384, NV = D / 64, KS = D / 16;
    const int j16 = tid & 15, lr = tid >> 4;
    f32x4 g[NV], bt[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { g[i] = *reinterpret_cast<const f32x4*>(gamma + 4 * (j16 + 16 * i)); bt[i] = *reinterpret_cast<const f32x4*>(beta + 4 * (j16 + 16 * i)); }
#pragma unroll
    for (int pass = 0; pass < BM / 16; ++pass) {
        const int trow = pass * 16 + lr, row = bm0 + trow;
        float* xp = x + (size_t)min(row, M - 1) * D;
        f32x4 v[NV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = 4 * (j16 + 16 * i);
            v[i] = *reinterpret_cast<const f32x4*>(xp + c) + *reinterpret_cast<const f32x4*>(T + trow * ROWS_TLD + c);
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
        if (row < M) {
#pragma unroll
            for (int i = 0; i < NV; ++i) *reinterpret_cast<f32x4*>(xp + 4 * (j16 + 16 * i)) = v[i];
        }
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
        const float mean = s * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float a = v[i][0] - mean, b = v[i][1] - mean, c = v[i][2] - mean, d = v[i][3] - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
        q += __shfl_xor(q, 1); q += __shfl_xor(q, 2); q += __shfl_xor(q, 4); q += __shfl_xor(q, 8);
        const float rstd = 1.0f / sqrtf(q * (1.0f / D) + eps);
        if (row < M) {
            _Float16* base = Af + (size_t)(row >> 5) * KS * 2 * 512 + (row & 31) * 8;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = 4 * (j16 + 16 * i);
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float inv = rstd * g[i][e]; y[e] = v[i][e] * inv + (bt[i][e] - mean * inv); }
                h16x4 hi, lo;
                h3_split(y, hi, lo);
                _Float16* d = base + (size_t)(c >> 4) * 2 * 512 + ((c >> 3) & 1) * 256 + (c & 4);
                *reinterpret_cast<h16x4*>(d) = hi;
                *reinterpret_cast<h16x4*>(d + 512) = lo;
            }
        }
    }
}

}  // namespace uu3d
