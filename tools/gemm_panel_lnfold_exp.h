// tools/gemm_panel_lnfold_exp.h -- EXPERIMENT RECORD (docs/HISTORY.md E.11): the first form of the row-panel GEMM, with the
// LayerNorm statistics + split in its own prologue (LayerNorm folded into the operand).  Superseded by
// csrc/uu3d_gemm_panel.h + ln_split_frag_kernel; kept because the numbers quoted in DESIGN.md come from it.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops [-DSTAMP] [-DUU3D_PANEL_LOADALL] -o tools/gemm_panel_lnfold_exp tools/gemm_panel_lnfold_exp.hip
//
// The tiled kernels of uu3d_gemm_h3.h restart a 12-iteration k-loop in every 64 x 128 tile, re-stage (and
// re-normalise, re-split) the same A rows in each of the N / 128 workgroups along N, and move four f16 planes
// through LDS per k-tile.  Here the roles are fixed differently:
//
//   * a wave owns a PANEL of 32 token rows over their full length K: it loads them once (coalesced, transposed
//     through LDS into the MFMA A-fragment layout), splits them into the f16 hi / lo planes and keeps the 2 x 24
//     A fragments (192 registers) resident for the whole kernel;
//   * LayerNorm is FOLDED into the Dense (same identity as gemm_h3_lnfold_kernel):
//         LN(x) W + b = rstd (d (gamma o W)) - rstd mean_d (gamma^T W) + (beta^T W + b),   d = x - x[0]
//     the GEMM runs on the shifted raw rows d against W' = gamma o W, the row sums of d and d^2 are accumulated
//     while the fragments are split (shifting by the row's first element keeps the one-pass variance
//     well conditioned), and the epilogue applies rstd / mean with the two per-column vectors -- no row-statistics
//     launch, no LayerNorm arithmetic, no gamma / beta loads in the prologue;
//   * the weight operand streams past: it is stored at commit time in FRAGMENT ORDER
//         [32-column chunk][k-half (192)][16-deep k-slice][plane hi|lo][lane][8 halfs]        (1 KiB per fragment)
//     so that a half-chunk is one linear 24 KiB piece of memory that goes global -> LDS by global_load_lds_dwordx4
//     with no swizzle, and a B fragment is one conflict-free ds_read_b128 at lane * 16;
//   * a workgroup = 4 waves = 128 rows shares the weight stream through a 3-slot LDS ring (72 KiB), one barrier per
//     half-chunk (36 MFMAs per wave), half-chunk t+2 in flight.  The resident fragments put a wave at ~270 registers,
//     i.e. ONE wave per SIMD / one workgroup per CU (capped at 256 registers the compiler spills into the main loop and
//     every scratch reload costs a vmcnt(0) that drains the DMA prefetch): the loop is software pipelined instead;
//   * per 16-deep k-slice a wave issues 2 ds_read_b128 and 3 MFMAs (ah*bh -> acc0; ah*bl, al*bh -> acc1), the guide's
//     "two ds_read_b128 per MFMA gap are free" regime; the main loop has no VALU work besides addressing.
//
// Grid: (M / 128) row tiles x S column ranges of N / (32 S) chunks each.
#pragma once
#include "../uplift-upsample-3dhpe_amd/csrc/uu3d_gemm_h3.h"
#include <type_traits>

namespace uu3d {

static constexpr int PANEL_K = 384;                  // contraction length (d_t)
static constexpr int PANEL_KS = PANEL_K / 16;        // 24 k-slices
static constexpr int PANEL_HALF_BYTES = 12 * 2 * 1024;   // one half-chunk: 12 slices x 2 planes x 1 KiB
static constexpr int PANEL_SLOTS = 6;                // ring depth: half-chunks t+1 .. t+5 in flight while t is consumed
static constexpr size_t PANEL_LDS_BYTES = (size_t)PANEL_SLOTS * PANEL_HALF_BYTES;   // 72 KiB
static constexpr int PANEL_APASS_K = 64;             // A staging pass: 32 rows x 64 floats per wave
static constexpr int PANEL_ALD = PANEL_APASS_K + 4;  // padded row (floats): 68 -> 16 distinct 16-byte slots per read group
static constexpr int PANEL_BIAS_FLOATS = 1024 + 4 * 64;   // per-column epilogue vectors (g | b', <= 16 chunks) + (rstd, rstd * mean) of the 128 rows
static constexpr size_t PANEL_LDS_TOTAL = PANEL_LDS_BYTES + PANEL_BIAS_FLOATS * sizeof(float);

// halfs in the fragment-ordered operand of an N x 384 Dense (N multiple of 32)
__host__ __device__ inline constexpr size_t panel_operand_halfs(int N) { return (size_t)(N / 32) * 2 * (PANEL_HALF_BYTES / 2); }

// Host side: fragment-ordered planes from the transposed, padded Bt[N][Kp] (k contiguous) used by the tiled kernels.
// hi / lo as produced by the commit-time split (lo pre-scaled by 2048).
inline void panel_pack_operand(const _Float16* Bh, const _Float16* Bl, int N, int Kp, _Float16* out) {
    for (int c = 0; c < N / 32; ++c)
        for (int hf = 0; hf < 2; ++hf)
            for (int kk = 0; kk < 12; ++kk)
                for (int p = 0; p < 2; ++p)
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 8; ++j) {
                            const int n = 32 * c + (l & 31), k = hf * 192 + kk * 16 + (l >> 5) * 8 + j;
                            out[((((size_t)(c * 2 + hf) * 12 + kk) * 2 + p) * 64 + l) * 8 + j] = (p ? Bl : Bh)[(size_t)n * Kp + k];
                        }
}

struct PanelEpStore {          // out[row][col] = v
    float* __restrict__ out; int ldo;
    __device__ __forceinline__ void store(int row, int col, float v) const { out[(size_t)row * ldo + col] = v; }
};
struct PanelEpReluSplit {      // ReLU(v) as the two f16 planes the next GEMM reads
    _Float16* __restrict__ Oh; _Float16* __restrict__ Ol; int ldo;
    __device__ __forceinline__ void store(int row, int col, float x) const {
        const float v = fmaxf(x, 0.f);
        const _Float16 h = h3_hi(v);
        Oh[(size_t)row * ldo + col] = h;
        Ol[(size_t)row * ldo + col] = (_Float16)((v - (float)h) * H3_SCALE);
    }
};

#ifdef UU3D_PANEL_ACC3
#define PANEL_ACC1(r) (acc1[r] + acc2[r])
#else
#define PANEL_ACC1(r) acc1[r]
#endif
#ifdef UU3D_PANEL_STAMP
__device__ unsigned long long panel_clk[8];   // tools/gemm_panel_exp: s_memtime ticks in prologue / loop / ..., summed over workgroups
#define PANEL_STAMP(...) __VA_ARGS__
#else
#define PANEL_STAMP(...)
#endif

// X [M][ldx] f32 rows of length 384; Bf the fragment-ordered planes of W' = gamma o W; gcol[n] = sum_k gamma_k W[k][n];
// bcol[n] = sum_k beta_k W[k][n] + b[n].
template <class EP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
gemm_h3_panel_ln_kernel(const float* __restrict__ X, const int ldx, const float eps, const _Float16* __restrict__ Bf,
                        const float* __restrict__ gcol, const float* __restrict__ bcol, const int M, const int m_tiles,
                        const int splits, const int chunks_per_wg, const EP ep)
{
#ifndef UU3D_PANEL_NOFLUSH
    h3_flush_f16_denormals();
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char psm[];

    const int id = blockIdx.x;
    const int xcd = id & 7, slot_id = id >> 3;
    const int ns = slot_id % splits;
    const int bm = (slot_id / splits) * 8 + xcd;           // the column ranges of one row tile share an XCD (A rows hit its L2)
    if (bm >= m_tiles) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = bm * 128 + wave * 32;                 // this wave's panel
    const int chunk0 = ns * chunks_per_wg;
    const int T = 2 * chunks_per_wg;                       // half-chunks this workgroup consumes

    // ---- weight stream: half-chunk t -> ring slot t % PANEL_SLOTS; each wave moves 6 of its 24 pieces of 1 KiB ----
    const unsigned char* bsrc = reinterpret_cast<const unsigned char*>(Bf) + (size_t)chunk0 * 2 * PANEL_HALF_BYTES + (wave * 6) * 1024 + lane * 16;
    auto dma = [&](int t, int slot) __attribute__((always_inline)) {
        const unsigned char* s = bsrc + (size_t)min(t, T - 1) * PANEL_HALF_BYTES;
        unsigned char* d = psm + slot * PANEL_HALF_BYTES + (wave * 6) * 1024;
#pragma unroll
        for (int p = 0; p < 6; ++p)
            __builtin_amdgcn_global_load_lds((h3_glb_void*)(s + p * 1024), (h3_lds_void*)(d + p * 1024), 16, 0, 0);
    };
    PANEL_STAMP(const long long c_start = clock64();)
#pragma unroll
    for (int t = 0; t < PANEL_SLOTS - 2; ++t) dma(t, t);   // land while the panel is prepared (the last two slots stage A)

    // ---- A panel: 32 rows x 384 floats -> split fragments in registers, row sums on the way ----
    h16x8 ah[PANEL_KS], al[PANEL_KS];
    float s1 = 0.f, s2 = 0.f;
    {
        float* st = reinterpret_cast<float*>(psm + (PANEL_SLOTS - 2) * PANEL_HALF_BYTES) + wave * (32 * PANEL_ALD);      // 8.5 KiB per wave
        const int fr = lane & 31, fk = (lane >> 5) * 8;
        // the row loads run three passes (24 loads, 96 registers) ahead of the staging: one exposed global latency
        constexpr int NPASS = PANEL_K / PANEL_APASS_K;
#ifdef UU3D_PANEL_LOADALL
        constexpr int NV = 6;
#else
        constexpr int NV = 3;
#endif
        f32x4 v[NV][8];
        auto load_pass = [&](int p) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {                  // 32 rows x 16 float4: 16 lanes read 256 B of one row
                const int r = i * 4 + (lane >> 4), c4 = lane & 15;
                v[p % NV][i] = *reinterpret_cast<const f32x4*>(X + (size_t)min(row0 + r, M - 1) * ldx + p * PANEL_APASS_K + c4 * 4);
            }
        };
#ifdef UU3D_PANEL_LOADALL
        load_pass(0); load_pass(1); load_pass(2); load_pass(3); load_pass(4); load_pass(5);
#else
        load_pass(0); load_pass(1); load_pass(2);
#endif
        float x0 = 0.f;
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = i * 4 + (lane >> 4), c4 = lane & 15;
                *reinterpret_cast<f32x4*>(st + r * PANEL_ALD + c4 * 4) = v[p % NV][i];
            }
            if (NV == 3 && p + 3 < NPASS) load_pass(p + 3);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (p == 0) x0 = st[fr * PANEL_ALD];           // the row's first element: the shift
#pragma unroll
            for (int q = 0; q < PANEL_APASS_K / 16; ++q) {
                const f32x4 lo4 = *reinterpret_cast<const f32x4*>(st + fr * PANEL_ALD + q * 16 + fk);
                const f32x4 hi4 = *reinterpret_cast<const f32x4*>(st + fr * PANEL_ALD + q * 16 + fk + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = (e < 4 ? lo4[e & 3] : hi4[e & 3]) - x0;
                    s1 += d; s2 = fmaf(d, d, s2);
                    const _Float16 h = h3_hi(d);
                    ah[p * 4 + q][e] = h;
                    al[p * 4 + q][e] = (_Float16)((d - (float)h) * H3_SCALE);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    // per-column vectors of this workgroup's columns and the row statistics go to LDS behind the ring
    float* gcol_s = reinterpret_cast<float*>(psm + PANEL_LDS_BYTES);       // [512] g | [512] b'
    float2* rstat_s = reinterpret_cast<float2*>(gcol_s + 1024);            // [128] (rstd, rstd * mean_d)
    for (int i = tid; i < chunks_per_wg * 32; i += 256) { gcol_s[i] = gcol[chunk0 * 32 + i]; gcol_s[512 + i] = bcol[chunk0 * 32 + i]; }
    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
    {
        const float mean = s1 * (1.0f / PANEL_K);
        const float var = fmaxf(s2 * (1.0f / PANEL_K) - mean * mean, 0.f);
        const float rstd = rsqrtf(var + eps);
        if (lane < 32) rstat_s[wave * 32 + lane] = make_float2(rstd, rstd * mean);
    }

    // ---- main loop ----
    // Every wait is "vmcnt(6 x half-chunks left in flight)": half-chunk t was issued before t+1 .. t+4, whose pieces are
    // the newest loads at that point, and vector memory operations retire in order -- wherever the compiler puts the epilogue's
    // stores, the wait can only become stricter, never weaker.  The stores of chunk c are issued at the top of the
    // next iteration, BEFORE its DMA, so that they are older than every load a later wait leaves in flight.
    __syncthreads();                                       // every wave is done with the staging area; LDS vectors visible
    PANEL_STAMP(const long long c_pro = clock64();)
    dma(PANEL_SLOTS - 2, PANEL_SLOTS - 2);
    int slot_r = 0, slot_w = PANEL_SLOTS - 1;              // slot consumed / refilled in the current step
    const int crow = (lane >> 5) * 4, ccol = lane & 31;
    const int valid = min(32, M - row0);                   // wave-uniform: rows of this panel that exist (<= 0: none)

    // One output element of chunk c (C/D register r): LayerNorm applied to the accumulated products, then the epilogue.
    auto emit = [&](int c, int r, const f32x16& p0, const f32x16& p1, float g, float bb) __attribute__((always_inline)) {
        const int lr = 8 * (r >> 2) + crow + (r & 3);
        const float2 rs = rstat_s[wave * 32 + lr];          // (rstd, rstd * mean_d): re-read per element, 32 registers the loop cannot spare
        ep.store(row0 + lr, (chunk0 + c) * 32 + ccol, fmaf(rs.x, p0[r] + p1[r] * (1.0f / H3_SCALE), fmaf(-rs.y, g, bb)));
    };
    // The epilogue of chunk c-1 is INTERLEAVED with the MFMAs of chunk c (two accumulator sets): an MFMA holds the
    // vector issue port for 8 of its 32 cycles, the ~10 VALU operations and the store of one output element fit into the
    // rest.  Issued as a block between two chunks it cost 1340 cycles per chunk with the matrix pipe idle (measured).
    // WHOLE = every row of the panel exists (no predicates); otherwise the stores are predicated and issued as a block.
    auto chunk = [&](auto whole_tag, int c, f32x16& acc0, f32x16& acc1, const f32x16& p0, const f32x16& p1) __attribute__((always_inline)) {
        constexpr bool WHOLE = decltype(whole_tag)::value;
        const bool prev = c > 0;
        const float pg = gcol_s[max(c - 1, 0) * 32 + ccol], pb = gcol_s[512 + max(c - 1, 0) * 32 + ccol];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int t = 2 * c + hf;
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(6 * (PANEL_SLOTS - 2)) : "memory");   // half-chunk t landed (this wave's pieces); own reads of t-1 returned
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();                                  // ... everybody's; the slot refilled below was last read in t-1
            __builtin_amdgcn_sched_barrier(0);
            if (!WHOLE && hf == 0 && prev) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (8 * (r >> 2) + crow + (r & 3) < valid) emit(c - 1, r, p0, p1, pg, pb);
                __builtin_amdgcn_sched_barrier(0);
            }
            dma(t + PANEL_SLOTS - 1, slot_w);
            if (hf == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
            }
            // Fragment reads run two k-slices ahead of the MFMAs that consume them.  hipcc sinks plain loads down to
            // their use (ds_read, lgkmcnt(0), 3 MFMAs: the LDS latency exposed 12 times per half-chunk at one wave per
            // SIMD), so the reads are asm with counted waits; LDS returns in order, and each wait names the fragments it
            // releases so that the MFMAs cannot move above it.
            const unsigned sb = (unsigned)(uintptr_t)(h3_lds_void*)(psm + slot_r * PANEL_HALF_BYTES + lane * 16);
            h16x8 bh[3], bl[3];
#define UU3D_PANEL_READ(i, kk) \
            asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" \
                         : "=&v"(bh[i]), "=&v"(bl[i]) : "v"(sb), "i"((kk) * 2048), "i"((kk) * 2048 + 1024))
            UU3D_PANEL_READ(0, 0);
            UU3D_PANEL_READ(1, 1);
#pragma unroll
            for (int kk = 0; kk < 12; ++kk) {
                if (kk + 2 < 12) {
                    UU3D_PANEL_READ((kk + 2) % 3, kk + 2);
                    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bh[kk % 3]), "+v"(bl[kk % 3]));
                } else if (kk + 1 < 12) {
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bh[kk % 3]), "+v"(bl[kk % 3]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bh[kk % 3]), "+v"(bl[kk % 3]));
                }
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[hf * 12 + kk], bh[kk % 3], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[hf * 12 + kk], bl[kk % 3], acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[hf * 12 + kk], bh[kk % 3], acc1, 0, 0, 0);
                if (WHOLE && hf * 12 + kk < 16 && prev) emit(c - 1, hf * 12 + kk, p0, p1, pg, pb);
            }
#undef UU3D_PANEL_READ
            slot_r = slot_r + 1 == PANEL_SLOTS ? 0 : slot_r + 1;
            slot_w = slot_w + 1 == PANEL_SLOTS ? 0 : slot_w + 1;
        }
    };
    f32x16 a0, a1, b0, b1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { b0[r] = 0.f; b1[r] = 0.f; }
    auto run = [&](auto whole_tag) __attribute__((always_inline)) {
        for (int c = 0; c < chunks_per_wg; c += 2) {
            chunk(whole_tag, c, a0, a1, b0, b1);
            if (c + 1 < chunks_per_wg) chunk(whole_tag, c + 1, b0, b1, a0, a1);
        }
    };
    if (valid == 32) run(std::true_type{}); else run(std::false_type{});
    PANEL_STAMP(const long long c_loop = clock64();)
    {   // last chunk
        const int c = chunks_per_wg - 1;
        const float g = gcol_s[c * 32 + ccol], bb = gcol_s[512 + c * 32 + ccol];
        if (c & 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (8 * (r >> 2) + crow + (r & 3) < valid) emit(c, r, b0, b1, g, bb);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (8 * (r >> 2) + crow + (r & 3) < valid) emit(c, r, a0, a1, g, bb);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the clamped tail DMAs must not outlive the LDS allocation
    PANEL_STAMP(if (tid == 0) { atomicAdd(&panel_clk[0], (unsigned long long)(c_pro - c_start)); atomicAdd(&panel_clk[1], (unsigned long long)(c_loop - c_pro));
        atomicAdd(&panel_clk[4], (unsigned long long)(clock64() - c_loop)); atomicAdd(&panel_clk[5], 1ull); })
}

}  // namespace uu3d
