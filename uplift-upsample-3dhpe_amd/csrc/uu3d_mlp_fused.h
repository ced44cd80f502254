// uu3d_mlp_fused.h -- vit.MLP of a temporal block (vision_transformer.py:46-68: Dense 384 -> 768, ReLU, Dense 768 -> 384) as ONE
// f16x3 kernel: the 9088 x 768 hidden activations never leave the CU.
//
// Before: fc1 on the row-panel kernel (23 us, hidden written as two f16 planes, 28 MB) + fc2 on the tiled LDS-DMA kernel
// (31 us: 64 x 128 tiles re-stream the weight for every 64 rows, 245 MB of L2 -> LDS traffic, MFMA busy 0.15).
//
// Structure = the row-panel kernel's (uu3d_gemm_panel.h), with the HIDDEN dimension split over workgroups:
//   * workgroup = 4 waves x 32 token rows, one slice of 256 hidden units (3 slices): 71 row tiles x 3 = 213 workgroups;
//   * a wave keeps the 2 x 24 A fragments of its panel (LayerNorm output, fragment ordered, written by ln_split_frag)
//     resident: 192 registers;
//   * every product is computed TRANSPOSED -- the weight fragment is the MFMA's A operand, the token fragment its B operand
//     (the same bytes as in the panel kernel, arguments swapped) -- so the C/D registers of fc1, H^T[hidden][token], hold
//     for ONE token (lane & 31) 16 hidden units: bias, ReLU and the hi / lo split are per-register work, and the split
//     registers ARE the B operand of fc2 (Y^T = W2^T H^T) with the k order 8 (j >> 2) + 4 h + (j & 3) inside a 16-deep
//     slice (MICROARCH guide, "an accumulator tile as the next MFMA's operand"); the fc2 weight fragments are packed in
//     that order at commit time.  The hidden slice of a panel is 8 chunks x 2 slices x 2 planes = 128 registers;
//   * fc1: 8 chunks (32 hidden units) x 24 k-slices; fc2: 12 output chunks x 16 k-slices, both streamed through the same
//     3-slot LDS ring in k-steps of 24 slices (48 KiB; fc2: 1.5 chunks per step), 16 steps per workgroup, LDS-DMA, counted
//     waits, one barrier per step, fragment reads two slices ahead by asm -- all as in the panel kernel;
//   * the result of a slice is a PARTIAL sum of fc2 (256 of the 768 hidden units): it goes to slab[slice][token][384] as
//     16-byte stores (lane = token, 4 consecutive output channels per register group); the three slabs, the bias and the
//     residual are added in a fixed order by the LayerNorm pass that follows anyway (ln_res_split_frag_kernel), which
//     writes the residual stream and the next LayerNorm-fed GEMM's A fragments.
#pragma once
#include "uu3d_gemm_panel.h"

namespace uu3d {

static constexpr int MLPF_SLICES = 3;                  // hidden slices (workgroups per row tile)
static constexpr int MLPF_HC = 8;                      // 32-unit hidden chunks per slice (h_t = 768 = 3 * 8 * 32)
static constexpr int MLPF_OC = 12;                     // 32-column output chunks (d_t = 384)
// halfs of the fc2 operand: [slice][out chunk][16 k-slices][plane][lane][8]
__host__ __device__ inline constexpr size_t mlpf_w2_halfs() { return (size_t)MLPF_SLICES * MLPF_OC * 16 * 2 * 512; }

// Host side: fc2 fragments from the transposed, padded planes Bt[n = out][k = hidden] (row stride Kp) of the tiled kernels.
// Element j of lane (row n = 32 c + (l & 31), half g = l >> 5) of k-slice t of hidden slice s is hidden unit
// 256 s + 16 t + 8 (j >> 2) + 4 g + (j & 3): the order in which fc1's accumulator registers hold them.
inline void mlpf_pack_w2(const _Float16* Bh, const _Float16* Bl, int Kp, _Float16* out) {
    for (int s = 0; s < MLPF_SLICES; ++s)
        for (int c = 0; c < MLPF_OC; ++c)
            for (int t = 0; t < 16; ++t)
                for (int p = 0; p < 2; ++p)
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 8; ++j) {
                            const int n = 32 * c + (l & 31), k = 256 * s + 16 * t + 8 * (j >> 2) + 4 * (l >> 5) + (j & 3);
                            out[(((((size_t)s * MLPF_OC + c) * 16 + t) * 2 + p) * 64 + l) * 8 + j] = (p ? Bl : Bh)[(size_t)n * Kp + k];
                        }
}

// Af: LayerNorm output, fragment ordered (K = 384).  W1f: the row-panel operand of fc1 (panel_pack_operand: chunk c at
// c * 48 KiB).  W2f: mlpf_pack_w2.  b1: fc1 bias [768].  slab: [3][M][384] floats.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
mlp_fused_h3_kernel(const _Float16* __restrict__ Af, const _Float16* __restrict__ W1f, const _Float16* __restrict__ W2f,
                    const float* __restrict__ b1, float* __restrict__ slab, const int M, const int m_tiles)
{
    static_assert(PANEL_SS == 24 && PANEL_SLOTS == 3, "k-steps of 24 slices in a 3-slot ring");
    constexpr int KS = 24;                                 // k-slices of fc1 (K = 384)
    constexpr int T = MLPF_HC + (MLPF_OC * 16) / PANEL_SS; // k-steps: 8 (fc1) + 8 (fc2)
    h3_flush_f16_denormals();
    extern __shared__ __attribute__((aligned(16))) unsigned char psm[];

    const int id = blockIdx.y * gridDim.x + blockIdx.x;   // balanced contiguous blocks of work items per XCD (see gemm_h3_panel_kernel)
    const int total = m_tiles * MLPF_SLICES, per = (total + 7) >> 3;
    const int u = (id & 7) * per + (id >> 3);
    if ((id >> 3) >= per || u >= total) return;
    const int bm = u / MLPF_SLICES, hs = u - bm * MLPF_SLICES;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = bm * 128 + wave * 32;

    // ---- weight stream: steps 0..7 = fc1 chunks 8 hs + t of W1f, steps 8..15 = 24-slice pieces of this slice's W2f ----
    const unsigned char* src1 = reinterpret_cast<const unsigned char*>(W1f) + (size_t)(MLPF_HC * hs) * PANEL_STEP_BYTES + (wave * PANEL_PIECES) * 1024;
    const unsigned char* src2 = reinterpret_cast<const unsigned char*>(W2f) + (size_t)hs * (MLPF_OC * 16 * 2048) + (wave * PANEL_PIECES) * 1024;
    const unsigned lane16 = (unsigned)lane * 16u;
    auto dma1 = [&](int t, int slot, int p) __attribute__((always_inline)) {
        const int tc = min(t, T - 1);
        const unsigned char* s = (tc < MLPF_HC ? src1 + (size_t)tc * PANEL_STEP_BYTES : src2 + (size_t)(tc - MLPF_HC) * PANEL_STEP_BYTES) + (p >> 2) * 4096;
        unsigned char* d = psm + slot * PANEL_STEP_BYTES + (wave * PANEL_PIECES) * 1024 + (p >> 2) * 4096;
        switch (p & 3) {
            case 0: __builtin_amdgcn_global_load_lds((h3_glb_void*)(s + lane16), (h3_lds_void*)d, 16, 0, 0); break;
            case 1: __builtin_amdgcn_global_load_lds((h3_glb_void*)(s + lane16), (h3_lds_void*)d, 16, 1024, 0); break;
            case 2: __builtin_amdgcn_global_load_lds((h3_glb_void*)(s + lane16), (h3_lds_void*)d, 16, 2048, 0); break;
            default: __builtin_amdgcn_global_load_lds((h3_glb_void*)(s + lane16), (h3_lds_void*)d, 16, 3072, 0); break;
        }
    };

    // ---- A panel straight into registers, then the ring's first two steps (vector memory retires in order) ----
    h16x8 ah[KS], al[KS];
    {
        const int panel = min(row0, M - 1) >> 5;
        const h16x8* ap = reinterpret_cast<const h16x8*>(Af) + (size_t)panel * KS * 2 * 64 + lane;
#pragma unroll
        for (int q = 0; q < KS; ++q) { ah[q] = ap[(q * 2 + 0) * 64]; al[q] = ap[(q * 2 + 1) * 64]; }
    }
#pragma unroll
    for (int t = 0; t < PANEL_SLOTS - 1; ++t)
#pragma unroll
        for (int p = 0; p < PANEL_PIECES; ++p) dma1(t, t, p);
    float* bias_s = reinterpret_cast<float*>(psm + PANEL_RING_BYTES);      // this slice's 256 fc1 biases
    if (tid < 256) bias_s[tid] = b1[256 * hs + tid];

    const int g = lane >> 5;
    int slot_r = 0, slot_w = PANEL_SLOTS - 1;
    auto step_begin = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(PANEL_PIECES * (PANEL_SLOTS - 2)) : "memory");     // this step landed (own pieces); own reads of the previous one returned
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto step_end = [&]() __attribute__((always_inline)) {
        slot_r = slot_r + 1 == PANEL_SLOTS ? 0 : slot_r + 1;
        slot_w = slot_w + 1 == PANEL_SLOTS ? 0 : slot_w + 1;
    };
#define UU3D_MLPF_READ(i, kk) \
    asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" \
                 : "=&v"(wh[i]), "=&v"(wl[i]) : "v"(sb), "i"((kk) * 2048), "i"((kk) * 2048 + 1024))

    // ================= fc1: H^T chunk c = W1^T[32 hidden][384] X^T, then bias + ReLU + split into fc2's B fragments =================
    h16x8 hh[MLPF_HC][2], hl[MLPF_HC][2];                  // hidden slice of this panel: [chunk][16-deep k-slice] hi / lo
    // registers 8 s + j of a chunk's accumulator -> element j of k-slice s (hidden 16 s + 8 (j >> 2) + 4 g + (j & 3))
    auto finish_reg = [&](int c, int r, const f32x16& p0, const f32x16& p1, const f32x4 (&bv)[4]) __attribute__((always_inline)) {
        const float v = fmaxf(p0[r] + p1[r] * (1.0f / H3_SCALE) + bv[r >> 2][r & 3], 0.f);
        const _Float16 h = h3_hi(v);
        hh[c][r >> 3][r & 7] = h;
        hl[c][r >> 3][r & 7] = (_Float16)((v - (float)h) * H3_SCALE);
    };
    f32x16 fa0, fa1, fb0, fb1;
    auto fc1_chunk = [&](int c, f32x16& acc0, f32x16& acc1, const f32x16& p0, const f32x16& p1) __attribute__((always_inline)) {
        // the previous chunk's 16 bias values of this lane (hidden 32 (c-1) + 8 j + 4 g + i): read BEFORE the step's wait, whose
        // lgkmcnt(0) covers them -- nothing but the asm fragment reads may be outstanding under the counted waits below
        f32x4 bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const f32x4*>(bias_s + 32 * max(c - 1, 0) + 8 * j + 4 * g);
        step_begin();
        asm volatile("" : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]));      // defined from here on: no compiler wait later
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        const unsigned sb = (unsigned)(uintptr_t)(h3_lds_void*)(psm + slot_r * PANEL_STEP_BYTES + lane * 16);
        h16x8 wh[3], wl[3];
        UU3D_MLPF_READ(0, 0);
        UU3D_MLPF_READ(1, 1);
#pragma unroll
        for (int kk = 0; kk < PANEL_SS; ++kk) {
            if (kk + 2 < PANEL_SS) UU3D_MLPF_READ((kk + 2) % 3, kk + 2);
            asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(wh[kk % 3]), "+v"(wl[kk % 3]) : "i"(panel_wait_count(kk)));
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[kk % 3], ah[kk], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[kk % 3], al[kk], acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[kk % 3], ah[kk], acc1, 0, 0, 0);
            if (c > 0 && kk < 16) finish_reg(c - 1, kk, p0, p1, bv);     // the previous chunk's epilogue, spread over this chunk's MFMAs
            if (kk & 1) dma1(c + PANEL_SLOTS - 1, slot_w, kk >> 1);
        }
        step_end();
    };
#pragma unroll
    for (int c = 0; c < MLPF_HC; c += 2) {
        fc1_chunk(c, fa0, fa1, fb0, fb1);
        fc1_chunk(c + 1, fb0, fb1, fa0, fa1);
    }
    {   // last chunk's epilogue
        f32x4 bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const f32x4*>(bias_s + 32 * (MLPF_HC - 1) + 8 * j + 4 * g);
#pragma unroll
        for (int r = 0; r < 16; ++r) finish_reg(MLPF_HC - 1, r, fb0, fb1, bv);
    }

    // ================= fc2 (partial): Y^T chunk n = W2^T[32 out][256 hidden of this slice] H^T =================
    // flattened slice index f = 16 n + t; step u holds f in [24 u, 24 u + 24).  Three chunks = two steps = one unit.
    const int valid = min(32, M - row0);
    const int token = row0 + (lane & 31);
    float* const srow = slab + ((size_t)hs * M + min(token, M - 1)) * 384 + 4 * g;
    auto store_chunk = [&](int n, const f32x16& a0, const f32x16& a1) __attribute__((always_inline)) {
        if ((lane & 31) < valid) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = a0[4 * j + i] + a1[4 * j + i] * (1.0f / H3_SCALE);
                *reinterpret_cast<f32x4*>(srow + 32 * n + 8 * j) = v;
            }
        }
    };
    for (int unit = 0; unit < MLPF_OC / 3; ++unit) {        // runtime loop: the body (48 slices) is unrolled, hidden fragments indexed statically
        f32x16 y0[3], y1[3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) { y0[c][r] = 0.f; y1[c][r] = 0.f; }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            step_begin();
            const unsigned sb = (unsigned)(uintptr_t)(h3_lds_void*)(psm + slot_r * PANEL_STEP_BYTES + lane * 16);
            h16x8 wh[3], wl[3];
            UU3D_MLPF_READ(0, 0);
            UU3D_MLPF_READ(1, 1);
#pragma unroll
            for (int kk = 0; kk < PANEL_SS; ++kk) {
                const int f = 24 * st + kk, c = f >> 4, t = f & 15;        // chunk inside the unit, k-slice inside the chunk
                if (kk + 2 < PANEL_SS) UU3D_MLPF_READ((kk + 2) % 3, kk + 2);
                asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(wh[kk % 3]), "+v"(wl[kk % 3]) : "i"(panel_wait_count(kk)));
                y0[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[kk % 3], hh[t >> 1][t & 1], y0[c], 0, 0, 0);
                y1[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[kk % 3], hl[t >> 1][t & 1], y1[c], 0, 0, 0);
                y1[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[kk % 3], hh[t >> 1][t & 1], y1[c], 0, 0, 0);
                if (kk & 1) dma1(MLPF_HC + 2 * unit + st + PANEL_SLOTS - 1, slot_w, kk >> 1);
                if (t == 15) store_chunk(3 * unit + c, y0[c], y1[c]);
            }
            step_end();
        }
    }
#undef UU3D_MLPF_READ
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the clamped tail DMAs must not outlive the LDS allocation
}

// ------------------------------------------------------------------------------------------------
// x[row] += bias + slab0[row] + slab1[row] + slab2[row] (fixed order), optionally xa[row] = x[row] + pe[row % period], then
// LayerNorm (two-pass, eps inside the root) of x (of xa when given), written as the fragment-ordered hi / lo planes of the next
// panel GEMM's A operand: ln_split_frag_kernel with the fused MLP's combine in front.  D = 16 KS.
template <int KS, int ROWS = 8>
__global__ void __launch_bounds__(16 * ROWS)
ln_res_split_frag_kernel(float* __restrict__ x, const int ld, const int M, const float eps, const float* __restrict__ bias,
                         const float* __restrict__ slab, float* __restrict__ xa, const float* __restrict__ pe, const int period,
                         const float* __restrict__ gamma, const float* __restrict__ beta, _Float16* __restrict__ Af)
{
    constexpr int D = 16 * KS, NV = D / 64;
    h3_flush_f16_denormals();
    const int tid = threadIdx.x, j = tid & 15, lr = tid >> 4;
    const int row = blockIdx.x * ROWS + lr;
    const int rc = min(row, M - 1);
    const size_t ro = (size_t)rc * ld, so = (size_t)rc * D, ss = (size_t)M * D;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = 4 * (j + 16 * i);
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + ro + c), bv = *reinterpret_cast<const f32x4*>(bias + c);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(slab + so + c), s1 = *reinterpret_cast<const f32x4*>(slab + ss + so + c),
                    s2 = *reinterpret_cast<const f32x4*>(slab + 2 * ss + so + c);
        f32x4 y = xv + (((s0 + s1) + s2) + bv);            // residual + (fc2 output incl. bias), slices in order
        if (row < M) *reinterpret_cast<f32x4*>(x + ro + c) = y;
        if (xa != nullptr) {
            y = y + *reinterpret_cast<const f32x4*>(pe + (size_t)(rc % period) * ld + c);
            if (row < M) *reinterpret_cast<f32x4*>(xa + ro + c) = y;
        }
        v[i] = y;
        s += (y[0] + y[1]) + (y[2] + y[3]);
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
    _Float16* base = Af + (size_t)(row >> 5) * KS * 2 * 512 + (row & 31) * 8;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = 4 * (j + 16 * i);
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + c);
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float inv = rstd * g[e]; y[e] = v[i][e] * inv + (bt[e] - mean * inv); }
        h16x4 hi, lo;
        h3_split(y, hi, lo);
        _Float16* d = base + (size_t)(c >> 4) * 2 * 512 + ((c >> 3) & 1) * 256 + (c & 4);
        *reinterpret_cast<h16x4*>(d) = hi;
        *reinterpret_cast<h16x4*>(d + 512) = lo;
    }
}

}  // namespace uu3d
