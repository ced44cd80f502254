// uu3d_gemm_panel2.h -- the row-panel f16x3 GEMM of uu3d_gemm_panel.h with HALF the LDS fragment reads per MFMA.
//
// gemm_h3_panel_kernel keeps a wave's 32-row A panel over the whole contraction (K = 384: 192 registers) resident and reads two
// weight fragments (2 KiB) from LDS per three MFMAs; four waves do that against one LDS pipe while the LDS-DMA writes the next
// chunk into it, and the loop runs at ~63 % of the matrix rate the chip sustains (26.6 against 16.8 ns per MFMA; a probe build
// that issues every MFMA twice per fragment read, tools/gemm_panel2_exp with -DUU3D_PANEL_PROBE_DOUBLE, ran at 21.8 ns).  Here the
// same 192 registers hold TWO row panels over HALF the contraction:
//   * a workgroup = 4 waves = 2 pairs of row panels (64 tokens) x 2 halves of K: wave (q, h) keeps the 12 k-slices
//     [12 h, 12 h + 12) of panels 2 q and 2 q + 1 resident and computes, for every 32-column chunk, both panels' PARTIAL products
//     over its half of K: 6 MFMAs per fragment pair read;
//   * the weight stream is unchanged: fragment-ordered chunks of 48 KiB through the 3-slot LDS ring by LDS-DMA, 12 pieces of 1 KiB
//     per wave and k-step, counted vmcnt waits, one barrier per chunk; a wave reads the 12 slices of its half;
//   * the partial tiles of a chunk meet through a 16 KiB LDS buffer [pair][panel][lane][16 floats]: for chunk c the wave with
//     h != (c & 1) writes its 2 x 16 combined values (acc0 + acc1 / 2048, + bias) at the end of the chunk, the wave with
//     h == (c & 1) reads them after the next barrier, adds its own and runs the epilogue interleaved with the MFMAs of chunk c + 1
//     -- the roles alternate, so the buffer needs no second copy (the writer of step c + 1 is the reader of step c + 1's start) and
//     both waves carry half the epilogues;
//   * LDS: 144 KiB ring + 16 KiB partials = 160 KiB exactly, so the bias does not go through LDS: the SENDING wave loads its
//     lane's bias value by name at the start of the step and adds it to the partial tiles at the end (vmcnt(12): everything but
//     the step's DMA pieces has retired by then).
// Sum order: (slices 0..11) + (slices 12..23) instead of one chain over 24 -- deterministic, the last bits differ from
// gemm_h3_panel_kernel.  K = 384 only (every LayerNorm-fed Dense layer of the temporal / strided blocks).
// RESULT (tools/gemm_panel2_exp, M = 9088, N = 1152, S = 3): correct (max error 1.2e-6), 34.1 us against 30.1 us of
// gemm_h3_panel_kernel.  A second accumulator set per panel does not fit (512 registers + scratch: 49.7 us), so the partial tiles
// are combined at the end of every chunk with the matrix pipe drained; that costs more than the halved fragment reads give.
// NOT part of the library.
// (The first form of this file split K over two waves PER SIMD, 8 waves of one panel each, to fill issue gaps: 31.0 against
// 29.5 us at M = 9088, N = 1152 -- the loop does not wait for issue slots.  DESIGN.md section 11.)
#pragma once
#ifdef UU3D_PANEL_PROBE_DOUBLE
#include "uu3d_gemm_panel_probe.h"      // gemm_h3_panel_kernel with every MFMA issued twice per fragment read
#else
#include "../../uplift-upsample-3dhpe_amd/csrc/uu3d_gemm_panel.h"
#endif

namespace uu3d {

static constexpr int PANEL2_X_BYTES = 2 * 2 * 64 * 16 * 4;                        // partial tiles [pair][panel][lane][16] floats
static constexpr size_t PANEL2_LDS_TOTAL = PANEL_RING_BYTES + PANEL2_X_BYTES;   // 160 KiB
static_assert(PANEL_SS == 24 && PANEL_SLOTS == 3 && PANEL_PIECES == 12, "one chunk of K = 384 per k-step, 3 ring slots, 12 pieces per wave");
constexpr int panel2_wait_count(const int g) { return g + 2 < 12 ? 4 : (g + 1 < 12 ? 2 : 0); }

// C[M][N] = A[M][384] B + colv;  A / B fragment ordered as for gemm_h3_panel_kernel, same grid and block size.
template <class EP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
gemm_h3_panel2_kernel(const _Float16* __restrict__ Af, const _Float16* __restrict__ Bf, const float* __restrict__ colv,
                      const int M, const int m_tiles, const int splits, const int chunks_per_wg, const EP ep)
{
    constexpr int KS = 24, HS = 12;                        // k-slices of the product / of one wave
    h3_flush_f16_denormals();                              // the epilogue may split its result
    extern __shared__ __attribute__((aligned(16))) unsigned char psm[];

    const int id = blockIdx.y * gridDim.x + blockIdx.x;    // balanced contiguous blocks of work items per XCD (see gemm_h3_panel_kernel)
    const int total = m_tiles * splits, per = (total + 7) >> 3;
    const int u = (id & 7) * per + (id >> 3);
    if ((id >> 3) >= per || u >= total) return;
    const int bm = u / splits, ns = u - bm * splits;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pq = wave & 1, kh = wave >> 1;               // pair of row panels, half of the contraction
    const int row0 = bm * 128 + pq * 64;                   // first row of the pair
    const int chunk0 = ns * chunks_per_wg;
    const int T = chunks_per_wg;                           // k-steps (one chunk each)

    // ---- weight stream: chunk t -> ring slot t % 3; each wave moves 12 of its 48 pieces of 1 KiB ----
    const unsigned char* bsrc = reinterpret_cast<const unsigned char*>(Bf) + (size_t)chunk0 * PANEL_STEP_BYTES + (wave * PANEL_PIECES) * 1024;
    const unsigned lane16 = (unsigned)lane * 16u;
    auto dma1 = [&](int t, int slot, int p) __attribute__((always_inline)) {       // piece p (0..11) of this wave's share of k-step t
        const unsigned char* s = bsrc + (size_t)min(t, T - 1) * PANEL_STEP_BYTES + (p >> 2) * 4096;
        unsigned char* d = psm + slot * PANEL_STEP_BYTES + (wave * PANEL_PIECES) * 1024 + (p >> 2) * 4096;
        switch (p & 3) {
            case 0: __builtin_amdgcn_global_load_lds((h3_glb_void*)(s + lane16), (h3_lds_void*)d, 16, 0, 0); break;
            case 1: __builtin_amdgcn_global_load_lds((h3_glb_void*)(s + lane16), (h3_lds_void*)d, 16, 1024, 0); break;
            case 2: __builtin_amdgcn_global_load_lds((h3_glb_void*)(s + lane16), (h3_lds_void*)d, 16, 2048, 0); break;
            default: __builtin_amdgcn_global_load_lds((h3_glb_void*)(s + lane16), (h3_lds_void*)d, 16, 3072, 0); break;
        }
    };

    // ---- this wave's half of the two A panels: 2 x 2 x 12 fragments straight into registers (panels past M: clamped, never stored) ----
    h16x8 ah[2][HS], al[2][HS];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int panel = min(row0 + 32 * j, M - 1) >> 5;
        const h16x8* ap = reinterpret_cast<const h16x8*>(Af) + ((size_t)panel * KS + kh * HS) * 2 * 64 + lane;
#pragma unroll
        for (int q = 0; q < HS; ++q) { ah[j][q] = ap[(q * 2 + 0) * 64]; al[j][q] = ap[(q * 2 + 1) * 64]; }
    }
#pragma unroll
    for (int t = 0; t < PANEL_SLOTS - 1; ++t)
#pragma unroll
        for (int p = 0; p < PANEL_PIECES; ++p) dma1(t, t, p);

    float* const xbuf = reinterpret_cast<float*>(psm + PANEL_RING_BYTES) + (pq * 128 + lane) * 16;     // this lane's 16 partial values of panel 0; panel 1: + 64 * 16
    const unsigned xa = (unsigned)(uintptr_t)(h3_lds_void*)xbuf;
    int slot_r = 0, slot_w = PANEL_SLOTS - 1;
    const int crow = (lane >> 5) * 4, ccol = lane & 31;
    const int valid = min(64, M - row0);                   // wave-uniform: rows of this pair that exist (<= 0: none)
    const float* const bias_p = colv + chunk0 * 32 + ccol;

    struct Acc { f32x16 a0[2], a1[2]; };                   // hi-hi and cross-term accumulators of the two panels
    // KH = this wave's half (compile time inside), PAR = parity of the chunk, WHOLE = every row of the pair exists.  own = this
    // wave's combined partial tiles of the chunk it finishes next (a second accumulator set instead costs 32 registers more: scratch)
    auto chunk = [&](auto kh_tag, auto par_tag, auto whole_tag, int c, Acc& acc, f32x16 (&own)[2]) __attribute__((always_inline)) {
        constexpr int KH = decltype(kh_tag)::value, PAR = decltype(par_tag)::value;
        constexpr bool WHOLE = decltype(whole_tag)::value;
        constexpr bool OWN_THIS = (PAR == KH);             // this wave finishes chunk c (in the next step); otherwise it sends its partials
        const bool fin = !OWN_THIS && c > 0;               // ... and finishes chunk c - 1 during this step
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(PANEL_PIECES * (PANEL_SLOTS - 2)) : "memory");   // chunk c landed (own pieces); own LDS traffic retired
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                      // ... everybody's; the partner's partial tiles of chunk c - 1 are in the buffer
        __builtin_amdgcn_sched_barrier(0);
        float bias_s = 0.f;
        if (!OWN_THIS)                                     // the sender's bias of chunk c: issued before this step's stores and DMAs
            asm volatile("global_load_dword %0, %1, off" : "=v"(bias_s) : "v"(bias_p + c * 32) : "memory");
        f32x4 xp[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) xp[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (fin) {
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48"
                         : "=&v"(xp[0][0]), "=&v"(xp[0][1]), "=&v"(xp[0][2]), "=&v"(xp[0][3]) : "v"(xa) : "memory");
            asm volatile("ds_read_b128 %0, %4 offset:4096\n\tds_read_b128 %1, %4 offset:4112\n\tds_read_b128 %2, %4 offset:4128\n\tds_read_b128 %3, %4 offset:4144"
                         : "=&v"(xp[1][0]), "=&v"(xp[1][1]), "=&v"(xp[1][2]), "=&v"(xp[1][3]) : "v"(xa) : "memory");
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc.a0[j][r] = 0.f; acc.a1[j][r] = 0.f; }
        const unsigned sb = (unsigned)(uintptr_t)(h3_lds_void*)(psm + slot_r * PANEL_STEP_BYTES + (KH * HS) * 2048 + lane * 16);
        h16x8 bh[3], bl[3];
#define UU3D_PANEL2_READ(i, kk) \
        asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" \
                     : "=&v"(bh[i]), "=&v"(bl[i]) : "v"(sb), "i"((kk) * 2048), "i"((kk) * 2048 + 1024))
        UU3D_PANEL2_READ(0, 0);
        UU3D_PANEL2_READ(1, 1);
        auto emit = [&](int j, int r) __attribute__((always_inline)) {
            const int lr = 32 * j + 8 * (r >> 2) + crow + (r & 3);
            const float v = own[j][r] + xp[j][r >> 2][r & 3];   // the partner's partial carries the bias
            if (WHOLE || lr < valid) ep.store(row0 + lr, (chunk0 + c - 1) * 32 + ccol, v);
        };
#pragma unroll
        for (int kk = 0; kk < HS; ++kk) {
            if (kk + 2 < HS) UU3D_PANEL2_READ((kk + 2) % 3, kk + 2);
            if (kk == 0)    // LDS returns in order: the partial tiles (issued first) are back with the first fragment pair
                asm volatile("s_waitcnt lgkmcnt(%10)" : "+v"(bh[0]), "+v"(bl[0]), "+v"(xp[0][0]), "+v"(xp[0][1]), "+v"(xp[0][2]), "+v"(xp[0][3]),
                             "+v"(xp[1][0]), "+v"(xp[1][1]), "+v"(xp[1][2]), "+v"(xp[1][3]) : "i"(panel2_wait_count(0)));
            else
                asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(bh[kk % 3]), "+v"(bl[kk % 3]) : "i"(panel2_wait_count(kk)));
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc.a0[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j][kk], bh[kk % 3], acc.a0[j], 0, 0, 0);
                acc.a1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j][kk], bl[kk % 3], acc.a1[j], 0, 0, 0);
                acc.a1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[j][kk], bh[kk % 3], acc.a1[j], 0, 0, 0);
            }
            if (fin && kk < 8) {                           // chunk c - 1's epilogue in the shadow of these MFMAs: 4 of its 32 stores per slice
                emit(0, 2 * kk); emit(0, 2 * kk + 1); emit(1, 2 * kk); emit(1, 2 * kk + 1);
            }
            dma1(c + PANEL_SLOTS - 1, slot_w, kk);        // refill of the slot read in step c - 1, spread over the step
        }
#undef UU3D_PANEL2_READ
        if (OWN_THIS) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) own[j][r] = acc.a0[j][r] + acc.a1[j][r] * (1.0f / H3_SCALE);
        } else {                                           // send this wave's partial tiles of chunk c (+ bias) to its partner
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(bias_s) : "i"(PANEL_PIECES) : "memory");       // all but the step's DMA pieces retired: the bias is here
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = (acc.a0[j][4 * g + i] + acc.a1[j][4 * g + i] * (1.0f / H3_SCALE)) + bias_s;
                    *reinterpret_cast<f32x4*>(xbuf + j * 1024 + 4 * g) = v;
                }
        }
        slot_r = slot_r + 1 == PANEL_SLOTS ? 0 : slot_r + 1;
        slot_w = slot_w + 1 == PANEL_SLOTS ? 0 : slot_w + 1;
    };

    auto run = [&](auto kh_tag, auto whole_tag) __attribute__((always_inline)) {
        constexpr int KH = decltype(kh_tag)::value;
        constexpr bool WHOLE = decltype(whole_tag)::value;
        Acc A;
        f32x16 own[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) own[j][r] = 0.f;
        for (int c = 0; c < chunks_per_wg; c += 2) {
            chunk(kh_tag, std::integral_constant<int, 0>{}, whole_tag, c, A, own);
            if (c + 1 < chunks_per_wg) chunk(kh_tag, std::integral_constant<int, 1>{}, whole_tag, c + 1, A, own);
        }
        // ---- last chunk: its owner takes the partner's partial tiles after one more barrier ----
        const int c = chunks_per_wg - 1;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // the clamped tail DMAs must not outlive the LDS allocation
        __builtin_amdgcn_s_barrier();
        if ((c & 1) == KH) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lr = 32 * j + 8 * (r >> 2) + crow + (r & 3);
                    const float v = own[j][r] + xbuf[j * 1024 + r];
                    if (WHOLE || lr < valid) ep.store(row0 + lr, (chunk0 + c) * 32 + ccol, v);
                }
        }
    };
    if (kh == 0) { if (valid == 64) run(std::integral_constant<int, 0>{}, std::true_type{}); else run(std::integral_constant<int, 0>{}, std::false_type{}); }
    else         { if (valid == 64) run(std::integral_constant<int, 1>{}, std::true_type{}); else run(std::integral_constant<int, 1>{}, std::false_type{}); }
}

}  // namespace uu3d
