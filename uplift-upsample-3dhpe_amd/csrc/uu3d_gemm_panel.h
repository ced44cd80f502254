// uu3d_gemm_panel.h -- f16x3 "row panel" GEMM: A fragments resident in registers, the weight operand streamed.
//
// The tiled kernels of uu3d_gemm_h3.h restart a 12-iteration k-loop in every 64 x 128 tile, re-stage the same A rows
// in each of the N / 128 workgroups along N, and move four f16 planes through LDS per k-tile.  Here the roles are
// fixed differently:
//
//   * BOTH operands are stored in MFMA FRAGMENT ORDER, one 1 KiB fragment = [lane][8 halfs]:
//         A (activations):  [32-row panel][16-deep k-slice][plane hi|lo]            written by the producing kernel
//         B (weights):      [32-column chunk][k-step of 12 slices][k-slice][plane]  written once at commit time
//     so a wave loads its panel with 2 K/16 fully coalesced 16-byte loads straight into the registers the MFMAs read
//     (no LDS transposition, no arithmetic), and a k-step of B is one linear 24 KiB piece that goes global -> LDS by
//     global_load_lds_dwordx4 with no swizzle; a B fragment is one conflict-free ds_read_b128 at lane * 16;
//   * a wave owns a PANEL of 32 token rows over the full contraction length and keeps its 2 K/16 A fragments
//     resident for the whole kernel (K = 384: 192 registers);
//   * a workgroup = 4 waves = 128 rows shares the weight stream through a 3-slot LDS ring (3 x 48 KiB), one barrier per
//     k-step of 24 slices (72 MFMAs per wave), two k-steps in flight (6 slots of 12 slices: 3 % slower);
//   * per k-slice a wave issues 2 ds_read_b128 and 3 MFMAs (ah*bh -> acc0; ah*bl, al*bh -> acc1); the fragment reads
//     run two slices ahead (asm with counted lgkmcnt waits: hipcc sinks plain loads to their use);
//   * the epilogue of chunk c-1 is interleaved with the MFMAs of chunk c (two accumulator sets): issued as a block
//     between chunks it cost 1340 cycles per chunk with the matrix pipe idle;
//   * ~330 registers per wave, so ONE wave per SIMD / one workgroup per CU: everything is software pipelined instead
//     of relying on co-resident waves.
//
// Grid: (M / 128) row tiles x S column ranges of N / (32 S) chunks each, launched as (8 S, ceil(ceil(items / 8) / S)).  Measured (tools/gemm_panel_exp): see DESIGN.md.
// Forms that measured slower and live in tools/ now: LayerNorm statistics + split in the kernel's own prologue
// (tools/gemm_panel_lnfold_exp.h), LayerNorm folded into the operand with producer-side fragment stores ("LNF") and the
// accumulating variant for the residual Dense layers (git history of round 1; docs/HISTORY.md E.11).
#pragma once
#include "uu3d_gemm_h3.h"
#include <type_traits>

namespace uu3d {

#ifndef UU3D_PANEL_SS
#define UU3D_PANEL_SS 24
#endif
static constexpr int PANEL_SS = UU3D_PANEL_SS;           // k-slices per k-step (12 or 24)
static constexpr int PANEL_STEP_BYTES = PANEL_SS * 2 * 1024;   // one k-step of B: slices x 2 planes x 1 KiB
static constexpr int PANEL_SLOTS = 144 / (2 * PANEL_SS); // ring depth (144 KiB): all but one slot in flight while one is consumed
static constexpr int PANEL_PIECES = PANEL_SS / 2;        // 1 KiB pieces of a k-step each of the 4 waves moves
static constexpr size_t PANEL_RING_BYTES = (size_t)PANEL_SLOTS * PANEL_STEP_BYTES;   // 144 KiB
static constexpr int PANEL_COLV_FLOATS = 1024;           // per-column epilogue vector of this workgroup's columns (<= 32 chunks)
static constexpr size_t PANEL_LDS_TOTAL = PANEL_RING_BYTES + PANEL_COLV_FLOATS * sizeof(float);

// halfs in the fragment-ordered operands (A is allocated in whole 32-row panels)
__host__ __device__ inline constexpr size_t panel_b_halfs(int N, int K) { return (size_t)(N / 32) * (K / 16) * 2 * 512; }
__host__ __device__ inline constexpr size_t panel_a_halfs(int M, int K) { return (size_t)((M + 31) / 32) * (K / 16) * 2 * 512; }
// element (row, k) of the hi plane in the A operand; the lo plane element is 512 halfs further
__host__ __device__ inline size_t panel_a_index(int row, int k, int K) {
    return ((((size_t)(row >> 5) * (K / 16) + (k >> 4)) * 2) * 64 + ((k >> 3) & 1) * 32 + (row & 31)) * 8 + (k & 7);
}

// Host side: fragment-ordered B planes from the transposed, padded Bt[N][Kp] (k contiguous) of the tiled kernels.
// hi / lo as produced by the commit-time split (lo pre-scaled by 2048).  K % 192 == 0, N % 32 == 0.
inline void panel_pack_operand(const _Float16* Bh, const _Float16* Bl, int N, int K, int Kp, _Float16* out) {
    const int steps = K / 192;
    for (int c = 0; c < N / 32; ++c)
        for (int st = 0; st < steps; ++st)
            for (int kk = 0; kk < 12; ++kk)
                for (int p = 0; p < 2; ++p)
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 8; ++j) {
                            const int n = 32 * c + (l & 31), k = st * 192 + kk * 16 + (l >> 5) * 8 + j;
                            out[((((size_t)(c * steps + st) * 12 + kk) * 2 + p) * 64 + l) * 8 + j] = (p ? Bl : Bh)[(size_t)n * Kp + k];
                        }
}

// ---- epilogues: v = acc + colv[col] ----
// Addresses are a scalar base + a 32-bit BYTE offset per lane (global_store ... saddr): a 64-bit address pair per unrolled
// output register cost the fc1 kernel its register budget (512 + scratch).  The launcher keeps M * ldo * 4 below 2^32.
struct PanelEpBias {           // out[row][col] = v
    static constexpr bool kResidual = false; static constexpr int kStores = 1;
    float* __restrict__ out; int ldo;
    __device__ __forceinline__ void store(int row, int col, float v) const {
        const unsigned b = ((unsigned)row * (unsigned)ldo + (unsigned)col) * 4u;
        *reinterpret_cast<float*>(reinterpret_cast<char*>(out) + b) = v;
    }
};
struct PanelEpBiasRelu {       // out[row][col] = max(v, 0)   (training-mode forward: the hidden activations stay f32 for the backward pass)
    static constexpr bool kResidual = false; static constexpr int kStores = 1;
    float* __restrict__ out; int ldo;
    __device__ __forceinline__ void store(int row, int col, float v) const {
        const unsigned b = ((unsigned)row * (unsigned)ldo + (unsigned)col) * 4u;
        *reinterpret_cast<float*>(reinterpret_cast<char*>(out) + b) = fmaxf(v, 0.f);
    }
};
struct PanelEpBiasSplitQ {     // [q | k | v] as row-major hi / lo planes for attn_h3_kernel; q (columns < qcols) times qscale = log2(e) / sqrt(d_h)
    static constexpr bool kResidual = false; static constexpr int kStores = 2;
    _Float16* __restrict__ Oh; _Float16* __restrict__ Ol; int ldo, qcols; float qscale;
    __device__ __forceinline__ void store(int row, int col, float x) const {
        const float v = col < qcols ? x * qscale : x;
        const _Float16 h = h3_hi(v);
        const unsigned b = ((unsigned)row * (unsigned)ldo + (unsigned)col) * 2u;
        *reinterpret_cast<_Float16*>(reinterpret_cast<char*>(Oh) + b) = h;
        *reinterpret_cast<_Float16*>(reinterpret_cast<char*>(Ol) + b) = (_Float16)((v - (float)h) * H3_SCALE);
    }
};
struct PanelEpBiasReluSplit {  // ReLU(v) as row-major hi / lo planes [M][ldo] (the A operand of the tiled LDS-DMA kernel)
    static constexpr bool kResidual = false; static constexpr int kStores = 2;
    _Float16* __restrict__ Oh; _Float16* __restrict__ Ol; int ldo;
    __device__ __forceinline__ void store(int row, int col, float x) const {
        const float v = fmaxf(x, 0.f);
        const _Float16 h = h3_hi(v);
        const unsigned b = ((unsigned)row * (unsigned)ldo + (unsigned)col) * 2u;
        *reinterpret_cast<_Float16*>(reinterpret_cast<char*>(Oh) + b) = h;
        *reinterpret_cast<_Float16*>(reinterpret_cast<char*>(Ol) + b) = (_Float16)((v - (float)h) * H3_SCALE);
    }
};
struct PanelEpBiasResidual {   // x[row][col] += v   (the attention projection on the residual stream, in place: every element is read and written by one lane)
    static constexpr bool kResidual = true; static constexpr int kStores = 1;
    float* __restrict__ x; int ldo;
    // Requests x[row][col] into an ACCUMULATION register and returns at once: the value is valid only after a counted vmcnt wait
    // that names the register (gemm_h3_panel_kernel does both).  Written as asm because (a) a load hipcc can see gets its own
    // s_waitcnt, and with the arch registers full it parks the value in an AGPR right away -- a vmcnt(0) directly behind the load,
    // which drains the weight ring; (b) "=a" puts it where the kernel has room (one wave per SIMD: 256 + 256 registers).
    __device__ __forceinline__ void request(int row, int col, float& dst) const {
        const unsigned b = ((unsigned)row * (unsigned)ldo + (unsigned)col) * 4u;
        asm volatile("global_load_dword %0, %1, %2" : "=a"(dst) : "v"(b), "s"(x) : "memory");
    }
    __device__ __forceinline__ void request8(int row, int col, float& dst) const {          // the same into an architectural register (gemm_h3_panel8_kernel: 256 registers per wave)
        const unsigned b = ((unsigned)row * (unsigned)ldo + (unsigned)col) * 4u;
        asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(b), "s"(x) : "memory");
    }
    __device__ __forceinline__ void store(int row, int col, float v) const {
        const unsigned b = ((unsigned)row * (unsigned)ldo + (unsigned)col) * 4u;
        *reinterpret_cast<float*>(reinterpret_cast<char*>(x) + b) = v;
    }
};
// x[row][col] += v, then -- by the workgroup that owns the whole rows (one column range per row tile: gemm_h3_panel8_kernel with
// splits == 1) -- LayerNorm 2 of the finished rows, split and written as the next panel GEMM's A fragments: the ln_split_frag launch
// between the projection and the MLP disappears (throughput schedule; round 4).  Af may be the buffer the projection's own A
// panels came from: a workgroup has read its rows' fragments into registers long before it rewrites them.
struct PanelEpBiasResidualLn : PanelEpBiasResidual {
    const float* __restrict__ gamma; const float* __restrict__ beta; float eps; _Float16* __restrict__ Af;
};
template <class EP> struct panel_ln_tail { static constexpr bool value = false; };
template <> struct panel_ln_tail<PanelEpBiasResidualLn> { static constexpr bool value = true; };

#ifndef UU3D_PANEL_LOO
#define UU3D_PANEL_LOO 0       // tools/panel8_exp: leave-out timing builds (results wrong), bit mask: 1 no refill DMA, 2 no epilogue stores, 4 no fragment reads, 8 no barrier, 16 no MFMA
#endif
#ifdef UU3D_PANEL_STAMP
__device__ unsigned long long panel_stamps[1024 * 8];   // tools/panel8_exp: s_memrealtime (100 MHz) per work item: entry, requests issued, first k-step landed, loop done, end; [5] = s_memtime ticks of the loop
#define PANEL_STAMP(...) __VA_ARGS__
#else
#define PANEL_STAMP(...)
#endif

// Counted LDS waits of the main loop.  Issue order of a k-step (all asm, in program order): B(0) B(1), then per k-slice g:
// B(g+2) | wait B(g) | 3 MFMAs, where B(g) = the two fragment reads of slice g.  LDS returns in order, so a wait for an
// operation = lgkmcnt(number of operations issued after it).
constexpr int panel_wait_count(const int g) { return g + 2 < PANEL_SS ? 4 : (g + 1 < PANEL_SS ? 2 : 0); }

// C[M][N] = A[M][K] B + colv,  K = 16 KS (KS % 24 == 0), A / B fragment ordered (see top).
// CPW > 0: chunks per workgroup known at compile time -- the chunk loop is unrolled completely.  Needed by epilogues that LOAD
// (EP::kResidual): with a loop, hipcc's wait insertion puts s_waitcnt vmcnt(0) at the loop header for the loads in flight across
// the back edge, which drains the weight ring once per iteration; in straight-line code it counts exactly.
template <int KS, class EP, int CPW = 0>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
gemm_h3_panel_kernel(const _Float16* __restrict__ Af, const _Float16* __restrict__ Bf, const float* __restrict__ colv,
                     const int M, const int m_tiles, const int splits, const int chunks_per_wg_, const EP ep)
{
    const int chunks_per_wg = CPW > 0 ? CPW : chunks_per_wg_;
    constexpr int SPC = KS / PANEL_SS;                           // k-steps per chunk
    h3_flush_f16_denormals();                              // the epilogue may split its result
    extern __shared__ __attribute__((aligned(16))) unsigned char psm[];

    // Work item u = row tile * S + column range.  The dispatcher deals workgroups to the 8 XCDs round robin; XCD x takes
    // the contiguous items [x per, (x + 1) per): the column ranges of a row tile share an L2 (A panels), and no XCD gets
    // more than ceil(items / 8) workgroups (dealing whole row tiles gave 33 on two XCDs at 82 tiles x 3: a second round).
    const int id = blockIdx.y * gridDim.x + blockIdx.x;
    const int total = m_tiles * splits, per = (total + 7) >> 3;
    const int u = (id & 7) * per + (id >> 3);
    if ((id >> 3) >= per || u >= total) return;
    const int bm = u / splits, ns = u - bm * splits;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the LDS-DMA destinations (M0) then come from SALU arithmetic
    const int row0 = bm * 128 + wave * 32;                 // this wave's panel
    const int chunk0 = ns * chunks_per_wg;
    const int T = SPC * chunks_per_wg;                     // k-steps this workgroup consumes

    // ---- weight stream: k-step t -> ring slot t % PANEL_SLOTS; each wave moves 6 of its 24 pieces of 1 KiB ----
    // the global address of a piece = scalar base (SGPR arithmetic) + lane * 16 as a 32-bit offset: no 64-bit VGPR address pairs
    const unsigned char* bsrc = reinterpret_cast<const unsigned char*>(Bf) + (size_t)chunk0 * SPC * PANEL_STEP_BYTES + (wave * PANEL_PIECES) * 1024;
    const unsigned lane16 = (unsigned)lane * 16u;
    auto dma1 = [&](int t, int slot, int p) __attribute__((always_inline)) {       // piece p of this wave's share of k-step t
        const unsigned char* s = bsrc + (size_t)min(t, T - 1) * PANEL_STEP_BYTES + (p >> 2) * 4096;
        unsigned char* d = psm + slot * PANEL_STEP_BYTES + (wave * PANEL_PIECES) * 1024 + (p >> 2) * 4096;
        // the instruction's immediate offset advances the global and the LDS address alike: one M0 value / base per 4 pieces
        // instead of a v_readfirstlane + s_mov m0 + 64-bit add for every piece (31.4 -> 30.1 us for the QKV projection)
        switch (p & 3) {
            case 0: __builtin_amdgcn_global_load_lds((h3_glb_void*)(s + lane16), (h3_lds_void*)d, 16, 0, 0); break;
            case 1: __builtin_amdgcn_global_load_lds((h3_glb_void*)(s + lane16), (h3_lds_void*)d, 16, 1024, 0); break;
            case 2: __builtin_amdgcn_global_load_lds((h3_glb_void*)(s + lane16), (h3_lds_void*)d, 16, 2048, 0); break;
            default: __builtin_amdgcn_global_load_lds((h3_glb_void*)(s + lane16), (h3_lds_void*)d, 16, 3072, 0); break;
        }
    };
    auto dma = [&](int t, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < PANEL_PIECES; ++p) dma1(t, slot, p);
    };
    PANEL_STAMP(const unsigned long long st_entry = __builtin_amdgcn_s_memrealtime(); unsigned long long st_first = 0;)

    // ---- A panel: 2 KS fragments straight into registers (panels past M: clamped to the last one, never stored) ----
    h16x8 ah[KS], al[KS];
    {
        const int panel = min(row0, M - 1) >> 5;
        const h16x8* ap = reinterpret_cast<const h16x8*>(Af) + (size_t)panel * KS * 2 * 64 + lane;
#pragma unroll
        for (int q = 0; q < KS; ++q) { ah[q] = ap[(q * 2 + 0) * 64]; al[q] = ap[(q * 2 + 1) * 64]; }
    }
    // the ring's first five k-steps go out AFTER the panel loads: vector memory retires in order, so the panel (needed
    // first) must not queue behind 120 KiB of weights
#pragma unroll
    for (int t = 0; t < PANEL_SLOTS - 1; ++t) dma(t, t);
    float* colv_s = reinterpret_cast<float*>(psm + PANEL_RING_BYTES);
    for (int i = tid; i < chunks_per_wg * 32; i += 256) colv_s[i] = colv[chunk0 * 32 + i];
    // ---- main loop ----
    // Every wait is "vmcnt(6 x k-steps left in flight)": k-step t was issued before t+1 .. t+4, whose pieces are the
    // newest loads at that point, and vector memory operations retire in order -- wherever the compiler puts the
    // epilogue's stores (or the panel loads above), the wait can only become stricter, never weaker.
    int slot_r = 0, slot_w = PANEL_SLOTS - 1;              // slot consumed / refilled in the current step
    const int crow = (lane >> 5) * 4, ccol = lane & 31;
    const int valid = min(32, M - row0);                   // wave-uniform: rows of this panel that exist (<= 0: none)
    PANEL_STAMP(const unsigned long long st_issued = __builtin_amdgcn_s_memrealtime(); const unsigned long long ck0 = __builtin_amdgcn_s_memtime();)

    auto emit = [&](int c, int r, const f32x16& p0, const f32x16& p1, float cv, float res) __attribute__((always_inline)) {
        float v = p0[r] + p1[r] * (1.0f / H3_SCALE) + cv;
        if constexpr (EP::kResidual) v += res;
#if UU3D_PANEL_LOO & 2
        if (v != 12345.678f) return;
#endif
        ep.store(row0 + 8 * (r >> 2) + crow + (r & 3), (chunk0 + c) * 32 + ccol, v);
    };
    // kResidual: the 16 residual values of chunk c are requested when chunk c STARTS and added when it is emitted, a whole chunk
    // (>= 72 MFMAs) later: read at the point of use, the wait for them would also wait for every weight piece issued before them
    // -- the whole ring -- since vector memory returns in order.  They are covered by the SAME counted wait as the ring: chunk c's
    // requests are older than its 12 refill pieces, so "all but the newest 12 have landed" at the start of chunk c + 1 includes
    // them.  Rows past M: clamped (read, never stored).
    auto fetch = [&](int c, float (&res)[16]) __attribute__((always_inline)) {
        if constexpr (EP::kResidual) {
#pragma unroll
            for (int r = 0; r < 16; ++r) ep.request(min(row0 + 8 * (r >> 2) + crow + (r & 3), M - 1), (chunk0 + c) * 32 + ccol, res[r]);
        }
    };
    // the residual registers become values here: `asm` names them so that nothing reads them earlier
#define UU3D_PANEL_RES16(r) "+a"(r[0]), "+a"(r[1]), "+a"(r[2]), "+a"(r[3]), "+a"(r[4]), "+a"(r[5]), "+a"(r[6]), "+a"(r[7]), \
                            "+a"(r[8]), "+a"(r[9]), "+a"(r[10]), "+a"(r[11]), "+a"(r[12]), "+a"(r[13]), "+a"(r[14]), "+a"(r[15])
    // WHOLE = every row of the panel exists: the 16 stores of chunk c-1 are spread over the first 16 k-slices of chunk c
    // (an MFMA holds the vector issue port for 8 of its 32 cycles).  Otherwise they are predicated and issued as a block.
    auto chunk = [&](auto whole_tag, int c, f32x16& acc0, f32x16& acc1, const f32x16& p0, const f32x16& p1, float (&rcur)[16], float (&rprev)[16]) __attribute__((always_inline)) {
        constexpr bool WHOLE = decltype(whole_tag)::value;
        const bool prev = c > 0;
        const float pcv = colv_s[max(c - 1, 0) * 32 + ccol];                       // the previous chunk's bias
#pragma unroll
        for (int st = 0; st < SPC; ++st) {
            if (EP::kResidual && st == 0) asm volatile("s_waitcnt vmcnt(%16) lgkmcnt(0)" : UU3D_PANEL_RES16(rprev) : "i"(PANEL_PIECES * (PANEL_SLOTS - 2)) : "memory");
            else
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(PANEL_PIECES * (PANEL_SLOTS - 2)) : "memory");   // k-step t landed (this wave's pieces); own reads of t-1 returned
            __builtin_amdgcn_sched_barrier(0);
#if !(UU3D_PANEL_LOO & 8)
            __builtin_amdgcn_s_barrier();                                  // ... everybody's; the slot refilled below was last read in t-1
#endif
            __builtin_amdgcn_sched_barrier(0);
            PANEL_STAMP(if (c == 0 && st == 0) st_first = __builtin_amdgcn_s_memrealtime();)
            if (!WHOLE && st == 0 && prev) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (8 * (r >> 2) + crow + (r & 3) < valid) emit(c - 1, r, p0, p1, pcv, rprev[r]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (st == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
                fetch(c, rcur);
            }
            const unsigned sb = (unsigned)(uintptr_t)(h3_lds_void*)(psm + slot_r * PANEL_STEP_BYTES + lane * 16);
            h16x8 bh[3], bl[3];
#define UU3D_PANEL_READ(i, kk) \
            asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" \
                         : "=&v"(bh[i]), "=&v"(bl[i]) : "v"(sb), "i"((kk) * 2048), "i"((kk) * 2048 + 1024))
            UU3D_PANEL_READ(0, 0);
            UU3D_PANEL_READ(1, 1);
#pragma unroll
            for (int kk = 0; kk < PANEL_SS; ++kk) {
#if !(UU3D_PANEL_LOO & 4)
                if (kk + 2 < PANEL_SS) UU3D_PANEL_READ((kk + 2) % 3, kk + 2);
                asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(bh[kk % 3]), "+v"(bl[kk % 3]) : "i"(panel_wait_count(kk)));
#endif
#if !(UU3D_PANEL_LOO & 16)
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[st * PANEL_SS + kk], bh[kk % 3], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[st * PANEL_SS + kk], bl[kk % 3], acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[st * PANEL_SS + kk], bh[kk % 3], acc1, 0, 0, 0);
#else
                asm volatile("" : "+v"(acc0), "+v"(acc1) : "v"(ah[st * PANEL_SS + kk]), "v"(al[st * PANEL_SS + kk]), "v"(bh[kk % 3]), "v"(bl[kk % 3]));
#endif
                if (WHOLE && st * PANEL_SS + kk < 16 && prev) emit(c - 1, st * PANEL_SS + kk, p0, p1, pcv, rprev[(st * PANEL_SS + kk) & 15]);
#if !(UU3D_PANEL_LOO & 1)
                if (kk & 1) dma1(c * SPC + st + PANEL_SLOTS - 1, slot_w, kk >> 1);   // the refill of the slot read in step t-1, spread over the step (-2 %)
#endif
            }
#undef UU3D_PANEL_READ
            slot_r = slot_r + 1 == PANEL_SLOTS ? 0 : slot_r + 1;
            slot_w = slot_w + 1 == PANEL_SLOTS ? 0 : slot_w + 1;
        }
    };
    f32x16 a0, a1, b0, b1;
    float ra[16], rb[16];                                  // residual values of the chunk in a / b (dead unless EP::kResidual)
#pragma unroll
    for (int r = 0; r < 16; ++r) { b0[r] = 0.f; b1[r] = 0.f; ra[r] = 0.f; rb[r] = 0.f; }
    auto run = [&](auto whole_tag) __attribute__((always_inline)) {
        if constexpr (CPW > 0) {
#pragma unroll
            for (int c = 0; c < CPW; c += 2) {
                chunk(whole_tag, c, a0, a1, b0, b1, ra, rb);
                if (c + 1 < CPW) chunk(whole_tag, c + 1, b0, b1, a0, a1, rb, ra);
            }
        } else {
            for (int c = 0; c < chunks_per_wg; c += 2) {
                chunk(whole_tag, c, a0, a1, b0, b1, ra, rb);
                if (c + 1 < chunks_per_wg) chunk(whole_tag, c + 1, b0, b1, a0, a1, rb, ra);
            }
        }
    };
    if (valid == 32) run(std::true_type{}); else run(std::false_type{});
    PANEL_STAMP(const unsigned long long st_loop = __builtin_amdgcn_s_memrealtime(); const unsigned long long ck1 = __builtin_amdgcn_s_memtime();)
    {   // last chunk
        const int c = chunks_per_wg - 1;
        const float cv = colv_s[c * 32 + ccol];
        if constexpr (EP::kResidual) {
            if (c & 1) asm volatile("s_waitcnt vmcnt(0)" : UU3D_PANEL_RES16(rb) :: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" : UU3D_PANEL_RES16(ra) :: "memory");
        }
        if (c & 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (8 * (r >> 2) + crow + (r & 3) < valid) emit(c, r, b0, b1, cv, rb[r]);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (8 * (r >> 2) + crow + (r & 3) < valid) emit(c, r, a0, a1, cv, ra[r]);
        }
    }
#undef UU3D_PANEL_RES16
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the clamped tail DMAs must not outlive the LDS allocation
    PANEL_STAMP(if (tid == 0 && u < 1024) { unsigned long long* o = panel_stamps + u * 8; o[0] = st_entry; o[1] = st_issued; o[2] = st_first; o[3] = st_loop; o[4] = __builtin_amdgcn_s_memrealtime(); o[5] = ck1 - ck0; })
}

// ------------------------------------------------------------------------------------------------
// LayerNorm (two-pass, eps inside the root) of M rows of D = 16 KS floats, written as the fragment-ordered hi / lo
// planes of a panel GEMM's A operand.  ROWS rows per workgroup, 16 threads per row, 4-float pieces interleaved across
// them, so that a wave's stores for one piece index are runs of 64 contiguous bytes of the operand.  At M = 9088:
// 6.6 us with 8 or 4 rows per workgroup, 7.0 with 16, 7.9 with 32 -- the floor of 28 MB through the Infinity Cache in
// one short launch.
// One row of ln_split_frag_body for the 16 threads (j = 0 .. 15) that share it: LayerNorm of x[row][0 .. 16 KS) and the fragment-ordered
// hi / lo planes of the row (rows >= M: read clamped, nothing written).  Same arithmetic, same order as ln_split_frag_body.
template <int KS>
__device__ __forceinline__ void ln_split_frag_row(const float* __restrict__ x, const int ld, const int M, const float eps,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  _Float16* __restrict__ Af, const int row, const int j)
{
    constexpr int D = 16 * KS, NV = D / 64;
    const float* p = x + (size_t)min(row, M - 1) * ld;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) { v[i] = *reinterpret_cast<const f32x4*>(p + 4 * (j + 16 * i)); s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]); }
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
    if (row >= M) return;
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

template <int KS, int ROWS>
__device__ __forceinline__ void ln_split_frag_body(const float* __restrict__ x, const int ld, const int M, const float eps,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   _Float16* __restrict__ Af, float2* __restrict__ stats)
{
    constexpr int D = 16 * KS, NV = D / 64;                // float4 pieces per thread
    h3_flush_f16_denormals();
    const int tid = threadIdx.x, j = tid & 15, lr = tid >> 4;
    const int row = blockIdx.x * ROWS + lr;
    const float* p = x + (size_t)min(row, M - 1) * ld;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) { v[i] = *reinterpret_cast<const f32x4*>(p + 4 * (j + 16 * i)); s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]); }
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
    if (stats != nullptr && j == 0 && row < M) stats[row] = make_float2(mean, rstd);      // (training: the backward pass reads them)
    _Float16* base = Af + (size_t)(row >> 5) * KS * 2 * 512 + (row & 31) * 8;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = 4 * (j + 16 * i);                    // first of 4 consecutive k
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

template <int KS, int ROWS = 16>
__global__ void __launch_bounds__(16 * ROWS)
ln_split_frag_kernel(const float* __restrict__ x, const int ld, const int M, const float eps,
                     const float* __restrict__ gamma, const float* __restrict__ beta, _Float16* __restrict__ Af)
{
    ln_split_frag_body<KS, ROWS>(x, ld, M, eps, gamma, beta, Af, nullptr);
}
// the same, also writing the rows' (mean, 1 / sqrt(var + eps)) -- the training-mode forward keeps them for the backward pass
template <int KS, int ROWS = 16>
__global__ void __launch_bounds__(16 * ROWS)
ln_split_frag_stats_kernel(const float* __restrict__ x, const int ld, const int M, const float eps,
                           const float* __restrict__ gamma, const float* __restrict__ beta, _Float16* __restrict__ Af, float2* __restrict__ stats)
{
    ln_split_frag_body<KS, ROWS>(x, ld, M, eps, gamma, beta, Af, stats);
}

}  // namespace uu3d
