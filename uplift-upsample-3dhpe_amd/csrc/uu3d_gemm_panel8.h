// uu3d_gemm_panel8.h -- the row-panel f16x3 GEMM of uu3d_gemm_panel.h with TWO waves per SIMD (round 4).
//
// gemm_h3_panel_kernel keeps a wave's whole 32 x 384 A panel in registers (192 of them), which leaves room for one wave per
// SIMD: every LDS-DMA issue, every epilogue store and every VALU instruction of that wave (4 cycles each when a wave is alone on
// its SIMD) comes straight out of its MFMA stream -- the matrix pipe was busy 0.27 of the time.  Here the contraction is SPLIT
// OVER A PAIR OF WAVES: waves w and w + 4 own the same 32 token rows, wave group h = w >> 2 the k-slices [12 h, 12 h + 12).
// A wave holds 24 A fragments (96 registers), fits in 256 registers, and eight waves = two per SIMD share the same 128 rows and
// the same weight stream as before: while one wave of a SIMD issues a DMA piece, waits for LDS or stores, the other one's MFMAs
// run.  Same operand formats as gemm_h3_panel_kernel (fragment-ordered A panels and B chunks), same grid.
//
//   * per chunk of 32 columns a wave does 12 k-slices x 3 MFMAs on its half of the contraction; the two partial sums of a row
//     pair meet through a 2 KiB-per-wave exchange area in LDS: each wave FINALISES 16 of the 32 rows (accumulator registers
//     0..7) and SENDS the other 16 (registers 8..15, already combined acc0 + acc1 / 2048).  Group 1 loads its A fragments with
//     the row index flipped by 16 (lane ^ 16), so in both groups "registers 0..7" are the rows the wave keeps: one code path;
//   * the ring (3 chunks x 48 KiB, chunk layout [slice][plane][lane][8]) is refilled in HALF-chunks of 24 KiB: a chunk is read
//     in two halves (k-slices kk < 6 / kk >= 6 of each group) separated by a barrier, and the half that the barrier retires is
//     requested again right behind it -- four half-chunks (96 KiB) are in flight or landed ahead of the reader at any time, a
//     piece has ~2.25 chunk times to land;
//   * two barriers per chunk.  At the barrier that opens half-interval j every wave has waited for its own pieces of half-chunk
//     j + 1 (strict counted vmcnt: the newest 9 operations are the 3 x 3 pieces of the three younger half-chunks), so behind it
//     half-chunk j + 1 is complete for everybody: the first fragment reads of a chunk are issued BEFORE the barrier that opens it.
//     The exchange area is written in the first half-interval of the next chunk and read in the second: one 16 KiB buffer,
//     every write / read pair separated by a barrier;
//   * LDS = 144 KiB ring + 16 KiB exchange = all 160 KiB: a chunk's per-column value (bias) and its residual values are requested by
//     name into registers when the chunk starts and become values at a counted wait 1.5 chunks later, where they are added;
//   * CPW (chunks per workgroup) is a template parameter: the chunk loop is straight-line code (by-name loads across a loop's
//     back edge are not safe: hipcc may move a register that has not landed).
#pragma once
#include "uu3d_gemm_panel.h"

namespace uu3d {

static constexpr int P8_CHUNK_BYTES = 24 * 2048;                      // one 32-column chunk of B: 24 slices x 2 planes x 1 KiB
static constexpr size_t P8_RING_BYTES = 3 * (size_t)P8_CHUNK_BYTES;   // 144 KiB
static constexpr size_t P8_XCHG_BYTES = 8 * 2048;                     // 8 waves x 64 lanes x 8 floats
static constexpr size_t P8_LDS_TOTAL = P8_RING_BYTES + P8_XCHG_BYTES; // 163840 = the whole LDS of a CU

#ifndef UU3D_P8_LOO
#define UU3D_P8_LOO 0          // tools/panel8_exp: leave-one-out timing builds (results wrong): 1 no DMA, 2 no stores, 3 no mid barrier, 4 no exchange
#endif

// VAR (measurement switches, tools/panel8_exp): bit 0 = a scheduling barrier behind every k-slice (the MFMAs stay between the reads
// and waits they were written between), bit 1 = the A fragments are requested by name BEHIND the first chunk's weights and waited for
// slice by slice inside chunk 0 (the first MFMA starts when 6 KiB of weights and 2 KiB of A per wave have landed, not 36 KiB).
template <class EP, int CPW, int VAR = 0>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
gemm_h3_panel8_kernel(const _Float16* __restrict__ Af, const _Float16* __restrict__ Bf, const float* __restrict__ colv,
                      const int M, const int m_tiles, const int splits, const EP ep)
{
    constexpr int KS = 24, HS = 12;
    PANEL_STAMP(const unsigned long long st_entry = __builtin_amdgcn_s_memrealtime();)
    h3_flush_f16_denormals();
    extern __shared__ __attribute__((aligned(16))) unsigned char psm[];

    // work item -> (row tile, column range): as gemm_h3_panel_kernel (contiguous items per XCD)
    const int id = blockIdx.y * gridDim.x + blockIdx.x;
    const int total = m_tiles * splits, per = (total + 7) >> 3;
    const int u = (id & 7) * per + (id >> 3);
    if ((id >> 3) >= per || u >= total) return;
    const int bm = u / splits, ns = u - bm * splits;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = wave >> 2, q = wave & 3;                 // k-half / panel of the row tile
    const int row0 = bm * 128 + q * 32;
    const int chunk0 = ns * CPW;

    // Everything below exists twice: for row tiles whose 128 rows all exist (exact operation counts in the waits) and for the ragged
    // last tile (predicated stores, strict counts).  The branch sits in FRONT of the first request by name: a register that has been
    // requested must not cross a control-flow join before its wait (the allocator may copy or spill it there -- measured: 17
    // scratch stores of fragments that had not landed when the branch sat behind the prologue).
    const bool whole = __builtin_amdgcn_readfirstlane((int)(M - bm * 128 >= 128));
    auto body = [&](auto whole_tag) __attribute__((always_inline)) {
    // ---- weight stream: half-chunk g = 2 c + j -> the k-slices [6 j, 6 j + 6) of both groups of chunk c; this wave moves the
    // 3 KiB at (12 h + 6 j) slices + q * 3 KiB of it (the same offset in the chunk's ring slot)
    const unsigned wofs = (unsigned)(h * HS * 2048 + q * 3072);
    const unsigned char* bsrc = reinterpret_cast<const unsigned char*>(Bf) + (size_t)chunk0 * P8_CHUNK_BYTES + wofs;
    const unsigned lane16 = (unsigned)lane * 16u;
    auto dma1 = [&](int c, int j, int slot, int i) __attribute__((always_inline)) {     // piece i (0..2) of half j of chunk c (clamped: a chunk past the end re-reads the last one into a free slot)
        const unsigned char* s = bsrc + (size_t)min(c, CPW - 1) * P8_CHUNK_BYTES + j * (6 * 2048);
        unsigned char* d = psm + slot * P8_CHUNK_BYTES + wofs + j * (6 * 2048);
        switch (i) {
            case 0: __builtin_amdgcn_global_load_lds((h3_glb_void*)(s + lane16), (h3_lds_void*)d, 16, 0, 0); break;
            case 1: __builtin_amdgcn_global_load_lds((h3_glb_void*)(s + lane16), (h3_lds_void*)d, 16, 1024, 0); break;
            default: __builtin_amdgcn_global_load_lds((h3_glb_void*)(s + lane16), (h3_lds_void*)d, 16, 2048, 0); break;
        }
    };

    // ---- A half panel: 24 fragments straight into registers; group 1 with the row index flipped by 16 ----
    constexpr bool STREAM = (VAR & 2) != 0;
    h16x8 ah[HS], al[HS];
    if constexpr (STREAM) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int i = 0; i < 3; ++i) dma1(0, g, 0, i);
        const int panel = __builtin_amdgcn_readfirstlane(min(row0, M - 1) >> 5);
        const unsigned char* abase = reinterpret_cast<const unsigned char*>(Af) + ((size_t)panel * KS + HS * h) * 2048;
        const unsigned avo = (unsigned)(lane ^ (h << 4)) * 16u;
#pragma unroll
        for (int s = 0; s < HS; ++s) {
            const unsigned char* ps = abase + s * 2048;
            asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024" : "=&v"(ah[s]), "=&v"(al[s]) : "v"(avo), "s"(ps) : "memory");
        }
    } else {
        const int panel = min(row0, M - 1) >> 5;
        const h16x8* ap = reinterpret_cast<const h16x8*>(Af) + (size_t)panel * KS * 2 * 64 + (lane ^ (h << 4));
#pragma unroll
        for (int s = 0; s < HS; ++s) { ah[s] = ap[((HS * h + s) * 2 + 0) * 64]; al[s] = ap[((HS * h + s) * 2 + 1) * 64]; }
    }
    // per-column vector (bias): one value per lane and chunk, requested by name when its chunk starts (no LDS left for it)
    const int ccol = lane & 31;
    float cvr[2] = {0.f, 0.f};
    // half-chunks 0 .. 4 (chunks 0, 1 and the first half of chunk 2) behind the panel: vector memory returns in order
#pragma unroll
    for (int g = STREAM ? 2 : 0; g < 5; ++g)
#pragma unroll
        for (int i = 0; i < 3; ++i) dma1(g >> 1, g & 1, g >> 1, i);

    const int crow = (lane >> 5) * 4;
    const int rbase = row0 + 16 * h + crow;                // row of accumulator register r (< 8): rbase + 8 (r >> 2) + (r & 3)
    const int valid = M - rbase;                           // register r exists iff 8 (r >> 2) + (r & 3) < valid
    unsigned char* const xmine = psm + P8_RING_BYTES + wave * 2048 + lane16;
    unsigned char* const xpart = psm + P8_RING_BYTES + (wave ^ 4) * 2048 + lane16;
    const unsigned rd0 = (unsigned)(uintptr_t)(h3_lds_void*)(psm + h * HS * 2048 + lane16);   // fragment reads of this group in ring slot 0

    f32x16 a0, a1, b0 = {}, b1 = {};                        // (chunk 0 is handed b0 / b1 as its "previous" accumulators and never reads them)
    float res[2][8];
#pragma unroll
    for (int r = 0; r < 8; ++r) { res[0][r] = 0.f; res[1][r] = 0.f; }
    h16x8 bh[3], bl[3];
#define UU3D_P8_READ(i, sb, kk) \
    asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" \
                 : "=&v"(bh[i]), "=&v"(bl[i]) : "v"(sb), "i"((kk) * 2048), "i"((kk) * 2048 + 1024))
    static_assert(CPW >= 1 && CPW <= 12, "chunks per workgroup");

    // first barrier: half-chunks 0 and 1 (chunk 0) landed = everything but the newest 9 pieces
    PANEL_STAMP(const unsigned long long st_issued = __builtin_amdgcn_s_memrealtime();)
    asm volatile("s_waitcnt vmcnt(%0)" :: "i"(STREAM ? 2 * HS + 9 : 9) : "memory");      // (STREAM: chunk 0's weights are older than the A fragments and 9 pieces)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    PANEL_STAMP(const unsigned long long st_first = __builtin_amdgcn_s_memrealtime(); const unsigned long long ck0 = __builtin_amdgcn_s_memtime();)
    UU3D_P8_READ(0, rd0, 0);
    UU3D_P8_READ(1, rd0, 1);

    // emit(c, r): register r (< 8) of chunk c's result = own partial sum + the partner's + bias (+ residual)
    auto finish = [&](int c, int r, const f32x16& p0, const f32x16& p1, const float (&rv)[8], const float (&rs)[8]) __attribute__((always_inline)) {
        float v = (p0[r] + p1[r] * (1.0f / H3_SCALE)) + rv[r] + cvr[c & 1];
        if constexpr (EP::kResidual) v += rs[r];
#if UU3D_P8_LOO != 2
        ep.store(rbase + 8 * (r >> 2) + (r & 3), (chunk0 + c) * 32 + ccol, v);
#else
        if (v == 12345.678f) ep.store(rbase + 8 * (r >> 2) + (r & 3), (chunk0 + c) * 32 + ccol, v);
#endif
    };

    // one chunk.  cur = (x0, x1) accumulates chunk c; prv = (p0, p1) holds chunk c - 1, which is sent / finished meanwhile.
    auto chunk = [&](auto c_tag, f32x16& x0, f32x16& x1, const f32x16& p0, const f32x16& p1) __attribute__((always_inline)) {
        constexpr bool WHOLE = decltype(whole_tag)::value;
        constexpr int c = decltype(c_tag)::value;
        constexpr int slot = c % 3, pslot = (c + 2) % 3, nslot = (c + 1) % 3;
        // Vector-memory operations per half-interval, in issue order: [first half of chunk c] RQ requests (bias + residual), 3 pieces;
        // [second half] 8 kStores stores of chunk c - 1 (c > 0; exactly that many only when no row is predicated), 3 pieces.  The
        // barrier that opens a half-interval needs the pieces issued FOUR half-intervals earlier: vmcnt(operations of the three
        // in between).  Counting too few (the predicated case: ST8 = 0) only waits for more than necessary.
        constexpr int RQ = EP::kResidual ? 9 : 1, ST8 = WHOLE ? 8 * EP::kStores : 0;
        const unsigned sb = rd0 + slot * P8_CHUNK_BYTES, nb = rd0 + nslot * P8_CHUNK_BYTES;
        float (&rcur)[8] = res[c & 1];
        float (&rprv)[8] = res[(c & 1) ^ 1];
        // ---- barrier B_c: half-chunk 2 c + 1 landed (own pieces); own reads of chunk c - 1 and of the exchange area returned
        //      (all but the 4 reads already issued for this chunk) ----
        if constexpr (c > 0) {
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(4)" :: "i"(9 + RQ + ST8 * ((c >= 2) + (c >= 3))) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { x0[r] = 0.f; x1[r] = 0.f; }
        {
            const unsigned b = (unsigned)((chunk0 + c) * 32 + ccol) * 4u;
            asm volatile("global_load_dword %0, %1, %2" : "=v"(cvr[c & 1]) : "v"(b), "s"(colv) : "memory");
        }
        if constexpr (EP::kResidual) {
#pragma unroll
            for (int r = 0; r < 8; ++r) ep.request8(min(rbase + 8 * (r >> 2) + (r & 3), M - 1), (chunk0 + c) * 32 + ccol, rcur[r]);
        }
#pragma unroll
        for (int kk = 0; kk < 6; ++kk) {
            UU3D_P8_READ((kk + 2) % 3, sb, kk + 2);
            asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(bh[kk % 3]), "+v"(bl[kk % 3]) : "i"(c > 0 && (kk == 2 || kk == 3) ? 6 : 4));   // (the two sends sit between B(3) and B(4))
            if constexpr (STREAM && c == 0)      // this slice's A fragments: younger are the later slices', the bias values, 9 pieces, the requests and pieces of this chunk so far
                asm volatile("s_waitcnt vmcnt(%2)" : "+v"(ah[kk]), "+v"(al[kk]) : "i"((HS - 1 - kk) * 2 + 9 + RQ + (kk <= 3 ? 0 : kk - 3)));
            x0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kk], bh[kk % 3], x0, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kk], bl[kk % 3], x1, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[kk], bh[kk % 3], x1, 0, 0, 0);
            if (c > 0 && kk == 1) {                        // send rows 16 .. 31 of chunk c - 1 (its MFMAs have drained by now)
#if UU3D_P8_LOO != 4
                f32x4 s0, s1;
#pragma unroll
                for (int e = 0; e < 4; ++e) { s0[e] = p0[8 + e] + p1[8 + e] * (1.0f / H3_SCALE); s1[e] = p0[12 + e] + p1[12 + e] * (1.0f / H3_SCALE); }
                asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:1024" :: "v"((unsigned)(uintptr_t)(h3_lds_void*)xmine), "v"(s0), "v"(s1) : "memory");
#endif
            }
#if UU3D_P8_LOO != 1
            if (kk >= 3) dma1(c + 2, 1, pslot, kk - 3);    // second half of chunk c + 2 into the slot chunk c - 1 was read from
#endif
            if constexpr ((VAR & 1) != 0) __builtin_amdgcn_sched_barrier(0);
        }
        // ---- barrier B'_c: half-chunk 2 c + 2 landed; everybody has read the first halves of chunk c and written the exchange area ----
        // (LDS operations return in order: the two sends are older than the reads of slices 6 and 7, the only 4 left outstanding here)
#if UU3D_P8_LOO != 3
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(4)" :: "i"(9 + (c >= 1 ? 2 : 1) * RQ + ST8 * (c >= 2)) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#endif
        f32x4 r0, r1;                                      // the partner's partial sums of rows this wave finishes
        float rv[8];
        if constexpr (c > 0) {
#if UU3D_P8_LOO != 4
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "=&v"(r0), "=&v"(r1) : "v"((unsigned)(uintptr_t)(h3_lds_void*)xpart) : "memory");
#else
            r0 = f32x4{0.f, 0.f, 0.f, 0.f}; r1 = r0;
#endif
        }
        // LDS operations in flight behind the barrier, oldest first: B(6) B(7) [X X] B(8) ...; the stores of chunk c - 1 go out at
        // k-slices 7 .. 9 so that the half-interval's three pieces (k-slices 9 .. 11) are its last vector-memory operations
#pragma unroll
        for (int kk = 6; kk < HS; ++kk) {
            if (kk + 2 < HS) UU3D_P8_READ((kk + 2) % 3, sb, kk + 2);
            const int young = (kk + 1 < HS ? 2 : 0) + (kk + 2 < HS ? 2 : 0);                       // B(kk + 1), B(kk + 2)
            if (c > 0 && kk == 6) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(bh[kk % 3]), "+v"(bl[kk % 3]));      // ... and the two exchange reads
            else if (c > 0 && kk == 7) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bh[kk % 3]), "+v"(bl[kk % 3]), "+v"(r0), "+v"(r1));   // B(7) and the exchange reads (older than B(8))
            else asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(bh[kk % 3]), "+v"(bl[kk % 3]) : "i"(young));
            if constexpr (STREAM && c == 0)
                asm volatile("s_waitcnt vmcnt(%2)" : "+v"(ah[kk]), "+v"(al[kk]) : "i"((HS - 1 - kk) * 2 + 9 + RQ + (kk <= 9 ? 3 : kk - 6)));
            x0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kk], bh[kk % 3], x0, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kk], bl[kk % 3], x1, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[kk], bh[kk % 3], x1, 0, 0, 0);
            if constexpr (c > 0) {
                if (kk == 7) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { rv[e] = r0[e]; rv[4 + e] = r1[e]; }
                    // the bias / residual values of chunk c - 1 (requested two half-intervals ago): younger than the last of them are that
                    // half-interval's 3 pieces, the 3 pieces + stores of the next one and the RQ requests + 3 pieces of this chunk
                    if constexpr (EP::kResidual)
                        asm volatile("s_waitcnt vmcnt(%9)" : "+v"(rprv[0]), "+v"(rprv[1]), "+v"(rprv[2]), "+v"(rprv[3]), "+v"(rprv[4]), "+v"(rprv[5]), "+v"(rprv[6]), "+v"(rprv[7]), "+v"(cvr[(c - 1) & 1])
                                     : "i"(9 + RQ + ST8 * (c >= 2)) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%1)" : "+v"(cvr[(c - 1) & 1]) : "i"(9 + RQ + ST8 * (c >= 2)) : "memory");
                }
                if (kk >= 7 && kk <= 9) {                  // registers 0 1 2 | 3 4 5 | 6 7
#pragma unroll
                    for (int e = 0; e < 3; ++e) {
                        const int r = (kk - 7) * 3 + e;
                        if (r < 8 && (WHOLE || 8 * (r >> 2) + (r & 3) < valid)) finish(c - 1, r, p0, p1, rv, rprv);
                    }
                }
            }
#if UU3D_P8_LOO != 1
            if (kk >= 9) dma1(c + 3, 0, slot, kk - 9);     // first half of chunk c + 3 into the slot being read (its first halves are retired)
#endif
            if constexpr ((VAR & 1) != 0) __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (c + 1 < CPW) {                       // the next chunk's first fragments: its first half landed one barrier ago
            UU3D_P8_READ(0, nb, 0);
            UU3D_P8_READ(1, nb, 1);
        }
    };

    {
#define UU3D_P8_CHUNK(C) if constexpr ((C) < CPW) { if constexpr (((C) & 1) == 0) chunk(std::integral_constant<int, (C)>{}, a0, a1, b0, b1); \
                                                    else chunk(std::integral_constant<int, (C)>{}, b0, b1, a0, a1); }
        UU3D_P8_CHUNK(0) UU3D_P8_CHUNK(1) UU3D_P8_CHUNK(2) UU3D_P8_CHUNK(3) UU3D_P8_CHUNK(4) UU3D_P8_CHUNK(5)
        UU3D_P8_CHUNK(6) UU3D_P8_CHUNK(7) UU3D_P8_CHUNK(8) UU3D_P8_CHUNK(9) UU3D_P8_CHUNK(10) UU3D_P8_CHUNK(11)
#undef UU3D_P8_CHUNK
        // ---- the last chunk: send, barrier, receive, finish ----
        PANEL_STAMP(const unsigned long long st_loop = __builtin_amdgcn_s_memrealtime(); const unsigned long long ck1 = __builtin_amdgcn_s_memtime();)
        constexpr int c = CPW - 1;
        const f32x16& p0 = (c & 1) ? b0 : a0;
        const f32x16& p1 = (c & 1) ? b1 : a1;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // (the clamped tail pieces must not outlive the LDS allocation either)
        __builtin_amdgcn_s_barrier();                      // everybody's reads of the exchange area (chunk c - 1) returned
        f32x4 s0, s1, r0, r1;
#pragma unroll
        for (int e = 0; e < 4; ++e) { s0[e] = p0[8 + e] + p1[8 + e] * (1.0f / H3_SCALE); s1[e] = p0[12 + e] + p1[12 + e] * (1.0f / H3_SCALE); }
        asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)" :: "v"((unsigned)(uintptr_t)(h3_lds_void*)xmine), "v"(s0), "v"(s1) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r0), "=&v"(r1) : "v"((unsigned)(uintptr_t)(h3_lds_void*)xpart) : "memory");
        float rv[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { rv[e] = r0[e]; rv[4 + e] = r1[e]; }
        float (&rl)[8] = res[c & 1];
        asm volatile("" : "+v"(cvr[c & 1]));               // (requested when the chunk started; the vmcnt(0) above covers it)
        if (EP::kResidual) asm volatile("" : "+v"(rl[0]), "+v"(rl[1]), "+v"(rl[2]), "+v"(rl[3]), "+v"(rl[4]), "+v"(rl[5]), "+v"(rl[6]), "+v"(rl[7]));
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if (decltype(whole_tag)::value || 8 * (r >> 2) + (r & 3) < valid) finish(c, r, p0, p1, rv, rl);
        if constexpr (panel_ln_tail<EP>::value) {
            // LayerNorm 2 of this row tile (the workgroup owns its rows: splits == 1, all 12 chunks): every store of the tile is
            // complete and visible to the workgroup behind the wait + barrier (workgroup scope: one CU, one L1, write-through)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#pragma unroll 2
            for (int pass = 0; pass < 4; ++pass)
                ln_split_frag_row<KS>(ep.x, ep.ldo, M, ep.eps, ep.gamma, ep.beta, ep.Af, bm * 128 + pass * 32 + (tid >> 4), tid & 15);
        }
        PANEL_STAMP(asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (tid == 0 && u < 1024) { unsigned long long* o = panel_stamps + u * 8; o[0] = st_entry; o[1] = st_issued; o[2] = st_first; o[3] = st_loop; o[4] = __builtin_amdgcn_s_memrealtime(); o[5] = ck1 - ck0; })
    };
    };
    if (whole) body(std::true_type{}); else body(std::false_type{});
#undef UU3D_P8_READ
}

}  // namespace uu3d
