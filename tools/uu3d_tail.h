// uu3d_tail.h -- the LAST strided transformer block (u_u_t.py:93-160, called from :369-386) and the central-frame head
// (u_u_t.py:414-416) as ONE launch whose workgroups cooperate INSIDE an XCD.
//
// Before: 9 launches of 5-17 us each for < 1 GFLOP (h36m_351 at batch 128: 384 token rows in, 128 out; 76 us of the 0.96 ms
// forward): LayerNorm + QKV, attention, projection (+ its split-K combine), LayerNorm + fc1, strided convolution (+ combine),
// head (+ combine).  Every one of them is a memory round trip behind a kernel boundary.
//
// Here the batch is cut into 8 GROUPS of ceil(B / 8) sequences; a group's whole chain runs on the workgroups of ONE XCD, so
// that every hand-off between its phases stays in that XCD's L2:
//   * a workgroup reads the id of the XCD it is actually running on (s_getreg_b32 HW_REG_XCC_ID) and claims a group for THAT
//     XCD (compare-and-swap on the group's owner word); it only ever works on groups its own XCD owns.  Placement is read,
//     never assumed: any dispatch order / workgroup -> XCD map gives the same results, an XCD that gets no workgroup (or gets
//     them late) just leaves its group to be claimed by one that has finished (speed, not correctness);
//   * the tasks of a phase are handed out by a ticket counter per (group, phase); a finished task drains its stores
//     (s_waitcnt vmcnt(0): the XCD's L2 has them -- the vector L1 is write-through), then adds 1 to the phase's done counter;
//     the next phase starts when done == number of tasks.  Consumers read everything another workgroup produced with sc1
//     loads (bypass the CU's own L1, served by the shared L2).  No device-scope fence anywhere (round 2 measured those at
//     0.2-0.5 ms per forward): the L2 IS the coherence point for the CUs that share it;
//   * every producer leaves the id of its XCD next to its done count and every consumer compares them with its own
//     (TailCtl::err bit 1) -- tests/test_tail_gpu.py fails if a foreign id is ever observed;
//   * weights never depend on a previous phase: a workgroup takes its ticket for the NEXT phase and issues that task's weight
//     fragment loads BEFORE it waits for the current phase to complete.
// Arithmetic: f16x3 products (uu3d_gemm_h3.h) on 32 x 32 x 16 MFMAs with both operands straight from memory into the
// registers the MFMA reads (weights in fragment order, uu3d_gemm_panel.h: one coalesced 1 KiB load per fragment; activations
// 32 bytes of a row per lane), the contraction split over 1, 2 or 4 waves of the workgroup and combined in wave order
// through LDS (deterministic); attention (<= 32 tokens) in exact f32 on the vector ALU (online softmax).
#pragma once
#include <type_traits>
#include "uu3d_gemm_h3.h"

namespace uu3d {

static constexpr int TAIL_GROUPS = 8, TAIL_PHASES = 6, TAIL_STAMPS = 64;
enum { TP_QKV = 0, TP_ATTN = 1, TP_PROJ = 2, TP_FC1 = 3, TP_CONV = 4, TP_HEAD = 5 };
enum { TAIL_ERR_TIMEOUT = 1, TAIL_ERR_FOREIGN_XCC = 2 };

struct TailCtl {                           // zeroed before every launch (hipMemsetAsync in uu3d_forward); 128-byte lines
    unsigned owner[TAIL_GROUPS];           // 0 = unowned, else 1 + the XCC id of the XCD that works on the group
    unsigned err, pad0[23];
    struct Line { unsigned v[32]; };
    struct TLine { unsigned v[8][8]; };    // [owner XCC][phase]
    TLine ticket[TAIL_GROUPS];             // [group].v[xcc][phase]: next task, counted separately per CLAIMING XCD: a workgroup draws its
                                           // first tickets together with its claim; if the claim fails (a foreign XCD owns the group) the
                                           // tickets it drew are from a counter nobody else uses
    Line done[TAIL_GROUPS];                // [group].v[phase]: finished tasks
    unsigned stamp[TAIL_GROUPS][TAIL_PHASES][TAIL_STAMPS];   // 1 + XCC id of the workgroup that ran task t (t < 64)
    unsigned census[TAIL_GROUPS];          // workgroups seen per XCC id (diagnostics: uu3d_tail_status)
    unsigned pad1[24];
};
static_assert(sizeof(TailCtl) % 16 == 0, "memset size");

struct TailParams {
    int B, G;                              // sequences, sequences per group
    int L_in, L_out, stride, pad_left, res_lo;
    int n_out;                             // 3 J
    float* x;                              // [B L_in][384]  block input (+ PE), updated in place by the projection
    float* qkv;                            // [B L_in][1152]
    float* o;                              // attention output as A FRAGMENTS: [group][32-row tile of the group's rows][24 k-slices][hi | lo][64 lanes][8 halfs]
    float* hb;                             // [B L_in][768]
    float* part;                           // [2][B L_out][384] partial sums of the convolution (two halves of K = 2304)
    float* out;                            // [B L_out][n_out]
    const float *ln1_g, *ln1_b, *bqkv, *bp, *ln2_g, *ln2_b, *b1, *b2, *bh;
    const _Float16 *wqkv_f, *wp_f, *w1_f, *wc_f, *wh_f;      // fragment-ordered planes (panel_pack_operand)
    TailCtl* ctl;
    unsigned long long* dbg;               // STAMP builds (tools/tail_exp.hip): [workgroup][64] s_memrealtime ticks (10 ns); else unused
};

namespace tail {

constexpr int D = 384, H = 768, KS_D = D / 16, KS_C = 3 * H / 16;     // k-slices: 24 (K = 384), 144 (K = 2304)
constexpr unsigned SPIN_LIMIT = 1u << 21;

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u; }   // HW_REG_XCC_ID[3:0]

// 16-byte load that bypasses this CU's L1 (buffer_load_dwordx4 ... sc1): for bytes another workgroup of the XCD stored in this launch
struct Sc1Buf {
    __amdgpu_buffer_rsrc_t r;
    __device__ __forceinline__ explicit Sc1Buf(const void* p) : r(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7ffffffc, 0x00020000)) {}
    __device__ __forceinline__ f32x4 ld(unsigned byte_off) const {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16 /* sc1 */));
    }
};
__device__ __forceinline__ unsigned ld_u32_agent(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// LDS (all dynamic, 16-byte aligned pieces): [0, BIG): the A fragments of the task's row tile; red: partial tiles of a unit whose
// K is split over waves
constexpr int AFRAG_BYTES = 24 * 2 * 1024;               // A fragments of one 32-row tile, K = 384: [k-slice][plane][lane][8 halfs]
constexpr int BIG_BYTES = AFRAG_BYTES;
constexpr int RED_BYTES = 4 * 16 * 64 * 4;
constexpr int MAX_L = 4;                                 // keys per sequence the unrolled attention holds in registers
constexpr int LDS_BYTES = BIG_BYTES + RED_BYTES + 256 + 2 * 384 * 4;   // + broadcast words + LayerNorm gamma | beta

// ---- the product of one unit: rows (lane & 31) of a 32-row tile x 32 columns, k-slices [0, SPW) of this wave -------------------
template <int SPW>
__device__ __forceinline__ void mfma_unit(const h16x8 (&ah)[SPW], const h16x8 (&al)[SPW], const h16x8 (&bh)[SPW], const h16x8 (&bl)[SPW],
                                          f32x16& acc) {
    f32x16 a0, a1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
#pragma unroll
    for (int q = 0; q < SPW; ++q) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[q], bh[q], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[q], bl[q], a1, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[q], bh[q], a1, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = a0[i] + a1[i] * (1.0f / H3_SCALE);
}
__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, h16x8& hi, h16x8& lo) {
    h16x4 h0, l0, h1, l1;
    h3_split(a, h0, l0); h3_split(b, h1, l1);
    hi = (h16x8){h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    lo = (h16x8){l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
}
// weight fragments of chunk c, k-slices [s0, s0 + SPW) of an operand with KS k-slices per chunk
template <int SPW>
__device__ __forceinline__ void load_b(const _Float16* __restrict__ Bf, const int KS, const int c, const int s0, const int lane,
                                       h16x8 (&bh)[SPW], h16x8 (&bl)[SPW]) {
    const h16x8* bp = reinterpret_cast<const h16x8*>(Bf) + ((size_t)c * KS + s0) * 2 * 64 + lane;
#pragma unroll
    for (int q = 0; q < SPW; ++q) { bh[q] = bp[(q * 2 + 0) * 64]; bl[q] = bp[(q * 2 + 1) * 64]; }
}

}  // namespace tail

// One workgroup = 4 waves = one wave per SIMD (a wave keeps the 48 weight fragments of its 32-column chunk in registers);
// grid = number of CUs.
template <bool STAMP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
strided_tail_kernel_t(const TailParams p)
{
    using namespace tail;
    h3_flush_f16_denormals();
    extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];
    h16x8* const afr = reinterpret_cast<h16x8*>(tsm);                               // [24][2][64]
    float (*red)[16][64] = reinterpret_cast<float (*)[16][64]>(tsm + BIG_BYTES);
    int* const ish = reinterpret_cast<int*>(tsm + BIG_BYTES + RED_BYTES);           // [0]: broadcast, [8 ..]: first tickets
    float* const gbs = reinterpret_cast<float*>(tsm + BIG_BYTES + RED_BYTES + 256); // gamma[384] | beta[384] of the current LayerNorm
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, g2 = lane >> 5;
    TailCtl* const ctl = p.ctl;
    const unsigned xcc = xcc_id();
    const int ngroups = (p.B + p.G - 1) / p.G;
    unsigned visited = 0;
    // STAMP builds: slot id of this workgroup's row <- s_memrealtime (first write wins: the first task of a phase)
    auto stamp = [&](int id) { if (STAMP) { if (tid == 0 && id < 64 && p.dbg[(size_t)blockIdx.x * 64 + id] == 0) p.dbg[(size_t)blockIdx.x * 64 + id] = __builtin_amdgcn_s_memrealtime(); } };
    stamp(0);                                                          // 0: start

    bool first_claim = true;
    for (;;) {
        // Claim + the first ticket of EVERY phase in ONE round trip: lane 0 .. 5 draw tickets of the candidate group (from this
        // XCD's own counters), lane 8 runs the compare-and-swap on its owner word.  First the preferred group (== XCC id), then --
        // after this XCD's group is done, or if a foreign XCD holds the preferred one -- whatever is unowned.
        if (wave == 0) {
            int cand = -1;
            if (first_claim) { cand = (int)(xcc & 7u); if (cand >= ngroups) cand = -1; }
            if (cand < 0) {
                for (int i = 0; i < TAIL_GROUPS && cand < 0; ++i) {
                    const int g = (int)((xcc + i) & 7u);
                    if (g >= ngroups || (visited >> g) & 1u) continue;
                    const unsigned cur = ld_u32_agent(&ctl->owner[g]);
                    if (cur == 0u || cur == xcc + 1u) cand = g;
                }
            }
            int got = -1;
            if (cand >= 0) {
                unsigned tk = 0, own = 0;
                if (lane < TAIL_PHASES) tk = atomicAdd(&ctl->ticket[cand].v[xcc & 7u][lane], 1u);
                if (lane == 8) own = atomicCAS(&ctl->owner[cand], 0u, xcc + 1u);
                own = __shfl(own, 8);
                if (own == 0u || own == xcc + 1u) { got = cand; if (lane < TAIL_PHASES) ish[8 + lane] = (int)tk; }
                else visited |= 1u << cand;                            // foreign: never look at it again (wave 0's copy of `visited`)
            }
            if (lane == 0) ish[0] = got, ish[1] = cand;
        }
        first_claim = false;
        __syncthreads();
        const int grp = ish[0], cand_seen = ish[1];
        __syncthreads();
        if (grp < 0) { if (cand_seen < 0) break; visited |= 1u << cand_seen; continue; }
        visited |= 1u << grp;
        stamp(2);                                                      // 2: group claimed, tickets drawn

        const int seq0 = grp * p.G, nb = min(p.G, p.B - seq0);
        const int R = nb * p.L_in, Ro = nb * p.L_out;                 // rows of this group: block input / block output
        const int row_in0 = seq0 * p.L_in, row_out0 = seq0 * p.L_out;
        const int RT = (R + 31) >> 5, RTo = (Ro + 31) >> 5;
        const int RTG = (p.G * p.L_in + 31) >> 5;                      // row tiles a full group has (fragment images per group)
        int ntask[TAIL_PHASES];
        ntask[TP_QKV] = RT * 9;                                       // (row tile, 4 chunks of 32 columns): one chunk per wave
        ntask[TP_ATTN] = (R + 15) >> 4;                               // 16 rows x 8 heads x 2 halves of the head dimension = 256 threads
        ntask[TP_PROJ] = RT * 12;                                     // (row tile, chunk), K over the 4 waves
        ntask[TP_FC1] = RT * 6;
        ntask[TP_CONV] = RTo * 12 * 2;                                // (row tile, chunk, half of K = 2304), K half over 4 waves
        ntask[TP_HEAD] = RTo * ((p.n_out + 31) >> 5);

        auto first_task = [&](int ph) -> int { return ish[8 + ph]; };
        auto next_task = [&](int ph) -> int {
            if (tid == 0) ish[0] = (int)atomicAdd(&ctl->ticket[grp].v[xcc & 7u][ph], 1u);
            __syncthreads();
            const int t = ish[0];
            __syncthreads();
            return t;
        };
        // wait until every task of phase ph has published; compare the producers' XCC ids with ours
        auto wait_phase = [&](int ph) {
            if (wave == 0) {
                const unsigned need = (unsigned)ntask[ph];
                unsigned spins = 0;
                while (ld_u32_agent(&ctl->done[grp].v[ph]) < need) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > SPIN_LIMIT) { if (lane == 0) atomicOr(&ctl->err, (unsigned)TAIL_ERR_TIMEOUT); break; }
                }
                if (lane < min(ntask[ph], TAIL_STAMPS)) {
                    const unsigned s = ld_u32_agent(&ctl->stamp[grp][ph][lane]);
                    if (s != xcc + 1u) atomicOr(&ctl->err, (unsigned)TAIL_ERR_FOREIGN_XCC);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    // no instruction: keeps the loads below the poll
            __syncthreads();
        };
        auto publish = [&](int ph, int t) {
            if (tid == 0 && t < TAIL_STAMPS) __hip_atomic_store(&ctl->stamp[grp][ph][t], xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave: its stores are in the XCD's L2
            __syncthreads();
            if (tid == 0) atomicAdd(&ctl->done[grp].v[ph], 1u);
            stamp(8 + 8 * ph + 7);
        };
        // combine the 4 partial tiles of a unit in wave order; true for the wave that then owns the result
        auto combine4 = [&](f32x16& acc) -> bool {
            if (wave != 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) red[wave][i][lane] = acc[i];
            }
            __syncthreads();
            if (wave == 0) {
                for (int u = 1; u < 4; ++u)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[i] += red[u][i][lane];
            }
            __syncthreads();                                          // red is reused by the next task
            return wave == 0;
        };
        // 48 consecutive channels [kb, kb + 48) of tile row tr, split and stored as A fragments (thread = (tr, kb / 48))
        auto write_afrag = [&](const int tr, const int kb, const f32x4 (&v)[12]) {
#pragma unroll
            for (int m = 0; m < 6; ++m) {
                h16x8 hi, lo;
                split8(v[2 * m], v[2 * m + 1], hi, lo);
                const int k = kb + 8 * m;
                h16x8* d = afr + ((k >> 4) * 2) * 64 + tr + 32 * ((k >> 3) & 1);
                d[0] = hi; d[64] = lo;
            }
        };
        // this wave's 32 x 32 tile: A fragments from LDS, the chunk's weight fragments from registers
        auto mfma_lds = [&](const h16x8 (&bh)[24], const h16x8 (&bl)[24], f32x16& acc) {
            f32x16 a0, a1;
#pragma unroll
            for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
#pragma unroll
            for (int q = 0; q < 24; ++q) {
                const h16x8 ah = afr[(q * 2) * 64 + lane], al = afr[(q * 2 + 1) * 64 + lane];
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[q], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[q], a1, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[q], a1, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = a0[i] + a1[i] * (1.0f / H3_SCALE);
        };

        // LayerNorm-fed Dense layer, out = LN(x) W + bias.  A task = one 32-row tile x 4 chunks of 32 columns (one per wave): the
        // workgroup normalises and splits the tile ONCE (thread = (row, 48 channels)), the fragments go through LDS.
        // planes == false: out[row][col] = v (f32, leading dimension ldo); true: relu(v) as f16 hi / lo planes [rows][ldo]
        auto ln_dense = [&](const int ph, const int prev, const int nchunks, const _Float16* __restrict__ Wf, const float* __restrict__ gamma,
                            const float* __restrict__ beta, const float* __restrict__ bias, float* __restrict__ outp, const int ldo, const bool planes) {
            const Sc1Buf bx(p.x);
            const int tpr = nchunks >> 2;                              // tasks per row tile
            if (tid < 96) { *reinterpret_cast<f32x4*>(gbs + 4 * tid) = *reinterpret_cast<const f32x4*>(gamma + 4 * tid);
                            *reinterpret_cast<f32x4*>(gbs + D + 4 * tid) = *reinterpret_cast<const f32x4*>(beta + 4 * tid); }
            bool gb_ready = false;                                     // (the first phase has no other barrier in front of the first read)
            int t = first_task(ph);
            bool waited = prev < 0;
            while (t < ntask[ph]) {
                const int rt = t / tpr, c = (t - rt * tpr) * 4 + wave;
                const float bv = bias[c * 32 + r];                     // (every weight-side load goes out before the wait / the operand rows)
                h16x8 bh[24], bl[24];
                const int tr = tid >> 3, kb = (tid & 7) * 48;
                const int grow = row_in0 + min(rt * 32 + tr, R - 1);
                f32x4 v[12];
                // vector memory returns in order: where nothing has to be waited for, the operand rows (needed first) go out before
                // the 48 weight fragments; behind a phase barrier the weights go out before the wait
                if (waited) {
#pragma unroll
                    for (int e = 0; e < 12; ++e) v[e] = bx.ld(((unsigned)grow * D + kb + 4 * e) * 4u);
                    load_b<24>(Wf, KS_D, c, 0, lane, bh, bl);
                } else {
                    load_b<24>(Wf, KS_D, c, 0, lane, bh, bl);
                    stamp(8 + 8 * ph + 0);
                    wait_phase(prev); waited = true;
                    stamp(8 + 8 * ph + 1);
#pragma unroll
                    for (int e = 0; e < 12; ++e) v[e] = bx.ld(((unsigned)grow * D + kb + 4 * e) * 4u);
                }
                float s1 = 0.f;
#pragma unroll
                for (int e = 0; e < 12; ++e) s1 += (v[e][0] + v[e][1]) + (v[e][2] + v[e][3]);
                s1 += __shfl_xor(s1, 1); s1 += __shfl_xor(s1, 2); s1 += __shfl_xor(s1, 4);
                const float mean = s1 * (1.0f / D);
                float s2 = 0.f;
#pragma unroll
                for (int e = 0; e < 12; ++e)
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const float a = v[e][i] - mean; s2 += a * a; }
                s2 += __shfl_xor(s2, 1); s2 += __shfl_xor(s2, 2); s2 += __shfl_xor(s2, 4);
                const float rstd = 1.0f / sqrtf(s2 * (1.0f / D) + 1e-5f);
                if (!gb_ready) { __syncthreads(); gb_ready = true; }
#pragma unroll
                for (int e = 0; e < 12; ++e)
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const float inv = rstd * gbs[kb + 4 * e + i]; v[e][i] = v[e][i] * inv + (gbs[D + kb + 4 * e + i] - mean * inv); }
                stamp(8 + 8 * ph + 2);
                write_afrag(tr, kb, v);
                __syncthreads();
                stamp(8 + 8 * ph + 3);
                f32x16 acc;
                mfma_lds(bh, bl, acc);
                stamp(8 + 8 * ph + 4);
                {
                    const int col = c * 32 + r;
                    if (!planes) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int lr = rt * 32 + 8 * (i >> 2) + 4 * g2 + (i & 3);
                            if (lr < R) outp[(size_t)(row_in0 + lr) * ldo + col] = acc[i] + bv;
                        }
                    } else {
                        _Float16* const oh = reinterpret_cast<_Float16*>(outp);
                        _Float16* const ol = oh + (size_t)p.B * p.L_in * ldo;
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int lr = rt * 32 + 8 * (i >> 2) + 4 * g2 + (i & 3);
                            const float y = fmaxf(acc[i] + bv, 0.f);
                            const _Float16 h = h3_hi(y);
                            if (lr < R) { const size_t o = (size_t)(row_in0 + lr) * ldo + col; oh[o] = h; ol[o] = (_Float16)((y - (float)h) * H3_SCALE); }
                        }
                    }
                }
                stamp(8 + 8 * ph + 5);
                publish(ph, t);                                        // (its barrier also frees the fragments for the next task)
                t = next_task(ph);
            }
        };
        // ================= phase 0: qkv = LN1(x) Wqkv + b =================
        ln_dense(TP_QKV, -1, 36, p.wqkv_f, p.ln1_g, p.ln1_b, p.bqkv, p.qkv, 3 * D, false);
        // ================= phase 1: attention (<= MAX_L keys, exact f32).  thread = (row, head, half of the head's 48 channels): every
        // load of a thread in one batch, q . k completed across the two halves by one shuffle; the context rows leave as the
        // projection's A fragments (f16 hi / lo, fragment order) =================
        {
            const Sc1Buf bqkv(p.qkv);
            int t = first_task(TP_ATTN);
            bool waited = false;
            while (t < ntask[TP_ATTN]) {
                stamp(8 + 8 * TP_ATTN + 0);
                if (!waited) { wait_phase(TP_QKV); waited = true; }
                stamp(8 + 8 * TP_ATTN + 1);
                const int lrow_u = t * 16 + (tid >> 4), hd = (tid >> 1) & 7, hf = tid & 1;
                const int lrow = min(lrow_u, R - 1), sb = lrow / p.L_in;
                const unsigned cb = (unsigned)(hd * 48 + hf * 24) * 4u;                    // this thread's 24 channels, bytes into q (k, v: + D, + 2 D floats)
                f32x4 q[6], acc[6];
#pragma unroll
                for (int e = 0; e < 6; ++e) q[e] = bqkv.ld((unsigned)(row_in0 + lrow) * (3 * D * 4u) + cb + 16u * e);
                auto attend = [&](auto lk_tag) __attribute__((always_inline)) {
                    constexpr int LK = decltype(lk_tag)::value;
                    f32x4 kk[LK][6], vv[LK][6];
#pragma unroll
                    for (int j = 0; j < LK; ++j) {
                        const unsigned ro = (unsigned)(row_in0 + sb * LK + j) * (3 * D * 4u) + cb;
#pragma unroll
                        for (int e = 0; e < 6; ++e) { kk[j][e] = bqkv.ld(ro + D * 4u + 16u * e); vv[j][e] = bqkv.ld(ro + 2 * D * 4u + 16u * e); }
                    }
                    const float scale = 1.44269504088896341f / sqrtf(48.f);               // logits in units of log 2
                    float sj[LK], m = -INFINITY;
#pragma unroll
                    for (int j = 0; j < LK; ++j) {
                        float sacc = 0.f;
#pragma unroll
                        for (int e = 0; e < 6; ++e) sacc += (q[e][0] * kk[j][e][0] + q[e][1] * kk[j][e][1]) + (q[e][2] * kk[j][e][2] + q[e][3] * kk[j][e][3]);
                        sacc += __shfl_xor(sacc, 1);                                      // the other half of the head (same sum in both lanes: x + y == y + x)
                        sj[j] = sacc * scale;
                        m = fmaxf(m, sj[j]);
                    }
                    float l = 0.f;
#pragma unroll
                    for (int j = 0; j < LK; ++j) { sj[j] = exp2f(sj[j] - m); l += sj[j]; }
                    const float inv = 1.0f / l;
#pragma unroll
                    for (int e = 0; e < 6; ++e) acc[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < LK; ++j) {
                        const float pj = sj[j] * inv;
#pragma unroll
                        for (int e = 0; e < 6; ++e) acc[e] = acc[e] + vv[j][e] * pj;
                    }
                };
                switch (p.L_in) {
                    case 1: attend(std::integral_constant<int, 1>{}); break;
                    case 2: attend(std::integral_constant<int, 2>{}); break;
                    case 3: attend(std::integral_constant<int, 3>{}); break;
                    default: attend(std::integral_constant<int, 4>{}); break;
                }
                if (lrow_u < R) {
                    // one fragment image (48 KiB) per 32-row tile of the GROUP's rows: tile index = group * tiles per group + local tile
                    h16x8* const fo = reinterpret_cast<h16x8*>(p.o) + (size_t)(grp * RTG + (lrow >> 5)) * (24 * 2 * 64) + (lrow & 31);
#pragma unroll
                    for (int mm = 0; mm < 3; ++mm) {
                        h16x8 hi, lo;
                        split8(acc[2 * mm], acc[2 * mm + 1], hi, lo);
                        const int k = hd * 48 + hf * 24 + 8 * mm;
                        h16x8* d = fo + ((k >> 4) * 2) * 64 + 32 * ((k >> 3) & 1);
                        d[0] = hi; d[64] = lo;
                    }
                }
                publish(TP_ATTN, t);
                t = next_task(TP_ATTN);
            }
        }
        // ================= phase 2: x += o Wp + bp: A fragments straight from memory, K over the 4 waves =================
        {
            const Sc1Buf bo(p.o);
            int t = first_task(TP_PROJ);
            bool waited = false;
            while (t < ntask[TP_PROJ]) {
                const int rt = t / 12, c = t - rt * 12, s0 = wave * 6;
                const float bias = p.bp[c * 32 + r];
                h16x8 bh[6], bl[6];
                load_b<6>(p.wp_f, KS_D, c, s0, lane, bh, bl);
                float old[16];                                                           // the residual rows (wave 0 stores the tile): not written by
                if (wave == 0) {                                                         // anybody before this task, so they go out before the wait too
#pragma unroll
                    for (int i = 0; i < 16; ++i) old[i] = p.x[(size_t)(row_in0 + min(rt * 32 + 8 * (i >> 2) + 4 * g2 + (i & 3), R - 1)) * D + c * 32 + r];
                }
                stamp(8 + 8 * TP_PROJ + 0);
                if (!waited) { wait_phase(TP_ATTN); waited = true; }
                stamp(8 + 8 * TP_PROJ + 1);
                h16x8 ah[6], al[6];
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const unsigned off = ((unsigned)(grp * RTG + rt) * (24 * 2 * 64) + (unsigned)((s0 + q) * 2) * 64 + lane) * 16u;
                    ah[q] = __builtin_bit_cast(h16x8, bo.ld(off)); al[q] = __builtin_bit_cast(h16x8, bo.ld(off + 1024u));
                }
                f32x16 out;
                mfma_unit<6>(ah, al, bh, bl, out);
                stamp(8 + 8 * TP_PROJ + 4);
                if (combine4(out)) {
                    const int col = c * 32 + r;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int lr = rt * 32 + 8 * (i >> 2) + 4 * g2 + (i & 3);
                        if (lr < R) p.x[(size_t)(row_in0 + lr) * D + col] = old[i] + (out[i] + bias);
                    }
                }
                publish(TP_PROJ, t);
                t = next_task(TP_PROJ);
            }
        }
        // ================= phase 3: hb = relu(LN2(x) W1 + b1), as f16 hi / lo planes (the convolution's A operand) =================
        ln_dense(TP_FC1, TP_PROJ, 24, p.w1_f, p.ln2_g, p.ln2_b, p.b1, p.hb, H, true);
        // ================= phase 4: partial sums of the strided 3-tap convolution (ZeroPadding1D + Conv1D, u_u_t.py:126-131); the A
        // operand = 16 bytes per lane and k-slice of the hidden activations' f16 planes, no arithmetic =================
        {
            const Sc1Buf bhb(p.hb);
            const unsigned lo_off = (unsigned)p.B * p.L_in * H * 2u;             // bytes from the hi plane to the lo plane
            int t = first_task(TP_CONV);
            bool waited = false;
            while (t < ntask[TP_CONV]) {
                const int hf = t & 1, uc = t >> 1, rt = uc / 12, c = uc - rt * 12;
                const int s0 = hf * (KS_C / 2) + wave * 18;                      // absolute k-slice of K = 2304
                h16x8 bh[18], bl[18];
                load_b<18>(p.wc_f, KS_C, c, s0, lane, bh, bl);
                stamp(8 + 8 * TP_CONV + 0);
                if (!waited) { wait_phase(TP_FC1); waited = true; }
                stamp(8 + 8 * TP_CONV + 1);
                const int lro = min(rt * 32 + r, Ro - 1);                        // output row (b, tt) of the group
                const int b = lro / p.L_out, tt = lro - b * p.L_out;
                const int t0 = tt * p.stride - p.pad_left;
                h16x8 ah[18], al[18];
#pragma unroll
                for (int q = 0; q < 18; ++q) {
                    const int k = 16 * (s0 + q) + 8 * g2, j = k / H, cc = k - j * H;
                    const int src = t0 + j;
                    const bool ok = src >= 0 && src < p.L_in;
                    const unsigned off = ((unsigned)(row_in0 + b * p.L_in + min(max(src, 0), p.L_in - 1)) * H + cc) * 2u;
                    const f32x4 vh = bhb.ld(off), vl = bhb.ld(lo_off + off);
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    ah[q] = __builtin_bit_cast(h16x8, ok ? vh : z); al[q] = __builtin_bit_cast(h16x8, ok ? vl : z);
                }
                f32x16 acc;
                mfma_unit<18>(ah, al, bh, bl, acc);
                stamp(8 + 8 * TP_CONV + 4);
                if (combine4(acc)) {
                    const int col = c * 32 + r;
                    float* pp = p.part + (size_t)hf * p.B * p.L_out * D;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int lr = rt * 32 + 8 * (i >> 2) + 4 * g2 + (i & 3);
                        if (lr < Ro) pp[(size_t)(row_out0 + lr) * D + col] = acc[i];
                    }
                }
                publish(TP_CONV, t);
                t = next_task(TP_CONV);
            }
        }
        // ================= phase 5: y = x[identity rows] + conv + b2 (u_u_t.py:138-156); out = y Wh + bh (:414-416) =================
        {
            const int hchunks = (p.n_out + 31) >> 5;
            const Sc1Buf bx(p.x), bpart(p.part);
            int t = first_task(TP_HEAD);
            bool waited = false;
            while (t < ntask[TP_HEAD]) {
                const int rt = t / hchunks, c = t - rt * hchunks, s0 = wave * 6;
                const float hbias = p.bh[min(c * 32 + r, p.n_out - 1)];
                h16x8 bh[6], bl[6];
                load_b<6>(p.wh_f, KS_D, c, s0, lane, bh, bl);
                stamp(8 + 8 * TP_HEAD + 0);
                if (!waited) { wait_phase(TP_CONV); waited = true; }
                stamp(8 + 8 * TP_HEAD + 1);
                const int lro = min(rt * 32 + r, Ro - 1);
                const int b = lro / p.L_out, tt = lro - b * p.L_out;
                const unsigned xrow = (unsigned)(row_in0 + b * p.L_in + tt * p.stride + p.res_lo);
                const unsigned prow = (unsigned)(row_out0 + lro);
                const unsigned half = (unsigned)p.B * p.L_out * D * 4u;
                h16x8 ah[6], al[6];
                {
                    f32x4 xa[6], xb[6], pa[6], pb[6], qa[6], qb[6];
#pragma unroll
                    for (int q = 0; q < 6; ++q) {
                        const unsigned kb = (16 * (s0 + q) + 8 * g2) * 4u;
                        xa[q] = bx.ld(xrow * (D * 4u) + kb); xb[q] = bx.ld(xrow * (D * 4u) + kb + 16u);
                        pa[q] = bpart.ld(prow * (D * 4u) + kb); pb[q] = bpart.ld(prow * (D * 4u) + kb + 16u);
                        qa[q] = bpart.ld(half + prow * (D * 4u) + kb); qb[q] = bpart.ld(half + prow * (D * 4u) + kb + 16u);
                    }
#pragma unroll
                    for (int q = 0; q < 6; ++q) {
                        const int k = 16 * (s0 + q) + 8 * g2;
                        const f32x4 ca = *reinterpret_cast<const f32x4*>(p.b2 + k), cb = *reinterpret_cast<const f32x4*>(p.b2 + k + 4);
                        split8(xa[q] + ((pa[q] + qa[q]) + ca), xb[q] + ((pb[q] + qb[q]) + cb), ah[q], al[q]);
                    }
                }
                f32x16 acc;
                mfma_unit<6>(ah, al, bh, bl, acc);
                if (combine4(acc)) {
                    const int col = c * 32 + r;
                    if (col < p.n_out) {
                        // a bounded spin that gave up anywhere in this launch means some phase was consumed unfinished: the
                        // predictions are poisoned instead of silently wrong (uu3d_tail_status names the reason)
                        const float bias = (ld_u32_agent(&ctl->err) & (unsigned)TAIL_ERR_TIMEOUT) ? __builtin_nanf("") : hbias;
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int lr = rt * 32 + 8 * (i >> 2) + 4 * g2 + (i & 3);
                            if (lr < Ro) p.out[(size_t)(row_out0 + lr) * p.n_out + col] = acc[i] + bias;
                        }
                    }
                }
                publish(TP_HEAD, t);
                t = next_task(TP_HEAD);
            }
        }
        // Stay until the group is complete.  A workgroup without tasks would otherwise go looking for unowned groups at once and take
        // the group of an XCD whose workgroups merely started a few microseconds late (seen in tools/tail_exp: that group then ran on
        // the idle workgroups of a foreign XCD alone, 20 us late).  Stealing is for XCDs that never show up.
        if (wave == 0) {
            unsigned spins = 0;
            while (ld_u32_agent(&ctl->done[grp].v[TP_HEAD]) < (unsigned)ntask[TP_HEAD]) {
                __builtin_amdgcn_s_sleep(64);
                if (++spins > SPIN_LIMIT) { if (lane == 0) atomicOr(&ctl->err, (unsigned)TAIL_ERR_TIMEOUT); break; }
            }
        }
        __syncthreads();
    }
    if (tid == 0) atomicAdd(&ctl->census[xcc & 7u], 1u);             // diagnostics (uu3d_tail_status): workgroups seen per XCC id
}

}  // namespace uu3d
