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
#include "uu3d_gemm_h3.h"

namespace uu3d {

static constexpr int TAIL_GROUPS = 8, TAIL_PHASES = 6, TAIL_STAMPS = 64;
enum { TP_QKV = 0, TP_ATTN = 1, TP_PROJ = 2, TP_FC1 = 3, TP_CONV = 4, TP_HEAD = 5 };
enum { TAIL_ERR_TIMEOUT = 1, TAIL_ERR_FOREIGN_XCC = 2 };

struct TailCtl {                           // zeroed before every launch (hipMemsetAsync in uu3d_forward); 128-byte lines
    unsigned owner[TAIL_GROUPS];           // 0 = unowned, else 1 + the XCC id of the XCD that works on the group
    unsigned err, pad0[23];
    struct Line { unsigned v[32]; };
    Line ticket[TAIL_GROUPS];              // [group].v[phase]: next task
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
    float* o;                              // [B L_in][384]
    float* hb;                             // [B L_in][768]
    float* part;                           // [2][B L_out][384] partial sums of the convolution (two halves of K = 2304)
    float* out;                            // [B L_out][n_out]
    const float *ln1_g, *ln1_b, *bqkv, *bp, *ln2_g, *ln2_b, *b1, *b2, *bh;
    const _Float16 *wqkv_f, *wp_f, *w1_f, *wc_f, *wh_f;      // fragment-ordered planes (panel_pack_operand)
    TailCtl* ctl;
    unsigned long long* dbg;               // STAMP builds (tools/tail_exp.hip): [workgroup][32] s_memrealtime ticks (10 ns); else unused
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

struct Shared {                            // static part of the LDS
    float red[4][16][64];                  // partial tiles of the waves of a unit
    float stat[2][4][32];                  // LayerNorm partial sums [pass][wave][row]
    float gam[384], bet[384];              // LayerNorm parameters of the current phase
    int bcast;
    int first[TAIL_PHASES];                // this workgroup's first ticket of every phase (taken together, up front)
};

// ---- claim a group for this XCD: the preferred one (group == XCC id) first -------------------------------------------------
__device__ __forceinline__ int claim_group(TailCtl* ctl, const unsigned xcc, unsigned& visited, const int ngroups) {
    // one vector load of all owner words; compare-and-swap only where it can succeed
    for (int i = 0; i < TAIL_GROUPS; ++i) {
        const int g = (int)((xcc + i) & 7u);
        if (g >= ngroups || (visited >> g) & 1u) continue;
        unsigned cur = ld_u32_agent(&ctl->owner[g]);
        if (cur == 0u) cur = atomicCAS(&ctl->owner[g], 0u, xcc + 1u) == 0u ? xcc + 1u : ld_u32_agent(&ctl->owner[g]);
        if (cur == xcc + 1u) { visited |= 1u << g; return g; }
    }
    return -1;
}

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

// One workgroup = 4 waves = one wave per SIMD (~300 registers); grid = number of CUs.
template <bool STAMP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
strided_tail_kernel_t(const TailParams p)
{
    using namespace tail;
    h3_flush_f16_denormals();
    __shared__ Shared sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, g2 = lane >> 5;
    TailCtl* const ctl = p.ctl;
    const unsigned xcc = xcc_id();
    if (tid == 0) atomicAdd(&ctl->census[xcc & 7u], 1u);
    const int ngroups = (p.B + p.G - 1) / p.G;
    unsigned visited = 0;
    int stamp_i = 0;
    auto stamp = [&]() { if (STAMP) { if (tid == 0 && stamp_i < 32) p.dbg[(size_t)blockIdx.x * 32 + stamp_i] = __builtin_amdgcn_s_memrealtime(); ++stamp_i; } };
    stamp();                                                           // 0: start

    for (;;) {
        if (tid == 0) sh.bcast = claim_group(ctl, xcc, visited, ngroups);
        __syncthreads();
        const int grp = sh.bcast;
        __syncthreads();
        if (grp < 0) break;
        visited |= 1u << grp;
        stamp();                                                       // 1: group claimed
        // the first ticket of EVERY phase in one vector atomic: no ticket round trip on the path between two phases
        if (tid < TAIL_PHASES) sh.first[tid] = (int)atomicAdd(&ctl->ticket[grp].v[tid], 1u);
        __syncthreads();
        stamp();                                                       // 2: tickets

        const int seq0 = grp * p.G, nb = min(p.G, p.B - seq0);
        const int R = nb * p.L_in, Ro = nb * p.L_out;                 // rows of this group: block input / block output
        const int row_in0 = seq0 * p.L_in, row_out0 = seq0 * p.L_out;
        const int RT = (R + 31) >> 5, RTo = (Ro + 31) >> 5;
        int ntask[TAIL_PHASES];
        ntask[TP_QKV] = (RT * 36 + 1) >> 1;                           // 2 units (row tile, 32-column chunk) per workgroup, K over wave pairs
        ntask[TP_ATTN] = (R * 8 + 255) >> 8;                          // one thread per (row, head)
        ntask[TP_PROJ] = RT * 12;                                     // one unit per workgroup, K over its 4 waves
        ntask[TP_FC1] = (RT * 24 + 1) >> 1;                           // 2 units per workgroup, K over wave pairs
        ntask[TP_CONV] = RTo * 12 * 2;                                // (row tile, chunk, half of K = 2304), K half over 4 waves
        ntask[TP_HEAD] = RTo * ((p.n_out + 31) >> 5);

        auto first_task = [&](int ph) -> int { return sh.first[ph]; };
        auto next_task = [&](int ph) -> int {
            if (tid == 0) sh.bcast = (int)atomicAdd(&ctl->ticket[grp].v[ph], 1u);
            __syncthreads();
            const int t = sh.bcast;
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
            stamp();
        };
        // LayerNorm statistics of the rows of a unit whose K = 384 is spread over KSPL waves (wave index inside the unit: kp):
        // two-pass, partial sums through LDS.  s1 = this wave's partial sum; sq(mean) = its partial sum of squared deviations.
        // every wave of the workgroup calls this (barriers).
        auto ln_stats = [&](float s1, auto&& sq, const int KSPL, float& mean, float& rstd) {
            s1 += __shfl_xor(s1, 32);
            if (g2 == 0) sh.stat[0][wave][r] = s1;
            __syncthreads();
            float tot = 0.f;
            const int w0 = wave - (wave % KSPL);
            for (int i = 0; i < KSPL; ++i) tot += sh.stat[0][w0 + i][r];
            mean = tot * (1.0f / D);
            float v = sq(mean);
            v += __shfl_xor(v, 32);
            if (g2 == 0) sh.stat[1][wave][r] = v;
            __syncthreads();
            float vt = 0.f;
            for (int i = 0; i < KSPL; ++i) vt += sh.stat[1][w0 + i][r];
            rstd = 1.0f / sqrtf(vt * (1.0f / D) + 1e-5f);
        };
        // combine the KSPL partial tiles of a unit in wave order; true for the wave that then owns the result
        auto combine = [&](f32x16& acc, const int KSPL) -> bool {
            if (KSPL == 1) return true;
            const int kp = wave % KSPL, w0 = wave - kp;
            if (kp != 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) sh.red[wave][i][lane] = acc[i];
            }
            __syncthreads();
            if (kp == 0) {
                for (int u = 1; u < KSPL; ++u)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[i] += sh.red[w0 + u][i][lane];
            }
            __syncthreads();                                          // red is reused by the next task
            return kp == 0;
        };

        // LayerNorm-fed Dense layer: out[row][col] = epi(LN(x)[row] . W[:, col] + bias[col]); 2 units (row tile, 32-column chunk) per
        // workgroup, K = 384 over a wave pair (12 k-slices each)
        auto ln_dense = [&](const int ph, const int prev, const int nchunks, const _Float16* __restrict__ Wf, const float* __restrict__ gamma,
                            const float* __restrict__ beta, const float* __restrict__ bias, float* __restrict__ outp, const int ldo, const bool relu) {
            if (tid < 96) { *reinterpret_cast<f32x4*>(sh.gam + 4 * tid) = *reinterpret_cast<const f32x4*>(gamma + 4 * tid);
                            *reinterpret_cast<f32x4*>(sh.bet + 4 * tid) = *reinterpret_cast<const f32x4*>(beta + 4 * tid); }
            const Sc1Buf bx(p.x);
            int t = first_task(ph);
            bool waited = prev < 0;
            while (t < ntask[ph]) {
                const int u = t * 2 + (wave >> 1), rt = u / nchunks, c = u - rt * nchunks, s0 = (wave & 1) * 12;
                const bool act = rt < RT;
                h16x8 bh[12], bl[12];
                load_b<12>(Wf, KS_D, act ? c : 0, s0, lane, bh, bl);
                if (!waited) { wait_phase(prev); waited = true; stamp(); }
                const int lrow = min(rt * 32 + r, R - 1), grow = row_in0 + lrow;
                f32x4 xa[12], xb[12];
                float s1 = 0.f;
#pragma unroll
                for (int q = 0; q < 12; ++q) {
                    const unsigned off = ((unsigned)grow * D + 16 * (s0 + q) + 8 * g2) * 4u;
                    xa[q] = bx.ld(off); xb[q] = bx.ld(off + 16u);
                }
#pragma unroll
                for (int q = 0; q < 12; ++q) s1 += ((xa[q][0] + xa[q][1]) + (xa[q][2] + xa[q][3])) + ((xb[q][0] + xb[q][1]) + (xb[q][2] + xb[q][3]));
                float mean, rstd;
                ln_stats(s1, [&](float mu) { float v = 0.f;
#pragma unroll
                    for (int q = 0; q < 12; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const float a = xa[q][e] - mu, b = xb[q][e] - mu; v += a * a + b * b; }
                    return v; }, 2, mean, rstd);
                h16x8 ah[12], al[12];
#pragma unroll
                for (int q = 0; q < 12; ++q) {
                    const int k = 16 * (s0 + q) + 8 * g2;
                    const f32x4 ga = *reinterpret_cast<const f32x4*>(sh.gam + k), gb = *reinterpret_cast<const f32x4*>(sh.gam + k + 4);
                    const f32x4 ba = *reinterpret_cast<const f32x4*>(sh.bet + k), bb = *reinterpret_cast<const f32x4*>(sh.bet + k + 4);
                    f32x4 ya, yb;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float ia = rstd * ga[e], ib = rstd * gb[e];
                        ya[e] = xa[q][e] * ia + (ba[e] - mean * ia);
                        yb[e] = xb[q][e] * ib + (bb[e] - mean * ib);
                    }
                    split8(ya, yb, ah[q], al[q]);
                }
                f32x16 acc;
                mfma_unit<12>(ah, al, bh, bl, acc);
                if (combine(acc, 2) && act) {
                    const int col = c * 32 + r;
                    const float bv = bias[col];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int lr = rt * 32 + 8 * (i >> 2) + 4 * g2 + (i & 3);
                        const float v = acc[i] + bv;
                        if (lr < R) outp[(size_t)(row_in0 + lr) * ldo + col] = relu ? fmaxf(v, 0.f) : v;
                    }
                }
                publish(ph, t);
                t = next_task(ph);
            }
        };
        // ================= phase 0: qkv = LN1(x) Wqkv + b =================
        ln_dense(TP_QKV, -1, 36, p.wqkv_f, p.ln1_g, p.ln1_b, p.bqkv, p.qkv, 3 * D, false);
        // ================= phase 1: attention, one thread per (row, head), online softmax in f32 =================
        {
            const Sc1Buf bqkv(p.qkv);
            int t = first_task(TP_ATTN);
            bool waited = false;
            while (t < ntask[TP_ATTN]) {
                if (!waited) { wait_phase(TP_QKV); waited = true; stamp(); }
                const int idx = t * 256 + tid, lrow = min(idx >> 3, R - 1), hd = idx & 7;
                const int b = lrow / p.L_in;
                const unsigned qoff = ((unsigned)(row_in0 + lrow) * (3 * D) + hd * 48) * 4u;
                f32x4 q[12], acc[12];
#pragma unroll
                for (int e = 0; e < 12; ++e) { q[e] = bqkv.ld(qoff + 16u * e); acc[e] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
                float m = -INFINITY, l = 0.f;
                const float scale = 1.44269504088896341f / sqrtf(48.f);         // logits in units of log 2
                for (int j = 0; j < p.L_in; ++j) {
                    const unsigned koff = ((unsigned)(row_in0 + b * p.L_in + j) * (3 * D) + D + hd * 48) * 4u;
                    f32x4 kv[12], vv[12];
#pragma unroll
                    for (int e = 0; e < 12; ++e) { kv[e] = bqkv.ld(koff + 16u * e); vv[e] = bqkv.ld(koff + D * 4u + 16u * e); }
                    float s = 0.f;
#pragma unroll
                    for (int e = 0; e < 12; ++e) s += (q[e][0] * kv[e][0] + q[e][1] * kv[e][1]) + (q[e][2] * kv[e][2] + q[e][3] * kv[e][3]);
                    s *= scale;
                    const float mn = fmaxf(m, s), corr = exp2f(m - mn), pj = exp2f(s - mn);
                    l = l * corr + pj; m = mn;
#pragma unroll
                    for (int e = 0; e < 12; ++e) acc[e] = acc[e] * corr + vv[e] * pj;
                }
                const float inv = 1.0f / l;
                if ((idx >> 3) < R) {
                    float* op = p.o + (size_t)(row_in0 + lrow) * D + hd * 48;
#pragma unroll
                    for (int e = 0; e < 12; ++e) *reinterpret_cast<f32x4*>(op + 4 * e) = acc[e] * inv;
                }
                publish(TP_ATTN, t);
                t = next_task(TP_ATTN);
            }
        }
        // ================= phase 2: x += o Wp + bp =================
        {
            const Sc1Buf bo(p.o);
            int t = first_task(TP_PROJ);
            bool waited = false;
            while (t < ntask[TP_PROJ]) {
                const int rt = t / 12, c = t - rt * 12, s0 = wave * 6;
                h16x8 bh[6], bl[6];
                load_b<6>(p.wp_f, KS_D, c, s0, lane, bh, bl);
                if (!waited) { wait_phase(TP_ATTN); waited = true; stamp(); }
                const int lrow = min(rt * 32 + r, R - 1), grow = row_in0 + lrow;
                h16x8 ah[6], al[6];
                {
                    f32x4 xa[6], xb[6];
#pragma unroll
                    for (int q = 0; q < 6; ++q) {
                        const unsigned off = ((unsigned)grow * D + 16 * (s0 + q) + 8 * g2) * 4u;
                        xa[q] = bo.ld(off); xb[q] = bo.ld(off + 16u);
                    }
#pragma unroll
                    for (int q = 0; q < 6; ++q) split8(xa[q], xb[q], ah[q], al[q]);
                }
                f32x16 acc;
                mfma_unit<6>(ah, al, bh, bl, acc);
                if (combine(acc, 4)) {
                    const int col = c * 32 + r;
                    const float bias = p.bp[col];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int lr = rt * 32 + 8 * (i >> 2) + 4 * g2 + (i & 3);
                        if (lr < R) { float* xp = p.x + (size_t)(row_in0 + lr) * D + col; *xp = *xp + (acc[i] + bias); }
                    }
                }
                publish(TP_PROJ, t);
                t = next_task(TP_PROJ);
            }
        }
        // ================= phase 3: hb = relu(LN2(x) W1 + b1) =================
        ln_dense(TP_FC1, TP_PROJ, 24, p.w1_f, p.ln2_g, p.ln2_b, p.b1, p.hb, H, true);
        // ================= phase 4: partial sums of the strided 3-tap convolution (ZeroPadding1D + Conv1D, u_u_t.py:126-131) =================
        {
            const Sc1Buf bhb(p.hb);
            int t = first_task(TP_CONV);
            bool waited = false;
            while (t < ntask[TP_CONV]) {
                const int hf = t & 1, uc = t >> 1, rt = uc / 12, c = uc - rt * 12;
                const int s0 = hf * (KS_C / 2) + wave * 18;                      // absolute k-slice of K = 2304
                h16x8 bh[18], bl[18];
                load_b<18>(p.wc_f, KS_C, c, s0, lane, bh, bl);
                if (!waited) { wait_phase(TP_FC1); waited = true; stamp(); }
                const int lro = min(rt * 32 + r, Ro - 1);                        // output row (b, tt) of the group
                const int b = lro / p.L_out, tt = lro - b * p.L_out;
                const int t0 = tt * p.stride - p.pad_left;
                h16x8 ah[18], al[18];
                {
                    f32x4 xa[18], xb[18];
                    bool ok[18];
#pragma unroll
                    for (int q = 0; q < 18; ++q) {
                        const int k = 16 * (s0 + q) + 8 * g2, j = k / H, cc = k - j * H;
                        const int src = t0 + j;
                        ok[q] = src >= 0 && src < p.L_in;
                        const unsigned off = ((unsigned)(row_in0 + b * p.L_in + min(max(src, 0), p.L_in - 1)) * H + cc) * 4u;
                        xa[q] = bhb.ld(off); xb[q] = bhb.ld(off + 16u);
                    }
#pragma unroll
                    for (int q = 0; q < 18; ++q) {
                        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                        split8(ok[q] ? xa[q] : z, ok[q] ? xb[q] : z, ah[q], al[q]);
                    }
                }
                f32x16 acc;
                mfma_unit<18>(ah, al, bh, bl, acc);
                if (combine(acc, 4)) {
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
                h16x8 bh[6], bl[6];
                load_b<6>(p.wh_f, KS_D, c, s0, lane, bh, bl);
                if (!waited) { wait_phase(TP_CONV); waited = true; stamp(); }
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
                if (combine(acc, 4)) {
                    const int col = c * 32 + r;
                    if (col < p.n_out) {
                        const float bias = p.bh[col];
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
    }
}

}  // namespace uu3d
