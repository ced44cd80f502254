// uu3d_gemm.h -- exact-f32 MFMA GEMM for gfx950 with fused A-operand loaders and epilogues.
//
// C[M][N] = Aop[M][K] * W[K][N] (+ epilogue).  W is stored TRANSPOSED and padded in HBM as
// Bt[Np][Kp] (Np multiple of 128, Kp multiple of 32, zero filled) so that both operands are
// read as 16-byte, k-contiguous fragments.
//
// Arithmetic: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: bit-for-bit an fmaf chain).
// A wave owns TM x TN tiles of 32x32.  Per 8-deep k-slice a lane reads ONE float4 per
// operand tile: lane (r = lane&31, h = lane>>5) takes k = 8*kk + 4*h + s for MFMA step
// s = 0..3.  The MFMA contracts over "k slots", so any permutation of k that is the same
// for A and B is legal; this one makes every LDS fragment read a ds_read_b128.
//
// LDS tiles are [rows][32] floats with a row stride of 36 floats: the 16 lanes of every
// ds_read_b128 group then hit 16 distinct 16-byte slots (36 = 4 * 9, 9 odd) -- no bank
// conflicts, and the staging ds_write_b128 of 8 lanes covers one contiguous 128-byte row.
//
// Workgroup = 256 threads = 4 waves as 2 (M) x 2 (N); LDS double buffered, next k-tile's
// global loads are issued before the MFMAs of the current one and written after them.
//
// Block -> tile mapping is XCD aware: block ids are dealt round-robin over the 8 XCDs, so
// id % 8 selects the XCD; all N-tiles of one M-tile go to the same XCD so the A rows are
// fetched into ONE L2 (weights are read by every XCD regardless).
#pragma once
#include "uu3d_dropout.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace uu3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static constexpr int GEMM_BK = 32;
static constexpr int GEMM_LD = 36;

__host__ __device__ inline constexpr size_t gemm_lds_bytes(int BM, int BN) {
    return (size_t)2 * (BM + BN) * GEMM_LD * sizeof(float);
}

// ------------------------------------------------------------------------------------
// A-operand loaders.  prep(row) builds a per-row context once.  Staging is split in two so
// that nothing waits on a global load before the MFMAs of the current k-tile have issued:
//   issue(ctx, k)        -> Raw : only address arithmetic + global loads (no use of the data)
//   finish(ctx, k, raw)  -> A[row][k .. k+3] ready for LDS (transforms run AFTER the MFMAs)
// Rows >= M are clamped to row M-1 (their products are never stored); k >= K reads as zero.
// ------------------------------------------------------------------------------------
struct ALoadPlain {
    const float* __restrict__ A;
    int lda, M, K;
    struct Ctx { const float* p; };
    struct Raw { f32x4 x; };
    __device__ __forceinline__ Ctx prep(int row) const {
        Ctx c; c.p = A + (size_t)min(row, M - 1) * lda; return c;
    }
    __device__ __forceinline__ Raw issue(const Ctx& c, int k) const {
        Raw r; r.x = *reinterpret_cast<const f32x4*>(c.p + min(k, K - 4)); return r;
    }
    __device__ __forceinline__ f32x4 finish(const Ctx&, int k, const Raw& r) const {
        return (k < K) ? r.x : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
};

// LayerNorm fused into the load: Keras' non-fused formula
//   inv = rsqrt(var + eps) * gamma ;  y = x * inv + (beta - mean * inv)
// with (mean, rstd) per row precomputed by row_stats_kernel.
struct ALoadLayerNorm {
    const float* __restrict__ A;
    const float2* __restrict__ stats;   // (mean, rstd) per row
    const float* __restrict__ gamma;
    const float* __restrict__ beta;
    int lda, M, K;
    struct Ctx { const float* p; float mean, rstd; };
    struct Raw { f32x4 x, g, b; };
    __device__ __forceinline__ Ctx prep(int row) const {
        const int rc = min(row, M - 1);
        Ctx c; c.p = A + (size_t)rc * lda; const float2 s = stats[rc]; c.mean = s.x; c.rstd = s.y;
        return c;
    }
    __device__ __forceinline__ Raw issue(const Ctx& c, int k) const {
        const int kc = min(k, K - 4);
        Raw r;
        r.x = *reinterpret_cast<const f32x4*>(c.p + kc);
        r.g = *reinterpret_cast<const f32x4*>(gamma + kc);
        r.b = *reinterpret_cast<const f32x4*>(beta + kc);
        return r;
    }
    __device__ __forceinline__ f32x4 finish(const Ctx& c, int k, const Raw& r) const {
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float inv = c.rstd * r.g[e];
            y[e] = r.x[e] * inv + (r.b[e] - c.mean * inv);
        }
        return (k < K) ? y : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
};

// Strided k=3 convolution as a 3-tap GEMM over gathered rows (ZeroPadding1D + Conv1D
// 'valid'): output row (b, t) contracts over k = j*C + c with source row t*stride + j - pad_left
// of sequence b (zero outside [0, L_in)).
struct ALoadConv3 {
    const float* __restrict__ Hin;      // (B * L_in, C)
    int C, L_in, L_out, stride, pad_left, M, K;   // M = B * L_out, K = 3 * C
    struct Ctx { int base_row; int t0; };          // base_row = b * L_in ; t0 = t*stride - pad_left
    struct Raw { f32x4 x; int ok; };
    __device__ __forceinline__ Ctx prep(int row) const {
        const int rc = min(row, M - 1);
        Ctx c; const int b = rc / L_out; const int t = rc - b * L_out;
        c.base_row = b * L_in; c.t0 = t * stride - pad_left;
        return c;
    }
    __device__ __forceinline__ Raw issue(const Ctx& c, int k) const {
        const int kc = min(k, K - 4);
        const int j = kc / C;
        const int cc = kc - j * C;
        const int src = c.t0 + j;
        Raw r;
        r.ok = (src >= 0) & (src < L_in) & (k < K);
        const int srcc = min(max(src, 0), L_in - 1);
        r.x = *reinterpret_cast<const f32x4*>(Hin + (size_t)(c.base_row + srcc) * C + cc);
        return r;
    }
    __device__ __forceinline__ f32x4 finish(const Ctx&, int, const Raw& r) const {
        return r.ok ? r.x : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
};

// ------------------------------------------------------------------------------------
// Epilogues.  To keep the 16 accumulator rows of a lane from turning into 16 dependent
// load->wait->store round trips, an epilogue is split into loads that do not depend on the
// accumulator and the final store:
//   colv(col)            -> float2 : per-column constants (bias, ...), once per tile column
//   pre(rowc, col)       -> float2 : per-element inputs (residual, PE, mask ...); rowc is a
//                                    CLAMPED (always valid) row, all 16 issued together
//   store(row, col, acc, colv, pre) : only called for row < M
// ------------------------------------------------------------------------------------
struct EpBias {            // out = acc + bias
    float* __restrict__ out; const float* __restrict__ bias; int ldo;
    __device__ __forceinline__ float2 colv(int col) const { return make_float2(bias[col], 0.f); }
    __device__ __forceinline__ float2 pre(int, int) const { return make_float2(0.f, 0.f); }
    __device__ __forceinline__ void store(int row, int col, float v, float2 cv, float2) const {
        out[(size_t)row * ldo + col] = v + cv.x;
    }
};
struct EpBiasRelu {        // out = max(acc + bias, 0)
    float* __restrict__ out; const float* __restrict__ bias; int ldo;
    __device__ __forceinline__ float2 colv(int col) const { return make_float2(bias[col], 0.f); }
    __device__ __forceinline__ float2 pre(int, int) const { return make_float2(0.f, 0.f); }
    __device__ __forceinline__ void store(int row, int col, float v, float2 cv, float2) const {
        out[(size_t)row * ldo + col] = fmaxf(v + cv.x, 0.f);
    }
};
// x = x + (acc + bias), in place; optionally a second stream out2 = x_new + pe2[row % period]
// (the next strided block's positional encoding, folded in here).
struct EpBiasResidual {
    float* x; const float* __restrict__ bias; int ld;
    float* out2; const float* __restrict__ pe2; int period;
    __device__ __forceinline__ float2 colv(int col) const { return make_float2(bias[col], 0.f); }
    __device__ __forceinline__ float2 pre(int rowc, int col) const {
        float2 p; p.x = x[(size_t)rowc * ld + col];
        p.y = (out2 != nullptr) ? pe2[(size_t)(rowc % period) * ld + col] : 0.f;
        return p;
    }
    __device__ __forceinline__ void store(int row, int col, float v, float2 cv, float2 p) const {
        const size_t o = (size_t)row * ld + col;
        const float y = p.x + (v + cv.x);
        x[o] = y;
        if (out2 != nullptr) out2[o] = y + p.y;
    }
};
// spatial_to_temporal_fc + strided-input token blend + temporal PE (u_u_t.py:332,344-352):
//   t = acc + bias ; x = m ? t : token ; x += pe[row % N]
struct EpSpatialToTemporal {
    float* __restrict__ x; const float* __restrict__ bias; int ld;
    const uint8_t* __restrict__ mask;     // per row (B*N), nullptr when no strided input
    const float* __restrict__ token; const float* __restrict__ pe; int period;
    const uint8_t* __restrict__ keep = nullptr;     // training with TOKEN_MASK_RATE > 0 (token_keep_kernel): real rows with keep == 0 become the masked-token value
    const float* __restrict__ mtoken = nullptr;     // that value: the learnable masked token (LEARNABLE_MASKED_TOKEN), nullptr = 0
    __device__ __forceinline__ float2 colv(int col) const {
        return make_float2(bias[col], mask != nullptr ? token[col] : 0.f);
    }
    __device__ __forceinline__ float2 pre(int rowc, int col) const {
        float2 p; p.x = pe[(size_t)(rowc % period) * ld + col];
        p.y = (mask != nullptr && mask[rowc] == 0) ? 0.f : ((keep != nullptr && keep[rowc] == 0) ? 2.f : 1.f);
        return p;
    }
    __device__ __forceinline__ void store(int row, int col, float v, float2 cv, float2 p) const {
        const float t = (p.y == 1.f) ? (v + cv.x) : (p.y == 0.f ? cv.y : (mtoken != nullptr ? mtoken[col] : 0.f));
        x[(size_t)row * ld + col] = t + p.x;
    }
};
// strided block tail (u_u_t.py:138-156): out = identity + (acc + bias) (+ next block's PE)
// identity row of output (b, t) is row  b*L_in + t*stride + lo  of the block's input stream.
struct EpConvResidual {
    float* __restrict__ out; const float* __restrict__ bias; int ld;
    const float* __restrict__ xin; int L_in, L_out, stride, lo;
    const float* __restrict__ pe_next;    // (L_out, ld) or nullptr
    const float* __restrict__ gate = nullptr; float keep = 1.f;      // training: DropPath on the MLP branch, one gate per sequence (u_u_t.py:136-137)
    DropCfg drop{};                                                   // training: Dropout on the convolution's output (u_u_t.py:88-89), in front of DropPath
    __device__ __forceinline__ float2 colv(int col) const { return make_float2(bias[col], 0.f); }
    __device__ __forceinline__ float2 pre(int rowc, int col) const {
        const int b = rowc / L_out; const int t = rowc - b * L_out;
        float2 p; p.x = xin[(size_t)(b * L_in + t * stride + lo) * ld + col];
        p.y = (pe_next != nullptr) ? pe_next[(size_t)t * ld + col] : 0.f;
        return p;
    }
    __device__ __forceinline__ void store(int row, int col, float v, float2 cv, float2 p) const {
        float z = v + cv.x;
        if (drop.on()) z *= drop_factor(drop, (unsigned long long)row * (unsigned)ld + (unsigned)col);
        if (gate != nullptr) z = (z / keep) * gate[row / L_out];
        float y = p.x + z;
        if (pe_next != nullptr) y += p.y;
        out[(size_t)row * ld + col] = y;
    }
};

// split-K partial sums: slice blockIdx.y writes its raw accumulators to slab[slice][row][col]
struct EpSlab {
    float* __restrict__ slab; int ld; size_t slice_stride;
    __device__ __forceinline__ float2 colv(int) const { return make_float2(0.f, 0.f); }
    __device__ __forceinline__ float2 pre(int, int) const { return make_float2(0.f, 0.f); }
    __device__ __forceinline__ void store(int row, int col, float v, float2, float2) const {
        slab[blockIdx.y * slice_stride + (size_t)row * ld + col] = v;
    }
};

// Deterministic split-K combine: slices are summed in slice order, then the real epilogue runs.
template <class EP>
__global__ void __launch_bounds__(256)
splitk_reduce_kernel(const float* __restrict__ slab, const int slices, const size_t slice_stride,
                     const int M, const int N, const int ld, const EP ep)
{
    __builtin_amdgcn_s_setreg((1 /*MODE*/) | (6 << 6) | ((2 - 1) << 11), 0);   // f16 denormals flush: an epilogue may split its result (h3_flush_f16_denormals)
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * N) return;
    const int row = idx / N, col = idx - row * N;
    const float* p = slab + (size_t)row * ld + col;
    // eight slices in flight, added in slice order (as a plain loop the thread waited for one load after the other: 10 us for a
    // 22-slice combine of 384 x 384 = 1.2 TB/s)
    float v = p[0];
    int s = 1;
    for (; s + 7 < slices; s += 8) {
        float a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = p[(size_t)(s + u) * slice_stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += a[u];
    }
    for (; s + 1 < slices; s += 2) { const float a0 = p[(size_t)s * slice_stride], a1 = p[(size_t)(s + 1) * slice_stride]; v += a0; v += a1; }
    if (s < slices) v += p[(size_t)s * slice_stride];
    ep.store(row, col, v, ep.colv(col), ep.pre(row, col));
}

// The same combine for MANY slices of a SMALL result (tall-skinny weight gradients: 32 x 32 ... 64 x 96 outputs over
// tens of thousands of rows): 16 lanes per output element, lane q adds slices q, q+16, ... in order, the 16 lane sums
// are then added in lane order -- deterministic, and 16 x shorter than the one-thread-per-element loop.
template <class EP>
__global__ void __launch_bounds__(256)
splitk_reduce16_kernel(const float* __restrict__ slab, const int slices, const size_t slice_stride,
                       const int M, const int N, const int ld, const EP ep)
{
    __shared__ float red[16][17];
    const int el = threadIdx.x & 15, q = threadIdx.x >> 4;
    const int idx = blockIdx.x * 16 + el;
    const bool ok = idx < M * N;
    const int row = ok ? idx / N : 0, col = ok ? idx - row * N : 0;
    const float* p = slab + (size_t)row * ld + col;
    float v = 0.f;
    if (ok) {                                               // four of this lane's slices in flight, added in slice order
        int s = q;
        for (; s + 48 < slices; s += 64) {
            const float a0 = p[(size_t)s * slice_stride], a1 = p[(size_t)(s + 16) * slice_stride];
            const float a2 = p[(size_t)(s + 32) * slice_stride], a3 = p[(size_t)(s + 48) * slice_stride];
            v += a0; v += a1; v += a2; v += a3;
        }
        for (; s < slices; s += 16) v += p[(size_t)s * slice_stride];
    }
    red[q][el] = v;
    __syncthreads();
    if (q == 0 && ok) {
        float t = red[0][el];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += red[k][el];
        ep.store(row, col, t, ep.colv(col), ep.pre(row, col));
    }
}

// ------------------------------------------------------------------------------------
// kernel
// ------------------------------------------------------------------------------------
template <int BM, int BN, class AL, class EP>
__global__ void __launch_bounds__(256)
gemm_f32_kernel(const AL al, const float* __restrict__ Bt, const int M, const int N, const int Kp,
                const int m_tiles, const int n_tiles, const int kt_per_split, const EP ep)
{
    constexpr int LD = GEMM_LD;
    constexpr int TM = BM / 64, TN = BN / 64;     // 32x32 MFMA tiles per wave
    constexpr int AI = BM / 32, BI = BN / 32;     // 16-byte staging loads per thread per k-tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                    // [2][BM][LD]
    float* Bs = smem + 2 * BM * LD;      // [2][BN][LD]

    // XCD-aware tile mapping (see header).
    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int slot = id >> 3;
    const int bn = slot % n_tiles;
    const int bm = (slot / n_tiles) * 8 + xcd;
    if (bm >= m_tiles) return;
    const int bm0 = bm * BM, bn0 = bn * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int srow = tid >> 3;           // staging row 0..31
    const int scol = (tid & 7) * 4;      // staging k offset

    typename AL::Ctx actx[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) actx[i] = al.prep(bm0 + srow + 32 * i);
    const float* bptr[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) bptr[i] = Bt + (size_t)(bn0 + srow + 32 * i) * Kp + scol;

    typename AL::Raw ra[AI];
    f32x4 rb[BI];
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // split-K: blockIdx.y owns k-tiles [kt_lo, KT) of this slice (kt_per_split == all of them
    // when the GEMM is not split; a split GEMM's epilogue is EpSlab and a reduce pass follows)
    const int kt_lo = blockIdx.y * kt_per_split;
    const int KT = min(Kp / GEMM_BK, kt_lo + kt_per_split);

    // prologue: first tile -> LDS buffer of its parity
#pragma unroll
    for (int i = 0; i < AI; ++i) ra[i] = al.issue(actx[i], kt_lo * GEMM_BK + scol);
#pragma unroll
    for (int i = 0; i < BI; ++i) rb[i] = *reinterpret_cast<const f32x4*>(bptr[i] + kt_lo * GEMM_BK);
    As += (kt_lo & 1) * BM * LD; Bs += (kt_lo & 1) * BN * LD;   // make "cur = (kt - kt_lo) & 1" below
#pragma unroll
    for (int i = 0; i < AI; ++i)
        *reinterpret_cast<f32x4*>(&As[(srow + 32 * i) * LD + scol]) = al.finish(actx[i], kt_lo * GEMM_BK + scol, ra[i]);
#pragma unroll
    for (int i = 0; i < BI; ++i) *reinterpret_cast<f32x4*>(&Bs[(srow + 32 * i) * LD + scol]) = rb[i];
    __syncthreads();

    const int fr = lane & 31;            // fragment row
    const int fk = (lane >> 5) * 4;      // fragment k offset inside an 8-slice

    As -= (kt_lo & 1) * BM * LD; Bs -= (kt_lo & 1) * BN * LD;
    for (int kt = kt_lo; kt < KT; ++kt) {
        const int cur = kt & 1;
        // Next k-tile's global loads are issued first; the last iteration re-loads its own
        // tile (cheap, L1/L2 hit) so the loop body stays branch-free.
        const int k0 = min(kt + 1, KT - 1) * GEMM_BK;
#pragma unroll
        for (int i = 0; i < AI; ++i) ra[i] = al.issue(actx[i], k0 + scol);
#pragma unroll
        for (int i = 0; i < BI; ++i) rb[i] = *reinterpret_cast<const f32x4*>(bptr[i] + k0);

        const float* Ac = As + cur * BM * LD + (wm * (BM / 2) + fr) * LD + fk;
        const float* Bc = Bs + cur * BN * LD + (wn * (BN / 2) + fr) * LD + fk;
#pragma unroll
        for (int kk = 0; kk < GEMM_BK / 8; ++kk) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(Ac + i * 32 * LD + kk * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bc + j * 32 * LD + kk * 8);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        }
        // stage the prefetched tile into the other buffer (harmless rewrite on the last pass:
        // nobody reads that buffer again)
        const int nxt = cur ^ 1;
#pragma unroll
        for (int i = 0; i < AI; ++i)
            *reinterpret_cast<f32x4*>(&As[nxt * BM * LD + (srow + 32 * i) * LD + scol]) =
                al.finish(actx[i], k0 + scol, ra[i]);
#pragma unroll
        for (int i = 0; i < BI; ++i)
            *reinterpret_cast<f32x4*>(&Bs[nxt * BN * LD + (srow + 32 * i) * LD + scol]) = rb[i];
        __syncthreads();
    }

    // epilogue: C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
    // Interior tiles (the common case) take a branch-free path: per-lane predicates would make
    // the compiler fence every store with s_waitcnt vmcnt(0).
    const int crow0 = bm0 + wm * (BM / 2) + 4 * (lane >> 5);
    const int ccol0 = bn0 + wn * (BN / 2) + (lane & 31);
    if (bm0 + BM <= M && bn0 + BN <= N) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = ccol0 + j * 32;
                const float2 cv = ep.colv(col);
                float2 pr[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) pr[r] = ep.pre(crow0 + i * 32 + (r & 3) + 8 * (r >> 2), col);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ep.store(crow0 + i * 32 + (r & 3) + 8 * (r >> 2), col, acc[i][j][r], cv, pr[r]);
            }
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = ccol0 + j * 32;
                if (col < N) {
                    const float2 cv = ep.colv(col);
                    float2 pr[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        pr[r] = ep.pre(min(crow0 + i * 32 + (r & 3) + 8 * (r >> 2), M - 1), col);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = crow0 + i * 32 + (r & 3) + 8 * (r >> 2);
                        if (row < M) ep.store(row, col, acc[i][j][r], cv, pr[r]);
                    }
                }
            }
    }
}

}  // namespace uu3d
