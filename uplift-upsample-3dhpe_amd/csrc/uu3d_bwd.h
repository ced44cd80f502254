// uu3d_bwd.h -- building blocks of the backward pass, exact f32.
//
//   gemm_tn_kernel     C[P][Q] = sum_r A[r][P]^T B[r][Q]   (weight gradients dW = X^T dY; the
//                      contraction runs over the TOKEN rows, so both operands are staged
//                      row-major [r][.] and the MFMA fragments are read down the columns)
//   colsum_kernel      bias / LayerNorm-beta style gradients: column sums over rows (optionally
//                      with a row period, for positional-encoding gradients)
//   ln_bwd_kernel      LayerNorm backward for one row per wave (dx) + per-block partial dgamma/dbeta
//   attn_bwd_kernel    softmax-attention backward for one (sequence, head) per workgroup
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "uu3d_gemm.h"
#include "uu3d_gemm_h3.h"

namespace uu3d {

// ------------------------------------------------------------------------------------
// A-side loaders for gemm_tn: load(r, p) -> A[r][p .. p+3] (p multiple of 4), zero outside.
// ------------------------------------------------------------------------------------
// fetch() issues the loads without a branch (indices clamped, validity kept as a flag) and finish() does the arithmetic and
// the zeroing: a kernel can then put ALL the loads of a k-step in flight before it waits for the first (with the test inside
// load() hipcc emitted one divergent branch per call and waited for each call's data inside it -- four serial memory round
// trips per k-step).  col() holds what depends on the thread's column only (it is fixed over the k-loop).
struct TnNoCol {};
struct TnLoadPlain {
    const float* __restrict__ A; int lda, R, P;
    struct Raw { f32x4 x; bool ok; };
    typedef TnNoCol Col;
    __device__ __forceinline__ Col col(int) const { return Col{}; }
    __device__ __forceinline__ Raw fetch(int r, int p) const {
        const bool ok = r < R && p < P;
        return Raw{*reinterpret_cast<const f32x4*>(A + (ok ? (size_t)r * lda + p : 0)), ok};
    }
    __device__ __forceinline__ f32x4 finish(const Raw& w, const Col&) const { return w.ok ? w.x : (f32x4){0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ f32x4 load(int r, int p) const { return finish(fetch(r, p), col(p)); }
};
// LayerNorm output recomputed on the fly: A[r][p] = LN(x)[r][p] (for dW of an LN-fed Dense)
struct TnLoadLayerNorm {
    const float* __restrict__ X; const float2* __restrict__ stats;
    const float* __restrict__ gamma; const float* __restrict__ beta; int ldx, R, P;
    struct Raw { f32x4 x; float2 s; bool ok; };
    struct Col { f32x4 g, b; };
    __device__ __forceinline__ Col col(int p) const {
        const int pp = p < P ? p : 0;
        return Col{*reinterpret_cast<const f32x4*>(gamma + pp), *reinterpret_cast<const f32x4*>(beta + pp)};
    }
    __device__ __forceinline__ Raw fetch(int r, int p) const {
        const bool ok = r < R && p < P;
        return Raw{*reinterpret_cast<const f32x4*>(X + (ok ? (size_t)r * ldx + p : 0)), stats[ok ? r : 0], ok};
    }
    __device__ __forceinline__ f32x4 finish(const Raw& w, const Col& c) const {
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float inv = w.s.y * c.g[e]; y[e] = w.x[e] * inv + (c.b[e] - w.s.x * inv); }
        return w.ok ? y : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __device__ __forceinline__ f32x4 load(int r, int p) const { return finish(fetch(r, p), col(p)); }
};
// rows of the zero-padded, strided k=3 convolution input (same gather as ALoadConv3)
struct TnLoadConv3 {
    const float* __restrict__ Hin; int C, L_in, L_out, stride, pad_left, R, P;   // R = B*L_out, P = 3*C
    struct Raw { f32x4 x; bool ok; };
    typedef TnNoCol Col;
    __device__ __forceinline__ Col col(int) const { return Col{}; }
    __device__ __forceinline__ Raw fetch(int r, int p) const {
        const int b = r / L_out, t = r - b * L_out;
        const int j = p / C, cc = p - j * C;
        const int src = t * stride - pad_left + j;
        const bool ok = r < R && p < P && src >= 0 && src < L_in;
        return Raw{*reinterpret_cast<const f32x4*>(Hin + (ok ? (size_t)(b * L_in + src) * C + cc : 0)), ok};
    }
    __device__ __forceinline__ f32x4 finish(const Raw& w, const Col&) const { return w.ok ? w.x : (f32x4){0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ f32x4 load(int r, int p) const { return finish(fetch(r, p), col(p)); }
};

// plain store epilogue for split-K reduce / direct store: C[p][q] = v
struct EpStore {
    float* __restrict__ out; int ldo;
    __device__ __forceinline__ float2 colv(int) const { return make_float2(0.f, 0.f); }
    __device__ __forceinline__ float2 pre(int, int) const { return make_float2(0.f, 0.f); }
    __device__ __forceinline__ void store(int row, int col, float v, float2, float2) const { out[(size_t)row * ldo + col] = v; }
};

// C[P][3 seg] stored as three separate [P][seg] tensors (q | k | v weight gradients of one GEMM over the concatenated d q|k|v)
struct EpStore3 {
    float* __restrict__ out0; float* __restrict__ out1; float* __restrict__ out2; int seg;
    __device__ __forceinline__ float2 colv(int) const { return make_float2(0.f, 0.f); }
    __device__ __forceinline__ float2 pre(int, int) const { return make_float2(0.f, 0.f); }
    __device__ __forceinline__ void store(int row, int col, float v, float2, float2) const {
        const int s = col / seg, c = col - s * seg;
        (s == 0 ? out0 : (s == 1 ? out1 : out2))[(size_t)row * seg + c] = v;
    }
};

// C[P][Q] (+= over blockIdx.y slices into slabs) ; 64x64 tile, 4 waves 2x2, BK = 32 rows of R.
// LDS: TA[2][32][68], TB[2][32][68] (r-major).  MFMA 32x32x2: A operand element (i = p, k = r),
// lane (p = lane & 31, h) reads TA[8kk + 4h + s][p] for step s -- ds_read_b32 down a column.
template <class AL, class EP>
__global__ void __launch_bounds__(256)
gemm_tn_kernel(const AL al, const float* __restrict__ Bm, const int ldb, const int R, const int P, const int Q,
               const int p_tiles, const int q_tiles, const int kt_per_split, const EP ep)
{
    constexpr int LD = 68;
    __shared__ __attribute__((aligned(16))) float TA[2 * 32 * LD];
    __shared__ __attribute__((aligned(16))) float TB[2 * 32 * LD];
    const int tile = blockIdx.x;
    const int bp = tile / q_tiles, bq = tile - bp * q_tiles;
    const int p0 = bp * 64, q0 = bq * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int srow = tid >> 4, scol = (tid & 15) * 4;          // 16 rows x 64 columns per pass, 2 passes
    const int KT_all = (R + 31) / 32;
    const int kt_lo = blockIdx.y * kt_per_split;
    const int KT = min(KT_all, kt_lo + kt_per_split);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    typename AL::Raw ra[2];
    const typename AL::Col colc = al.col(p0 + scol);
    f32x4 rb[2];
    bool okb[2];
    auto issue = [&](int kt) {                       // loads only (see the loaders): the loader's arithmetic runs in stage()
        const int r0 = kt * 32;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = r0 + srow + 16 * i;
            ra[i] = al.fetch(r, p0 + scol);
            okb[i] = r < R && q0 + scol < Q;
            rb[i] = *reinterpret_cast<const f32x4*>(Bm + (okb[i] ? (size_t)r * ldb + q0 + scol : 0));
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<f32x4*>(&TA[buf * 32 * LD + (srow + 16 * i) * LD + scol]) = al.finish(ra[i], colc);
            *reinterpret_cast<f32x4*>(&TB[buf * 32 * LD + (srow + 16 * i) * LD + scol]) = okb[i] ? rb[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    if (kt_lo < KT) { issue(kt_lo); stage(kt_lo & 1); }
    __syncthreads();
    const int fp = wm * 32 + (lane & 31), fq = wn * 32 + (lane & 31), fh = (lane >> 5) * 4;
    for (int kt = kt_lo; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) issue(kt + 1);
        const float* Ac = TA + cur * 32 * LD + fh * LD + fp;
        const float* Bc = TB + cur * 32 * LD + fh * LD + fq;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ac[(8 * kk + s) * LD], Bc[(8 * kk + s) * LD], acc, 0, 0, 0);
        if (kt + 1 < KT) stage(cur ^ 1);
        __syncthreads();
    }
    const int crow0 = p0 + wm * 32 + 4 * (lane >> 5), col = q0 + wn * 32 + (lane & 31);
    if (col < Q) {
        const float2 cv = ep.colv(col);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = crow0 + (r & 3) + 8 * (r >> 2);
            if (row < P) ep.store(row, col, acc[r], cv, ep.pre(row, col));
        }
    }
}

// The same product on the f16 matrix cores (f16x3, uu3d_gemm_h3.h): 128 x 128 tile, 4 waves 2 x 2, wave tile 64 x 64 = 2 x 2
// v_mfma_f32_32x32x16_f16 tiles, BK = 32 rows of R per step.  The contraction index is the ROW of both operands, so an
// MFMA fragment (8 consecutive k of one output row / column) runs DOWN a column of the row-major tiles: the tiles are
// staged as f16 planes [hi | lo] x [A | B], [32 rows][128 columns] with 320-byte rows, and the fragments come back
// transposed by gfx950's ds_read_b64_tr_b16 (per 16-lane group a 4-row x 16-column block, delivered column-major: lane
// 4q + p of the group supplies the address of row q, columns 4p .. 4p+3; lane i receives column i).  With 320-byte rows
// the four rows of a block fall on four different 16-bank groups: every read is conflict-free.  The instruction needs
// EXEC all ones -- there is no divergent code around the reads, out-of-range rows / columns are staged as zeros.
static constexpr int TNH_ROW_BYTES = 320;
static constexpr int TNH_PLANE_BYTES = 32 * TNH_ROW_BYTES;
static constexpr int TNH_STAGE_BYTES = 4 * TNH_PLANE_BYTES;          // A hi, A lo, B hi, B lo
static constexpr size_t TNH_LDS_BYTES = 2 * TNH_STAGE_BYTES;
typedef __fp16 tnh_fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) tnh_fp16x4 tnh_lds_fp16x4;

template <class AL, class EP>
__global__ void __launch_bounds__(256, 2)
gemm_tn_h3_kernel(const AL al, const float* __restrict__ Bm, const int ldb, const int R, const int P, const int Q,
                  const int p_tiles, const int q_tiles, const int kt_per_split, const EP ep)
{
    h3_flush_f16_denormals();
    extern __shared__ __attribute__((aligned(16))) unsigned char tnh_smem[];
    const int tile = blockIdx.x;
    const int bp = tile / q_tiles, bq = tile - bp * q_tiles;
    const int p0 = bp * 128, q0 = bq * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int srow = tid >> 5, scol = (tid & 31) * 4;          // 8 rows x 128 columns per pass, 4 passes
    const int KT_all = (R + 31) / 32;
    const int kt_lo = blockIdx.y * kt_per_split;
    const int KT = min(KT_all, kt_lo + kt_per_split);

    f32x16 acc0[2][2], acc1[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }
    typename AL::Raw ra[4];
    const typename AL::Col colc = al.col(p0 + scol);
    f32x4 rb[4];
    bool okb[4];
    auto issue = [&](int kt) {                       // loads only: the arithmetic of the loader runs in stage(), behind the MFMAs
        const int r0 = kt * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + srow + 8 * i;
            ra[i] = al.fetch(r, p0 + scol);
            okb[i] = r < R && q0 + scol < Q;
            rb[i] = *reinterpret_cast<const f32x4*>(Bm + (okb[i] ? (size_t)r * ldb + q0 + scol : 0));
        }
    };
    auto stage = [&](int buf) {
        unsigned char* st = tnh_smem + buf * TNH_STAGE_BYTES + srow * TNH_ROW_BYTES + scol * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            h16x4 hi, lo;
            h3_split(al.finish(ra[i], colc), hi, lo);
            *reinterpret_cast<h16x4*>(st + 8 * i * TNH_ROW_BYTES) = hi;
            *reinterpret_cast<h16x4*>(st + 8 * i * TNH_ROW_BYTES + TNH_PLANE_BYTES) = lo;
            h3_split(okb[i] ? rb[i] : (f32x4){0.f, 0.f, 0.f, 0.f}, hi, lo);
            *reinterpret_cast<h16x4*>(st + 8 * i * TNH_ROW_BYTES + 2 * TNH_PLANE_BYTES) = hi;
            *reinterpret_cast<h16x4*>(st + 8 * i * TNH_ROW_BYTES + 3 * TNH_PLANE_BYTES) = lo;
        }
    };
    if (kt_lo < KT) { issue(kt_lo); stage(kt_lo & 1); }
    __syncthreads();
    // this lane's address inside a plane for the block (k-slice 0, first four rows) of MFMA tile 0 of the wave
    const int frow = 8 * (lane >> 5) + ((lane & 15) >> 2), fcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const int fa = frow * TNH_ROW_BYTES + (wm * 64 + fcol) * 2, fb = frow * TNH_ROW_BYTES + (wn * 64 + fcol) * 2 + 2 * TNH_PLANE_BYTES;
    auto frag = [&](const unsigned char* base) -> h16x8 {          // 8 consecutive k (rows) of this lane's column
        const tnh_fp16x4 u = __builtin_amdgcn_ds_read_tr16_b64_v4f16((tnh_lds_fp16x4*)(base));
        const tnh_fp16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((tnh_lds_fp16x4*)(base + 4 * TNH_ROW_BYTES));
        const h16x4 a = __builtin_bit_cast(h16x4, u), b = __builtin_bit_cast(h16x4, v);
        return (h16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    };
    for (int kt = kt_lo; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) issue(kt + 1);
        const unsigned char* sa = tnh_smem + cur * TNH_STAGE_BYTES + fa;
        const unsigned char* sb = tnh_smem + cur * TNH_STAGE_BYTES + fb;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            h16x8 ah[2], al_[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = frag(sa + s * 16 * TNH_ROW_BYTES + i * 64);
                al_[i] = frag(sa + s * 16 * TNH_ROW_BYTES + i * 64 + TNH_PLANE_BYTES);
                bh[i] = frag(sb + s * 16 * TNH_ROW_BYTES + i * 64);
                bl[i] = frag(sb + s * 16 * TNH_ROW_BYTES + i * 64 + TNH_PLANE_BYTES);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc0[i][j], 0, 0, 0);
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc1[i][j], 0, 0, 0);
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al_[i], bh[j], acc1[i][j], 0, 0, 0);
                }
        }
        if (kt + 1 < KT) stage(cur ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int crow0 = p0 + wm * 64 + 32 * i + 4 * (lane >> 5), col = q0 + wn * 64 + 32 * j + (lane & 31);
            if (col < Q) {
                const float2 cv = ep.colv(col);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = crow0 + (r & 3) + 8 * (r >> 2);
                    if (row < P) ep.store(row, col, acc0[i][j][r] + acc1[i][j][r] * (1.0f / H3_SCALE), cv, ep.pre(row, col));
                }
            }
        }
}

// out[c] (period == 0) or out[(r % period)][c] = sum over rows r of scale(r) * X[r][c].
// One workgroup per 64 columns x row-slice; slices are combined in order by a second pass
// (deterministic).  mask (optional, per row): only rows with mask[r] == want contribute.
static __global__ void __launch_bounds__(256)
colsum_kernel(const float* __restrict__ X, const int ldx, const int R, const int C, const int period,
              const uint8_t* __restrict__ mask, const int want, float* __restrict__ partial, const int slices)
{
    // grid: (ceil(C/64), slices) ; thread (lane = column, 4 row-lanes)
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int P = period > 0 ? period : 1;
    const int rows_per_slice = ((R + slices - 1) / slices + P - 1) / P * P;     // multiple of the period
    const int r_lo = blockIdx.y * rows_per_slice, r_hi = min(R, r_lo + rows_per_slice);
    __shared__ float red[4][64];
    for (int ph = 0; ph < P; ++ph) {
        float s = 0.f;
        if (c < C)
            for (int r = r_lo + ph + rl * P; r < r_hi; r += 4 * P)
                if (mask == nullptr || (int)(mask[r] != 0) == want) s += X[(size_t)r * ldx + c];
        red[rl][threadIdx.x & 63] = s;
        __syncthreads();
        if (rl == 0 && c < C)
            partial[((size_t)blockIdx.y * P + ph) * C + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        __syncthreads();
    }
}
// Periodic column sums (positional-encoding gradients: out[ph][c] = sum over the nb = R / P samples b of X[b * P + ph][c]) with
// C % 4 == 0 and no mask: one thread per 4 outputs and slice of samples, 16-byte loads a sample apart, no LDS, no barrier.
// colsum_kernel walks the P phases one after the other with two barriers each (23 - 40 us for the 71 x 384 temporal encoding).
// grid (ceil(P * C / 4 / 256), slices); partial[slice][P * C].
static __global__ void __launch_bounds__(256)
colsum_period4_kernel(const float* __restrict__ X, const int ldx, const int nb, const int P, const int C, float* __restrict__ partial,
                      const int slices)
{
    const int c4 = C >> 2, o = blockIdx.x * 256 + threadIdx.x;
    if (o >= P * c4) return;
    const int ph = o / c4, c = (o - ph * c4) * 4;
    const int per = (nb + slices - 1) / slices;
    const int b_lo = blockIdx.y * per, b_hi = min(nb, b_lo + per);
    const float* x = X + (size_t)ph * ldx + c;
    const size_t bs = (size_t)P * ldx;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    int b = b_lo;
    for (; b + 3 < b_hi; b += 4) {
        s0 += *reinterpret_cast<const f32x4*>(x + (size_t)b * bs);
        s1 += *reinterpret_cast<const f32x4*>(x + (size_t)(b + 1) * bs);
        s2 += *reinterpret_cast<const f32x4*>(x + (size_t)(b + 2) * bs);
        s3 += *reinterpret_cast<const f32x4*>(x + (size_t)(b + 3) * bs);
    }
    for (; b < b_hi; ++b) s0 += *reinterpret_cast<const f32x4*>(x + (size_t)b * bs);
    *reinterpret_cast<f32x4*>(partial + (size_t)blockIdx.y * P * C + (size_t)ph * C + c) = (s0 + s1) + (s2 + s3);
}
// Fast path of colsum_kernel for the common case (no period, no mask, C % 4 == 0): 16-byte loads, 4 independent
// partial sums per thread.  A workgroup covers cgs = min(64, C / 4) column groups and gives the other 256 / cgs thread
// rows to more matrix rows, so narrow matrices (the spatial stack's 32 .. 96 columns over 77 k rows) use every lane;
// grid (ceil(C / (4 cgs)), slices).
static __global__ void __launch_bounds__(256)
colsum4_kernel(const float* __restrict__ X, const int ldx, const int R, const int C, float* __restrict__ partial, const int slices,
               const int cgs)
{
    const int rlanes = 256 / cgs;
    const int cg = threadIdx.x % cgs, rl = threadIdx.x / cgs;
    const int c = (blockIdx.x * cgs + cg) * 4;
    const int rows_per_slice = (R + slices - 1) / slices;
    const int r_lo = blockIdx.y * rows_per_slice, r_hi = min(R, r_lo + rows_per_slice);
    __shared__ f32x4 red[256];
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    if (c < C && rl < rlanes) {
        int r = r_lo + rl;
        for (; r + 3 * rlanes < r_hi; r += 4 * rlanes) {
            s0 += *reinterpret_cast<const f32x4*>(X + (size_t)r * ldx + c);
            s1 += *reinterpret_cast<const f32x4*>(X + (size_t)(r + rlanes) * ldx + c);
            s2 += *reinterpret_cast<const f32x4*>(X + (size_t)(r + 2 * rlanes) * ldx + c);
            s3 += *reinterpret_cast<const f32x4*>(X + (size_t)(r + 3 * rlanes) * ldx + c);
        }
        for (; r < r_hi; r += rlanes) s0 += *reinterpret_cast<const f32x4*>(X + (size_t)r * ldx + c);
    }
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rl == 0 && c < C) {
        f32x4 t = red[cg];
        for (int k = 1; k < rlanes; ++k) t += red[k * cgs + cg];              // fixed order: deterministic
        *reinterpret_cast<f32x4*>(partial + (size_t)blockIdx.y * C + c) = t;
    }
}
// where element c of a combined result goes: one tensor, or up to three of `seg` elements each (the q | k | v bias gradients)
struct ReduceOut {
    float* p0; float* p1; float* p2; int seg;
    __host__ __device__ __forceinline__ float* at(int c) const { const int s = c / seg; return (s == 0 ? p0 : (s == 1 ? p1 : p2)) + (c - s * seg); }
};
// out[i] (+)= sum_k partial[k * pstride + i], i < n.  Thread (column c of 16, lane q of 16): lane q adds the
// slices k = q, q+QL, ... in order; the QL lane sums are then added in lane order -> deterministic.
template <int QL>
__global__ void __launch_bounds__(16 * QL)
reduce_partials_kernel(const float* __restrict__ partial, const int n, const size_t pstride, const int slices,
                       const ReduceOut out, const int accumulate)
{
    __shared__ float red[QL][17];
    const int cl = threadIdx.x & 15, q = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    // four slices in flight per lane (the loop is the latency of its loads: a 500-slice combine of a 32-float result
    // ran 32 dependent rounds on 2 workgroups); s0..s3 are added in a fixed order
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < n) {
        const float* p = partial + c;
        int k = q;
        for (; k + 3 * QL < slices; k += 4 * QL) {
            const float a0 = p[(size_t)k * pstride], a1 = p[(size_t)(k + QL) * pstride];
            const float a2 = p[(size_t)(k + 2 * QL) * pstride], a3 = p[(size_t)(k + 3 * QL) * pstride];
            s0 += a0; s1 += a1; s2 += a2; s3 += a3;
        }
        for (; k < slices; k += QL) s0 += p[(size_t)k * pstride];
    }
    red[q][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (q == 0 && c < n) {
        float t = red[0][cl];
#pragma unroll
        for (int k = 1; k < QL; ++k) t += red[k][cl];
        float* o = out.at(c);
        *o = accumulate ? *o + t : t;
    }
}
// out[0..n) (+)= sum over `slices` partial rows: 64 lanes per column when there are many slices, 16 otherwise.
inline void launch_reduce_partials(const float* partial, int n, size_t pstride, int slices, float* out, int accumulate, hipStream_t stream);
inline void launch_reduce_partials(const float* partial, int n, size_t pstride, int slices, const ReduceOut out, int accumulate, hipStream_t stream) {
    if (slices >= 128) hipLaunchKernelGGL(reduce_partials_kernel<64>, dim3((n + 15) / 16), dim3(1024), 0, stream, partial, n, pstride, slices, out, accumulate);
    else hipLaunchKernelGGL(reduce_partials_kernel<16>, dim3((n + 15) / 16), dim3(256), 0, stream, partial, n, pstride, slices, out, accumulate);
}
inline void launch_reduce_partials(const float* partial, int n, size_t pstride, int slices, float* out, int accumulate, hipStream_t stream) {
    launch_reduce_partials(partial, n, pstride, slices, ReduceOut{out, nullptr, nullptr, n > 0 ? n : 1}, accumulate, stream);
}

// ------------------------------------------------------------------------------------
// LayerNorm backward.  y = xhat * gamma + beta, xhat = (x - mean) * rstd (stats from row_stats_kernel).
//   g   = dy * gamma ;  dx = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat))
//   dgamma = sum_r dy * xhat ; dbeta = sum_r dy
// One wave per row, RPW rows per wave, 4 waves per workgroup; each workgroup writes its partial
// dgamma / dbeta to partial[wg][2][D]; reduce_partials_kernel adds them in a fixed order (deterministic).
// dx is written, or res + dx when accumulate != 0 (residual-stream gradient; res may be dx_out itself: each element is read
// by the lane that writes it).
// ------------------------------------------------------------------------------------
// gated.out (optional) additionally receives the DropPath-gated copy of the result the next Dense-layer backward reads:
// (dx / keep) * gate[row / rps], the arithmetic of scale_rows_kernel.
struct LnBwdGated { const float* gate; float keep; int rps; float* out; };

template <int MAXV>
__global__ void __launch_bounds__(256)
ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float2* __restrict__ stats,
              const float* __restrict__ gamma, const int ld, const int D, const int M, const int rows_per_wave,
              float* dx_out, const float* res, const int accumulate, float* __restrict__ partial, const LnBwdGated gated)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 dg[MAXV], db[MAXV], gm[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        dg[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; db[i] = dg[i];
        const int c = (i * 64 + lane) * 4;
        gm[i] = (c < D) ? *reinterpret_cast<const f32x4*>(gamma + c) : dg[i];
    }
    const int row0 = (blockIdx.x * 4 + wave) * rows_per_wave;
    // two rows per iteration, every load of both (x, dy, the residual) issued before the first use: with one row after the other
    // the wave waited for each row's loads in turn (12 us for 4544 x 384 = 2.6 TB/s)
    for (int rr = 0; rr < rows_per_wave; rr += 2) {
        int row[2]; bool ok[2]; float2 st[2]; f32x4 xv[2][MAXV], dv[2][MAXV], old[2][MAXV]; float gt[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            row[u] = row0 + rr + u;
            ok[u] = row[u] < M && rr + u < rows_per_wave;            // wave-uniform
            const int rw = ok[u] ? row[u] : 0;
            st[u] = stats[rw];
            gt[u] = gated.out != nullptr ? gated.gate[rw / gated.rps] : 0.f;
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int c = (i * 64 + lane) * 4;
                xv[u][i] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[u][i] = xv[u][i]; old[u][i] = xv[u][i];
                if (c < D) {
                    xv[u][i] = *reinterpret_cast<const f32x4*>(x + (size_t)rw * ld + c);
                    dv[u][i] = *reinterpret_cast<const f32x4*>(dy + (size_t)rw * ld + c);
                    if (accumulate) old[u][i] = *reinterpret_cast<const f32x4*>(res + (size_t)rw * ld + c);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!ok[u]) continue;
            f32x4 xh[MAXV], g[MAXV];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int c = (i * 64 + lane) * 4;
                xh[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; g[i] = xh[i];
                if (c < D) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        xh[i][e] = (xv[u][i][e] - st[u].x) * st[u].y;
                        g[i][e] = dv[u][i][e] * gm[i][e];
                        s1 += g[i][e]; s2 += g[i][e] * xh[i][e];
                        dg[i][e] += dv[u][i][e] * xh[i][e]; db[i][e] += dv[u][i][e];
                    }
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
            const float m1 = s1 / (float)D, m2 = s2 / (float)D;
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int c = (i * 64 + lane) * 4;
                if (c < D) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = st[u].y * (g[i][e] - m1 - xh[i][e] * m2);
                    if (accumulate) o += old[u][i];
                    *reinterpret_cast<f32x4*>(dx_out + (size_t)row[u] * ld + c) = o;
                    if (gated.out != nullptr) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (o[e] / gated.keep) * gt[u];
                        *reinterpret_cast<f32x4*>(gated.out + (size_t)row[u] * ld + c) = o;
                    }
                }
            }
        }
    }
    // combine the 4 waves' partial dgamma/dbeta through LDS, in wave order
    __shared__ __attribute__((aligned(16))) float red[4][2][MAXV * 256];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        *reinterpret_cast<f32x4*>(&red[wave][0][(i * 64 + lane) * 4]) = dg[i];
        *reinterpret_cast<f32x4*>(&red[wave][1][(i * 64 + lane) * 4]) = db[i];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 2 * D; idx += 256) {
        const int which = idx / D, c = idx - which * D;
        partial[((size_t)blockIdx.x * 2 + which) * D + c] = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
    }
}

// LayerNorm backward for NARROW rows (D == 4 * LPR <= 64, the spatial stack's D = 32): LPR lanes per row, 256 / LPR rows
// per workgroup at once (the one-row-per-wave kernel used 8 of 64 lanes and ran its rows one after the other: 43 us for a
// 30 MB pass).  Row group rg of workgroup b takes the rows b*RG*rpg + rr*RG + rg, so that a wave reads consecutive rows.
// dx is bit-identical to ln_bwd_kernel (the LPR-lane butterfly is the tail of the 64-lane one); the dgamma / dbeta partials
// are combined over the row groups in group order.
template <int LPR>
__global__ void __launch_bounds__(256)
ln_bwd_narrow_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float2* __restrict__ stats,
                     const float* __restrict__ gamma, const int ld, const int M, const int rows_per_group,
                     float* dx_out, const float* res, const int accumulate, float* __restrict__ partial, const LnBwdGated gated)
{
    constexpr int RG = 256 / LPR, D = 4 * LPR;
    const int l = threadIdx.x % LPR, rg = threadIdx.x / LPR;
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + 4 * l);
    f32x4 dg = (f32x4){0.f, 0.f, 0.f, 0.f}, db = dg;
    const int base = blockIdx.x * RG * rows_per_group + rg;

    constexpr int NU = 4;
    // NU rows per iteration, every load of all of them issued before the first use: the loop was the latency of one row's loads after the
    // other (15 us for the spatial stack's 77 k rows = 3.3 TB/s)
    for (int rr = 0; rr < rows_per_group; rr += NU) {
        int row[NU]; bool ok[NU]; float2 st[NU]; f32x4 xv[NU], dv[NU], old[NU]; float gt[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            row[u] = base + (rr + u) * RG;
            ok[u] = row[u] < M && rr + u < rows_per_group;
            const int rw = ok[u] ? row[u] : 0;
            st[u] = stats[rw];
            xv[u] = *reinterpret_cast<const f32x4*>(x + (size_t)rw * ld + 4 * l);
            dv[u] = *reinterpret_cast<const f32x4*>(dy + (size_t)rw * ld + 4 * l);
            old[u] = accumulate ? *reinterpret_cast<const f32x4*>(res + (size_t)rw * ld + 4 * l) : (f32x4){0.f, 0.f, 0.f, 0.f};
            gt[u] = gated.out != nullptr ? gated.gate[rw / gated.rps] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (!ok[u]) { xv[u] = (f32x4){st[u].x, st[u].x, st[u].x, st[u].x}; dv[u] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
            f32x4 xh, g;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xh[e] = (xv[u][e] - st[u].x) * st[u].y;
                g[e] = dv[u][e] * gm[e];
                s1 += g[e]; s2 += g[e] * xh[e];
                dg[e] += dv[u][e] * xh[e]; db[e] += dv[u][e];
            }
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
            const float m1 = s1 / (float)D, m2 = s2 / (float)D;
            if (ok[u]) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = st[u].y * (g[e] - m1 - xh[e] * m2);
                if (accumulate) o += old[u];
                *reinterpret_cast<f32x4*>(dx_out + (size_t)row[u] * ld + 4 * l) = o;
                if (gated.out != nullptr) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (o[e] / gated.keep) * gt[u];
                    *reinterpret_cast<f32x4*>(gated.out + (size_t)row[u] * ld + 4 * l) = o;
                }
            }
        }
    }
    __shared__ __attribute__((aligned(16))) float red[RG][2][D];
    *reinterpret_cast<f32x4*>(&red[rg][0][4 * l]) = dg;
    *reinterpret_cast<f32x4*>(&red[rg][1][4 * l]) = db;
    __syncthreads();
    if (threadIdx.x < 2 * D) {
        const int which = threadIdx.x / D, c = threadIdx.x - which * D;
        float t = red[0][which][c];
#pragma unroll 8
        for (int k = 1; k < RG; ++k) t += red[k][which][c];
        partial[((size_t)blockIdx.x * 2 + which) * D + c] = t;
    }
}

// ------------------------------------------------------------------------------------
// Generic softmax attention, forward and backward, one (sequence, head) per workgroup, thread = row.
// Used for the training path of BOTH stacks (spatial: L = 17, d_h = 4; temporal/strided: L <= 128,
// d_h = 48).  qkv rows are [q | k | v] with head h at channels [h*DH, (h+1)*DH).
//   forward : S = Q K^T / sqrt(d_h) (+ (1-mask) * -1e9), P = softmax(S), O = P V
//   backward: dV = P^T dO ; dP = dO V^T ; dS = P * (dP - rowsum(dP * P)) ; dQ = dS K / sqrt(d) ; dK = dS^T Q / sqrt(d)
// LDS: K, V, (Q, dO) tiles [L][DH+4] and the P / dS matrices [L][L+1].
// ------------------------------------------------------------------------------------
template <int DH>
__host__ __device__ inline size_t attn_generic_lds_bytes(int L, bool backward) {
    const size_t tile = (size_t)L * (DH + 4);
    return sizeof(float) * (backward ? 4 * tile + 2 * (size_t)L * (L + 1) : 2 * tile);
}

template <int DH>
__global__ void __launch_bounds__(128)
attn_generic_fwd_kernel(const float* __restrict__ qkv, const int ld, const int D, const int L, const int H,
                        const uint8_t* __restrict__ key_mask, float* __restrict__ out, const int ldo,
                        const int pack, const int total, const DropCfg drop = DropCfg{})
{
    // `pack` (sequence, head) pairs share a workgroup, L threads each (L = 17 leaves 119 of 128 lanes busy instead of 17)
    constexpr int LD = DH + 4;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int slot = threadIdx.x / L, i = threadIdx.x - slot * L;
    const int bh = blockIdx.x * pack + slot;
    const bool live = slot < pack && bh < total;
    float* Ks = sm + (size_t)min(slot, pack - 1) * 2 * L * LD; float* Vs = Ks + L * LD;
    const int b = live ? bh / H : 0, h = live ? bh - b * H : 0;
    const float* base = qkv + (size_t)b * L * ld + h * DH;
    float q[DH], o[DH];
    if constexpr (DH == 4) {                 // a head's row is one 16-byte piece: thread i moves row i (4 loads instead of 12 scalar ones)
        if (live) {
            *reinterpret_cast<f32x4*>(&Ks[i * LD]) = *reinterpret_cast<const f32x4*>(base + (size_t)i * ld + D);
            *reinterpret_cast<f32x4*>(&Vs[i * LD]) = *reinterpret_cast<const f32x4*>(base + (size_t)i * ld + 2 * D);
            const f32x4 q4 = *reinterpret_cast<const f32x4*>(base + (size_t)i * ld);
#pragma unroll
            for (int c = 0; c < DH; ++c) q[c] = q4[c];
        }
    } else if (live) {
        for (int idx = i; idx < L * DH; idx += L) {
            const int r = idx / DH, c = idx - r * DH;
            Ks[r * LD + c] = base[(size_t)r * ld + D + c];
            Vs[r * LD + c] = base[(size_t)r * ld + 2 * D + c];
        }
#pragma unroll
        for (int c = 0; c < DH; ++c) q[c] = base[(size_t)i * ld + c];
    }
    __syncthreads();
    if (!live) return;
#pragma unroll
    for (int c = 0; c < DH; ++c) o[c] = 0.f;
    const float rsq = 1.0f / sqrtf((float)DH);          // multiplications by reciprocals and exp2 instead of divisions and libm expf
    float mx = -INFINITY;
    for (int j = 0; j < L; ++j) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < DH; ++c) s = fmaf(q[c], Ks[j * LD + c], s);
        s = s * rsq;
        if (key_mask != nullptr) s += (key_mask[(size_t)b * L + j] ? 0.0f : 1.0f) * -1e9f;
        mx = fmaxf(mx, s);
    }
    float sum = 0.f;
    for (int j = 0; j < L; ++j) {           // recompute (no per-thread score array of run-time length)
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < DH; ++c) s = fmaf(q[c], Ks[j * LD + c], s);
        s = s * rsq;
        if (key_mask != nullptr) s += (key_mask[(size_t)b * L + j] ? 0.0f : 1.0f) * -1e9f;
        const float e = __builtin_amdgcn_exp2f((s - mx) * 1.44269504088896341f);
        sum += e;
        // Dropout on the attention WEIGHTS (after the softmax, vision_transformer.py:126-129): the row sum is that of the full row
        const float ed = drop.on() ? e * drop_factor(drop, ((unsigned long long)bh * L + i) * L + j) : e;
#pragma unroll
        for (int c = 0; c < DH; ++c) o[c] = fmaf(ed, Vs[j * LD + c], o[c]);
    }
    const float rsum = 1.0f / sum;
    if constexpr (DH == 4) {
        *reinterpret_cast<f32x4*>(out + ((size_t)b * L + i) * ldo + h * DH) = (f32x4){o[0] * rsum, o[1] * rsum, o[2] * rsum, o[3] * rsum};
    } else {
#pragma unroll
        for (int c = 0; c < DH; ++c) out[((size_t)b * L + i) * ldo + h * DH + c] = o[c] * rsum;
    }
}

template <int DH>
__global__ void __launch_bounds__(128)
attn_generic_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dO, const int ld, const int D,
                        const int L, const int H, const uint8_t* __restrict__ key_mask,
                        float* __restrict__ dqkv /* same layout as qkv */, const int ldo,
                        const int pack, const int total, const DropCfg drop = DropCfg{})
{
    // with Dropout on the attention weights (P' = M * P, O = P' V): dV = P'^T dO, dP = M * (dO V^T), dS = P * (dP - rowsum(dP * P))
    constexpr int LD = DH + 4;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int LP = L + 1;
    const int slot = threadIdx.x / L, i = threadIdx.x - slot * L;       // `pack` pairs per workgroup, see the forward kernel
    const int bh = blockIdx.x * pack + slot;
    const bool live = slot < pack && bh < total;
    float* Qs = sm + (size_t)min(slot, pack - 1) * (4 * L * LD + 2 * L * LP);
    float* Ks = Qs + L * LD; float* Vs = Ks + L * LD; float* Gs = Vs + L * LD;   // Gs = dO
    float* Pm = Gs + L * LD;                 // [L][L+1]
    float* Sm = Pm + L * LP;                 // dS
    const int b = live ? bh / H : 0, h = live ? bh - b * H : 0;
    const float* base = qkv + (size_t)b * L * ld + h * DH;
    if constexpr (DH == 4) {
        if (live) {
            *reinterpret_cast<f32x4*>(&Qs[i * LD]) = *reinterpret_cast<const f32x4*>(base + (size_t)i * ld);
            *reinterpret_cast<f32x4*>(&Ks[i * LD]) = *reinterpret_cast<const f32x4*>(base + (size_t)i * ld + D);
            *reinterpret_cast<f32x4*>(&Vs[i * LD]) = *reinterpret_cast<const f32x4*>(base + (size_t)i * ld + 2 * D);
            *reinterpret_cast<f32x4*>(&Gs[i * LD]) = *reinterpret_cast<const f32x4*>(dO + ((size_t)b * L + i) * ldo + h * DH);
        }
    } else if (live)
        for (int idx = i; idx < L * DH; idx += L) {
            const int r = idx / DH, c = idx - r * DH;
            Qs[r * LD + c] = base[(size_t)r * ld + c];
            Ks[r * LD + c] = base[(size_t)r * ld + D + c];
            Vs[r * LD + c] = base[(size_t)r * ld + 2 * D + c];
            Gs[r * LD + c] = dO[((size_t)b * L + r) * ldo + h * DH + c];
        }
    __syncthreads();
    const float rsq = 1.0f / sqrtf((float)DH);
    if (live) {
        float q[DH], g[DH];
#pragma unroll
        for (int c = 0; c < DH; ++c) { q[c] = Qs[i * LD + c]; g[c] = Gs[i * LD + c]; }
        float mx = -INFINITY;
        for (int j = 0; j < L; ++j) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < DH; ++c) s = fmaf(q[c], Ks[j * LD + c], s);
            s = s * rsq;
            if (key_mask != nullptr) s += (key_mask[(size_t)b * L + j] ? 0.0f : 1.0f) * -1e9f;
            Pm[i * LP + j] = s;
            mx = fmaxf(mx, s);
        }
        float sum = 0.f;
        for (int j = 0; j < L; ++j) { const float e = __builtin_amdgcn_exp2f((Pm[i * LP + j] - mx) * 1.44269504088896341f); Pm[i * LP + j] = e; sum += e; }
        float delta = 0.f;
        const float rsum = 1.0f / sum;
        for (int j = 0; j < L; ++j) {
            const float pij = Pm[i * LP + j] * rsum;
            float dp = 0.f;
#pragma unroll
            for (int c = 0; c < DH; ++c) dp = fmaf(g[c], Vs[j * LD + c], dp);
            if (drop.on()) dp *= drop_factor(drop, ((unsigned long long)bh * L + i) * L + j);
            Pm[i * LP + j] = pij; Sm[i * LP + j] = dp;
            delta = fmaf(pij, dp, delta);
        }
        float dq[DH];
#pragma unroll
        for (int c = 0; c < DH; ++c) dq[c] = 0.f;
        for (int j = 0; j < L; ++j) {
            const float ds = Pm[i * LP + j] * (Sm[i * LP + j] - delta);
            Sm[i * LP + j] = ds;
#pragma unroll
            for (int c = 0; c < DH; ++c) dq[c] = fmaf(ds, Ks[j * LD + c], dq[c]);
        }
        if constexpr (DH == 4) {
            *reinterpret_cast<f32x4*>(dqkv + ((size_t)b * L + i) * ld + h * DH) = (f32x4){dq[0] * rsq, dq[1] * rsq, dq[2] * rsq, dq[3] * rsq};
        } else {
#pragma unroll
            for (int c = 0; c < DH; ++c) dqkv[((size_t)b * L + i) * ld + h * DH + c] = dq[c] * rsq;
        }
    }
    __syncthreads();
    if (live) {                                  // thread = key row j
        float dk[DH], dv[DH];
#pragma unroll
        for (int c = 0; c < DH; ++c) { dk[c] = 0.f; dv[c] = 0.f; }
        for (int r = 0; r < L; ++r) {
            const float ds = Sm[r * LP + i];
            float pr = Pm[r * LP + i];
            if (drop.on()) pr *= drop_factor(drop, ((unsigned long long)bh * L + r) * L + i);
#pragma unroll
            for (int c = 0; c < DH; ++c) { dk[c] = fmaf(ds, Qs[r * LD + c], dk[c]); dv[c] = fmaf(pr, Gs[r * LD + c], dv[c]); }
        }
        if constexpr (DH == 4) {
            *reinterpret_cast<f32x4*>(dqkv + ((size_t)b * L + i) * ld + D + h * DH) = (f32x4){dk[0] * rsq, dk[1] * rsq, dk[2] * rsq, dk[3] * rsq};
            *reinterpret_cast<f32x4*>(dqkv + ((size_t)b * L + i) * ld + 2 * D + h * DH) = (f32x4){dv[0], dv[1], dv[2], dv[3]};
        } else {
#pragma unroll
            for (int c = 0; c < DH; ++c) {
                dqkv[((size_t)b * L + i) * ld + D + h * DH + c] = dk[c] * rsq;
                dqkv[((size_t)b * L + i) * ld + 2 * D + h * DH + c] = dv[c];
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// The spatial stack's attention backward: L = 17 joints, head dim 4, no key mask -- attn_generic_bwd_kernel<4> with every
// loop unrolled over a compile-time L, the probability / dS rows of a query in REGISTERS, K / V / Q / dO rows read as 16-byte
// LDS broadcasts in batches, and P^T / dS^T written to LDS once for the per-key pass.  Same arithmetic order as the generic
// kernel (bitwise equal results); 45.8 -> see DESIGN.md section 9.  `pack` (sequence, head) pairs per workgroup of 128 threads.
template <int L>
__host__ __device__ inline constexpr size_t attn_small_bwd_lds_bytes() { return (size_t)(4 * L * 4 + 2 * L * ((L + 3) / 4 * 4)) * sizeof(float); }
template <int L>
__global__ void __launch_bounds__(128)
attn_small_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dO, const int ld, const int D, const int H,
                      float* __restrict__ dqkv, const int ldo, const int pack, const int total)
{
    constexpr int DH = 4, LP = (L + 3) / 4 * 4;          // padded row of the transposed P / dS tiles
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int slot = threadIdx.x / L, i = threadIdx.x - slot * L;
    const int bh = blockIdx.x * pack + slot;
    const bool live = slot < pack && bh < total;
    float* Qs = sm + (size_t)min(slot, pack - 1) * (4 * L * 4 + 2 * L * LP);
    float* Ks = Qs + L * 4; float* Vs = Ks + L * 4; float* Gs = Vs + L * 4;
    float* Pt = Gs + L * 4;                   // [key][query]
    float* St = Pt + L * LP;                  // dS^T
    const int b = live ? bh / H : 0, h = live ? bh - b * H : 0;
    const float* base = qkv + ((size_t)b * L + i) * ld + h * DH;
    f32x4 q = {0.f, 0.f, 0.f, 0.f}, g = q;
    if (live) {
        q = *reinterpret_cast<const f32x4*>(base);
        g = *reinterpret_cast<const f32x4*>(dO + ((size_t)b * L + i) * ldo + h * DH);
        *reinterpret_cast<f32x4*>(&Qs[i * 4]) = q;
        *reinterpret_cast<f32x4*>(&Ks[i * 4]) = *reinterpret_cast<const f32x4*>(base + D);
        *reinterpret_cast<f32x4*>(&Vs[i * 4]) = *reinterpret_cast<const f32x4*>(base + 2 * D);
        *reinterpret_cast<f32x4*>(&Gs[i * 4]) = g;
    }
    __syncthreads();
    const float rsq = 1.0f / sqrtf((float)DH);
    if (live) {
        float p[L], ds[L];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < L; ++j) {
            const f32x4 k = *reinterpret_cast<const f32x4*>(&Ks[j * 4]);
            float sc = 0.f;
#pragma unroll
            for (int c = 0; c < DH; ++c) sc = fmaf(q[c], k[c], sc);
            sc = sc * rsq;
            p[j] = sc; mx = fmaxf(mx, sc);
        }
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < L; ++j) { p[j] = __builtin_amdgcn_exp2f((p[j] - mx) * 1.44269504088896341f); sum += p[j]; }
        const float rsum = 1.0f / sum;
        float delta = 0.f;
#pragma unroll
        for (int j = 0; j < L; ++j) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&Vs[j * 4]);
            p[j] = p[j] * rsum;
            float dp = 0.f;
#pragma unroll
            for (int c = 0; c < DH; ++c) dp = fmaf(g[c], v[c], dp);
            ds[j] = dp;
            delta = fmaf(p[j], dp, delta);
        }
        f32x4 dq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < L; ++j) {
            const f32x4 k = *reinterpret_cast<const f32x4*>(&Ks[j * 4]);
            ds[j] = p[j] * (ds[j] - delta);
#pragma unroll
            for (int c = 0; c < DH; ++c) dq[c] = fmaf(ds[j], k[c], dq[c]);
            Pt[j * LP + i] = p[j]; St[j * LP + i] = ds[j];
        }
        *reinterpret_cast<f32x4*>(dqkv + ((size_t)b * L + i) * ld + h * DH) = (f32x4){dq[0] * rsq, dq[1] * rsq, dq[2] * rsq, dq[3] * rsq};
    }
    __syncthreads();
    if (live) {                                  // thread = key row i
        f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dv = dk;
#pragma unroll
        for (int r = 0; r < L; ++r) {
            const float dsr = St[i * LP + r], pr = Pt[i * LP + r];
            const f32x4 qq = *reinterpret_cast<const f32x4*>(&Qs[r * 4]), gg = *reinterpret_cast<const f32x4*>(&Gs[r * 4]);
#pragma unroll
            for (int c = 0; c < DH; ++c) { dk[c] = fmaf(dsr, qq[c], dk[c]); dv[c] = fmaf(pr, gg[c], dv[c]); }
        }
        *reinterpret_cast<f32x4*>(dqkv + ((size_t)b * L + i) * ld + D + h * DH) = (f32x4){dk[0] * rsq, dk[1] * rsq, dk[2] * rsq, dk[3] * rsq};
        *reinterpret_cast<f32x4*>(dqkv + ((size_t)b * L + i) * ld + 2 * D + h * DH) = dv;
    }
}

// ------------------------------------------------------------------------------------
// Softmax-attention backward on the matrix pipe (v_mfma_f32_16x16x4_f32, exact f32), one workgroup per (sequence,
// head), one wave per tile of 16 tokens -- the structure of attn_f32_kernel (uu3d_attn.h).  Two passes, both keep
// their L x L tiles in REGISTERS:
//   pass 1, wave = query tile: S^T = K Q^T and dP^T = V dO^T land as [query = lane & 15][key = 16 j + 4 g + r];
//           softmax over keys, delta = sum_k P dP, dS = P (dP - delta); row max / sum / delta go to LDS;
//           dQ = dS K uses the registers directly as the A operand (the P V trick of the forward kernel).
//   pass 2, wave = key tile: S = Q K^T and dP = dO V^T are RECOMPUTED in the other orientation,
//           [key = lane & 15][query = 16 j + 4 g + r], P and dS rebuilt from the saved row statistics, and
//           dK = dS^T Q, dV = P^T dO again consume the registers as A operands.
// The generic kernel holds P and dS in LDS and does one scalar LDS read per FMA: 147 us per launch at L = 71,
// B = 64; recomputing two small products instead costs ~240 extra MFMAs per wave and no L x L LDS traffic.
template <int NT, int DH>
__global__ void __launch_bounds__(64 * NT)
attn_bwd_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ dO, const int ld, const int D,
                     const int L, const int H, const uint8_t* __restrict__ key_mask,
                     float* __restrict__ dqkv, const int ldo)
{
    static_assert(DH % 16 == 0, "head dim must be a multiple of 16");
    constexpr int LD = DH + 4, F4 = DH / 4, KT = DH / 16, ROWS = NT * 16;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Qs = sm; float* Ks = Qs + ROWS * LD; float* Vs = Ks + ROWS * LD; float* Gs = Vs + ROWS * LD;
    float* Mx = Gs + ROWS * LD; float* Sum = Mx + ROWS; float* Dl = Sum + ROWS;

    const int bh = blockIdx.x, b = bh / H, h = bh - b * H, tid = threadIdx.x;
    const float* base = qkv + (size_t)b * L * ld + h * DH;
    for (int idx = tid; idx < ROWS * F4; idx += 64 * NT) {
        const int row = idx / F4, c4 = (idx - row * F4) * 4;
        f32x4 q = {0.f, 0.f, 0.f, 0.f}, k = q, v = q, g = q;
        if (row < L) {
            const float* p = base + (size_t)row * ld + c4;
            q = *reinterpret_cast<const f32x4*>(p);
            k = *reinterpret_cast<const f32x4*>(p + D);
            v = *reinterpret_cast<const f32x4*>(p + 2 * D);
            g = *reinterpret_cast<const f32x4*>(dO + ((size_t)b * L + row) * ldo + h * DH + c4);
        }
        *reinterpret_cast<f32x4*>(&Qs[row * LD + c4]) = q; *reinterpret_cast<f32x4*>(&Ks[row * LD + c4]) = k;
        *reinterpret_cast<f32x4*>(&Vs[row * LD + c4]) = v; *reinterpret_cast<f32x4*>(&Gs[row * LD + c4]) = g;
    }
    __syncthreads();

    const int lane = tid & 63, w = tid >> 6, qi = lane & 15, g = lane >> 4;
    const float scale_mul = 1.0f / sqrtf((float)DH);      // reciprocal multiplies and exp2 in both passes (same P in both)
    float* dq_out = dqkv + (size_t)b * L * ld + h * DH;

    // ---------------- pass 1: this wave's 16 queries against all keys ----------------
    {
        f32x4 qf[KT], gf[KT];
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            qf[t] = *reinterpret_cast<const f32x4*>(&Qs[(16 * w + qi) * LD + 16 * t + 4 * g]);
            gf[t] = *reinterpret_cast<const f32x4*>(&Gs[(16 * w + qi) * LD + 16 * t + 4 * g]);
        }
        f32x4 st[NT], dp[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, d = a;
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(&Ks[(16 * j + qi) * LD + 16 * t + 4 * g]);
                const f32x4 vf = *reinterpret_cast<const f32x4*>(&Vs[(16 * j + qi) * LD + 16 * t + 4 * g]);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s], qf[t][s], a, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[s], gf[t][s], d, 0, 0, 0);
                }
            }
            st[j] = a; dp[j] = d;
        }
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * j + 4 * g + r;
                // clamped, branch-free byte load (a test around it costs a branch + wait per key)
                const uint8_t mk = (key_mask != nullptr) ? key_mask[(size_t)b * L + min(key, L - 1)] : (uint8_t)1;
                const float v = (key < L) ? st[j][r] * scale_mul + (mk ? 0.0f : 1.0f) * -1e9f : -INFINITY;
                st[j][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float e = __builtin_amdgcn_exp2f((st[j][r] - mx) * 1.44269504088896341f); st[j][r] = e; sum += e; }
        sum += __shfl_xor(sum, 16); sum += __shfl_xor(sum, 32);
        const float rsum = 1.0f / sum;
        float delta = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) { st[j][r] = st[j][r] * rsum; delta = fmaf(st[j][r], dp[j][r], delta); }
        delta += __shfl_xor(delta, 16); delta += __shfl_xor(delta, 32);
        if (g == 0) { Mx[16 * w + qi] = mx; Sum[16 * w + qi] = rsum; Dl[16 * w + qi] = delta; }
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) st[j][r] = st[j][r] * (dp[j][r] - delta);          // dS
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    o = __builtin_amdgcn_mfma_f32_16x16x4f32(st[j][s], Ks[(16 * j + 4 * g + s) * LD + 16 * t + qi], o, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = 16 * w + 4 * g + r;
                if (q < L) dq_out[(size_t)q * ld + 16 * t + qi] = o[r] * scale_mul;
            }
        }
    }
    __syncthreads();

    // ---------------- pass 2: this wave's 16 keys against all queries ----------------
    {
        const int key = 16 * w + qi;
        f32x4 kf[KT], vf[KT];
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            kf[t] = *reinterpret_cast<const f32x4*>(&Ks[key * LD + 16 * t + 4 * g]);
            vf[t] = *reinterpret_cast<const f32x4*>(&Vs[key * LD + 16 * t + 4 * g]);
        }
        float madd = 0.f;
        if (key_mask != nullptr && key < L) madd = (key_mask[(size_t)b * L + key] ? 0.0f : 1.0f) * -1e9f;
        f32x4 pp[NT], ds[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, d = a;
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                const f32x4 qa = *reinterpret_cast<const f32x4*>(&Qs[(16 * j + qi) * LD + 16 * t + 4 * g]);
                const f32x4 ga = *reinterpret_cast<const f32x4*>(&Gs[(16 * j + qi) * LD + 16 * t + 4 * g]);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    a = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[s], kf[t][s], a, 0, 0, 0);      // S[query 16j+4g+r][key]
                    d = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s], vf[t][s], d, 0, 0, 0);      // dP[query][key]
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = 16 * j + 4 * g + r;
                float p = 0.f, dsv = 0.f;
                if (q < L && key < L) {
                    p = __builtin_amdgcn_exp2f(((a[r] * scale_mul + madd) - Mx[q]) * 1.44269504088896341f) * Sum[q];      // Sum holds 1 / sum
                    dsv = p * (d[r] - Dl[q]);
                }
                pp[j][r] = p; ds[j][r] = dsv;
            }
        }
        float* dk_out = dq_out + D;
        float* dv_out = dq_out + 2 * D;
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            f32x4 ok = {0.f, 0.f, 0.f, 0.f}, ov = ok;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    ok = __builtin_amdgcn_mfma_f32_16x16x4f32(ds[j][s], Qs[(16 * j + 4 * g + s) * LD + 16 * t + qi], ok, 0, 0, 0);
                    ov = __builtin_amdgcn_mfma_f32_16x16x4f32(pp[j][s], Gs[(16 * j + 4 * g + s) * LD + 16 * t + qi], ov, 0, 0, 0);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kr = 16 * w + 4 * g + r;
                if (kr < L) { dk_out[(size_t)kr * ld + 16 * t + qi] = ok[r] * scale_mul; dv_out[(size_t)kr * ld + 16 * t + qi] = ov[r]; }
            }
        }
    }
}
template <int DH>
__host__ __device__ inline constexpr size_t attn_bwd_mfma_lds_bytes(int NT) { return (size_t)NT * 16 * (4 * (DH + 4) + 3) * sizeof(float); }

}  // namespace uu3d
