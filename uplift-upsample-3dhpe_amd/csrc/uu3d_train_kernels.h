// uu3d_train_kernels.h -- small kernels, loaders and epilogues used only by the training step
// (training-mode forward that keeps activations, DropPath, and the backward chain).
#pragma once
#include "uu3d_dropout.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "uu3d_gemm.h"
#include "uu3d_gemm_h3.h"

namespace uu3d {

// ---- weight repacking on the device -------------------------------------------------------------
// The optimizer updates ONE flat master buffer (Keras layouts, inventory order).  The GEMM kernels
// want k-contiguous, padded operands; this table-driven kernel regenerates every packed operand from
// the master buffer after each optimizer step (padding stays zero from allocation).
//   kind 1: forward operand   Bt[(n0 + n) * ld + k]        = W[k][n]      (ld = Kp)
//   kind 4: backward operand  WK[k * ld + n0 + n]          = W[k][n]      (dX = dY W^T reads rows of W)
//   kind 5: conv-transpose    WT[c * ld + j * N + n]       = Wc[j][c][n]  (K = 3*C rows (j, c), ld = 3*N)
struct PackDesc { long long src, dst; int kind, K, N, ld, n0, C; };

// Blocks per descriptor (host and device agree through repack_blocks): kind 1 is a TRANSPOSE and goes through LDS in 32 x 32
// tiles, reads and writes both in 128-byte runs (as a flat copy its 10 M stores were 4 bytes each, ld apart: 138 us per
// optimizer step against 30 for the tiled form); the other kinds are copies with a different row stride, 1024 elements per block.
__host__ __device__ inline int repack_blocks(const int kind, const int K, const int N) {
    return kind == 1 ? ((K + 31) / 32) * ((N + 31) / 32) : (int)(((long long)K * N + 1023) / 1024);
}
static __global__ void __launch_bounds__(256)
repack_kernel(const float* __restrict__ master, float* __restrict__ arena, const PackDesc* __restrict__ desc,
              const int* __restrict__ blk_first, const int ndesc,
              _Float16* __restrict__ arena_h, const long long plane_halfs)    // f16x3 training: the same element as hi / lo halfs at the same
                                                                              // index of two planes (nullptr: none) -- one pass instead of a second
                                                                              // kernel over the whole arena (split_rows_kernel, 26 us per step)
{
    h3_flush_f16_denormals();
    auto put = [&](const long long at, const float v) __attribute__((always_inline)) {
        arena[at] = v;
        if (arena_h != nullptr) { const _Float16 h = h3_hi(v); arena_h[at] = h; arena_h[at + plane_halfs] = (_Float16)((v - (float)h) * H3_SCALE); }
    };
    int lo = 0, hi = ndesc - 1;                       // last descriptor whose first block <= blockIdx.x
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (blk_first[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
    const PackDesc d = desc[lo];
    const int bt = (int)blockIdx.x - blk_first[lo];
    if (d.kind == 1) {                                 // block-uniform
        __shared__ float tile[32][33];
        const int ntn = (d.N + 31) / 32, tk = bt / ntn, tn = bt - tk * ntn;
        const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = tk * 32 + r + 8 * i, n = tn * 32 + c;
            tile[r + 8 * i][c] = (k < d.K && n < d.N) ? master[d.src + (long long)k * d.N + n] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = tn * 32 + r + 8 * i, k = tk * 32 + c;
            if (k < d.K && n < d.N) put(d.dst + (long long)(d.n0 + n) * d.ld + k, tile[c][r + 8 * i]);
        }
        return;
    }
    const long long total = (long long)d.K * d.N;
    const long long base = (long long)bt * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long e = base + i * 256 + threadIdx.x;
        if (e >= total) break;
        const int k = (int)(e / d.N), n = (int)(e - (long long)k * d.N);
        const float v = master[d.src + e];
        long long o;
        if (d.kind == 4) o = (long long)k * d.ld + d.n0 + n;
        else { const int j = k / d.C, c = k - j * d.C; o = (long long)c * d.ld + (long long)j * d.N + n; }
        put(d.dst + o, v);
    }
}

// Operands of the fused f16x3 spatial stack (uu3d_spatial_h3.h) for the training-mode forward, regenerated from the master
// buffer after each optimizer step: per block the A-operand fragments of W^T ([n-tile][kk][plane hi / lo][lane][8], element
// = split(W[16 kk + 8 (lane >> 5) + e][32 nt + (lane & 31)]), SpatialFragLayoutH3) and the 352 LayerNorm parameters / biases
// in SpatialBlockLayoutV2 order.  Offsets are into the master buffer; < 0: tensor absent (qkv_bias false) -> zeros.
struct SpatialPackSrc { long long ln1_g, ln1_b, wq, bq, wk, bk, wv, bv, wp, bp, ln2_g, ln2_b, w1, b1, w2, b2; };
static __global__ void __launch_bounds__(256)
spatial_train_pack_kernel(const float* __restrict__ master, const SpatialPackSrc* __restrict__ src, _Float16* __restrict__ frag,
                          float* __restrict__ blocks, const int frag_stride, const int blk_stride)
{
    h3_flush_f16_denormals();
    const SpatialPackSrc d = src[blockIdx.x];
    _Float16* F = frag + (size_t)blockIdx.x * frag_stride;
    float* Bk = blocks + (size_t)blockIdx.x * blk_stride;
    // fragments: matrices (offset in halfs, source, K, N) in SpatialFragLayoutH3 order
    const long long ms[6] = {d.wq, d.wk, d.wv, d.wp, d.w1, d.w2};
    const int mk[6] = {32, 32, 32, 32, 32, 64}, mn[6] = {32, 32, 32, 32, 64, 32};
    int off = 0;
    for (int mi = 0; mi < 6; ++mi) {
        const int K = mk[mi], N = mn[mi], total = K * N;
        for (int e0 = threadIdx.x; e0 < total; e0 += 256) {
            // element index inside the matrix's fragments: ((nt * KK + kk) * 64 + lane) * 8 + e  (per plane)
            const int e = e0 & 7, lane = (e0 >> 3) & 63, t = e0 >> 9, KK = K / 16, kk = t % KK, nt = t / KK;
            const float x = master[ms[mi] + (long long)(16 * kk + 8 * (lane >> 5) + e) * N + 32 * nt + (lane & 31)];
            const _Float16 h = h3_hi(x);
            const size_t at = (size_t)off + (((size_t)(nt * KK + kk) * 2) * 64 + lane) * 8 + e;
            F[at] = h;
            F[at + 64 * 8] = (_Float16)((x - (float)h) * H3_SCALE);
        }
        off += 2 * total;
    }
    // parameters: ln1_g, ln1_b, ln2_g, ln2_b, bq, bk, bv, bp (32 each), b1 (64), b2 (32)
    const long long ps[10] = {d.ln1_g, d.ln1_b, d.ln2_g, d.ln2_b, d.bq, d.bk, d.bv, d.bp, d.b1, d.b2};
    const int pn[10] = {32, 32, 32, 32, 32, 32, 32, 32, 64, 32};
    int po = 0;
    for (int pi = 0; pi < 10; ++pi) {
        if ((int)threadIdx.x < pn[pi]) Bk[po + threadIdx.x] = ps[pi] >= 0 ? master[ps[pi] + threadIdx.x] : 0.f;
        po += pn[pi];
    }
}

// Row-panel GEMM operands (uu3d_gemm_panel.h: fragment-ordered f16 planes, [32-column chunk][k-step of 12 slices][k-slice][plane]
// [lane][8]) of the LayerNorm-fed Dense layers (fused q | k | v, fc1) for the training-mode forward, regenerated from the master
// buffer after each optimizer step.  K = 384.  src[part] = Keras (384, N / parts) kernels side by side along N.
struct PanelPackSrc { long long src[3]; long long dst; int N, parts; };
static __global__ void __launch_bounds__(256)
panel_train_pack_kernel(const float* __restrict__ master, const PanelPackSrc* __restrict__ desc, const int* __restrict__ blk_first, const int ndesc,
                        _Float16* __restrict__ out)
{
    h3_flush_f16_denormals();
    int lo = 0, hi = ndesc - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (blk_first[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
    const PanelPackSrc d = desc[lo];
    const int e0 = ((int)blockIdx.x - blk_first[lo]) * 256 + threadIdx.x;        // one (chunk, k-step, slice, lane) = 8 consecutive k of one column
    const int lane = e0 & 63, kk = (e0 >> 6) % 12, st = (e0 / (64 * 12)) & 1, c = e0 / (64 * 12 * 2);
    if (c >= d.N / 32) return;
    const int n = 32 * c + (lane & 31), k0 = st * 192 + kk * 16 + (lane >> 5) * 8;
    const int Np = d.N / d.parts, part = n / Np, nn = n - part * Np;
    const float* w = master + d.src[part] + (size_t)k0 * Np + nn;
    h16x8 hv, lv;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = w[(size_t)j * Np];
        const _Float16 h = h3_hi(x);
        hv[j] = h; lv[j] = (_Float16)((x - (float)h) * H3_SCALE);
    }
    _Float16* o = out + d.dst + ((((size_t)(c * 2 + st) * 12 + kk) * 2) * 64 + lane) * 8;
    *reinterpret_cast<h16x8*>(o) = hv;
    *reinterpret_cast<h16x8*>(o + 512) = lv;
}

// ---- DropPath (vision_transformer.py:16-43): gate = floor(u + keep) per leading-dim sample -------
static __global__ void __launch_bounds__(256)
droppath_gate_kernel(const float* __restrict__ u, const int n, const float keep, float* __restrict__ gate)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) gate[i] = (u != nullptr) ? floorf(u[i] + keep) : 1.0f;
}

// random token masking (u_u_t.py:287-311): keep[b, n] = 1 where the token entering the temporal transformer is spatial_to_temporal_fc's
// output, 0 where it is replaced -- by the strided-input token (stride_mask == 0) or by the masked-token value 0 (u < rate, never the
// central frame).  `keep` serves the forward epilogue and the backward row mask (d s2t_out = dX * keep).
static __global__ void __launch_bounds__(256)
token_keep_kernel(const float* __restrict__ u, const float rate, const uint8_t* __restrict__ stride_mask, const int rows, const int N,
                  uint8_t* __restrict__ keep, uint8_t* __restrict__ replaced)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= rows) return;
    const bool masked = (u[i] < rate) && (i % N != N / 2);
    const bool real = (stride_mask == nullptr || stride_mask[i] != 0);
    keep[i] = (uint8_t)(real && !masked);
    replaced[i] = (uint8_t)(real && masked);      // rows that carry the masked-token value: the rows its gradient sums (LEARNABLE_MASKED_TOKEN)
}

// every layer's gates of both stacks in one launch: layer i of the spatial stack has ns gates (u and gate at i * ns), the
// temporal stack's follow (u at Ls * ns + i * nt, gates in their own array); keep[i] >= 1 or u == nullptr: gate = 1
struct GateKeeps { float s[16], t[16]; };
static __global__ void __launch_bounds__(256)
droppath_gates_kernel(const float* __restrict__ u, const int Ls, const int ns, const int Lt, const int nt, const GateKeeps keeps,
                      float* __restrict__ gate_s, float* __restrict__ gate_t)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int total_s = Ls * ns;
    if (i >= total_s + Lt * nt) return;
    const bool sp = i < total_s;
    const int j = sp ? i : i - total_s;
    const float keep = sp ? keeps.s[j / ns] : keeps.t[j / nt];
    const float g = (u != nullptr && keep < 1.f) ? floorf(u[i] + keep) : 1.0f;
    (sp ? gate_s : gate_t)[j] = g;
}

// ---- elementwise helpers ------------------------------------------------------------------------
// x0 = kp @ We + be + pe[joint]   (u_u_t.py:321-323), rows = frames * J
static __global__ void __launch_bounds__(256)
embed_fwd_kernel(const float* __restrict__ kp, const float* __restrict__ We, const float* __restrict__ be,
                 const float* __restrict__ pe, const int rows, const int J, const int DS, float* __restrict__ out)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * DS) return;
    const int r = idx / DS, c = idx - r * DS;
    const float kx = kp[2 * r], ky = kp[2 * r + 1];
    out[idx] = (fmaf(ky, We[DS + c], kx * We[c]) + be[c]) + pe[(r % J) * DS + c];
}
// T[r][0..DS) = kx * dX0[r], T[r][DS..2DS) = ky * dX0[r]  (column sums give dWe (2, DS))
static __global__ void __launch_bounds__(256)
embed_bwd_prep_kernel(const float* __restrict__ kp, const float* __restrict__ dx0, const int rows, const int DS, float* __restrict__ T)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * DS) return;
    const int r = idx / DS, c = idx - r * DS;
    const float g = dx0[idx];
    T[(size_t)r * 2 * DS + c] = kp[2 * r] * g;
    T[(size_t)r * 2 * DS + DS + c] = kp[2 * r + 1] * g;
}
// y = LayerNorm(x) written out (spatial_norm, u_u_t.py:329), Keras non-fused formula
static __global__ void __launch_bounds__(256)
ln_apply_kernel(const float* __restrict__ x, const float2* __restrict__ stats, const float* __restrict__ g,
                const float* __restrict__ b, const int rows, const int D, float* __restrict__ y)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * D) return;
    const int r = idx / D, c = idx - r * D;
    const float2 s = stats[r];
    const float inv = s.y * g[c];
    y[idx] = x[idx] * inv + (b[c] - s.x * inv);
}
// out[r][c] = in[r][c] / keep * gate[r / rps]      (DropPath backward; gate == nullptr: plain copy)
// or, with row_mask: out = in where mask[r] == want else 0   (token blend backward)
static __global__ void __launch_bounds__(256)
scale_rows_kernel(const float* __restrict__ in, const int rows, const int D, const float* __restrict__ gate,
                  const float keep, const int rps, const uint8_t* __restrict__ row_mask, const int want, float* __restrict__ out)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * D) return;
    const int r = idx / D;
    float v = in[idx];
    if (gate != nullptr) v = (v / keep) * gate[r / rps];
    if (row_mask != nullptr && (int)(row_mask[r] != 0) != want) v = 0.f;
    out[idx] = v;
}
// gradient of the strided block's identity path (trim + MaxPool1D(pool 1, stride s)) as a gather:
//   dmid[b, r] = dout[b, (r - lo) / s] if (r - lo) % s == 0 and in range, else 0
static __global__ void __launch_bounds__(256)
identity_bwd_kernel(const float* __restrict__ dout, const int B, const int L_in, const int L_out, const int stride,
                    const int lo, const int D, float* __restrict__ dmid)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * L_in * D) return;
    const int row = idx / D, c = idx - row * D;
    const int b = row / L_in, r = row - b * L_in;
    const int q = r - lo;
    float v = 0.f;
    if (q >= 0 && q % stride == 0 && q / stride < L_out) v = dout[((size_t)b * L_out + q / stride) * D + c];
    dmid[idx] = v;
}
// ---- BatchNormalization in training mode (OUTPUT_BN heads, u_u_t.py:275-285,400-404,414-416; kl.BatchNormalization(momentum=0.1,
// epsilon=1e-5) on rank-3 / rank-2 inputs = Keras' non-fused path: batch mean and BIASED variance over every axis but the last, the same
// biased variance in the moving-average update, moving = moving * momentum + batch * (1 - momentum)).  Column statistics are two
// launch_colsum passes (sum, then sum of squared deviations: two-pass like tf.nn.moments) around these elementwise kernels.
static __global__ void __launch_bounds__(256)
bn_sqdev_kernel(const float* __restrict__ x, const float* __restrict__ sum, const float inv_rows, const long long n, const int D, float* __restrict__ out)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float d = x[i] - sum[i % D] * inv_rows;
    out[i] = d * d;
}
// mean, 1 / sqrt(var + eps) of the batch; moving statistics updated in place (the only weights a training-mode forward writes)
static __global__ void __launch_bounds__(256)
bn_stats_kernel(const float* __restrict__ sum, const float* __restrict__ sumsq, const float inv_rows, const float eps, const float momentum, const int D,
                float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ moving_mean, float* __restrict__ moving_var)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= D) return;
    const float m = sum[c] * inv_rows, v = sumsq[c] * inv_rows;
    mean[c] = m; rstd[c] = 1.0f / sqrtf(v + eps);
    moving_mean[c] = moving_mean[c] * momentum + m * (1.0f - momentum);
    moving_var[c] = moving_var[c] * momentum + v * (1.0f - momentum);
}
// inference-mode BatchNormalization (u_u_t.py:400-404,414-416 with training=False): mean / 1 / sqrt(var + eps) from the moving statistics
static __global__ void __launch_bounds__(256)
bn_moving_stats_kernel(const float* __restrict__ moving_mean, const float* __restrict__ moving_var, const float eps, const int D,
                       float* __restrict__ mean, float* __restrict__ rstd)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= D) return;
    mean[c] = moving_mean[c]; rstd[c] = 1.0f / sqrtf(moving_var[c] + eps);
}
static __global__ void __launch_bounds__(256)
bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                const float* __restrict__ beta, const long long n, const int D, float* __restrict__ y)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % D);
    y[i] = (x[i] - mean[c]) * rstd[c] * gamma[c] + beta[c];
}
// out = dy * xhat  (its column sum is d gamma; the column sum of dy is d beta)
static __global__ void __launch_bounds__(256)
bn_bwd_prod_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                   const long long n, const int D, float* __restrict__ out)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % D);
    out[i] = dy[i] * ((x[i] - mean[c]) * rstd[c]);
}
// dx (+)= gamma * rstd * (dy - dbeta / R - xhat * dgamma / R)
static __global__ void __launch_bounds__(256)
bn_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                 const float* __restrict__ gamma, const float* __restrict__ dgamma, const float* __restrict__ dbeta, const float inv_rows,
                 const long long n, const int D, const int accumulate, float* __restrict__ dx)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % D);
    const float xh = (x[i] - mean[c]) * rstd[c];
    const float v = gamma[c] * rstd[c] * (dy[i] - dbeta[c] * inv_rows - xh * dgamma[c] * inv_rows);
    dx[i] = accumulate ? dx[i] + v : v;
}

// out[r][0..ldo) = in[r][0..C) followed by zeros   (head gradients: 51 -> 64 columns)
static __global__ void __launch_bounds__(256)
pad_cols_kernel(const float* __restrict__ in, const int rows, const int C, const int ldo, float* __restrict__ out)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * ldo) return;
    const int r = idx / ldo, c = idx - r * ldo;
    out[idx] = c < C ? in[(size_t)r * C + c] : 0.f;
}
// a += b
static __global__ void __launch_bounds__(256)
add_inplace_kernel(float* __restrict__ a, const float* __restrict__ b, const long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] += b[i];
}

// ---- loaders / epilogues for the generic GEMM ---------------------------------------------------
__device__ __forceinline__ float gelu_exact(float h) { return 0.5f * h * (1.0f + erff(h * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad(float h) {
    return 0.5f * (1.0f + erff(h * 0.70710678118654752440f)) + h * 0.39894228040143267794f * expf(-0.5f * h * h);
}
// A = gelu(Hpre)  (fc2 input, vision_transformer.py:62-65)
struct ALoadGelu {
    const float* __restrict__ A; int lda, M, K;
    struct Ctx { const float* p; };
    struct Raw { f32x4 x; };
    __device__ __forceinline__ Ctx prep(int row) const { Ctx c; c.p = A + (size_t)min(row, M - 1) * lda; return c; }
    __device__ __forceinline__ Raw issue(const Ctx& c, int k) const { Raw r; r.x = *reinterpret_cast<const f32x4*>(c.p + min(k, K - 4)); return r; }
    __device__ __forceinline__ f32x4 finish(const Ctx&, int k, const Raw& r) const {
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = gelu_exact(r.x[e]);
        return (k < K) ? y : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
};
// transposed strided conv as a gather GEMM: row (b, src) of dH, k = j * Nn + n reads dz[b, t][n]
// with t = (src + pad_left - j) / stride when divisible and in range.
struct ALoadConvT {
    const float* __restrict__ dz; int Nn, L_in, L_out, stride, pad_left, M, K;   // M = B*L_in, K = 3*Nn
    struct Ctx { int b, src; };
    struct Raw { f32x4 x; int ok; };
    __device__ __forceinline__ Ctx prep(int row) const { const int rc = min(row, M - 1); Ctx c; c.b = rc / L_in; c.src = rc - c.b * L_in; return c; }
    __device__ __forceinline__ Raw issue(const Ctx& c, int k) const {
        const int kc = min(k, K - 4);
        const int j = kc / Nn, n = kc - j * Nn;
        const int q = c.src + pad_left - j;
        const int t = q / stride;
        Raw r;
        r.ok = (q >= 0) & (q - t * stride == 0) & (t < L_out) & (k < K);
        const int tc = min(max(t, 0), L_out - 1);
        r.x = *reinterpret_cast<const f32x4*>(dz + ((size_t)c.b * L_out + tc) * Nn + n);
        return r;
    }
    __device__ __forceinline__ f32x4 finish(const Ctx&, int, const Raw& r) const { return r.ok ? r.x : (f32x4){0.f, 0.f, 0.f, 0.f}; }
};
// A rows scaled to zero where the row mask says so (d s2t_out = dX ⊙ m), used by TN/NT through a copy instead.

// out = xin + ((acc + bias) / keep) * gate[row / rps]   (residual branch with DropPath, out of place)
struct EpBiasResGate {
    float* __restrict__ out; const float* __restrict__ xin; const float* __restrict__ bias; int ld;
    const float* __restrict__ gate; float keep; int rps;      // gate == nullptr: no DropPath layer
    float* out2; const float* __restrict__ pe2; int period;   // optional second stream out + pe2[row % period]
    DropCfg drop{};                                           // Dropout on (acc + bias), in front of DropPath (vision_transformer.py:153-154,65-66 then :186-193)
    __device__ __forceinline__ float2 colv(int col) const { return make_float2(bias[col], 0.f); }
    __device__ __forceinline__ float2 pre(int rowc, int col) const {
        float2 p; p.x = xin[(size_t)rowc * ld + col];
        p.y = (out2 != nullptr) ? pe2[(size_t)(rowc % period) * ld + col] : 0.f;
        return p;
    }
    __device__ __forceinline__ void store(int row, int col, float v, float2 cv, float2 p) const {
        float y = v + cv.x;
        if (drop.on()) y *= drop_factor(drop, (unsigned long long)row * (unsigned)ld + (unsigned)col);
        if (gate != nullptr) y = (y / keep) * gate[row / rps];
        const float o = p.x + y;
        out[(size_t)row * ld + col] = o;
        if (out2 != nullptr) out2[(size_t)row * ld + col] = o + p.y;
    }
};
struct EpAdd {              // out += acc
    float* out; int ldo;
    __device__ __forceinline__ float2 colv(int) const { return make_float2(0.f, 0.f); }
    __device__ __forceinline__ float2 pre(int rowc, int col) const { return make_float2(out[(size_t)rowc * ldo + col], 0.f); }
    __device__ __forceinline__ void store(int row, int col, float v, float2, float2 p) const { out[(size_t)row * ldo + col] = p.x + v; }
};
struct EpReluMask {         // out = acc * (H > 0) * scale   (scale: 1 / (1 - rate) of an inner Dropout behind the ReLU -- H is stored AFTER it, so H > 0 = kept and active)
    float* __restrict__ out; const float* __restrict__ H; int ldo; float scale = 1.f;
    __device__ __forceinline__ float2 colv(int) const { return make_float2(0.f, 0.f); }
    __device__ __forceinline__ float2 pre(int rowc, int col) const { return make_float2(H[(size_t)rowc * ldo + col], 0.f); }
    __device__ __forceinline__ void store(int row, int col, float v, float2, float2 p) const { out[(size_t)row * ldo + col] = p.x > 0.f ? v * scale : 0.f; }
};
struct EpGeluGrad {         // out = acc * gelu'(Hpre)
    float* __restrict__ out; const float* __restrict__ Hpre; int ldo;
    __device__ __forceinline__ float2 colv(int) const { return make_float2(0.f, 0.f); }
    __device__ __forceinline__ float2 pre(int rowc, int col) const { return make_float2(Hpre[(size_t)rowc * ldo + col], 0.f); }
    __device__ __forceinline__ void store(int row, int col, float v, float2, float2 p) const { out[(size_t)row * ldo + col] = v * gelu_grad(p.x); }
};
// TN A-side: gelu(Hpre) rows (dW2 of the spatial MLP)
struct TnLoadGelu {
    const float* __restrict__ A; int lda, R, P;
    struct Raw { f32x4 x; bool ok; };
    typedef TnNoCol Col;
    __device__ __forceinline__ Col col(int) const { return Col{}; }
    __device__ __forceinline__ Raw fetch(int r, int p) const {
        const bool ok = r < R && p < P;
        return Raw{*reinterpret_cast<const f32x4*>(A + (ok ? (size_t)r * lda + p : 0)), ok};
    }
    __device__ __forceinline__ f32x4 finish(const Raw& w, const Col&) const {
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = w.ok ? gelu_exact(w.x[e]) : 0.f;
        return y;
    }
    __device__ __forceinline__ f32x4 load(int r, int p) const { return finish(fetch(r, p), col(p)); }
};

// x[i] *= s over a range of the flat gradient buffer (s = 1 / loss scale, a power of two: exact).  A non-finite value -- a
// loss-scaled activation gradient beyond the f16 range makes the hi plane of the f16x3 split infinite -- raises *flag, which
// the guarded AdamW update (uu3d_adamw_update_guarded) reads on the device: the step is then skipped, no host sync.
static __global__ void __launch_bounds__(256)
scale_flat_kernel(float* __restrict__ x, const long long n, const float s, unsigned* __restrict__ flag)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const float v = x[i] * s;
        x[i] = v;
        if (flag != nullptr && !(fabsf(v) <= 3.4e38f)) atomicOr(flag, 1u);
    }
}

}  // namespace uu3d
