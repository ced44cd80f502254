// uu3d_misc.h -- small row-wise kernels: LayerNorm statistics and per-joint MPJPE.
#pragma once
#include <vector>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "uu3d_gemm.h"

namespace uu3d {

// (mean, 1/sqrt(var + eps)) per row, biased variance, two passes over registers:
// the statistics half of Keras' non-fused LayerNormalization (tf.nn.moments), consumed by
// ALoadLayerNorm.  One wave per row, 4 rows per workgroup.  D % 4 == 0, D <= 64 * 4 * MAXV.
template <int MAXV>
__global__ void __launch_bounds__(256)
row_stats_kernel(const float* __restrict__ x, const int ld, const int D, const int M, const float eps,
                 float2* __restrict__ stats)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* p = x + (size_t)row * ld;
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * 64 + lane) * 4;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < D) v[i] = *reinterpret_cast<const float4*>(p + c);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
            const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    if (lane == 0) stats[row] = make_float2(mean, 1.0f / sqrtf(q / (float)D + eps));
}

// The same statistics for NARROW rows (D == 4 * LPR <= 64: the spatial stack's D = 32): LPR lanes per row, 64 / LPR rows
// per wave at once instead of one row on 8 of 64 lanes.  Bit-identical to row_stats_kernel: the butterfly over the LPR
// lanes is the tail of the 64-lane one, whose other lanes hold zeros.
template <int LPR>
__global__ void __launch_bounds__(256)
row_stats_narrow_kernel(const float* __restrict__ x, const int ld, const int M, const float eps, float2* __restrict__ stats)
{
    constexpr int RG = 256 / LPR;
    const int l = threadIdx.x % LPR;
    const int row = blockIdx.x * RG + threadIdx.x / LPR;
    const bool ok = row < M;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) v = *reinterpret_cast<const float4*>(x + (size_t)row * ld + 4 * l);
    float s = (v.x + v.y) + (v.z + v.w);
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)(4 * LPR);
    const float a = v.x - mean, b = v.y - mean, cc = v.z - mean, d = v.w - mean;
    float q = (a * a + b * b) + (cc * cc + d * d);
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o);
    if (ok && l == 0) stats[row] = make_float2(mean, 1.0f / sqrtf(q / (float)(4 * LPR) + eps));
}

// Launch the row statistics: the narrow form when the rows are exactly 32 or 64 floats.
inline void launch_row_stats(const float* x, int ld, int D, int M, float eps, float2* stats, hipStream_t stream) {
    if (D == 32) hipLaunchKernelGGL(row_stats_narrow_kernel<8>, dim3((M + 31) / 32), dim3(256), 0, stream, x, ld, M, eps, stats);
    else if (D == 64) hipLaunchKernelGGL(row_stats_narrow_kernel<16>, dim3((M + 15) / 16), dim3(256), 0, stream, x, ld, M, eps, stats);
    else hipLaunchKernelGGL(row_stats_kernel<4>, dim3((M + 3) / 4), dim3(256), 0, stream, x, ld, D, M, eps, stats);
}

// metrics.mpjpe(normalize=False) (common/dataset/metrics.py:13-37), float64 arithmetic on
// f32 inputs exactly as the reference's numpy call after its astype(np.float64).
static __global__ void __launch_bounds__(256)
mpjpe_kernel(const float* __restrict__ pred, const float* __restrict__ gt, const int B, const int J,
             const int root, double* __restrict__ out)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * J) return;
    const int b = idx / J;
    const float* pp = pred + (size_t)idx * 3;
    const float* pr = pred + ((size_t)b * J + root) * 3;
    const float* gp = gt + (size_t)idx * 4;
    const float* gr = gt + ((size_t)b * J + root) * 4;
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double d = ((double)pp[c] - (double)pr[c]) - ((double)gp[c] - (double)gr[c]);
        acc += d * d;
    }
    out[idx] = (gp[3] > 0.f) ? sqrt(acc) : -1.0;
}

// Ordered compaction of the frames that carry real 2D input (stride_mask == 1): the "gather" of
// the learned-upsampling path.  spatial_transformation's result at masked frames is overwritten
// by the strided-input token (u_u_t.py:350, ToDo at :320), so the spatial stack only has to run
// on this list -- an exact, output-preserving saving.  One workgroup, ascending order, deterministic.
//   list[0 .. count) = indices of valid frames, count stored in list[total]
// Each thread owns kCompactPerThread consecutive frames (one 16-byte load), so up to 1024 * 16 frames take a
// out[row] = x[row] + pe[row % period]  (rows of D floats, D % 4 == 0): the strided positional encoding of a model without temporal blocks
static __global__ void __launch_bounds__(256)
add_period_kernel(const float* __restrict__ x, const float* __restrict__ pe, const int rows, const int D, const int period, float* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, per_row = (size_t)D / 4;
    if (i >= (size_t)rows * per_row) return;
    const size_t row = i / per_row, c = (i - row * per_row) * 4;
    *reinterpret_cast<f32x4*>(out + row * D + c) = *reinterpret_cast<const f32x4*>(x + row * D + c) + *reinterpret_cast<const f32x4*>(pe + (row % period) * D + c);
}

// Attention maps for return_attention=True (vision_transformer.py:117-130: softmax(q k^T / sqrt(d_h) + mask * -1e9), what MHA returns
// next to its output and UpliftUpsampleTransformer.call collects per temporal block, u_u_t.py:365,418-419).  NOT on the hot path: the
// forward's attention kernels never materialise the (L, L) tile; this kernel recomputes it from the block's q | k.  One workgroup per
// (sequence, head), K staged in LDS as f32, one thread per query row at a time; three passes over the row in the output buffer (raw
// logits and their maximum, exponentials and their sum, normalisation).  q / k either f32 [rows][ld] (planes_lo == nullptr) or f16 hi / lo
// planes with q already multiplied by log2(e) / sqrt(d_h) (the QKV epilogue of the f16x3 path): `qscale` and `base2` say which.
static __global__ void __launch_bounds__(256)
attn_probs_kernel(const void* __restrict__ qkv_hi, const _Float16* __restrict__ planes_lo, const int ld, const int D, const int L, const int H,
                  const int DH, const uint8_t* __restrict__ key_mask, const float qscale, const int base2, float* __restrict__ out)
{
    extern __shared__ float kls[];                                   // [L][DH + 1]
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const size_t tok0 = (size_t)b * L;
    const int KL = DH + 1;
    auto val = [&](size_t row, int col) -> float {
        const size_t o = row * ld + col;
        if (planes_lo == nullptr) return reinterpret_cast<const float*>(qkv_hi)[o];
        return (float)reinterpret_cast<const _Float16*>(qkv_hi)[o] + (float)planes_lo[o] * (1.0f / 2048.0f);
    };
    for (int i = threadIdx.x; i < L * DH; i += 256) { const int j = i / DH, c = i - j * DH; kls[j * KL + c] = val(tok0 + j, D + h * DH + c); }
    __syncthreads();
    const float neg = base2 ? -1e9f * 1.44269504088896341f : -1e9f;  // the mask term in the units of the logits
    for (int i = threadIdx.x; i < L; i += 256) {
        float* row = out + (((size_t)b * H + h) * L + i) * L;
        float q[64];
        for (int c = 0; c < DH; ++c) q[c] = val(tok0 + i, h * DH + c) * qscale;
        float mx = -INFINITY;
        for (int j = 0; j < L; ++j) {
            float s = 0.f;
            for (int c = 0; c < DH; ++c) s += q[c] * kls[j * KL + c];
            if (key_mask != nullptr && key_mask[tok0 + j] == 0) s += neg;
            row[j] = s; mx = fmaxf(mx, s);
        }
        float sum = 0.f;
        for (int j = 0; j < L; ++j) { const float e = base2 ? exp2f(row[j] - mx) : expf(row[j] - mx); row[j] = e; sum += e; }
        const float inv = 1.0f / sum;
        for (int j = 0; j < L; ++j) row[j] *= inv;
    }
}

// single pass: in-thread count, wave-level shuffle scan, one barrier for the 16 wave totals.
static constexpr int kCompactPerThread = 16;
static __global__ void __launch_bounds__(1024)
compact_frames_kernel(const uint8_t* __restrict__ mask, const int total, int* __restrict__ list)
{
    __shared__ int wave_sum[16];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int start = 0; start < total; start += 1024 * kCompactPerThread) {
        const int first = start + tid * kCompactPerThread;
        uint8_t v[kCompactPerThread];
        if (first + kCompactPerThread <= total && (((uintptr_t)(mask + first)) & 15) == 0) {
            const uint4 q = *reinterpret_cast<const uint4*>(mask + first);
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int e = 0; e < kCompactPerThread; ++e) v[e] = (uint8_t)((w[e >> 2] >> (8 * (e & 3))) & 0xffu);
        } else {
#pragma unroll
            for (int e = 0; e < kCompactPerThread; ++e) v[e] = (first + e < total) ? mask[first + e] : (uint8_t)0;
        }
        int cnt = 0;
#pragma unroll
        for (int e = 0; e < kCompactPerThread; ++e) cnt += (v[e] != 0);
        int incl = cnt;                                   // inclusive scan over the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        int off = base + incl - cnt;
        for (int w = 0; w < wave; ++w) off += wave_sum[w];
#pragma unroll
        for (int e = 0; e < kCompactPerThread; ++e) if (v[e] != 0) list[off++] = first + e;
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += wave_sum[w]; base += t; }
        __syncthreads();
    }
    if (tid == 0) list[total] = base;
}


// Window gather over a resident pose table: the device form of H36mSequenceGenerator's per-sample slicing
// (common/dataset/uplifiting_dataset.py:322-394).  poses (F, J, C) holds all videos back to back; window w is centred
// on frame `center` of video `video` and samples every `stride`-th frame:
//     f_n = center - ((N - 1) * stride) / 2 + n * stride,  n = 0 .. N-1
// Frames outside [0, video_len) are padding: pad_mask = 0 and the value is the nearest sampled in-range frame
// (pad_edge, numpy mode "edge" = padding_type "copy") or 0 ("zeros").  stride_mask[n] = ((n - N/2) * stride + shift)
// % mask_stride == 0 with the caller's shift (the centre frame index for globally aligned masks, :381-384; a random
// multiple of the stride in training, :386-392).  zero_masked also applies x * stride_mask (eval.py:67, train.py:474).
// flip: joints permuted by flip_order, channel 0 negated (:403-407).  One thread per (window, frame, joint).
struct WindowDesc { int32_t video, center, stride, mask_stride, mask_shift, flip; };
static __global__ void __launch_bounds__(256)
gather_windows_kernel(const float* __restrict__ poses, const int64_t* __restrict__ video_start, const int32_t* __restrict__ video_len,
                      const WindowDesc* __restrict__ win, const int32_t* __restrict__ flip_order,
                      const int B, const int N, const int J, const int C, const int pad_edge, const int zero_masked,
                      float* __restrict__ out, uint8_t* __restrict__ stride_mask, uint8_t* __restrict__ pad_mask)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * N * J) return;
    const int j = (int)(idx % J);
    const int n = (int)((idx / J) % N);
    const int w = (int)(idx / ((long)J * N));
    const WindowDesc d = win[w];
    const int len = video_len[d.video];
    const int f = d.center - ((N - 1) * d.stride) / 2 + n * d.stride;
    int src = f;
    if (f < 0) src = f + ((-f + d.stride - 1) / d.stride) * d.stride;                   // first sampled frame >= 0
    else if (f >= len) src = f - ((f - len + d.stride) / d.stride) * d.stride;          // last sampled frame < len
    const bool inside = (f >= 0) && (f < len);
    const bool have = inside || (pad_edge != 0 && src >= 0 && src < len);
    const int rel = (n - N / 2) * d.stride + d.mask_shift;
    int mod = rel % d.mask_stride; if (mod < 0) mod += d.mask_stride;                   // python's % on negative numbers
    const bool sm = (mod == 0);
    if (j == 0) {
        stride_mask[(long)w * N + n] = sm ? 1 : 0;
        if (pad_mask != nullptr) pad_mask[(long)w * N + n] = inside ? 1 : 0;
    }
    const int js = (d.flip && flip_order != nullptr) ? flip_order[j] : j;
    const float* p = poses + ((video_start[d.video] + (have ? src : 0)) * J + js) * C;
    float* o = out + idx * C;
    const bool keep = have && (!zero_masked || sm);
    for (int c = 0; c < C; ++c) {
        float v = keep ? p[c] : 0.f;
        if (c == 0 && d.flip) v = -v;
        o[c] = v;
    }
}

// World -> camera -> 2D of every joint of a batch of windows, one camera per window: the device form of
// tf_world_to_cam_and_2d (common/dataset/uplifiting_dataset.py:669-761; AMASS training draws a camera per sample).
// cam (B, 19) = quaternion (w, x, y, z), translation (3), intrinsics (12: res 2, focal 2, centre 2, radial 3, tangential 2 + 1 spare
// as stored by the reference: indices 7..18).  x_cam = qrot(qinverse(q), x_world - t); XX = clamp(x_cam.xy / x_cam.z, -1, 1);
// 2D = f * (XX * (1 + k . (r2, r2^2, r2^3) + p . XX) + p * r2) + c.
static __global__ void __launch_bounds__(256)
world_to_cam_2d_kernel(const float* __restrict__ world, const float* __restrict__ cams, const long per_window, const long total,
                       float* __restrict__ cam3d, float* __restrict__ kp2d)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const float* cm = cams + (idx / per_window) * 18;       // quaternion wxyz | translation | 11 intrinsics: the reference's vectors are 18 wide
    const float qw = cm[0], qx = -cm[1], qy = -cm[2], qz = -cm[3];                 // inverse of a unit quaternion
    const float vx = world[idx * 3] - cm[4], vy = world[idx * 3 + 1] - cm[5], vz = world[idx * 3 + 2] - cm[6];
    // uv = qvec x v ; uuv = qvec x uv ; out = v + 2 (w uv + uuv)        (tf_qrot, :697-705)
    const float ux = qy * vz - qz * vy, uy = qz * vx - qx * vz, uz = qx * vy - qy * vx;
    const float wx = qy * uz - qz * uy, wy = qz * ux - qx * uz, wz = qx * uy - qy * ux;
    const float X = vx + 2.f * (qw * ux + wx), Y = vy + 2.f * (qw * uy + wy), Z = vz + 2.f * (qw * uz + wz);
    if (cam3d != nullptr) { cam3d[idx * 3] = X; cam3d[idx * 3 + 1] = Y; cam3d[idx * 3 + 2] = Z; }
    if (kp2d == nullptr) return;
    const float* in = cm + 7;
    const float a = fminf(fmaxf(X / Z, -1.f), 1.f), b = fminf(fmaxf(Y / Z, -1.f), 1.f);
    const float r2 = a * a + b * b;
    const float radial = 1.f + (in[6] * r2 + in[7] * (r2 * r2) + in[8] * (r2 * r2 * r2));
    const float tan = in[9] * a + in[10] * b;
    kp2d[idx * 2] = in[2] * (a * (radial + tan) + in[9] * r2) + in[4];
    kp2d[idx * 2 + 1] = in[3] * (b * (radial + tan) + in[10] * r2) + in[5];
}

// ---- which HIP streams share a hardware queue -------------------------------------------------------------------------------
// HIP deals the streams of a process to GPU_MAX_HW_QUEUES (4) hardware queues, not in creation order, and two streams on one queue
// run in order -- a "side stream" on the caller's queue overlaps nothing.  The probe: a spin kernel on a, a tiny kernel on b, and
// HIP events tell whether b's kernel ran at once or behind the spin.
static __global__ void spin_kernel(const long long ticks)      // wall_clock64: 100 MHz
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
static __global__ void touch_kernel() {}

inline bool streams_share_queue(hipStream_t a, hipStream_t b)
{
    hipEvent_t e0, ea, eb;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&ea) != hipSuccess || hipEventCreate(&eb) != hipSuccess) return false;
    (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b);
    (void)hipEventRecord(e0, a);
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, 150000LL);          // 1.5 ms
    (void)hipEventRecord(ea, a);
    hipLaunchKernelGGL(touch_kernel, dim3(1), dim3(64), 0, b);
    (void)hipEventRecord(eb, b);
    (void)hipEventSynchronize(ea); (void)hipEventSynchronize(eb);
    float ta = 0.f, tb = 0.f;
    const bool ok = hipEventElapsedTime(&ta, e0, ea) == hipSuccess && hipEventElapsedTime(&tb, e0, eb) == hipSuccess;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(ea); (void)hipEventDestroy(eb);
    return ok && tb > 0.5f * ta;
}

// n non-blocking streams on hardware queues different from `main`'s and from each other's, as far as the device has them (up to
// `tries` streams are created and probed; the ones not taken are destroyed; what cannot be found distinct is filled with any stream)
inline int create_streams_on_other_queues(hipStream_t main, int n, hipStream_t* out, int tries = 12)
{
    std::vector<hipStream_t> spare;
    int have = 0;
    for (int i = 0; i < tries && have < n; ++i) {
        hipStream_t s;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return -1;
        bool clash = streams_share_queue(main, s);
        for (int k = 0; k < have && !clash; ++k) clash = streams_share_queue(out[k], s);
        if (clash) spare.push_back(s); else out[have++] = s;
    }
    const int distinct = have;
    while (have < n && !spare.empty()) { out[have++] = spare.back(); spare.pop_back(); }
    while (have < n) { if (hipStreamCreateWithFlags(&out[have], hipStreamNonBlocking) != hipSuccess) return -1; ++have; }
    for (hipStream_t s : spare) (void)hipStreamDestroy(s);
    (void)hipGetLastError();
    return distinct;
}

// Range guard of the f16x3 forward (include/uu3d.h, "RANGE CONTRACT"): any non-finite value in the outputs sets the model's sticky word.
// An activation beyond the f16 range becomes Inf in its hi plane and NaN one product later; LayerNorm and attention spread it over its whole
// sequence, so both outputs are looked at: `full` catches the residual stream overflowing in front of the full-sequence head alone.
static __global__ void __launch_bounds__(256)
range_check_kernel(const float* __restrict__ a, const long na, const float* __restrict__ b, const long nb, int* __restrict__ flag)
{
    const long stride = (long)gridDim.x * blockDim.x * 4;
    bool bad = false;
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < na + nb; i += stride) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long j = i + e;
            const float v = j < na ? a[j] : (j < na + nb ? b[j - na] : 0.f);
            bad |= !(fabsf(v) <= 3.0e38f);
        }
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
// uu3d_range_status: read AND clear the sticky word in one atomic (a flag raised by another stream between a separate copy and a memset would be lost)
static __global__ void range_take_kernel(int* __restrict__ flag, int* __restrict__ taken) { *taken = atomicExch(flag, 0); }

}  // namespace uu3d
