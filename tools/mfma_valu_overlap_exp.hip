// Round 4: do the matrix pipe and the vector ALU of one SIMD run at the same time?  The attention kernel's knock-outs
// (profiles/r04_attention_knockouts.txt) add up as a plain SUM of MFMA time, VALU time and LDS time; this asks the hardware directly.
// 256 workgroups x 8 waves (two waves per SIMD).  Wave role A (waves 0-3) and role B (waves 4-7) run one of
//     M  a chain of v_mfma_f32_32x32x16_f16 on three accumulators (operands in registers)
//     V  v_fma_f32 on 16 independent registers
//     X  v_exp_f32 on 16 independent registers (quarter rate)
//     L  ds_read_b128 + full wait (LDS round trips)
//     -  nothing
// and the launch is timed for A alone, B alone and both; "same wave" interleaves M with V / X in ONE instruction stream.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/mfma_valu_overlap_exp tools/mfma_valu_overlap_exp.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { R_NONE = 0, R_MFMA = 1, R_FMA = 2, R_EXP = 3, R_LDS = 4, R_MFMA_FMA = 5, R_MFMA_EXP = 6 };

template <int ROLE>
__device__ __forceinline__ float work(const h16x8* __restrict__ src, int iters, unsigned char* sm, int lane)
{
    float res = 0.f;
    if constexpr (ROLE == R_MFMA || ROLE == R_MFMA_FMA || ROLE == R_MFMA_EXP) {
        h16x8 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = src[i * 64 + lane]; b[i] = src[(4 + i) * 64 + lane]; }
        f32x16 c0, c1, c2; float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; c2[r] = 0.f; v[r] = 0.001f * (float)(lane + r); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {                     // 12 MFMAs per iteration
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k], b[k], c0, 0, 0, 0);
                if constexpr (ROLE == R_MFMA_FMA) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[r]) : "v"(0.999f));
                }
                if constexpr (ROLE == R_MFMA_EXP) {
#pragma unroll
                    for (int r = 0; r < 2; ++r) asm volatile("v_exp_f32 %0, %0" : "+v"(v[r]));
                }
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k], b[(k + 1) & 3], c1, 0, 0, 0);
                if constexpr (ROLE == R_MFMA_FMA) {
#pragma unroll
                    for (int r = 8; r < 16; ++r) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[r]) : "v"(0.999f));
                }
                if constexpr (ROLE == R_MFMA_EXP) {
#pragma unroll
                    for (int r = 2; r < 4; ++r) asm volatile("v_exp_f32 %0, %0" : "+v"(v[r]));
                }
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(k + 1) & 3], b[k], c2, 0, 0, 0);
                if constexpr (ROLE == R_MFMA_FMA) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[r]) : "v"(0.999f));
                }
                if constexpr (ROLE == R_MFMA_EXP) {
#pragma unroll
                    for (int r = 4; r < 6; ++r) asm volatile("v_exp_f32 %0, %0" : "+v"(v[r]));
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) res += c0[r] + c1[r] + c2[r] + v[r];
    } else if constexpr (ROLE == R_FMA || ROLE == R_EXP) {
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = 0.001f * (float)(lane + r);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < (ROLE == R_FMA ? 6 : 2); ++rep)          // 96 FMAs (384 issue cycles) or 32 exponentials (512) per iteration
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if constexpr (ROLE == R_FMA) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[r]) : "v"(0.999f));
                    else asm volatile("v_exp_f32 %0, %0" : "+v"(v[r]));
                }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) res += v[r];
    } else if constexpr (ROLE == R_LDS) {
        const unsigned sb = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(sm + lane * 16);
        f32x4 x[4];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 3; ++rep) {                // 3 round trips of 4 reads per iteration
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]) : "v"(sb) : "memory");
                res += x[0][0] + x[1][1] + x[2][2] + x[3][3];
            }
        }
    }
    return res;
}

template <int RA, int RB>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) kern(const h16x8* __restrict__ src, float* __restrict__ out, int itA, int itB)
{
    __shared__ __attribute__((aligned(16))) unsigned char sm[8192];
    for (int i = threadIdx.x; i < 8192 / 16; i += blockDim.x) reinterpret_cast<h16x8*>(sm)[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float r;
    if (w < 4) r = work<RA>(src, itA, sm, lane); else r = work<RB>(src, itB, sm, lane);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

static h16x8* dsrc; static float* dout;
template <int RA, int RB> static float run(int itA, int itB) {
    auto k = kern<RA, RB>;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, dsrc, dout, itA, itB);
    std::vector<float> t;
    for (int r = 0; r < 7; ++r) {
        CK(hipEventRecord(e0)); for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, dsrc, dout, itA, itB);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms * 100.f);
    }
    std::sort(t.begin(), t.end());
    return t[3];
}

int main() {
    std::vector<_Float16> h(64 * 1024); for (size_t i = 0; i < h.size(); ++i) h[i] = (_Float16)(0.01f * (float)((i * 37) % 199) - 1.f);
    CK(hipMalloc(&dsrc, h.size() * 2)); CK(hipMemcpy(dsrc, h.data(), h.size() * 2, hipMemcpyHostToDevice)); CK(hipMalloc(&dout, 256 * 512 * 4));
    const int N = 200;     // iterations: 2400 MFMAs (~79 k cycles), 19200 FMAs (~77 k issue cycles), 6400 exponentials (~102 k), 600 LDS round trips
    printf("two waves per SIMD; A = waves 0-3, B = waves 4-7; us per launch (median of 7 x 10)\n");
    const float m = run<R_MFMA, R_NONE>(N, 0), f = run<R_NONE, R_FMA>(0, N), x = run<R_NONE, R_EXP>(0, N), l = run<R_NONE, R_LDS>(0, N);
    printf("  alone:   MFMA %.1f   FMA %.1f   EXP %.1f   LDS round trips %.1f\n", m, f, x, l);
    printf("  A MFMA + B FMA : %.1f   (sum %.1f, max %.1f)\n", run<R_MFMA, R_FMA>(N, N), m + f, std::max(m, f));
    printf("  A MFMA + B EXP : %.1f   (sum %.1f, max %.1f)\n", run<R_MFMA, R_EXP>(N, N), m + x, std::max(m, x));
    printf("  A MFMA + B LDS : %.1f   (sum %.1f, max %.1f)\n", run<R_MFMA, R_LDS>(N, N), m + l, std::max(m, l));
    printf("  A MFMA + B MFMA: %.1f   (sum %.1f, max %.1f)\n", run<R_MFMA, R_MFMA>(N, N), m + m, m);
    printf("  A FMA  + B FMA : %.1f   (sum %.1f, max %.1f)\n", run<R_FMA, R_FMA>(N, N), f + f, f);
    printf("  A FMA  + B EXP : %.1f   (sum %.1f, max %.1f)\n", run<R_FMA, R_EXP>(N, N), f + x, std::max(f, x));
    printf("  A FMA  + B LDS : %.1f   (sum %.1f, max %.1f)\n", run<R_FMA, R_LDS>(N, N), f + l, std::max(f, l));
    printf("  same wave, 8 FMAs behind every MFMA (A only)  : %.1f   (MFMA alone %.1f; the FMAs alone would take %.1f)\n", run<R_MFMA_FMA, R_NONE>(N, 0), m, f);
    printf("  same wave, 2 EXPs behind every MFMA (A only)  : %.1f   (MFMA alone %.1f; the EXPs alone would take %.1f)\n", run<R_MFMA_EXP, R_NONE>(N, 0), m, x * 0.75f);
    printf("  same wave MFMA + FMA in A, the same in B      : %.1f\n", run<R_MFMA_FMA, R_MFMA_FMA>(N, N));
    return 0;
}
