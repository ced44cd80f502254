#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <int NACC, int SHAPE>
__global__ void __launch_bounds__(256) peak(float* out, int iters, unsigned long long* clk) {
    float a = threadIdx.x * 0.001f + 0.5f, b = 1.0f - threadIdx.x * 0.002f;
    f32x16 acc[NACC]; f32x4 acc4[NACC];
    for (int i = 0; i < NACC; ++i) { for (int r = 0; r < 16; ++r) acc[i][r] = 0.f; for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (SHAPE == 32) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                else acc4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[i], 0, 0, 0);
            }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0; for (int i = 0; i < NACC; ++i) { for (int r = 0; r < 16; ++r) s += acc[i][r]; for (int r = 0; r < 4; ++r) s += acc4[i][r]; }
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 100 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
template <int NACC, int SHAPE> void run(int wgs_per_cu, int iters) {
    float* out; unsigned long long* clk; CK(hipMalloc(&out, 256 * 256 * 16 * 4)); CK(hipMalloc(&clk, 16));
    int grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((peak<NACC, SHAPE>), dim3(grid), dim3(256), 0, 0, out, iters, clk);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((peak<NACC, SHAPE>), dim3(grid), dim3(256), 0, 0, out, iters, clk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    double flop = (double)grid * 4 * iters * 8 * NACC * (SHAPE == 32 ? 4096.0 : 2048.0);
    printf("shape %d nacc %d wgs/cu %d: %.3f ms  %.1f TF  clock %.0f MHz\n", SHAPE, NACC, wgs_per_cu, ms, flop / ms / 1e9, (double)h[0] / h[1] * 100.0);
    CK(hipFree(out)); CK(hipFree(clk));
}
int main() {
    run<1, 32>(1, 2000); run<1, 32>(4, 500); run<4, 32>(1, 500); run<4, 32>(2, 250);
    run<1, 16>(1, 4000); run<4, 16>(1, 1000); run<4, 16>(4, 250);
    run<1, 32>(4, 20);   // short kernel (~ like ours)
    return 0;
}
