// Does the register alignment of the A / B operands of v_mfma_f32_32x32x16_f16 change its issue rate?  One wave per SIMD,
// a chain of MFMAs with explicit registers; cycles per MFMA by s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define BODY(A, B, A2, B2) \
    "v_mfma_f32_32x32x16_f16 a[0:15], " A ", " B ", a[0:15]\n" \
    "v_mfma_f32_32x32x16_f16 a[16:31], " A ", " B2 ", a[16:31]\n" \
    "v_mfma_f32_32x32x16_f16 a[16:31], " A2 ", " B ", a[16:31]\n"

template <int V>
__global__ void __launch_bounds__(256) k(unsigned long long* out, int iters) {
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int i = 0; i < iters; ++i) {
        if (V == 0) asm volatile(BODY("v[8:11]", "v[16:19]", "v[12:15]", "v[20:23]") BODY("v[24:27]", "v[16:19]", "v[28:31]", "v[20:23]") BODY("v[32:35]", "v[16:19]", "v[36:39]", "v[20:23]") BODY("v[40:43]", "v[16:19]", "v[44:47]", "v[20:23]")
                                 ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31");
        if (V == 1) asm volatile(BODY("v[8:11]", "v[18:21]", "v[12:15]", "v[22:25]") BODY("v[28:31]", "v[18:21]", "v[32:35]", "v[22:25]") BODY("v[36:39]", "v[18:21]", "v[40:43]", "v[22:25]") BODY("v[44:47]", "v[18:21]", "v[48:51]", "v[22:25]")
                                 ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31");
        if (V == 2) asm volatile(BODY("a[32:35]", "v[16:19]", "a[36:39]", "v[20:23]") BODY("a[40:43]", "v[16:19]", "a[44:47]", "v[20:23]") BODY("a[48:51]", "v[16:19]", "a[52:55]", "v[20:23]") BODY("a[56:59]", "v[16:19]", "a[60:63]", "v[20:23]")
                                 ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31");
        if (V == 3) asm volatile(BODY("a[32:35]", "v[18:21]", "a[36:39]", "v[22:25]") BODY("a[40:43]", "v[18:21]", "a[44:47]", "v[22:25]") BODY("a[48:51]", "v[18:21]", "a[52:55]", "v[22:25]") BODY("a[56:59]", "v[18:21]", "a[60:63]", "v[22:25]")
                                 ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31");
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int V> int run(const char* what, unsigned long long* d) {
    const int iters = 200, grid = 256;
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, d, iters); CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, d, iters); CK(hipDeviceSynchronize());
    unsigned long long h[256]; CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    double s = 0; for (int i = 0; i < grid; ++i) s += (double)h[i];
    printf("%-70s %6.2f cycles per MFMA\n", what, s / grid / iters / 12);
    return 0;
}
int main() {
    unsigned long long* d; CK(hipMalloc(&d, 256 * 8));
    run<0>("A v[8k:..] B v[16:19]/v[20:23] (same bank phase), acc AGPR", d);
    run<1>("A v[..] B v[18:21]/v[22:25] (B shifted by 2 registers)", d);
    run<2>("A in AGPRs, B v[16:19]/v[20:23]", d);
    run<3>("A in AGPRs, B v[18:21]/v[22:25]", d);
    return 0;
}
