// Round 6: operand / result layout of v_mfma_f32_16x16x32_f16 on gfx950, checked element by element, and its issue rate beside the 32x32x16 form.
//   hipcc -O3 --offload-arch=gfx950 -o tools/mfma16_layout tools/mfma16_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// assumed: A (16 x 32): lane l holds row l % 16, k = 8 (l / 16) + j;  B (32 x 16): lane l holds column l % 16, k = 8 (l / 16) + j;
//          D (16 x 16): lane l holds column l % 16, rows 4 (l / 16) + r
__global__ void layout_kernel(const _Float16* A, const _Float16* B, float* D) {
    const int l = threadIdx.x;
    h16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = A[(l % 16) * 32 + 8 * (l / 16) + j]; b[j] = B[(8 * (l / 16) + j) * 16 + (l % 16)]; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l / 16) + r) * 16 + (l % 16)] = c[r];
}
template <int FORM>
__global__ void __launch_bounds__(256) rate_kernel(const h16x8* src, float* out, int iters) {
    const int l = threadIdx.x & 63;
    h16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = src[i * 64 + l]; b[i] = src[(4 + i) * 64 + l]; }
    float res = 0.f;
    if (FORM == 16) {
        f32x4 c[6] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 6; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i + it) & 3], c[i], 0, 0, 0);
        for (int i = 0; i < 6; ++i) res += c[i][0];
    } else {
        f32x16 c[3] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 3; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[(i + it) & 3], c[i], 0, 0, 0);
        for (int i = 0; i < 3; ++i) res += c[i][0];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = res;
}
int main() {
    std::vector<_Float16> A(16 * 32), B(32 * 16); std::vector<float> D(256), R(256, 0.f);
    for (int i = 0; i < 512; ++i) { A[i] = (_Float16)((i * 37 % 17 - 8) / 8.0f); B[i] = (_Float16)((i * 53 % 13 - 6) / 4.0f); }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < 32; ++k) s += (float)A[i * 32 + k] * (float)B[k * 16 + j]; R[i * 16 + j] = s; }
    _Float16 *dA, *dB; float* dD; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dD, 1024));
    CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD); CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 256; ++i) bad += D[i] != R[i];
    printf("v_mfma_f32_16x16x32_f16 layout (A: row l%%16, k 8(l/16)+j; B: col l%%16, k 8(l/16)+j; D: col l%%16, rows 4(l/16)+r): %d of 256 results differ\n", bad);
    h16x8* src; float* out; CK(hipMalloc(&src, 8 * 64 * 16)); CK(hipMemset(src, 0x3c, 8 * 64 * 16)); CK(hipMalloc(&out, 1024 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int form : {16, 32}) for (int waves : {4, 8}) {
        const int iters = 4000; float ms;
        auto launch = [&]() { if (form == 16) hipLaunchKernelGGL(rate_kernel<16>, dim3(256), dim3(64 * waves), 0, 0, src, out, iters); else hipLaunchKernelGGL(rate_kernel<32>, dim3(256), dim3(64 * waves), 0, 0, src, out, iters); };
        launch(); CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        const double flop = 256.0 * waves * iters * (form == 16 ? 6 * 2.0 * 16 * 16 * 32 : 3 * 2.0 * 32 * 32 * 16);
        printf("  %dx%d form, %d waves per CU: %.1f us, %.0f TFLOP/s (f16 passes)\n", form, form, waves, ms * 1e3, flop / (ms * 1e-3) * 1e-12);
    }
    return 0;
}
