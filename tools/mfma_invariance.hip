// Is v_mfma_f32_32x32x16_f16 bitwise invariant to WHERE a row / column sits in the tile?
// Test 1: B has 32 identical columns -> all columns of D = A B must be bitwise equal (column invariance).
// Test 2: A has 32 identical rows    -> all rows of D must be bitwise equal (row invariance).
// Random f16 data of mixed magnitudes (hi/lo-plane like), several accumulation steps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
// A: [steps][32][16], B: [steps][16][32] (f16), D: [32][32]
__global__ void k(const _Float16* A, const _Float16* B, int steps, float* D) {
    const int lane = threadIdx.x;
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int s = 0; s < steps; ++s) {
        h16x8 a, b;
        for (int e = 0; e < 8; ++e) {
            a[e] = A[(s * 32 + (lane & 31)) * 16 + 8 * (lane >> 5) + e];
            b[e] = B[(s * 16 + 8 * (lane >> 5) + e) * 32 + (lane & 31)];
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[r];
}
int main() {
    const int steps = 4, trials = 2000;
    _Float16 *dA, *dB; float* dD; CK(hipMalloc(&dA, steps * 512 * 2)); CK(hipMalloc(&dB, steps * 512 * 2)); CK(hipMalloc(&dD, 4096));
    std::vector<_Float16> A(steps * 512), B(steps * 512); std::vector<float> D(1024);
    srand(3);
    auto rnd = [] { float v = (rand() / (float)RAND_MAX * 2.f - 1.f); int e = rand() % 12; return v * (1.0f / (1 << e)) * 4.f; };
    long colbad = 0, rowbad = 0;
    for (int t = 0; t < trials; ++t) {
        // column invariance: every column of B equal to column 0
        for (auto& v : A) v = (_Float16)rnd();
        for (int s = 0; s < steps; ++s) for (int kk = 0; kk < 16; ++kk) { _Float16 v = (_Float16)rnd(); for (int j = 0; j < 32; ++j) B[(s * 16 + kk) * 32 + j] = v; }
        CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, steps, dD); CK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
        for (int i = 0; i < 32; ++i) for (int j = 1; j < 32; ++j) if (D[i * 32 + j] != D[i * 32]) { if (!colbad) printf("column variance: row %d col %d: %.9g vs col 0 %.9g\n", i, j, D[i * 32 + j], D[i * 32]); ++colbad; }
        // row invariance: every row of A equal to row 0
        for (auto& v : B) v = (_Float16)rnd();
        for (int s = 0; s < steps; ++s) for (int kk = 0; kk < 16; ++kk) { _Float16 v = (_Float16)rnd(); for (int i = 0; i < 32; ++i) A[(s * 32 + i) * 16 + kk] = v; }
        CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, steps, dD); CK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
        for (int j = 0; j < 32; ++j) for (int i = 1; i < 32; ++i) if (D[i * 32 + j] != D[j]) { if (!rowbad) printf("row variance: row %d col %d: %.9g vs row 0 %.9g\n", i, j, D[i * 32 + j], D[j]); ++rowbad; }
    }
    printf("v_mfma_f32_32x32x16_f16, %d trials x %d accumulation steps: column-variant elements %ld, row-variant elements %ld\n", trials, steps, colbad, rowbad);
    return 0;
}
