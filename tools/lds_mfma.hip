#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
// V=0: read 2 frags then 4 MFMA (compiler schedule); V=1: explicit register double-buffer of frags;
// V=2: 16x16x4 MFMA, 2x2 tiles of 16x16 per wave (same output tile), stride 40; V=3: as V=2 with double-buffer
template <int V>
__global__ void __launch_bounds__(256) k(float* out, int iters, unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) float As[2 * 64 * 40], Bs[2 * 64 * 40];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < 2 * 64 * 40; i += 256) { As[i] = i * 0.001f; Bs[i] = 1.f - i * 0.0007f; }
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    if (V < 2) {
        const float* Ac0 = As + (wm * 32 + (lane & 31)) * 36 + (lane >> 5) * 4;
        const float* Bc0 = Bs + (wn * 32 + (lane & 31)) * 36 + (lane >> 5) * 4;
        f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (V == 0) {
            for (int it = 0; it < iters; ++it) {
                const float* Ac = Ac0 + (it & 1) * 64 * 40; const float* Bc = Bc0 + (it & 1) * 64 * 40;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const f32x4 af = *(const f32x4*)(Ac + kk * 8), bf = *(const f32x4*)(Bc + kk * 8);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q], bf[q], acc, 0, 0, 0);
                }
            }
        } else {
            f32x4 af = *(const f32x4*)(Ac0), bf = *(const f32x4*)(Bc0);
            for (int it = 0; it < iters; ++it) {
                const float* Ac = Ac0 + (it & 1) * 64 * 40; const float* Bc = Bc0 + (it & 1) * 64 * 40;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const f32x4 an = *(const f32x4*)(Ac + ((kk + 1) & 3) * 8), bn = *(const f32x4*)(Bc + ((kk + 1) & 3) * 8);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q], bf[q], acc, 0, 0, 0);
                    af = an; bf = bn;
                }
            }
        }
        for (int r = 0; r < 16; ++r) s += acc[r];
    } else {
        const float* Ac0 = As + (wm * 32 + (lane & 15)) * 40 + (lane >> 4) * 4;
        const float* Bc0 = Bs + (wn * 32 + (lane & 15)) * 40 + (lane >> 4) * 4;
        f32x4 acc[2][2]; for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
        if (V == 2) {
            for (int it = 0; it < iters; ++it) {
                const float* Ac = Ac0 + (it & 1) * 64 * 40; const float* Bc = Bc0 + (it & 1) * 64 * 40;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {   // 16 k per step
                    f32x4 a0 = *(const f32x4*)(Ac + kk * 16), a1 = *(const f32x4*)(Ac + 16 * 40 + kk * 16);
                    f32x4 b0 = *(const f32x4*)(Bc + kk * 16), b1 = *(const f32x4*)(Bc + 16 * 40 + kk * 16);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q], b0[q], acc[0][0], 0, 0, 0);
                        acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q], b1[q], acc[0][1], 0, 0, 0);
                        acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q], b0[q], acc[1][0], 0, 0, 0);
                        acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q], b1[q], acc[1][1], 0, 0, 0);
                    }
                }
            }
        } else {
            f32x4 a0 = *(const f32x4*)(Ac0), a1 = *(const f32x4*)(Ac0 + 16 * 40), b0 = *(const f32x4*)(Bc0), b1 = *(const f32x4*)(Bc0 + 16 * 40);
            for (int it = 0; it < iters; ++it) {
                const float* Ac = Ac0 + (it & 1) * 64 * 40; const float* Bc = Bc0 + (it & 1) * 64 * 40;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int nk = ((kk + 1) & 1) * 16;
                    f32x4 a0n = *(const f32x4*)(Ac + nk), a1n = *(const f32x4*)(Ac + 16 * 40 + nk);
                    f32x4 b0n = *(const f32x4*)(Bc + nk), b1n = *(const f32x4*)(Bc + 16 * 40 + nk);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q], b0[q], acc[0][0], 0, 0, 0);
                        acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q], b1[q], acc[0][1], 0, 0, 0);
                        acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q], b0[q], acc[1][0], 0, 0, 0);
                        acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q], b1[q], acc[1][1], 0, 0, 0);
                    }
                    a0 = a0n; a1 = a1n; b0 = b0n; b1 = b1n;
                }
            }
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + tid] = s;
    if (blockIdx.x == 100 && tid == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
template <int V> void run(const char* name, int wgs_per_cu, int iters) {
    float* out; unsigned long long* clk; CK(hipMalloc(&out, 256 * 256 * 16 * 4)); CK(hipMalloc(&clk, 16));
    int grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<V>), dim3(grid), dim3(256), 0, 0, out, iters, clk);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<V>), dim3(grid), dim3(256), 0, 0, out, iters, clk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    double flop = (double)grid * 4 * iters * 16 * 4096.0;    // per iteration: 32 deep k, 32x32 outputs per wave
    double mhz = (double)h[0] / h[1] * 100.0;
    printf("%-34s wgs/cu %d: %.3f ms %6.1f TF clock %.0f MHz -> %.1f%% of pipe peak at that clock\n", name, wgs_per_cu, ms, flop / ms / 1e9, mhz, flop / ms / 1e9 / (65536 * mhz * 1e-6) * 100);
}
int main() {
    for (int w : {1, 2, 4}) {
        run<0>("32x32x2 read-then-mfma", w, 2000 / w);
        run<1>("32x32x2 frag double-buffer", w, 2000 / w);
        run<2>("16x16x4 2x2 read-then-mfma", w, 2000 / w);
        run<3>("16x16x4 2x2 frag double-buffer", w, 2000 / w);
    }
}
