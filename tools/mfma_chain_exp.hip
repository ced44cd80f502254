// Round 4: what does the f16x3 inner pattern cost on the matrix pipe by itself?  Per k-slice the panel kernels issue
//     acc0 += ah * bh;  acc1 += ah * bl;  acc1 += al * bh          (two of three MFMAs on the same accumulator)
// Variants (one wave per SIMD unless noted, operands random, 864 MFMAs per wave like the QKV launch):
//   0  the pattern as is, operands in registers                     3  the pattern + 2 ds_read_b128 per slice, counted waits (the kernels' loop)
//   1  three independent accumulators                               4  variant 0 with two waves per SIMD (half the MFMAs each)
//   2  order acc1, acc0, acc1 (the dependent pair separated)        5  variant 3 with two waves per SIMD
// Prints s_memtime cycles per MFMA (median over workgroups), wall time and the clock they imply (s_memrealtime = 100 MHz).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/mfma_chain_exp tools/mfma_chain_exp.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) kern(const h16x8* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ clk, int chunks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    constexpr int SL = WAVES == 4 ? 24 : 12;                 // k-slices per wave and chunk
    const int lane = threadIdx.x & 63;
    h16x8 ah[SL], al[SL];
#pragma unroll
    for (int s = 0; s < SL; ++s) { ah[s] = src[(s * 2) * 64 + lane]; al[s] = src[(s * 2 + 1) * 64 + lane]; }
    for (int i = threadIdx.x; i < 48 * 1024 / 16; i += blockDim.x) reinterpret_cast<h16x8*>(sm)[i] = src[i];
    __syncthreads();
    h16x8 bh[3], bl[3];
    bh[0] = src[4096 + lane]; bl[0] = src[4160 + lane]; bh[1] = src[4224 + lane]; bl[1] = src[4288 + lane]; bh[2] = src[4352 + lane]; bl[2] = src[4416 + lane];
    f32x16 a0, a1, a2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; a2[r] = 0.f; }
    const unsigned sb = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(sm + lane * 16);
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int c = 0; c < chunks; ++c) {
        if (V == 3 || V == 5) {
            asm volatile("ds_read_b128 %0, %2 offset:0\n\tds_read_b128 %1, %2 offset:1024" : "=&v"(bh[0]), "=&v"(bl[0]) : "v"(sb));
            asm volatile("ds_read_b128 %0, %2 offset:2048\n\tds_read_b128 %1, %2 offset:3072" : "=&v"(bh[1]), "=&v"(bl[1]) : "v"(sb));
        }
#pragma unroll
        for (int kk = 0; kk < SL; ++kk) {
            if (V == 3 || V == 5) {
                if (kk + 2 < SL) asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" : "=&v"(bh[(kk + 2) % 3]), "=&v"(bl[(kk + 2) % 3]) : "v"(sb), "i"((kk + 2) * 2048), "i"((kk + 2) * 2048 + 1024));
                asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(bh[kk % 3]), "+v"(bl[kk % 3]) : "i"(kk + 2 < SL ? 4 : (kk + 1 < SL ? 2 : 0)));
            }
            if (V == 1) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kk], bh[kk % 3], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kk], bl[kk % 3], a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[kk], bh[kk % 3], a2, 0, 0, 0);
            } else if (V == 2) {
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kk], bl[kk % 3], a1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kk], bh[kk % 3], a0, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[kk], bh[kk % 3], a1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kk], bh[kk % 3], a0, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kk], bl[kk % 3], a1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[kk], bh[kk % 3], a1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { clk[(blockIdx.x * WAVES + (threadIdx.x >> 6)) * 2] = (unsigned long long)(t1 - t0); clk[(blockIdx.x * WAVES + (threadIdx.x >> 6)) * 2 + 1] = (unsigned long long)(r1 - r0); }
}

// the same products on v_mfma_f32_16x16x32_f16: a 32 x 32 tile = 2 x 2 tiles of 16 x 16, 32-deep slices; per slice pair 12 MFMAs of half the size
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES) kern16(const h16x8* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ clk, int chunks)
{
    constexpr int SL = WAVES == 4 ? 12 : 6;                  // 32-deep k-slices per wave and chunk
    const int lane = threadIdx.x & 63;
    h16x8 ah[SL][2], al[SL][2];
#pragma unroll
    for (int s = 0; s < SL; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) { ah[s][i] = src[(s * 4 + i) * 64 + lane]; al[s][i] = src[(s * 4 + 2 + i) * 64 + lane]; }
    h16x8 bh[2], bl[2];
    bh[0] = src[4096 + lane]; bl[0] = src[4160 + lane]; bh[1] = src[4224 + lane]; bl[1] = src[4288 + lane];
    f32x4v a0[2][2], a1[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) { a0[i][j][r] = 0.f; a1[i][j][r] = 0.f; }
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int c = 0; c < chunks; ++c) {
#pragma unroll
        for (int kk = 0; kk < SL; ++kk) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    a0[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kk][i], bh[j], a0[i][j], 0, 0, 0);
                    a1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kk][i], bl[j], a1[i][j], 0, 0, 0);
                    a1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[kk][i], bh[j], a1[i][j], 0, 0, 0);
                }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += a0[i][j][r] + a1[i][j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { clk[(blockIdx.x * WAVES + (threadIdx.x >> 6)) * 2] = (unsigned long long)(t1 - t0); clk[(blockIdx.x * WAVES + (threadIdx.x >> 6)) * 2 + 1] = (unsigned long long)(r1 - r0); }
}
template <int WAVES> void run16(const h16x8* src, float* out, unsigned long long* clk, const char* what) {
    const int wgs = 256, chunks = 12, SL = WAVES == 4 ? 12 : 6;
    auto k = kern16<WAVES>;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k, dim3(wgs), dim3(64 * WAVES), 0, 0, src, out, clk, chunks);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k, dim3(wgs), dim3(64 * WAVES), 0, 0, src, out, clk, chunks);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(wgs * WAVES * 2); CK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> cyc, rt; for (int i = 0; i < wgs * WAVES; ++i) { cyc.push_back((double)h[2 * i]); rt.push_back((double)h[2 * i + 1]); }
    std::sort(cyc.begin(), cyc.end()); std::sort(rt.begin(), rt.end());
    const double mf = chunks * SL * 12.0, c = cyc[cyc.size() / 2], r = rt[rt.size() / 2];
    printf("%-64s %6.1f cycles / MFMA (per wave), loop %6.2f us, in-kernel clock %.2f GHz, launch %6.2f us\n", what, c / mf, r / 100.0, c / r / 10.0, ms * 1e3 / 200);
}

template <int V, int WAVES> void run(const h16x8* src, float* out, unsigned long long* clk, const char* what) {
    const int wgs = 256, chunks = 12, SL = WAVES == 4 ? 24 : 12;
    auto k = kern<V, WAVES>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    // several launches back to back so that the chip settles at the clock it holds under this load
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k, dim3(wgs), dim3(64 * WAVES), 160 * 1024, 0, src, out, clk, chunks);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k, dim3(wgs), dim3(64 * WAVES), 160 * 1024, 0, src, out, clk, chunks);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(wgs * WAVES * 2); CK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> cyc, rt; for (int i = 0; i < wgs * WAVES; ++i) { cyc.push_back((double)h[2 * i]); rt.push_back((double)h[2 * i + 1]); }
    std::sort(cyc.begin(), cyc.end()); std::sort(rt.begin(), rt.end());
    const double mf = chunks * SL * 3.0, c = cyc[cyc.size() / 2], r = rt[rt.size() / 2];
    printf("%-64s %6.1f cycles / MFMA (per wave), loop %6.2f us, in-kernel clock %.2f GHz, launch %6.2f us\n", what, c / mf, r / 100.0, c / r / 10.0, ms * 1e3 / 200);
}

int main() {
    std::vector<_Float16> h(8192 * 8); srand(1); for (auto& v : h) v = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    h16x8* src; float* out; unsigned long long* clk;
    CK(hipMalloc(&src, h.size() * 2)); CK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&clk, 256 * 8 * 16));
    run<0, 4>(src, out, clk, "0: acc0, acc1, acc1 (registers), 1 wave / SIMD");
    run<1, 4>(src, out, clk, "1: three independent accumulators, 1 wave / SIMD");
    run<2, 4>(src, out, clk, "2: acc1, acc0, acc1, 1 wave / SIMD");
    run<3, 4>(src, out, clk, "3: pattern + 2 ds_read_b128 per slice, 1 wave / SIMD");
    run<0, 8>(src, out, clk, "4: pattern (registers), 2 waves / SIMD, half the MFMAs each");
    run<3, 8>(src, out, clk, "5: pattern + LDS reads, 2 waves / SIMD");
    run16<4>(src, out, clk, "6: the same products on 16x16x32 MFMAs (registers), 1 wave / SIMD");
    run16<8>(src, out, clk, "7: 16x16x32, 2 waves / SIMD");
    run<0, 4>(src, out, clk, "0 again");
    run16<4>(src, out, clk, "6 again");
    return 0;
}
