// Experimental variants of the 64x64 f32-MFMA GEMM main loop (not product code).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// MODE 0: baseline (prefetch distance 1, 2 LDS buffers)
// MODE 1: no global loads inside the loop (LDS + MFMA + barrier ceiling)
// MODE 2: prefetch distance 2 (two register sets), 2 LDS buffers
// MODE 3: baseline + all 8 fragments of a k-tile read up front
// MODE 4: no loads, no barrier, no lds writes (pure LDS-read + MFMA)
template <int MODE>
__global__ void __launch_bounds__(256)
k64(const float* __restrict__ A, const float* __restrict__ Bt, float* __restrict__ C, int M, int N, int K, int mt, int nt, unsigned long long* clk)
{
    const unsigned long long t0_ = __builtin_amdgcn_s_memtime(), r0_ = __builtin_amdgcn_s_memrealtime();
    constexpr int BM = 64, BN = 64, LD = 36;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem; float* Bs = smem + 2 * BM * LD;
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int bn = slot % nt, bm = (slot / nt) * 8 + xcd;
    if (bm >= mt) return;
    const int bm0 = bm * BM, bn0 = bn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int srow = tid >> 3, scol = (tid & 7) * 4;
    const float* ap[2]; const float* bp[2];
    for (int i = 0; i < 2; ++i) { ap[i] = A + (size_t)min(bm0 + srow + 32 * i, M - 1) * K + scol; bp[i] = Bt + (size_t)(bn0 + srow + 32 * i) * K + scol; }
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int KT = K / 32;
    f32x4 ra0[2], rb0[2], ra1[2], rb1[2];
    auto issue = [&](f32x4 (&ra)[2], f32x4 (&rb)[2], int kt) {
        const int k0 = min(kt, KT - 1) * 32;
#pragma unroll
        for (int i = 0; i < 2; ++i) { ra[i] = *(const f32x4*)(ap[i] + k0); rb[i] = *(const f32x4*)(bp[i] + k0); }
    };
    auto stage = [&](const f32x4 (&ra)[2], const f32x4 (&rb)[2], int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *(f32x4*)&As[buf * BM * LD + (srow + 32 * i) * LD + scol] = ra[i];
            *(f32x4*)&Bs[buf * BN * LD + (srow + 32 * i) * LD + scol] = rb[i];
        }
    };
    const int fr = lane & 31, fk = (lane >> 5) * 4;
    auto compute = [&](int buf) {
        const float* Ac = As + buf * BM * LD + (wm * 32 + fr) * LD + fk;
        const float* Bc = Bs + buf * BN * LD + (wn * 32 + fr) * LD + fk;
        if (MODE == 3) {
            f32x4 af[4], bf[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) { af[kk] = *(const f32x4*)(Ac + kk * 8); bf[kk] = *(const f32x4*)(Bc + kk * 8); }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk][s], bf[kk][s], acc, 0, 0, 0);
        } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const f32x4 af = *(const f32x4*)(Ac + kk * 8), bf = *(const f32x4*)(Bc + kk * 8);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[s], acc, 0, 0, 0);
            }
        }
    };
    issue(ra0, rb0, 0); stage(ra0, rb0, 0);
    if (MODE == 2) { issue(ra1, rb1, 1); }
    __syncthreads();
    if (MODE == 0 || MODE == 3) {
        for (int kt = 0; kt < KT; ++kt) {
            issue(ra0, rb0, kt + 1);
            compute(kt & 1);
            stage(ra0, rb0, (kt & 1) ^ 1);
            __syncthreads();
        }
    } else if (MODE == 1) {
        for (int kt = 0; kt < KT; ++kt) {
            compute(kt & 1);
            stage(ra0, rb0, (kt & 1) ^ 1);
            __syncthreads();
        }
    } else if (MODE == 4) {
        for (int kt = 0; kt < KT; ++kt) compute(0);
    } else {   // MODE 2: at iteration kt, set (kt+1)&1 holds tile kt+1 (issued one iteration ago); issue tile kt+2 into the other set
        for (int kt = 0; kt < KT; kt += 2) {
            issue(ra0, rb0, kt + 2);        // ra0 free: tile kt was staged already
            compute(0);
            stage(ra1, rb1, 1);             // tile kt+1
            __syncthreads();
            issue(ra1, rb1, kt + 3);
            compute(1);
            stage(ra0, rb0, 0);             // tile kt+2
            __syncthreads();
        }
    }
    if (blockIdx.x == 1000 && threadIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - t0_; clk[1] = __builtin_amdgcn_s_memrealtime() - r0_; }
    const int crow0 = bm0 + wm * 32 + 4 * (lane >> 5), col = bn0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) { const int row = crow0 + (r & 3) + 8 * (r >> 2); if (bm0 + 64 <= M || row < M) C[(size_t)row * N + col] = acc[r]; }
}


// Persistent variant: grid = 1024 workgroups, each walks its XCD's tile list; the next tile's first
// k-tile is prefetched during the current tile's last k-iteration, epilogue stores overlap the next MFMAs.
template <int FULL>
__global__ void __launch_bounds__(256)
k64p(const float* __restrict__ A, const float* __restrict__ Bt, float* __restrict__ C, int M, int N, int K, int mt, int nt, unsigned long long* clk)
{
    constexpr int BM = 64, BN = 64, LD = 36;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem; float* Bs = smem + 2 * BM * LD;
    const int w = blockIdx.x, xcd = w & 7, j0 = w >> 3, step = gridDim.x >> 3;
    const int nbx = (mt - xcd + 7) / 8;            // M-tiles owned by this XCD
    const int count = nbx * nt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int srow = tid >> 3, scol = (tid & 7) * 4;
    const int fr = lane & 31, fk = (lane >> 5) * 4;
    const int KT = K / 32;
    if (j0 >= count) return;
    const float* ap[2]; const float* bp[2]; const float* apn[2]; const float* bpn[2];
    auto setup = [&](int i, const float* (&a)[2], const float* (&b)[2], int& bm0, int& bn0) {
        const int bm = xcd + 8 * (i / nt), bn = i % nt;
        bm0 = bm * BM; bn0 = bn * BN;
#pragma unroll
        for (int q = 0; q < 2; ++q) { a[q] = A + (size_t)min(bm0 + srow + 32 * q, M - 1) * K + scol; b[q] = Bt + (size_t)(bn0 + srow + 32 * q) * K + scol; }
    };
    int bm0, bn0, bm0n = 0, bn0n = 0;
    setup(j0, ap, bp, bm0, bn0);
    f32x4 ra[2], rb[2];
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) { ra[q] = *(const f32x4*)(ap[q]); rb[q] = *(const f32x4*)(bp[q]); }
#pragma unroll
    for (int q = 0; q < 2; ++q) { *(f32x4*)&As[(srow + 32 * q) * LD + scol] = ra[q]; *(f32x4*)&Bs[(srow + 32 * q) * LD + scol] = rb[q]; }
    __syncthreads();
    int it = 0;
    for (int i = j0; i < count; i += step) {
        const bool has_next = (i + step < count);
        for (int kt = 0; kt < KT; ++kt, ++it) {
            const bool last = (kt == KT - 1);
            if (FULL) {
                if (!last) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) { ra[q] = *(const f32x4*)(ap[q] + (kt + 1) * 32); rb[q] = *(const f32x4*)(bp[q] + (kt + 1) * 32); }
                } else if (has_next) {
                    setup(i + step, apn, bpn, bm0n, bn0n);
#pragma unroll
                    for (int q = 0; q < 2; ++q) { ra[q] = *(const f32x4*)(apn[q]); rb[q] = *(const f32x4*)(bpn[q]); }
                }
            }
            const int buf = it & 1;
            const float* Ac = As + buf * BM * LD + (wm * 32 + fr) * LD + fk;
            const float* Bc = Bs + buf * BN * LD + (wn * 32 + fr) * LD + fk;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const f32x4 af = *(const f32x4*)(Ac + kk * 8), bf = *(const f32x4*)(Bc + kk * 8);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[s], acc, 0, 0, 0);
            }
            if (FULL) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    *(f32x4*)&As[(buf ^ 1) * BM * LD + (srow + 32 * q) * LD + scol] = ra[q];
                    *(f32x4*)&Bs[(buf ^ 1) * BN * LD + (srow + 32 * q) * LD + scol] = rb[q];
                }
                __syncthreads();
            }
        }
        const int crow0 = bm0 + wm * 32 + 4 * (lane >> 5), col = bn0 + wn * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) { const int row = crow0 + (r & 3) + 8 * (r >> 2); if (bm0 + 64 <= M || row < M) C[(size_t)row * N + col] = acc[r]; acc[r] = 0.f; }
        if (FULL && has_next) { ap[0] = apn[0]; ap[1] = apn[1]; bp[0] = bpn[0]; bp[1] = bpn[1]; }
        if (FULL) { bm0 = bm0n; bn0 = bn0n; } else if (has_next) { setup(i + step, ap, bp, bm0, bn0); }
    }
    if (blockIdx.x == 1000 && threadIdx.x == 0) { clk[0] = 1; clk[1] = 1; }
}
template <int FULL> void runp(const char* name, float* dA, float* dB, float* dC, int M, int N, int K, int grid) {
    auto kern = k64p<FULL>; size_t lds = 2 * 128 * 36 * 4;
    int mt = (M + 63) / 64, nt = N / 64;
    static unsigned long long* dclk = nullptr; if (!dclk) CK(hipMalloc(&dclk, 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, dA, dB, dC, M, N, K, mt, nt, dclk);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, dA, dB, dC, M, N, K, mt, nt, dclk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    printf("  %-28s %7.1f us %6.1f TF (grid %d)\n", name, ms * 1e3, 2.0 * M * N * K / ms / 1e9, grid);
}


// Baseline structure (prefetch distance 1, 2 LDS buffers, non-persistent) with the 16x16x4 MFMA, stride-40 LDS rows.
__global__ void __launch_bounds__(256)
k64_16(const float* __restrict__ A, const float* __restrict__ Bt, float* __restrict__ C, int M, int N, int K, int mt, int nt, unsigned long long* clk)
{
    constexpr int BM = 64, BN = 64, LD = 40;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem; float* Bs = smem + 2 * BM * LD;
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int bn = slot % nt, bm = (slot / nt) * 8 + xcd;
    if (bm >= mt) return;
    const int bm0 = bm * BM, bn0 = bn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int srow = tid >> 3, scol = (tid & 7) * 4;
    const float* ap[2]; const float* bp[2];
    for (int i = 0; i < 2; ++i) { ap[i] = A + (size_t)min(bm0 + srow + 32 * i, M - 1) * K + scol; bp[i] = Bt + (size_t)(bn0 + srow + 32 * i) * K + scol; }
    f32x4 acc[2][2]; for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int KT = K / 32;
    f32x4 ra[2], rb[2];
    const int fr = lane & 15, fg = lane >> 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) { ra[i] = *(const f32x4*)(ap[i]); rb[i] = *(const f32x4*)(bp[i]); }
#pragma unroll
    for (int i = 0; i < 2; ++i) { *(f32x4*)&As[(srow + 32 * i) * LD + scol] = ra[i]; *(f32x4*)&Bs[(srow + 32 * i) * LD + scol] = rb[i]; }
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1, k0 = min(kt + 1, KT - 1) * 32;
#pragma unroll
        for (int i = 0; i < 2; ++i) { ra[i] = *(const f32x4*)(ap[i] + k0); rb[i] = *(const f32x4*)(bp[i] + k0); }
        const float* Ac = As + buf * BM * LD + (wm * 32 + fr) * LD + 4 * fg;
        const float* Bc = Bs + buf * BN * LD + (wn * 32 + fr) * LD + 4 * fg;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            f32x4 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) { af[i] = *(const f32x4*)(Ac + i * 16 * LD + kk * 16); bf[i] = *(const f32x4*)(Bc + i * 16 * LD + kk * 16); }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) { *(f32x4*)&As[(buf ^ 1) * BM * LD + (srow + 32 * i) * LD + scol] = ra[i]; *(f32x4*)&Bs[(buf ^ 1) * BN * LD + (srow + 32 * i) * LD + scol] = rb[i]; }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int row = bm0 + wm * 32 + 16 * i + 4 * fg + r, col = bn0 + wn * 32 + 16 * j + fr; if (row < M) C[(size_t)row * N + col] = acc[i][j][r]; }
}
void run16(const char* name, float* dA, float* dB, float* dC, int M, int N, int K) {
    size_t lds = 2 * 128 * 40 * 4;
    int mt = (M + 63) / 64, nt = N / 64, grid = ((mt + 7) / 8 * 8) * nt;
    static unsigned long long* dclk = nullptr; if (!dclk) CK(hipMalloc(&dclk, 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k64_16, dim3(grid), dim3(256), lds, 0, dA, dB, dC, M, N, K, mt, nt, dclk);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k64_16, dim3(grid), dim3(256), lds, 0, dA, dB, dC, M, N, K, mt, nt, dclk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    printf("  %-28s %7.1f us %6.1f TF\n", name, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
}

template <int MODE> void run(const char* name, float* dA, float* dB, float* dC, int M, int N, int K) {
    auto kern = k64<MODE>; size_t lds = 2 * 128 * 36 * 4;
    int mt = (M + 63) / 64, nt = N / 64, grid = ((mt + 7) / 8 * 8) * nt;
    static unsigned long long* dclk = nullptr; if (!dclk) CK(hipMalloc(&dclk, 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, dA, dB, dC, M, N, K, mt, nt, dclk);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, dA, dB, dC, M, N, K, mt, nt, dclk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    unsigned long long hc[2]; CK(hipMemcpy(hc, dclk, 16, hipMemcpyDeviceToHost));
    printf("  %-28s %7.1f us %6.1f TF  clock %.0f MHz (wg lifetime %.1f us)\n", name, ms * 1e3, 2.0 * M * N * K / ms / 1e9, (double)hc[0] / hc[1] * 100.0, hc[1] / 100.0);
}
int main() {
    const int Mmax = 10496, Nmax = 1280, Kmax = 2304;
    std::vector<float> h((size_t)Mmax * Kmax); srand(1); for (auto& v : h) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, h.size() * 4)); CK(hipMemcpy(dA, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dB, (size_t)Nmax * Kmax * 4)); CK(hipMemcpy(dB, h.data(), (size_t)Nmax * Kmax * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dC, (size_t)Mmax * Nmax * 4));
    int shapes[][3] = {{9088, 1152, 384}, {9088, 768, 384}, {9088, 384, 768}, {9088, 384, 384}};
    for (auto& s : shapes) {
        printf("M=%d N=%d K=%d\n", s[0], s[1], s[2]);
        run<0>("baseline", dA, dB, dC, s[0], s[1], s[2]);
        run16("baseline, 16x16x4 mfma", dA, dB, dC, s[0], s[1], s[2]);
        run<1>("no global loads in loop", dA, dB, dC, s[0], s[1], s[2]);
        run<4>("lds-read + mfma only", dA, dB, dC, s[0], s[1], s[2]);
        run<2>("prefetch distance 2", dA, dB, dC, s[0], s[1], s[2]);
        run<3>("fragments up front", dA, dB, dC, s[0], s[1], s[2]);
        runp<0>("persistent lds+mfma only", dA, dB, dC, s[0], s[1], s[2], 1024);
        runp<1>("persistent pipelined", dA, dB, dC, s[0], s[1], s[2], 1024);
        runp<1>("persistent pipelined", dA, dB, dC, s[0], s[1], s[2], 768);
    }
}
