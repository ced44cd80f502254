// Standalone micro-benchmark of uu3d::gemm_f32_kernel tile shapes on the model's GEMM shapes.
//   hipcc -O3 --offload-arch=gfx950 -o tools/gemm_bench tools/gemm_bench.hip && tools/gemm_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../uplift-upsample-3dhpe_amd/csrc/uu3d_gemm.h"
#include "gemm_persistent_exp.h"
using namespace uu3d;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

static float *dA, *dB, *dC, *dbias, *dg, *db; static float2* dstats;

template <int BM, int BN, class AL, class EP>
float run(const AL& al, const EP& ep, int M, int N, int K, int iters) {
    auto kern = gemm_f32_kernel<BM, BN, AL, EP>;
    size_t lds = gemm_lds_bytes(BM, BN);
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int mt = (M + BM - 1) / BM, nt = (N + BN - 1) / BN;
    int grid = ((mt + 7) / 8 * 8) * nt;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, al, dB, M, N, K, mt, nt, K / 32, ep);
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, al, dB, M, N, K, mt, nt, K / 32, ep);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

static int* dticket;
template <int TM, int TN, class AL, class EP>
float runp(const AL& al, const EP& ep, int M, int N, int K, int iters, int grid_cap) {
    auto kern = gemm_f32p_kernel<TM, TN, AL, EP>;
    size_t lds = gemmp_lds_bytes(TM, TN); if (lds < 33 * 1024) lds = 33 * 1024;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int mt = (M + 32 * TM - 1) / (32 * TM), nt = (N + 32 * TN - 1) / (32 * TN), total = mt * nt;
    int grid = total < grid_cap ? total : grid_cap;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemsetAsync(dticket, 0, 4096 * 4, 0));
    for (int i = 0; i < 3; ++i) { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, al, dB, M, N, K, nt, total, dticket + 100 + i, ep); }
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, al, dB, M, N, K, nt, total, dticket + i, ep); }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}
template <int TM, int TN>
void bothp(int M, int N, int K, int cap) {
    ALoadPlain ap{dA, K, M, K}; ALoadLayerNorm aln{dA, dstats, dg, db, K, M, K};
    EpBias ep{dC, dbias, N}; EpBiasResidual er{dC, dbias, N, nullptr, nullptr, 1};
    float t1 = runp<TM, TN>(ap, ep, M, N, K, 20, cap);
    float t2 = runp<TM, TN>(aln, ep, M, N, K, 20, cap);
    float t3 = runp<TM, TN>(ap, er, M, N, K, 20, cap);
    double fl = 2.0 * M * N * K;
    printf(" P%3dx%-3d plain+bias %7.1f us %6.1f TF | ln+bias %7.1f us %6.1f TF | plain+res %7.1f us %6.1f TF (grid cap %d)\n", 32 * TM, 32 * TN,
           t1 * 1e3, fl / t1 / 1e9, t2 * 1e3, fl / t2 / 1e9, t3 * 1e3, fl / t3 / 1e9, cap);
}

template <int BM, int BN, class AL, class EP>
float runcap(const AL& al, const EP& ep, int M, int N, int K, int iters, int cap) {
    auto kern = gemm_f32_kernel<BM, BN, AL, EP>;
    size_t lds = gemm_lds_bytes(BM, BN);
    size_t want = (160 * 1024) / (cap + 1) + 1024; if (want > lds) lds = want;   // > 160K/(cap+1) -> at most cap per CU
    if (lds > 160 * 1024) lds = 160 * 1024;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int mt = (M + BM - 1) / BM, nt = (N + BN - 1) / BN;
    int grid = ((mt + 7) / 8 * 8) * nt;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, al, dB, M, N, K, mt, nt, K / 32, ep);
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, al, dB, M, N, K, mt, nt, K / 32, ep);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}
template <int BM, int BN>
void capped(int M, int N, int K) {
    ALoadLayerNorm aln{dA, dstats, dg, db, K, M, K}; EpBias ep{dC, dbias, N};
    double fl = 2.0 * M * N * K;
    printf("  %3dx%-3d ln+bias by occupancy cap:", BM, BN);
    for (int cap = 1; cap <= 4; ++cap) { float t = runcap<BM, BN>(aln, ep, M, N, K, 20, cap); printf("  cap%d %6.1f us %5.1f TF", cap, t * 1e3, fl / t / 1e9); }
    printf("\n");
}
template <int BM, int BN>
void both(int M, int N, int K) {
    ALoadPlain ap{dA, K, M, K}; ALoadLayerNorm aln{dA, dstats, dg, db, K, M, K};
    EpBias ep{dC, dbias, N}; EpBiasResidual er{dC, dbias, N, nullptr, nullptr, 1};
    float t1 = run<BM, BN>(ap, ep, M, N, K, 20);
    float t2 = run<BM, BN>(aln, ep, M, N, K, 20);
    float t3 = run<BM, BN>(ap, er, M, N, K, 20);
    double fl = 2.0 * M * N * K;
    printf("  %3dx%-3d plain+bias %7.1f us %6.1f TF | ln+bias %7.1f us %6.1f TF | plain+res %7.1f us %6.1f TF\n", BM, BN,
           t1 * 1e3, fl / t1 / 1e9, t2 * 1e3, fl / t2 / 1e9, t3 * 1e3, fl / t3 / 1e9);
}

int main() {
    const int Mmax = 10496, Nmax = 1280, Kmax = 2304;
    std::vector<float> h((size_t)Mmax * Kmax);
    srand(1); for (auto& v : h) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    CK(hipMalloc(&dA, h.size() * 4)); CK(hipMemcpy(dA, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dB, (size_t)Nmax * Kmax * 4)); CK(hipMemcpy(dB, h.data(), (size_t)Nmax * Kmax * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dC, (size_t)Mmax * Nmax * 4)); CK(hipMemset(dC, 0, (size_t)Mmax * Nmax * 4));
    CK(hipMalloc(&dbias, Nmax * 4)); CK(hipMemset(dbias, 0, Nmax * 4));
    CK(hipMalloc(&dg, Kmax * 4)); CK(hipMemcpy(dg, h.data(), Kmax * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&db, Kmax * 4)); CK(hipMemcpy(db, h.data() + 5000, Kmax * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dticket, 4096 * 4));
    CK(hipMalloc(&dstats, Mmax * 8)); CK(hipMemcpy(dstats, h.data(), Mmax * 8, hipMemcpyHostToDevice));
    int shapes[][3] = {{9088, 1152, 384}, {9088, 768, 384}, {9088, 384, 768}, {9088, 384, 384}, {9088, 384, 544},
                       {2944, 1152, 384}, {2944, 384, 2304}, {384, 384, 2304}, {128, 384, 2304}, {10496, 1152, 384}};
    const char* only = getenv("SHAPES"); int nshape = only ? atoi(only) : 100; int si = 0;
    for (auto& s : shapes) {
        if (si++ >= nshape) break;
        printf("M=%d N=%d K=%d\n", s[0], s[1], s[2]);
        both<64, 64>(s[0], s[1], s[2]);
        capped<64, 64>(s[0], s[1], s[2]); capped<128, 64>(s[0], s[1], s[2]); capped<128, 128>(s[0], s[1], s[2]);
        if (getenv("NOP")) continue;
        bothp<2, 2>(s[0], s[1], s[2], 1024);
        bothp<1, 2>(s[0], s[1], s[2], 1024);
        bothp<2, 1>(s[0], s[1], s[2], 1024);
        bothp<1, 1>(s[0], s[1], s[2], 1024);
        bothp<2, 2>(s[0], s[1], s[2], 768);
    }
    return 0;
}
