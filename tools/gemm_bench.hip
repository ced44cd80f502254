// Standalone micro-benchmark of uu3d::gemm_f32_kernel tile shapes on the model's GEMM shapes.
//   hipcc -O3 --offload-arch=gfx950 -o tools/gemm_bench tools/gemm_bench.hip && tools/gemm_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../uplift-upsample-3dhpe_amd/csrc/uu3d_gemm.h"
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
    CK(hipMalloc(&dstats, Mmax * 8)); CK(hipMemcpy(dstats, h.data(), Mmax * 8, hipMemcpyHostToDevice));
    int shapes[][3] = {{9088, 1152, 384}, {9088, 768, 384}, {9088, 384, 768}, {9088, 384, 384}, {9088, 384, 544},
                       {2944, 1152, 384}, {2944, 384, 2304}, {384, 384, 2304}, {128, 384, 2304}, {10496, 1152, 384}};
    for (auto& s : shapes) {
        printf("M=%d N=%d K=%d\n", s[0], s[1], s[2]);
        both<128, 128>(s[0], s[1], s[2]);
        both<128, 64>(s[0], s[1], s[2]);
        both<64, 128>(s[0], s[1], s[2]);
        both<64, 64>(s[0], s[1], s[2]);
    }
    return 0;
}
