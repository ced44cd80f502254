#include <cstring>
// Standalone micro-benchmark of uu3d::gemm_f32_kernel tile shapes on the model's GEMM shapes.
//   hipcc -O3 --offload-arch=gfx950 -o tools/gemm_bench tools/gemm_bench.hip && tools/gemm_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../uplift-upsample-3dhpe_amd/csrc/uu3d_gemm.h"
#include "gemm_persistent_exp.h"
#include "../uplift-upsample-3dhpe_amd/csrc/uu3d_gemm_h3.h"
#include "gemm_planes_regstage_exp.h"
#include <cmath>
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
static _Float16 *dBh, *dBl;
static std::vector<float> hA, hB;

struct ALoadLNConstStats {   // experiment: LayerNorm loader without the per-row stats load
    const float* __restrict__ A; const float2* __restrict__ stats; const float* __restrict__ gamma; const float* __restrict__ beta; int lda, M, K;
    struct Ctx { const float* p; float mean, rstd; };
    struct Raw { f32x4 x, g, b; };
    __device__ __forceinline__ Ctx prep(int row) const { const int rc = min(row, M - 1); Ctx c; c.p = A + (size_t)rc * lda; c.mean = 0.1f; c.rstd = 1.3f; return c; }
    __device__ __forceinline__ Raw issue(const Ctx& c, int k) const { const int kc = min(k, K - 4); Raw r; r.x = *reinterpret_cast<const f32x4*>(c.p + kc); r.g = *reinterpret_cast<const f32x4*>(gamma + kc); r.b = *reinterpret_cast<const f32x4*>(beta + kc); return r; }
    __device__ __forceinline__ f32x4 finish(const Ctx& c, int k, const Raw& r) const { f32x4 y; for (int e = 0; e < 4; ++e) { const float inv = c.rstd * r.g[e]; y[e] = r.x[e] * inv + (r.b[e] - c.mean * inv); } return (k < K) ? y : (f32x4){0.f, 0.f, 0.f, 0.f}; }
};
struct ALoadLNNoGB {         // experiment: stats loaded, gamma = 1, beta = 0 (no gamma / beta loads)
    const float* __restrict__ A; const float2* __restrict__ stats; const float* __restrict__ gamma; const float* __restrict__ beta; int lda, M, K;
    struct Ctx { const float* p; float mean, rstd; };
    struct Raw { f32x4 x; };
    __device__ __forceinline__ Ctx prep(int row) const { const int rc = min(row, M - 1); Ctx c; c.p = A + (size_t)rc * lda; const float2 s = stats[rc]; c.mean = s.x; c.rstd = s.y; return c; }
    __device__ __forceinline__ Raw issue(const Ctx& c, int k) const { const int kc = min(k, K - 4); Raw r; r.x = *reinterpret_cast<const f32x4*>(c.p + kc); return r; }
    __device__ __forceinline__ f32x4 finish(const Ctx& c, int k, const Raw& r) const { f32x4 y; for (int e = 0; e < 4; ++e) { y[e] = r.x[e] * c.rstd - c.mean * c.rstd; } return (k < K) ? y : (f32x4){0.f, 0.f, 0.f, 0.f}; }
};
struct ALoadLNNoSel {        // experiment: full LayerNorm, no k < K select
    const float* __restrict__ A; const float2* __restrict__ stats; const float* __restrict__ gamma; const float* __restrict__ beta; int lda, M, K;
    struct Ctx { const float* p; float mean, rstd; };
    struct Raw { f32x4 x, g, b; };
    __device__ __forceinline__ Ctx prep(int row) const { const int rc = min(row, M - 1); Ctx c; c.p = A + (size_t)rc * lda; const float2 s = stats[rc]; c.mean = s.x; c.rstd = s.y; return c; }
    __device__ __forceinline__ Raw issue(const Ctx& c, int k) const { Raw r; r.x = *reinterpret_cast<const f32x4*>(c.p + k); r.g = *reinterpret_cast<const f32x4*>(gamma + k); r.b = *reinterpret_cast<const f32x4*>(beta + k); return r; }
    __device__ __forceinline__ f32x4 finish(const Ctx& c, int k, const Raw& r) const { f32x4 y; for (int e = 0; e < 4; ++e) { const float inv = c.rstd * r.g[e]; y[e] = r.x[e] * inv + (r.b[e] - c.mean * inv); } return y; }
};
template <int TM, int TN, class AL, class EP> float runh3(const AL& al, const EP& ep, int M, int N, int K, int iters);
template <int TM, int TN, class AL>
void detcheck(const char* name, int M, int N, int K) {
    AL al{dA, dstats, dg, db, K, M, K}; EpBias ep{dC, dbias, N};
    std::vector<float> c1((size_t)M * N), c2((size_t)M * N);
    size_t tot = 0;
    for (int rep = 0; rep < 3; ++rep) {
        runh3<TM, TN>(al, ep, M, N, K, 1); CK(hipMemcpy(c1.data(), dC, c1.size() * 4, hipMemcpyDeviceToHost));
        runh3<TM, TN>(al, ep, M, N, K, 1); CK(hipMemcpy(c2.data(), dC, c2.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < c1.size(); ++i) tot += (c1[i] != c2[i]);
    }
    printf("   det %-12s %3dx%-3d mismatches %zu\n", name, 64 * TM, 64 * TN, tot);
    if (getenv("FORENSIC") && tot && !strcmp(name, "ln") && TM == 1 && TN == 2) {
        // dump what a host-side least-squares needs to find which staged A values were wrong
        std::vector<float> c3((size_t)M * N);
        run<64, 64>(al, ep, M, N, K, 1); CK(hipMemcpy(c3.data(), dC, c3.size() * 4, hipMemcpyDeviceToHost));
        FILE* f = fopen("gpurun_out/forensic.bin", "wb");
        int hdr[4] = {M, N, K, 0}; std::vector<int> rows;
        for (int r = 0; r < M && rows.size() < 64; ++r) { bool bad = false; for (int n = 0; n < N; ++n) if (std::fabs(c1[(size_t)r * N + n] - c3[(size_t)r * N + n]) > 1e-3 || std::fabs(c2[(size_t)r * N + n] - c3[(size_t)r * N + n]) > 1e-3) bad = true; if (bad) rows.push_back(r); }
        hdr[3] = (int)rows.size(); fwrite(hdr, 4, 4, f); fwrite(rows.data(), 4, rows.size(), f);
        fwrite(hB.data(), 4, (size_t)N * K, f);                       // B[n][k]
        fwrite(hA.data(), 4, K, f); fwrite(hA.data() + 5000, 4, K, f); // gamma, beta
        for (int r : rows) { fwrite(hA.data() + (size_t)r * K, 4, K, f); fwrite(hA.data() + 2 * (size_t)r, 4, 2, f);
                             fwrite(c1.data() + (size_t)r * N, 4, N, f); fwrite(c2.data() + (size_t)r * N, 4, N, f); fwrite(c3.data() + (size_t)r * N, 4, N, f); }
        fclose(f);
    }
}

static _Float16 *dAh, *dAl, *dOh, *dOl;
struct EpSink {   // experiment: the whole GEMM but (practically) no output traffic
    float* __restrict__ out; int ldo;
    __device__ __forceinline__ float2 colv(int) const { return make_float2(0.f, 0.f); }
    __device__ __forceinline__ float2 pre(int, int) const { return make_float2(0.f, 0.f); }
    __device__ __forceinline__ void store(int row, int col, float acc, float2, float2) const { if (acc == 12345.678f) out[(size_t)row * ldo + col] = acc; }
};
template <int TM, int TN, class PL, class EP, int DEPTH = 2>
float runh3p(const PL& pl, const EP& ep, int M, int N, int K, int iters) {
    auto kern = gemm_h3p_kernel<TM, TN, PL, EP, DEPTH>;
    size_t lds = gemm_h3_lds_bytes(64 * TM, 64 * TN);
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int mt = (M + 64 * TM - 1) / (64 * TM), nt = (N + 64 * TN - 1) / (64 * TN);
    int grid = ((mt + 7) / 8 * 8) * nt;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, pl, dBh, dBl, M, N, K, mt, nt, K / 32, ep);
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, pl, dBh, dBl, M, N, K, mt, nt, K / 32, ep);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}
template <int TM, int TN>
void h3p(int M, int N, int K) {
    {
        std::vector<_Float16> bh((size_t)N * K), bl((size_t)N * K);
        for (size_t i = 0; i < (size_t)N * K; ++i) { float x = hB[i]; _Float16 h = (_Float16)x; bh[i] = h; bl[i] = (_Float16)((x - (float)h) * 2048.0f); }
        CK(hipMemcpy(dBh, bh.data(), bh.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dBl, bl.data(), bl.size() * 2, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(split_rows_kernel, dim3(((size_t)M * K / 4 + 255) / 256), dim3(256), 0, 0, dA, K, K, M, dAh, dAl, K);
    PLoadPlain pl{dAh, dAl, K, M}; ALoadPlain ap{dA, K, M, K};
    EpBias ep{dC, dbias, N}; EpBiasResidual er{dC, dbias, N, nullptr, nullptr, 1}; EpBiasReluSplit es{dOh, dOl, dbias, N};
    std::vector<float> c1((size_t)M * N), c2((size_t)M * N), c3((size_t)M * N);
    runh3p<TM, TN>(pl, ep, M, N, K, 1); CK(hipMemcpy(c1.data(), dC, c1.size() * 4, hipMemcpyDeviceToHost));
    runh3p<TM, TN>(pl, ep, M, N, K, 1); CK(hipMemcpy(c2.data(), dC, c2.size() * 4, hipMemcpyDeviceToHost));
    runh3<1, 1>(ap, ep, M, N, K, 1); CK(hipMemcpy(c3.data(), dC, c3.size() * 4, hipMemcpyDeviceToHost));
    size_t nd = 0, nx = 0;
    for (size_t i = 0; i < c1.size(); ++i) { nd += (c1[i] != c2[i]); nx += (c1[i] != c3[i]); }
    float t1 = runh3p<TM, TN>(pl, ep, M, N, K, 20), t2 = runh3p<TM, TN>(pl, er, M, N, K, 20), t3 = runh3p<TM, TN>(pl, es, M, N, K, 20);
    float t0 = runh3p<TM, TN, PLoadPlain, EpBias, 1>(pl, ep, M, N, K, 20);
    double fl = 2.0 * M * N * K;
    EpSink sink{dC, N};
    float t4 = runh3p<TM, TN>(pl, sink, M, N, K, 20);
#ifdef H3P_CLOCK
    { unsigned long long z[3] = {0, 0, 0}, h[3]; CK(hipMemcpyToSymbol(HIP_SYMBOL(h3p_clk), z, 24));
      runh3p<TM, TN>(pl, ep, M, N, K, 20); CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(h3p_clk), 24));
      printf("   shader clock inside the k loop: %.0f MHz; mean loop time per workgroup %.2f us (%llu samples)\n", (double)h[0] / h[1] * 100.0, (double)h[1] / h[2] / 100.0, h[2]); }
#endif
    printf("   depth-1 planes+bias %7.1f us %6.1f TF | depth-2 without output stores %7.1f us %6.1f TF\n", t0 * 1e3, fl / t0 / 1e9, t4 * 1e3, fl / t4 / 1e9);
    printf(" P%3dx%-3d planes+bias %7.1f us %6.1f TF | +res %7.1f us %6.1f TF | +relu,split %7.1f us %6.1f TF | rerun diff %zu, vs on-the-fly split diff %zu\n", 64 * TM, 64 * TN,
           t1 * 1e3, fl / t1 / 1e9, t2 * 1e3, fl / t2 / 1e9, t3 * 1e3, fl / t3 / 1e9, nd, nx);
}

template <int TM, int TN, class GL, class EP, int NBUF = 3>
float runh3g(const GL& gl, const EP& ep, int M, int N, int K, int iters) {
    auto kern = gemm_h3g_kernel<TM, TN, GL, EP, NBUF>;
    size_t lds = gemm_h3g_lds_bytes(64 * TM, 64 * TN, NBUF);
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int mt = (M + 64 * TM - 1) / (64 * TM), nt = (N + 64 * TN - 1) / (64 * TN);
    int grid = ((mt + 7) / 8 * 8) * nt;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, gl, dBh, dBl, M, N, K, mt, nt, K / 32, ep);
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, gl, dBh, dBl, M, N, K, mt, nt, K / 32, ep);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}
template <int TM, int TN>
void h3g(int M, int N, int K) {   // call after h3p<> (planes of A and B are in place)
    GLoadPlain gl{dAh, dAl, K, M}; ALoadPlain ap{dA, K, M, K};
    EpBias ep{dC, dbias, N}; EpBiasResidual er{dC, dbias, N, nullptr, nullptr, 1}; EpBiasReluSplit es{dOh, dOl, dbias, N};
    std::vector<float> c1((size_t)M * N), c2((size_t)M * N), c3((size_t)M * N);
    size_t nd = 0, nx = 0;
    for (int rep = 0; rep < 3; ++rep) {
        runh3g<TM, TN>(gl, ep, M, N, K, 1); CK(hipMemcpy(c1.data(), dC, c1.size() * 4, hipMemcpyDeviceToHost));
        runh3g<TM, TN>(gl, ep, M, N, K, 1); CK(hipMemcpy(c2.data(), dC, c2.size() * 4, hipMemcpyDeviceToHost));
        if (rep == 0) { runh3<1, 1>(ap, ep, M, N, K, 1); CK(hipMemcpy(c3.data(), dC, c3.size() * 4, hipMemcpyDeviceToHost)); }
        for (size_t i = 0; i < c1.size(); ++i) { nd += (c1[i] != c2[i]); nx += (c1[i] != c3[i]); }
    }
    float t1 = runh3g<TM, TN>(gl, ep, M, N, K, 20), t2 = runh3g<TM, TN>(gl, er, M, N, K, 20), t3 = runh3g<TM, TN>(gl, es, M, N, K, 20);
    double fl = 2.0 * M * N * K;
    float u1 = runh3g<TM, TN, GLoadPlain, EpBias, 2>(gl, ep, M, N, K, 20), u2 = runh3g<TM, TN, GLoadPlain, EpBiasResidual, 2>(gl, er, M, N, K, 20), u3 = runh3g<TM, TN, GLoadPlain, EpBiasReluSplit, 2>(gl, es, M, N, K, 20);
    { EpSink sink{dC, N}; float v1 = runh3g<TM, TN>(gl, sink, M, N, K, 20); printf("   3 buffers, no output stores: %7.1f us %6.1f TF\n", v1 * 1e3, fl / v1 / 1e9); }
    printf("   2 LDS buffers:      %7.1f us %6.1f TF | +res %7.1f us %6.1f TF | +relu,split %7.1f us %6.1f TF\n", u1 * 1e3, fl / u1 / 1e9, u2 * 1e3, fl / u2 / 1e9, u3 * 1e3, fl / u3 / 1e9);
    printf(" G%3dx%-3d dma+bias    %7.1f us %6.1f TF | +res %7.1f us %6.1f TF | +relu,split %7.1f us %6.1f TF | rerun diff %zu, vs on-the-fly split diff %zu\n", 64 * TM, 64 * TN,
           t1 * 1e3, fl / t1 / 1e9, t2 * 1e3, fl / t2 / 1e9, t3 * 1e3, fl / t3 / 1e9, nd, nx);
}
void lnsplit_time(int M, int D) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(ln_split_kernel<2>, dim3((M + 3) / 4), dim3(256), 0, 0, dA, D, D, M, 1e-6f, dg, db, dAh, dAl, D);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(ln_split_kernel<2>, dim3((M + 3) / 4), dim3(256), 0, 0, dA, D, D, M, 1e-6f, dg, db, dAh, dAl, D);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    printf(" ln_split M=%d D=%d: %6.1f us  %6.2f TB/s (8 B per element)\n", M, D, ms * 1e3, 8.0 * M * D / ms / 1e9);
}
template <int TM, int TN>
void detall(int M, int N, int K) {
    {
        std::vector<_Float16> bh((size_t)N * K), bl((size_t)N * K);
        for (size_t i = 0; i < (size_t)N * K; ++i) { float x = hB[i]; _Float16 h = (_Float16)x; bh[i] = h; bl[i] = (_Float16)((x - (float)h) * 2048.0f); }
        CK(hipMemcpy(dBh, bh.data(), bh.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dBl, bl.data(), bl.size() * 2, hipMemcpyHostToDevice));
    }
    detcheck<TM, TN, ALoadLayerNorm>("ln", M, N, K);
    detcheck<TM, TN, ALoadLNConstStats>("conststats", M, N, K);
    detcheck<TM, TN, ALoadLNNoGB>("no-gb", M, N, K);
    detcheck<TM, TN, ALoadLNNoSel>("no-select", M, N, K);
}
template <int TM, int TN, class AL, class EP>
float runh3(const AL& al, const EP& ep, int M, int N, int K, int iters) {
    auto kern = gemm_h3_kernel<TM, TN, AL, EP>;
    size_t lds = gemm_h3_lds_bytes(64 * TM, 64 * TN);
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int mt = (M + 64 * TM - 1) / (64 * TM), nt = (N + 64 * TN - 1) / (64 * TN);
    int grid = ((mt + 7) / 8 * 8) * nt;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, al, dBh, dBl, M, N, K, mt, nt, K / 32, ep);
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, al, dBh, dBl, M, N, K, mt, nt, K / 32, ep);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}
template <int TM, int TN>
void h3(int M, int N, int K) {
    // planes for this (N, K): rows of hB viewed as [N][K]
    std::vector<_Float16> bh((size_t)N * K), bl((size_t)N * K);
    for (size_t i = 0; i < (size_t)N * K; ++i) { float x = hB[i]; _Float16 h = (_Float16)x; bh[i] = h; bl[i] = (_Float16)((x - (float)h) * 2048.0f); }
    CK(hipMemcpy(dBh, bh.data(), bh.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dBl, bl.data(), bl.size() * 2, hipMemcpyHostToDevice));
    ALoadPlain ap{dA, K, M, K}; ALoadLayerNorm aln{dA, dstats, dg, db, K, M, K};
    EpBias ep{dC, dbias, N}; EpBiasResidual er{dC, dbias, N, nullptr, nullptr, 1};
    float t1 = runh3<TM, TN>(ap, ep, M, N, K, 20);
    // accuracy of plain+bias against float64 on a few rows (bias is zero)
    std::vector<float> c((size_t)4 * N);
    CK(hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    for (int r = 0; r < 4; ++r) for (int n = 0; n < N; ++n) {
        double ref = 0; for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)r * K + k] * (double)hB[(size_t)n * K + k];
        maxerr = std::fmax(maxerr, std::fabs(ref - c[(size_t)r * N + n])); maxref = std::fmax(maxref, std::fabs(ref));
    }
    {   // determinism + full comparison against the f32 kernel
        std::vector<float> c1((size_t)M * N), c2((size_t)M * N), c3((size_t)M * N);
        runh3<TM, TN>(ap, ep, M, N, K, 1); CK(hipMemcpy(c1.data(), dC, c1.size() * 4, hipMemcpyDeviceToHost));
        runh3<TM, TN>(ap, ep, M, N, K, 1); CK(hipMemcpy(c2.data(), dC, c2.size() * 4, hipMemcpyDeviceToHost));
        run<64, 64>(ap, ep, M, N, K, 1); CK(hipMemcpy(c3.data(), dC, c3.size() * 4, hipMemcpyDeviceToHost));
        size_t nd = 0; double md = 0, mf = 0;
        for (size_t i = 0; i < c1.size(); ++i) { if (c1[i] != c2[i]) ++nd; md = std::fmax(md, std::fabs((double)c1[i] - c2[i])); mf = std::fmax(mf, std::fabs((double)c1[i] - c3[i])); }
        printf("      rerun mismatches %zu (max %.2e), max |h3 - f32| %.2e\n", nd, md, mf);
        runh3<TM, TN>(aln, ep, M, N, K, 1); CK(hipMemcpy(c1.data(), dC, c1.size() * 4, hipMemcpyDeviceToHost));
        runh3<TM, TN>(aln, ep, M, N, K, 1); CK(hipMemcpy(c2.data(), dC, c2.size() * 4, hipMemcpyDeviceToHost));
        run<64, 64>(aln, ep, M, N, K, 1); CK(hipMemcpy(c3.data(), dC, c3.size() * 4, hipMemcpyDeviceToHost));
        nd = 0; md = 0; mf = 0;
        for (size_t i = 0; i < c1.size(); ++i) { if (c1[i] != c2[i]) ++nd; md = std::fmax(md, std::fabs((double)c1[i] - c2[i])); mf = std::fmax(mf, std::fabs((double)c1[i] - c3[i])); }
        printf("      LN: rerun mismatches %zu (max %.2e), max |h3 - f32| %.2e\n", nd, md, mf);
        if (getenv("DUMP") && nd) {
            int shown = 0;
            for (int r = 0; r < M && shown < 40; ++r) {
                int cnt1 = 0, cnt2 = 0, first = -1, last = -1;
                for (int n = 0; n < N; ++n) {
                    const bool b1 = std::fabs((double)c1[(size_t)r * N + n] - c3[(size_t)r * N + n]) > 1e-3;
                    const bool b2 = std::fabs((double)c2[(size_t)r * N + n] - c3[(size_t)r * N + n]) > 1e-3;
                    cnt1 += b1; cnt2 += b2; if (b1 || b2) { if (first < 0) first = n; last = n; }
                }
                if (cnt1 || cnt2) { printf("        row %5d (tile %3d, in-tile %3d): run1 bad %4d run2 bad %4d cols [%d..%d]\n", r, r / (64 * TM), r % (64 * TM), cnt1, cnt2, first, last); ++shown; }
            }
        }
    }
    float t2 = runh3<TM, TN>(aln, ep, M, N, K, 20);
    float t3 = runh3<TM, TN>(ap, er, M, N, K, 20);
    double fl = 2.0 * M * N * K;
    printf(" H%3dx%-3d plain+bias %7.1f us %6.1f TF | ln+bias %7.1f us %6.1f TF | plain+res %7.1f us %6.1f TF | rel err %.1e\n", 64 * TM, 64 * TN,
           t1 * 1e3, fl / t1 / 1e9, t2 * 1e3, fl / t2 / 1e9, t3 * 1e3, fl / t3 / 1e9, maxerr / maxref);
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
    std::vector<float>& h = hA; h.resize((size_t)Mmax * Kmax);
    srand(1); for (auto& v : h) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    CK(hipMalloc(&dA, h.size() * 4)); CK(hipMemcpy(dA, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dB, (size_t)Nmax * Kmax * 4)); CK(hipMemcpy(dB, h.data(), (size_t)Nmax * Kmax * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dC, (size_t)Mmax * Nmax * 4)); CK(hipMemset(dC, 0, (size_t)Mmax * Nmax * 4));
    CK(hipMalloc(&dbias, Nmax * 4)); CK(hipMemset(dbias, 0, Nmax * 4));
    CK(hipMalloc(&dg, Kmax * 4)); CK(hipMemcpy(dg, h.data(), Kmax * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&db, Kmax * 4)); CK(hipMemcpy(db, h.data() + 5000, Kmax * 4, hipMemcpyHostToDevice));
    hB.assign(hA.begin(), hA.begin() + (size_t)Nmax * Kmax);
    CK(hipMalloc(&dBh, (size_t)Nmax * Kmax * 2)); CK(hipMalloc(&dBl, (size_t)Nmax * Kmax * 2));
    CK(hipMalloc(&dticket, 4096 * 4));
    CK(hipMalloc(&dAh, (size_t)Mmax * Kmax * 2)); CK(hipMalloc(&dAl, (size_t)Mmax * Kmax * 2));
    CK(hipMalloc(&dOh, (size_t)Mmax * Nmax * 2)); CK(hipMalloc(&dOl, (size_t)Mmax * Nmax * 2));
    CK(hipMalloc(&dstats, Mmax * 8)); CK(hipMemcpy(dstats, h.data(), Mmax * 8, hipMemcpyHostToDevice));
    int shapes[][3] = {{9088, 1152, 384}, {9088, 768, 384}, {9088, 384, 768}, {9088, 384, 384}, {9088, 384, 544},
                       {2944, 1152, 384}, {2944, 384, 2304}, {384, 384, 2304}, {128, 384, 2304}, {10496, 1152, 384}};
    const char* only = getenv("SHAPES"); int nshape = only ? atoi(only) : 100; int si = 0;
    if (const char* one = getenv("SHAPE")) { sscanf(one, "%d,%d,%d", &shapes[0][0], &shapes[0][1], &shapes[0][2]); nshape = 1; }
    for (auto& s : shapes) {
        if (si++ >= nshape) break;
        printf("M=%d N=%d K=%d\n", s[0], s[1], s[2]);
        if (getenv("DET")) { detall<1, 1>(s[0], s[1], s[2]); detall<2, 1>(s[0], s[1], s[2]); detall<1, 2>(s[0], s[1], s[2]); detall<2, 2>(s[0], s[1], s[2]); continue; }
        both<64, 64>(s[0], s[1], s[2]);
        h3<1, 1>(s[0], s[1], s[2]); h3<2, 1>(s[0], s[1], s[2]); h3<1, 2>(s[0], s[1], s[2]); h3<2, 2>(s[0], s[1], s[2]);
        h3p<1, 1>(s[0], s[1], s[2]); h3p<2, 1>(s[0], s[1], s[2]); h3p<1, 2>(s[0], s[1], s[2]); h3p<2, 2>(s[0], s[1], s[2]);
        h3g<1, 1>(s[0], s[1], s[2]); h3g<2, 1>(s[0], s[1], s[2]); h3g<1, 2>(s[0], s[1], s[2]); h3g<2, 2>(s[0], s[1], s[2]);
        if (s[2] == 384) lnsplit_time(s[0], s[2]);
        if (getenv("NOP")) continue;
        bothp<2, 2>(s[0], s[1], s[2], 1024);
        bothp<1, 2>(s[0], s[1], s[2], 1024);
        bothp<2, 1>(s[0], s[1], s[2], 1024);
        bothp<1, 1>(s[0], s[1], s[2], 1024);
        bothp<2, 2>(s[0], s[1], s[2], 768);
    }
    return 0;
}
