// Standalone check + timing of the row-panel f16x3 GEMM (csrc/uu3d_gemm_panel.h) against the tiled kernel it replaces.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops -o tools/gemm_panel_exp tools/gemm_panel_exp.hip
//   tools/gemm_panel_exp [M] [N]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
#include "../uplift-upsample-3dhpe_amd/csrc/uu3d_gemm.h"
#include "../uplift-upsample-3dhpe_amd/csrc/uu3d_gemm_h3.h"
#ifdef STAMP
#define UU3D_PANEL_STAMP 1
#endif
#include "../uplift-upsample-3dhpe_amd/csrc/uu3d_gemm_panel.h"
using namespace uu3d;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <class F> float timeit(F launch, int iters = 20) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(e0)); for (int i = 0; i < iters; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / iters;
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 9088, N = argc > 2 ? atoi(argv[2]) : 1152, K = 384;
    std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> X((size_t)M * K), W((size_t)N * K), g(K), b(K), bias(N);
    for (int r = 0; r < M; ++r) { const float off = nd(rng), sc = 0.5f + fabsf(nd(rng)); for (int k = 0; k < K; ++k) X[(size_t)r * K + k] = off + sc * nd(rng); }
    for (auto& v : W) v = 0.05f * nd(rng);
    for (int k = 0; k < K; ++k) { g[k] = 1.f + 0.1f * nd(rng); b[k] = 0.1f * nd(rng); }
    for (auto& v : bias) v = 0.1f * nd(rng);
    std::vector<_Float16> Bh((size_t)N * K), Bl((size_t)N * K), Bf(panel_b_halfs(N, K));
    for (size_t i = 0; i < W.size(); ++i) { const _Float16 h = h3_hi(W[i]); Bh[i] = h; Bl[i] = (_Float16)((W[i] - (float)h) * H3_SCALE); }
    panel_pack_operand(Bh.data(), Bl.data(), N, K, K, Bf.data());
    std::vector<float2> stats(M);
    for (int r = 0; r < M; ++r) { double s = 0, v = 0; for (int k = 0; k < K; ++k) s += X[(size_t)r * K + k]; const double mean = s / K;
        for (int k = 0; k < K; ++k) { const double d = X[(size_t)r * K + k] - mean; v += d * d; } stats[r] = make_float2((float)mean, (float)(1.0 / sqrt(v / K + 1e-5))); }

    float *dX, *dg, *db, *dbias, *dC, *dC2; _Float16* dAf; _Float16 *dBh, *dBl, *dBf; float2* dstats;
    CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dg, K * 4)); CK(hipMalloc(&db, K * 4)); CK(hipMalloc(&dbias, N * 4));
    CK(hipMalloc(&dC, (size_t)M * N * 4)); CK(hipMalloc(&dC2, (size_t)M * N * 4)); CK(hipMalloc(&dstats, M * 8));
    CK(hipMalloc(&dBh, Bh.size() * 2)); CK(hipMalloc(&dBl, Bl.size() * 2)); CK(hipMalloc(&dBf, Bf.size() * 2));
    CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dg, g.data(), K * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dbias, bias.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dBh, Bh.data(), Bh.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dBl, Bl.data(), Bl.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dBf, Bf.data(), Bf.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dstats, stats.data(), M * 8, hipMemcpyHostToDevice));
    CK(hipMemset(dC, 0xff, (size_t)M * N * 4));
    CK(hipMalloc(&dAf, panel_a_halfs(M, K) * 2));

    const double fl = 2.0 * M * (double)N * K;
    // reference rows in double
    auto check = [&](const float* dOut, const char* tag) {
        std::vector<float> C((size_t)M * N); CK(hipMemcpy(C.data(), dOut, C.size() * 4, hipMemcpyDeviceToHost));
        double maxerr = 0; size_t bad = 0;
        for (int r = 0; r < M; r += (r < 256 || r > M - 256) ? 1 : 37) {
            double s = 0, v = 0; for (int k = 0; k < K; ++k) s += X[(size_t)r * K + k]; const double mean = s / K;
            for (int k = 0; k < K; ++k) { const double d = X[(size_t)r * K + k] - mean; v += d * d; } const double rstd = 1.0 / sqrt(v / K + 1e-5);
            std::vector<double> y(K); for (int k = 0; k < K; ++k) y[k] = (X[(size_t)r * K + k] - mean) * rstd * g[k] + b[k];
            for (int n = 0; n < N; ++n) { double acc = bias[n]; for (int k = 0; k < K; ++k) acc += y[k] * W[(size_t)n * K + k];
                const double e = fabs(acc - C[(size_t)r * N + n]); if (!(e < 1e-4)) ++bad; if (e > maxerr || e != e) maxerr = e; }
        }
        printf("  %s: max |err| vs float64 %.3e, entries over 1e-4: %zu\n", tag, maxerr, bad);
    };

    // the tiled kernel of the product path (row statistics precomputed)
    {
        ALoadLayerNorm al{dX, dstats, dg, db, K, M, K}; EpBias ep{dC2, dbias, N};
        auto kern = gemm_h3_kernel<1, 2, ALoadLayerNorm, EpBias>;
        constexpr size_t lds = gemm_h3_lds_bytes(64, 128);
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int mt = (M + 63) / 64, nt = N / 128;
        float ms = timeit([&] { hipLaunchKernelGGL(kern, dim3((mt + 7) / 8 * 8 * nt), dim3(256), lds, 0, al, dBh, dBl, M, N, K, mt, nt, K / 32, ep); });
        printf("tiled 64x128 (LN loader, stats given): %7.1f us  %6.1f TFLOP/s algorithmic\n", ms * 1e3, fl / ms / 1e9);
        check(dC2, "tiled");
    }
    {
        float msl = timeit([&] { hipLaunchKernelGGL(ln_split_frag_kernel<24>, dim3((M + 15) / 16), dim3(256), 0, 0, dX, K, M, 1e-5f, dg, db, dAf); });
        float msl8 = timeit([&] { hipLaunchKernelGGL((ln_split_frag_kernel<24, 8>), dim3((M + 7) / 8), dim3(128), 0, 0, dX, K, M, 1e-5f, dg, db, dAf); });
        float msl4 = timeit([&] { hipLaunchKernelGGL((ln_split_frag_kernel<24, 4>), dim3((M + 3) / 4), dim3(64), 0, 0, dX, K, M, 1e-5f, dg, db, dAf); });
        float msl32 = timeit([&] { hipLaunchKernelGGL((ln_split_frag_kernel<24, 32>), dim3((M + 31) / 32), dim3(512), 0, 0, dX, K, M, 1e-5f, dg, db, dAf); });
        printf("ln_split_frag: %7.1f us (16 rows per workgroup); 8 rows %7.1f, 4 rows %7.1f, 32 rows %7.1f\n", msl * 1e3, msl8 * 1e3, msl4 * 1e3, msl32 * 1e3);
        PanelEpBias ep{dC, N};
        auto kern = gemm_h3_panel_kernel<24, PanelEpBias>;
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PANEL_LDS_TOTAL));
        const int mt = (M + 127) / 128, nch = N / 32;
        for (int S : {2, 3, 4, 6, 9, 12}) {
            if (nch % S) continue;
            const int cpw = nch / S;
            CK(hipMemset(dC, 0xff, (size_t)M * N * 4));
            float ms = timeit([&] { hipLaunchKernelGGL(kern, dim3(8 * S, ((mt * S + 7) / 8 + S - 1) / S), dim3(256), PANEL_LDS_TOTAL, 0, dAf, dBf, dbias, M, mt, S, cpw, ep, 0, 0.f); });
            float ms2 = timeit([&] { hipLaunchKernelGGL(ln_split_frag_kernel<24>, dim3((M + 15) / 16), dim3(256), 0, 0, dX, K, M, 1e-5f, dg, db, dAf);
                                     hipLaunchKernelGGL(kern, dim3(8 * S, ((mt * S + 7) / 8 + S - 1) / S), dim3(256), PANEL_LDS_TOTAL, 0, dAf, dBf, dbias, M, mt, S, cpw, ep, 0, 0.f); });
            printf("panel S=%2d (%4d workgroups, %2d chunks each): %7.1f us  %6.1f TFLOP/s algorithmic;  with ln_split_frag in front %7.1f us\n", S, mt * S, cpw, ms * 1e3, fl / ms / 1e9, ms2 * 1e3);
            check(dC, "panel");
#ifdef STAMP
            { unsigned long long z[8] = {0}, h[8]; CK(hipMemcpyToSymbol(HIP_SYMBOL(panel_clk), z, 64));
              hipLaunchKernelGGL(kern, dim3(8 * S, ((mt * S + 7) / 8 + S - 1) / S), dim3(256), PANEL_LDS_TOTAL, 0, dAf, dBf, dbias, M, mt, S, cpw, ep, 0, 0.f); CK(hipDeviceSynchronize());
              CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(panel_clk), 64)); const double n = (double)h[5];
              printf("  per workgroup (s_memtime ticks): prologue %.0f, loop %.0f (per k-step %.0f), tail %.0f\n", h[0] / n, h[1] / n, h[1] / n / (2 * cpw), h[4] / n); }
#endif
        }
    }
    return 0;
}
