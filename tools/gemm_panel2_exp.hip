// Standalone check + timing of the K-split row-panel GEMM experiment (tools/r02_variants/uu3d_gemm_panel2.h) against
// gemm_h3_panel_kernel; -DUU3D_PANEL_PROBE_DOUBLE: the product kernel with every MFMA issued twice per fragment read.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops -o tools/gemm_panel2_exp tools/gemm_panel2_exp.hip
//   tools/gemm_panel2_exp [M] [N]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
#include "../uplift-upsample-3dhpe_amd/csrc/uu3d_gemm.h"
#include "../uplift-upsample-3dhpe_amd/csrc/uu3d_gemm_h3.h"
#include "r02_variants/uu3d_gemm_panel2.h"
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
    std::vector<float> X((size_t)M * K), W((size_t)N * K), g(K, 1.f), b(K, 0.f), bias(N);
    for (auto& v : X) v = nd(rng);
    for (auto& v : W) v = 0.05f * nd(rng);
    for (auto& v : bias) v = 0.1f * nd(rng);
    std::vector<_Float16> Bh((size_t)N * K), Bl((size_t)N * K), Bf(panel_b_halfs(N, K));
    for (size_t i = 0; i < W.size(); ++i) { const _Float16 h = h3_hi(W[i]); Bh[i] = h; Bl[i] = (_Float16)((W[i] - (float)h) * H3_SCALE); }
    panel_pack_operand(Bh.data(), Bl.data(), N, K, K, Bf.data());
    float *dX, *dg, *db, *dbias, *dC; _Float16 *dAf, *dBf;
    CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dg, K * 4)); CK(hipMalloc(&db, K * 4)); CK(hipMalloc(&dbias, N * 4));
    CK(hipMalloc(&dC, (size_t)M * N * 4)); CK(hipMalloc(&dBf, Bf.size() * 2)); CK(hipMalloc(&dAf, panel_a_halfs(M, K) * 2));
    CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dg, g.data(), K * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dbias, bias.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dBf, Bf.data(), Bf.size() * 2, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((ln_split_frag_kernel<24, 8>), dim3((M + 7) / 8), dim3(128), 0, 0, dX, K, M, 1e-5f, dg, db, dAf);
    const double fl = 2.0 * M * (double)N * K;
    auto check = [&](const char* tag) {
        std::vector<float> C((size_t)M * N); CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
        double maxerr = 0; size_t bad = 0;
        for (int r = 0; r < M; r += (r < 256 || r > M - 256) ? 1 : 37) {
            double s = 0, v = 0; for (int k = 0; k < K; ++k) s += X[(size_t)r * K + k]; const double mean = s / K;
            for (int k = 0; k < K; ++k) { const double d = X[(size_t)r * K + k] - mean; v += d * d; } const double rstd = 1.0 / sqrt(v / K + 1e-5);
            for (int n = 0; n < N; ++n) { double acc = bias[n]; for (int k = 0; k < K; ++k) acc += (X[(size_t)r * K + k] - mean) * rstd * W[(size_t)n * K + k];
                const double e = fabs(acc - C[(size_t)r * N + n]); if (!(e < 1e-4)) ++bad; if (e > maxerr || e != e) maxerr = e; }
        }
        printf("  %s: max |err| vs float64 %.3e, entries over 1e-4: %zu\n", tag, maxerr, bad);
    };
    PanelEpBias ep{dC, N};
    const int mt = (M + 127) / 128, nch = N / 32;
    auto k1 = gemm_h3_panel_kernel<24, PanelEpBias>;
    auto k2 = gemm_h3_panel2_kernel<PanelEpBias>;
    CK(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PANEL_LDS_TOTAL));
    CK(hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PANEL2_LDS_TOTAL));
    for (int S : {2, 3, 4, 6, 8, 9, 12}) {
        if (nch % S) continue;
        const int cpw = nch / S;
        const dim3 grid(8 * S, ((mt * S + 7) / 8 + S - 1) / S);
        CK(hipMemset(dC, 0xff, (size_t)M * N * 4));
        float ms = timeit([&] { hipLaunchKernelGGL(k1, grid, dim3(256), PANEL_LDS_TOTAL, 0, dAf, dBf, dbias, M, mt, S, cpw, ep); });
        printf("panel  S=%d (%4d workgroups, %2d chunks each): %7.1f us  %6.1f TFLOP/s algorithmic\n", S, mt * S, cpw, ms * 1e3, fl / ms / 1e9);
        check("one wave per SIMD");
        CK(hipMemset(dC, 0xff, (size_t)M * N * 4));
        ms = timeit([&] { hipLaunchKernelGGL(k2, grid, dim3(256), PANEL2_LDS_TOTAL, 0, dAf, dBf, dbias, M, mt, S, cpw, ep); });
        printf("panel2 S=%d (%4d workgroups, %2d chunks each): %7.1f us  %6.1f TFLOP/s algorithmic\n", S, mt * S, cpw, ms * 1e3, fl / ms / 1e9);
        check("two panels per wave, K split");
    }
    return 0;
}
