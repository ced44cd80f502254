// Round 4: the 8-wave row-panel GEMM (csrc/uu3d_gemm_panel8.h: contraction split over a pair of waves, two waves per SIMD)
// against the 4-wave kernel it replaces (csrc/uu3d_gemm_panel.h), with leave-one-out timing builds of both.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops [-DUU3D_PANEL_LOO=n] [-DUU3D_P8_LOO=n] -o tools/panel8_exp tools/panel8_exp.hip
//   tools/panel8_exp [M]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
#include <algorithm>
#include "../uplift-upsample-3dhpe_amd/csrc/uu3d_gemm.h"
#include "../uplift-upsample-3dhpe_amd/csrc/uu3d_gemm_h3.h"
#include "../uplift-upsample-3dhpe_amd/csrc/uu3d_gemm_panel.h"
#include "../uplift-upsample-3dhpe_amd/csrc/uu3d_gemm_panel8.h"
using namespace uu3d;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <class F> float timeit(F launch, int iters = 50) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) launch();
    CK(hipEventRecord(e0)); for (int i = 0; i < iters; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / iters;
}

struct Problem {
    int M, N, K = 384;
    std::vector<float> X, W, g, b, bias;
    float *dX, *dg, *db, *dbias; _Float16 *dAf, *dBf;
    Problem(int M_, int N_) : M(M_), N(N_) {
        std::mt19937 rng(1 + N); std::normal_distribution<float> nd(0.f, 1.f);
        X.resize((size_t)M * K); W.resize((size_t)N * K); g.resize(K); b.resize(K); bias.resize(N);
        for (int r = 0; r < M; ++r) { const float off = nd(rng), sc = 0.5f + fabsf(nd(rng)); for (int k = 0; k < K; ++k) X[(size_t)r * K + k] = off + sc * nd(rng); }
        for (auto& v : W) v = 0.05f * nd(rng);
        for (int k = 0; k < K; ++k) { g[k] = 1.f + 0.1f * nd(rng); b[k] = 0.1f * nd(rng); }
        for (auto& v : bias) v = 0.1f * nd(rng);
        std::vector<_Float16> Bh((size_t)N * K), Bl((size_t)N * K), Bf(panel_b_halfs(N, K));
        for (size_t i = 0; i < W.size(); ++i) { const _Float16 h = h3_hi(W[i]); Bh[i] = h; Bl[i] = (_Float16)((W[i] - (float)h) * H3_SCALE); }
        panel_pack_operand(Bh.data(), Bl.data(), N, K, K, Bf.data());
        CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dg, K * 4)); CK(hipMalloc(&db, K * 4)); CK(hipMalloc(&dbias, N * 4));
        CK(hipMalloc(&dBf, Bf.size() * 2)); CK(hipMalloc(&dAf, panel_a_halfs(M, K) * 2));
        CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dg, g.data(), K * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, b.data(), K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dbias, bias.data(), N * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dBf, Bf.data(), Bf.size() * 2, hipMemcpyHostToDevice));
        hipLaunchKernelGGL((ln_split_frag_kernel<24, 8>), dim3((M + 7) / 8), dim3(128), 0, 0, dX, K, M, 1e-5f, dg, db, dAf);
        CK(hipDeviceSynchronize());
    }
    // C = LN(X) W^T + bias (+ R) checked in double on a sample of rows
    void check(const float* dOut, const float* resid, const char* tag) const {
        std::vector<float> C((size_t)M * N); CK(hipMemcpy(C.data(), dOut, C.size() * 4, hipMemcpyDeviceToHost));
        double maxerr = 0; size_t bad = 0, nan = 0;
        for (int r = 0; r < M; r += (r < 300 || r > M - 300) ? 1 : 37) {
            double s = 0, v = 0; for (int k = 0; k < K; ++k) s += X[(size_t)r * K + k]; const double mean = s / K;
            for (int k = 0; k < K; ++k) { const double d = X[(size_t)r * K + k] - mean; v += d * d; } const double rstd = 1.0 / sqrt(v / K + 1e-5);
            std::vector<double> y(K); for (int k = 0; k < K; ++k) y[k] = (X[(size_t)r * K + k] - mean) * rstd * g[k] + b[k];
            for (int n = 0; n < N; ++n) { double acc = bias[n] + (resid ? resid[(size_t)r * N + n] : 0.0); for (int k = 0; k < K; ++k) acc += y[k] * W[(size_t)n * K + k];
                const double e = fabs(acc - C[(size_t)r * N + n]); if (e != e) ++nan; if (!(e < 1e-4)) ++bad; if (e > maxerr) maxerr = e; }
        }
        printf("    %-28s max |err| vs float64 %.3e, over 1e-4: %zu, NaN: %zu\n", tag, maxerr, bad, nan);
    }
};

template <class EP, int CPW, int VAR = 3>
static float run8(const Problem& p, int S, const EP& ep, int iters = 50) {
    auto kern = gemm_h3_panel8_kernel<EP, CPW, VAR>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P8_LDS_TOTAL));
    const int mt = (p.M + 127) / 128;
    return timeit([&] { hipLaunchKernelGGL(kern, dim3(8 * S, ((mt * S + 7) / 8 + S - 1) / S), dim3(512), P8_LDS_TOTAL, 0, p.dAf, p.dBf, p.dbias, p.M, mt, S, ep); }, iters);
}
template <class EP, int CPW = 0>
static float run4(const Problem& p, int S, const EP& ep, int iters = 50) {
    auto kern = gemm_h3_panel_kernel<24, EP, CPW>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PANEL_LDS_TOTAL));
    const int mt = (p.M + 127) / 128, cpw = (p.N / 32) / S;
    return timeit([&] { hipLaunchKernelGGL(kern, dim3(8 * S, ((mt * S + 7) / 8 + S - 1) / S), dim3(256), PANEL_LDS_TOTAL, 0, p.dAf, p.dBf, p.dbias, p.M, mt, S, cpw, ep); }, iters);
}

#ifdef UU3D_PANEL_STAMP
#include <algorithm>
__global__ void __launch_bounds__(256) empty_kernel(int* p) { if (p != nullptr && threadIdx.x == 12345) *p = 1; }
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) empty512_kernel(int* p) { if (p != nullptr && threadIdx.x == 12345) *p = 1; }
template <class L> static void single_vs_stream(const char* what, L launch) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) launch();
    float one = 0; for (int i = 0; i < 10; ++i) { CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); one += ms / 10; }
    CK(hipEventRecord(e0)); for (int i = 0; i < 100; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  %-40s one launch between events %6.2f us; 100 back to back %6.2f us each\n", what, one * 1e3, ms * 10);
}
// one launch (after warm-ups); the work items' stamps relative to the earliest entry, in microseconds
template <class L> static void stamp_report(const char* what, int items, L launch) {
    for (int i = 0; i < 5; ++i) launch();
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> z(1024 * 8, 0), h(1024 * 8);
    CK(hipMemcpyToSymbol(HIP_SYMBOL(panel_stamps), z.data(), z.size() * 8));
    launch(); CK(hipDeviceSynchronize());
    CK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(panel_stamps), h.size() * 8));
    unsigned long long t0 = ~0ull; for (int u = 0; u < items; ++u) if (h[u * 8]) t0 = std::min(t0, h[u * 8]);
    const char* names[5] = {"entry", "requests issued", "first k-step landed", "loop done", "end"};
    printf("  %s: stamps of %d work items, us after the first entry (min / median / max)\n", what, items);
    for (int k = 0; k < 5; ++k) { std::vector<double> v; for (int u = 0; u < items; ++u) if (h[u * 8]) v.push_back((double)(h[u * 8 + k] - t0) / 100.0);
        std::sort(v.begin(), v.end()); printf("    %-22s %6.2f %6.2f %6.2f\n", names[k], v.front(), v[v.size() / 2], v.back()); }
    { std::vector<double> v, w; for (int u = 0; u < items; ++u) if (h[u * 8]) { v.push_back((double)h[u * 8 + 5]); w.push_back((double)(h[u * 8 + 3] - h[u * 8 + 2]) / 100.0); }
      std::sort(v.begin(), v.end()); std::sort(w.begin(), w.end());
      printf("    loop: %.0f shader cycles in %.2f us (median) = %.2f GHz\n", v[v.size() / 2], w[w.size() / 2], v[v.size() / 2] / w[w.size() / 2] / 1000.0); }
}
#endif

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 9088;
#ifdef UU3D_PANEL_STAMP
    {
        Problem p(M, 1152);
        float* dC; CK(hipMalloc(&dC, (size_t)M * 1152 * 4));
        const int mt = (M + 127) / 128, S = 3;
        float te = timeit([&] { hipLaunchKernelGGL(empty_kernel, dim3(8 * S, ((mt * S + 7) / 8 + S - 1) / S), dim3(256), PANEL_LDS_TOTAL, 0, (int*)nullptr); }, 200);
        printf("empty kernel, same grid and LDS, back to back: %.2f us per launch\n", te * 1e3);
        auto k4 = gemm_h3_panel_kernel<24, PanelEpBias, 0>; CK(hipFuncSetAttribute((const void*)k4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PANEL_LDS_TOTAL));
        auto k8 = gemm_h3_panel8_kernel<PanelEpBias, 12>; CK(hipFuncSetAttribute((const void*)k8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P8_LDS_TOTAL));
        PanelEpBias ep{dC, 1152};
        const dim3 grid(8 * S, ((mt * S + 7) / 8 + S - 1) / S);
        single_vs_stream("empty 256 threads, 148 KiB LDS", [&] { hipLaunchKernelGGL(empty_kernel, grid, dim3(256), PANEL_LDS_TOTAL, 0, (int*)nullptr); });
        CK(hipFuncSetAttribute((const void*)empty512_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P8_LDS_TOTAL));
        single_vs_stream("empty 512 threads, 160 KiB LDS", [&] { hipLaunchKernelGGL(empty512_kernel, grid, dim3(512), P8_LDS_TOTAL, 0, (int*)nullptr); });
        single_vs_stream("empty 512 threads, 0 LDS", [&] { hipLaunchKernelGGL(empty512_kernel, grid, dim3(512), 0, 0, (int*)nullptr); });
        single_vs_stream("4-wave QKV", [&] { hipLaunchKernelGGL(k4, grid, dim3(256), PANEL_LDS_TOTAL, 0, p.dAf, p.dBf, p.dbias, p.M, mt, S, 12, ep); });
        single_vs_stream("8-wave QKV", [&] { hipLaunchKernelGGL(k8, grid, dim3(512), P8_LDS_TOTAL, 0, p.dAf, p.dBf, p.dbias, p.M, mt, S, ep); });
        stamp_report("4-wave QKV", mt * S, [&] { hipLaunchKernelGGL(k4, dim3(8 * S, ((mt * S + 7) / 8 + S - 1) / S), dim3(256), PANEL_LDS_TOTAL, 0, p.dAf, p.dBf, p.dbias, p.M, mt, S, 12, ep); });
        stamp_report("8-wave QKV", mt * S, [&] { hipLaunchKernelGGL(k8, dim3(8 * S, ((mt * S + 7) / 8 + S - 1) / S), dim3(512), P8_LDS_TOTAL, 0, p.dAf, p.dBf, p.dbias, p.M, mt, S, ep); });
        { auto k = gemm_h3_panel8_kernel<PanelEpBias, 12, 1>; CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P8_LDS_TOTAL));
          stamp_report("8-wave + sched barriers", mt * S, [&] { hipLaunchKernelGGL(k, grid, dim3(512), P8_LDS_TOTAL, 0, p.dAf, p.dBf, p.dbias, p.M, mt, S, ep); }); }
        { auto k = gemm_h3_panel8_kernel<PanelEpBias, 12, 2>; CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P8_LDS_TOTAL));
          stamp_report("8-wave + A streaming", mt * S, [&] { hipLaunchKernelGGL(k, grid, dim3(512), P8_LDS_TOTAL, 0, p.dAf, p.dBf, p.dbias, p.M, mt, S, ep); }); }
        { auto k = gemm_h3_panel8_kernel<PanelEpBias, 12, 3>; CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P8_LDS_TOTAL));
          stamp_report("8-wave + both", mt * S, [&] { hipLaunchKernelGGL(k, grid, dim3(512), P8_LDS_TOTAL, 0, p.dAf, p.dBf, p.dbias, p.M, mt, S, ep); }); }
        CK(hipFree(dC));
        return 0;
    }
#endif
    printf("panel8_exp: M = %d, UU3D_PANEL_LOO = %d, UU3D_P8_LOO = %d\n", M, UU3D_PANEL_LOO, UU3D_P8_LOO);
    {   // QKV shape: N = 1152, S = 3 -> 12 chunks per workgroup
        Problem p(M, 1152);
        const double fl = 2.0 * M * 1152.0 * 384;
        float *dC, *dC8; CK(hipMalloc(&dC, (size_t)M * 1152 * 4)); CK(hipMalloc(&dC8, (size_t)M * 1152 * 4));
        CK(hipMemset(dC, 0xff, (size_t)M * 1152 * 4)); CK(hipMemset(dC8, 0xff, (size_t)M * 1152 * 4));
        float t4 = run4(p, 3, PanelEpBias{dC, 1152});
        float t8 = run8<PanelEpBias, 12>(p, 3, PanelEpBias{dC8, 1152});
        printf("N 1152 (QKV), f32 epilogue:   4-wave %6.1f us (%5.1f TFLOP/s)   8-wave %6.1f us (%5.1f TFLOP/s)\n", t4 * 1e3, fl / t4 / 1e9, t8 * 1e3, fl / t8 / 1e9);
        p.check(dC, nullptr, "4-wave"); p.check(dC8, nullptr, "8-wave");
        {   // bitwise: the 8-wave sum order differs (two half sums), report the largest difference
            std::vector<float> a((size_t)M * 1152), b((size_t)M * 1152); CK(hipMemcpy(a.data(), dC, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), dC8, b.size() * 4, hipMemcpyDeviceToHost));
            double md = 0; for (size_t i = 0; i < a.size(); ++i) md = fmax(md, fabs((double)a[i] - b[i])); printf("    max |4-wave - 8-wave| = %.3e\n", md);
        }
        _Float16* dQ; CK(hipMalloc(&dQ, (size_t)M * 1152 * 4));
        PanelEpBiasSplitQ eq{dQ, dQ + (size_t)M * 1152, 1152, 384, 0.2f};
        float q4 = run4(p, 3, eq), q8 = run8<PanelEpBiasSplitQ, 12>(p, 3, eq);
        printf("N 1152 (QKV), split-plane epilogue: 4-wave %6.1f us   8-wave %6.1f us\n", q4 * 1e3, q8 * 1e3);
        float q4b = run4(p, 2, eq);
        printf("N 1152 (QKV), split-plane epilogue, S = 2 (142 workgroups): 4-wave %6.1f us\n", q4b * 1e3);
        // run-to-run determinism of the 8-wave kernel: 20 launches, bitwise
        {
            std::vector<float> a((size_t)M * 1152), b((size_t)M * 1152); size_t diff = 0;
            run8<PanelEpBias, 12>(p, 3, PanelEpBias{dC8, 1152}, 1); CK(hipMemcpy(a.data(), dC8, a.size() * 4, hipMemcpyDeviceToHost));
            for (int it = 0; it < 20; ++it) { CK(hipMemset(dC8, 0xff, (size_t)M * 1152 * 4)); run8<PanelEpBias, 12>(p, 3, PanelEpBias{dC8, 1152}, 1);
                CK(hipMemcpy(b.data(), dC8, b.size() * 4, hipMemcpyDeviceToHost)); for (size_t i = 0; i < a.size(); ++i) diff += (a[i] != b[i]) && !(a[i] != a[i] && b[i] != b[i]); }
            printf("    8-wave determinism: %zu differing values in 20 x %zu\n", diff, a.size());
        }
        CK(hipFree(dC)); CK(hipFree(dC8)); CK(hipFree(dQ));
    }
    {   // fc1 shape (training forward / non-fused path): N = 768, S = 3 -> 8 chunks
        Problem p(M, 768);
        const double fl = 2.0 * M * 768.0 * 384;
        float *dC, *dC8; CK(hipMalloc(&dC, (size_t)M * 768 * 4)); CK(hipMalloc(&dC8, (size_t)M * 768 * 4));
        float t4 = run4(p, 3, PanelEpBias{dC, 768}), t8 = run8<PanelEpBias, 8>(p, 3, PanelEpBias{dC8, 768});
        printf("N  768 (fc1), f32 epilogue:   4-wave %6.1f us (%5.1f TFLOP/s)   8-wave %6.1f us (%5.1f TFLOP/s)\n", t4 * 1e3, fl / t4 / 1e9, t8 * 1e3, fl / t8 / 1e9);
        p.check(dC, nullptr, "4-wave"); p.check(dC8, nullptr, "8-wave");
        CK(hipFree(dC)); CK(hipFree(dC8));
    }
    {   // projection shape: N = 384, x += A W + b in place; S = 3 / 1 -> 4 / 12 chunks
        Problem p(M, 384);
        const double fl = 2.0 * M * 384.0 * 384;
        std::vector<float> R((size_t)M * 384); std::mt19937 rng(7); std::normal_distribution<float> nd(0.f, 1.f); for (auto& v : R) v = nd(rng);
        float *dx; CK(hipMalloc(&dx, R.size() * 4));
        CK(hipMemcpy(dx, R.data(), R.size() * 4, hipMemcpyHostToDevice));
        run4<PanelEpBiasResidual, 4>(p, 3, PanelEpBiasResidual{dx, 384}, 1);   // (timeit warms up 5 times: 6 additions)
        {   std::vector<float> six(R.size()); std::vector<float> C(R.size()); CK(hipMemcpy(C.data(), dx, C.size() * 4, hipMemcpyDeviceToHost));
            // x_6 = R + 6 y  =>  check (x_6 - R) / 6 + R against y + R
            for (size_t i = 0; i < C.size(); ++i) C[i] = (C[i] - R[i]) / 6.f + R[i];
            float* dT; CK(hipMalloc(&dT, C.size() * 4)); CK(hipMemcpy(dT, C.data(), C.size() * 4, hipMemcpyHostToDevice)); p.check(dT, R.data(), "4-wave residual (x6 / 6)"); CK(hipFree(dT)); }
        CK(hipMemcpy(dx, R.data(), R.size() * 4, hipMemcpyHostToDevice));
        run8<PanelEpBiasResidual, 4>(p, 3, PanelEpBiasResidual{dx, 384}, 1);
        {   std::vector<float> C(R.size()); CK(hipMemcpy(C.data(), dx, C.size() * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < C.size(); ++i) C[i] = (C[i] - R[i]) / 6.f + R[i];
            float* dT; CK(hipMalloc(&dT, C.size() * 4)); CK(hipMemcpy(dT, C.data(), C.size() * 4, hipMemcpyHostToDevice)); p.check(dT, R.data(), "8-wave residual (x6 / 6)"); CK(hipFree(dT)); }
        CK(hipMemset(dx, 0, R.size() * 4));
        float t4 = run4<PanelEpBiasResidual, 4>(p, 3, PanelEpBiasResidual{dx, 384});
        CK(hipMemset(dx, 0, R.size() * 4));
        float t8 = run8<PanelEpBiasResidual, 4>(p, 3, PanelEpBiasResidual{dx, 384});
        CK(hipMemset(dx, 0, R.size() * 4));
        float t4s1 = run4<PanelEpBiasResidual, 12>(p, 1, PanelEpBiasResidual{dx, 384});
        CK(hipMemset(dx, 0, R.size() * 4));
        float t8s1 = run8<PanelEpBiasResidual, 12>(p, 1, PanelEpBiasResidual{dx, 384});
        printf("N  384 (projection + residual): S = 3: 4-wave %6.1f us  8-wave %6.1f us (%5.1f TFLOP/s);  S = 1: 4-wave %6.1f us  8-wave %6.1f us\n",
               t4 * 1e3, t8 * 1e3, fl / t8 / 1e9, t4s1 * 1e3, t8s1 * 1e3);
        CK(hipFree(dx));
    }
    for (int Mr : {M - 40, M - 100, 1000}) {   // ragged last row tile (predicated stores, strict waits), every chunk count the library instantiates
        Problem p(Mr, 1152);
        float* dC; CK(hipMalloc(&dC, (size_t)Mr * 1152 * 4 + 4096));
        printf("ragged M = %d (N 1152):\n", Mr);
        CK(hipMemset(dC, 0xff, (size_t)Mr * 1152 * 4 + 4096)); run8<PanelEpBias, 12>(p, 3, PanelEpBias{dC, 1152}, 1); p.check(dC, nullptr, "8-wave, 12 chunks (S = 3)");
        CK(hipMemset(dC, 0xff, (size_t)Mr * 1152 * 4 + 4096)); run8<PanelEpBias, 6>(p, 6, PanelEpBias{dC, 1152}, 1); p.check(dC, nullptr, "8-wave, 6 chunks (S = 6)");
        CK(hipMemset(dC, 0xff, (size_t)Mr * 1152 * 4 + 4096)); run8<PanelEpBias, 4>(p, 9, PanelEpBias{dC, 1152}, 1); p.check(dC, nullptr, "8-wave, 4 chunks (S = 9)");
        { unsigned tail[1024]; CK(hipMemcpy(tail, dC + (size_t)Mr * 1152, 4096, hipMemcpyDeviceToHost)); int touched = 0; for (unsigned v : tail) touched += v != 0xffffffffu; printf("    words written past the last row: %d\n", touched); }
        Problem p2(Mr, 768);
        CK(hipMemset(dC, 0xff, (size_t)Mr * 768 * 4)); run8<PanelEpBias, 8>(p2, 3, PanelEpBias{dC, 768}, 1); p2.check(dC, nullptr, "8-wave, 8 chunks (N 768)");
        Problem p3(Mr, 384);
        std::vector<float> R((size_t)Mr * 384); std::mt19937 rng(9); std::normal_distribution<float> nd(0.f, 1.f); for (auto& v : R) v = nd(rng);
        CK(hipMemcpy(dC, R.data(), R.size() * 4, hipMemcpyHostToDevice));
        { auto kern = gemm_h3_panel8_kernel<PanelEpBiasResidual, 12, 3>; CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P8_LDS_TOTAL));
          const int mt = (Mr + 127) / 128; hipLaunchKernelGGL(kern, dim3(8, (mt + 7) / 8), dim3(512), P8_LDS_TOTAL, 0, p3.dAf, p3.dBf, p3.dbias, Mr, mt, 1, PanelEpBiasResidual{dC, 384}); CK(hipDeviceSynchronize()); }
        p3.check(dC, R.data(), "8-wave residual, 12 chunks (S = 1)");
        CK(hipFree(dC));
    }
    {   // alternating A/B of the variants on the QKV launch with its real (split-plane) epilogue: 12 rounds x 20 launches each, medians
        Problem p(M, 1152);
        _Float16* dQ; CK(hipMalloc(&dQ, (size_t)M * 1152 * 4));
        PanelEpBiasSplitQ eq{dQ, dQ + (size_t)M * 1152, 1152, 384, 0.2f};
        std::vector<float> t[5];
        for (int round = 0; round < 12; ++round) {
            t[0].push_back(run4(p, 3, eq, 20));
            t[1].push_back(run8<PanelEpBiasSplitQ, 12, 0>(p, 3, eq, 20));
            t[2].push_back(run8<PanelEpBiasSplitQ, 12, 1>(p, 3, eq, 20));
            t[3].push_back(run8<PanelEpBiasSplitQ, 12, 2>(p, 3, eq, 20));
            t[4].push_back(run8<PanelEpBiasSplitQ, 12, 3>(p, 3, eq, 20));
        }
        const char* nm[5] = {"4-wave", "8-wave", "8-wave + sched barriers", "8-wave + A streaming", "8-wave + both"};
        for (int i = 0; i < 5; ++i) { std::sort(t[i].begin(), t[i].end()); printf("A/B QKV split-plane epilogue: %-26s median %6.2f us (min %6.2f, max %6.2f)\n", nm[i], t[i][6] * 1e3, t[i][0] * 1e3, t[i][11] * 1e3); }
        float* dC; CK(hipMalloc(&dC, (size_t)M * 1152 * 4)); CK(hipMemset(dC, 0xff, (size_t)M * 1152 * 4));
        run8<PanelEpBias, 12, 3>(p, 3, PanelEpBias{dC, 1152}, 1); p.check(dC, nullptr, "8-wave + both");
        CK(hipMemset(dC, 0xff, (size_t)M * 1152 * 4));
        run8<PanelEpBias, 12, 2>(p, 3, PanelEpBias{dC, 1152}, 1); p.check(dC, nullptr, "8-wave + A streaming");
        CK(hipFree(dQ)); CK(hipFree(dC));
    }
    return 0;
}
