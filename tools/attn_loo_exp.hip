// Round 4: where the f16x3 attention kernel's time goes (csrc/uu3d_attn_h3.h), by leave-one-out timing builds.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops [-DUU3D_ATTN_LOO=n] -o tools/attn_loo_exp[_n] tools/attn_loo_exp.hip
//   tools/attn_loo_exp            -> the two product launches: 71 tokens x batch 128 (3 waves, 4 workgroups per CU) and 351 tokens x batch 32 (11 waves)
// The knock-out builds compute nonsense; only their time is read.  The plain build checks the kernel against float64 on a sample first.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
#include <algorithm>
#include "../uplift-upsample-3dhpe_amd/csrc/uu3d_gemm.h"
#include "../uplift-upsample-3dhpe_amd/csrc/uu3d_attn_h3.h"
#include "attn_pp_exp.h"
using namespace uu3d;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <class F> static void timeit(const char* what, F launch, double mfma_per_launch, double bytes) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) launch();
    std::vector<float> t;
    for (int r = 0; r < 9; ++r) {
        CK(hipEventRecord(e0)); for (int i = 0; i < 20; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms * 50.f);
    }
    std::sort(t.begin(), t.end());
    printf("  %-34s LOO=%-3d  %6.2f us per launch (min %.2f, max %.2f of 9 x 20 back to back)   MFMA time at 33 cycles, 2.0 GHz, 1024 pipes: %.2f us   bytes: %.1f MB = %.2f us at 6 TB/s\n",
           what, (int)(UU3D_ATTN_LOO), t[4], t.front(), t.back(), mfma_per_launch * 33.0 / 2.0e3 / 1024.0, bytes / 1e6, bytes / 6e6);
}

static std::vector<_Float16> last_out;     // the previous run's output planes: attn_h3_pp_kernel must reproduce attn_h3_kernel bit for bit
template <int MW, int WPE, int PP = 0>
static void run(int B, int L, const char* what) {
    const int H = 8, D = 384, DH = 48;
    const size_t rows = (size_t)B * L;
    std::mt19937 rng(7); std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> x(rows * 3 * D);
    for (auto& v : x) v = nd(rng);
    const float qs = 1.44269504088896341f / sqrtf((float)DH);
    std::vector<_Float16> hi(x.size()), lo(x.size());
    for (size_t r = 0; r < rows; ++r) for (int c = 0; c < 3 * D; ++c) {
        const size_t i = r * 3 * D + c; const float v = c < D ? x[i] * qs : x[i];
        const _Float16 h = h3_hi(v); hi[i] = h; lo[i] = (_Float16)((v - (float)h) * H3_SCALE);
    }
    _Float16 *dh, *dl, *dout; CK(hipMalloc(&dh, hi.size() * 2)); CK(hipMalloc(&dl, lo.size() * 2));
    const size_t out_halfs = ((rows + 31) / 32) * 32 * (size_t)D * 2;
    CK(hipMalloc(&dout, out_halfs * 2)); CK(hipMemset(dout, 0, out_halfs * 2));
    CK(hipMemcpy(dh, hi.data(), hi.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dl, lo.data(), lo.size() * 2, hipMemcpyHostToDevice));
    auto k = PP == 8 ? attn_h3_pp_kernel<DH, false, 8> : PP == 4 ? attn_h3_pp_kernel<DH, false, 4> : attn_h3_kernel<DH, MW, WPE, false, false>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_h3_lds_bytes(ATTN_H3_MAX_L, DH)));
    const int nt = (L + 31) / 32, waves = PP ? PP : std::min(nt, MW);
    const size_t lds = attn_h3_lds_bytes(L, DH);
    // row-major output planes (frag = 0): hi plane, lo plane rows * D halfs further
    auto launch = [&] { hipLaunchKernelGGL(k, dim3(B * H), dim3(64 * waves), lds, 0, dh, dl, 3 * D, D, L, H, (const uint8_t*)nullptr, dout, rows * D, D, 0); };
    launch(); CK(hipDeviceSynchronize());
#if UU3D_ATTN_LOO == 0
    {   // a sample of (sequence, head, query) rows against float64 softmax(q k^T) v on the operands the kernel reads (hi + lo / 2048)
        std::vector<_Float16> o(out_halfs); CK(hipMemcpy(o.data(), dout, out_halfs * 2, hipMemcpyDeviceToHost));
        auto val = [&](size_t r, int c) { return (double)(float)hi[r * 3 * D + c] + (double)(float)lo[r * 3 * D + c] / 2048.0; };
        double maxerr = 0;
        for (int b : {0, B / 2, B - 1}) for (int h : {0, 5, 7}) for (int q : {0, 31, 32, L / 2, L - 1}) {
            std::vector<double> p(L); double mx = -1e300;
            for (int kk = 0; kk < L; ++kk) { double s = 0; for (int c = 0; c < DH; ++c) s += val((size_t)b * L + q, h * DH + c) * val((size_t)b * L + kk, D + h * DH + c); p[kk] = s; mx = std::max(mx, s); }
            double l = 0; for (int kk = 0; kk < L; ++kk) { p[kk] = exp2(p[kk] - mx); l += p[kk]; }
            for (int c = 0; c < DH; ++c) { double acc = 0; for (int kk = 0; kk < L; ++kk) acc += p[kk] * val((size_t)b * L + kk, 2 * D + h * DH + c);
                const size_t oi = ((size_t)b * L + q) * D + h * DH + c;
                const double got = (double)(float)o[oi] + (double)(float)o[rows * D + oi] / 2048.0;
                maxerr = std::max(maxerr, fabs(got - acc / l)); if (PP && getenv("PP_DEBUG") && c == 0) printf("      b %d h %d q %d c0: got %.6f want %.6f\n", b, h, q, got, acc / l); }
        }
        printf("  %-34s max |err| against float64 on 45 sampled rows: %.3e\n", what, maxerr);
        if (PP) { size_t diff = 0; if (last_out.size() != o.size()) diff = ~(size_t)0; else for (size_t i = 0; i < o.size(); ++i) diff += __builtin_bit_cast(unsigned short, o[i]) != __builtin_bit_cast(unsigned short, last_out[i]);
            printf("  %-34s halfs that differ from attn_h3_kernel's output: %zu\n", what, diff); }
        last_out = o;
    }
#endif
    // MFMAs per launch: S^T 9 per (query tile, key tile), O^T 12 (6 when the tile's second 16-key step is padding only)
    double mf = 0; for (int kt = 0; kt < nt; ++kt) mf += 9 + (32 * kt + 16 < L ? 12 : 6);
    mf *= (double)nt * B * H;
    const double bytes = (double)rows * 3 * D * 4 + (double)rows * D * 4;          // q, k, v planes in, context planes out
    timeit(what, launch, mf, bytes);
#ifdef UU3D_PP_STAMP
    if (PP) {
        unsigned long long z[8] = {0}, hh[8];
        CK(hipMemcpyToSymbol(HIP_SYMBOL(pp_stamps), z, 64)); launch(); CK(hipDeviceSynchronize()); CK(hipMemcpyFromSymbol(hh, HIP_SYMBOL(pp_stamps), 64));
        const double wg = (double)B * H, passes = (double)hh[4], its = (double)hh[5];
        printf("      wave 0, cycles (s_memtime): staging %.0f per workgroup; per pass: prologue %.0f, last tile + store %.0f; per key tile: phase 1 (S under B) %.0f, phase 2 (O under A) %.0f   [%.0f passes, %.0f iterations]\n",
               hh[6] / wg, hh[0] / passes, hh[3] / passes, hh[1] / its, hh[2] / its, passes, its);
    }
#endif
    CK(hipFree(dh)); CK(hipFree(dl)); CK(hipFree(dout));
}

int main() {
    run<3, 3>(128, 71, "71 tokens x 128 sequences");
    run<12, 3>(32, 351, "351 tokens x 32 sequences");
    run<12, 3, 4>(32, 351, "351 x 32, one stream per SIMD");
    run<12, 3, 8>(32, 351, "351 x 32, two streams per SIMD");
    run<12, 3>(128, 351, "351 tokens x 128 sequences");
    run<12, 3, 4>(128, 351, "351 x 128, one stream per SIMD");
    run<12, 3, 8>(128, 351, "351 x 128, two streams per SIMD");
    run<12, 3>(64, 130, "130 tokens x 64 sequences");
    run<12, 3, 4>(64, 130, "130 x 64, one stream per SIMD");
    run<12, 3, 8>(64, 130, "130 x 64, two streams per SIMD");
    run<12, 3>(16, 384, "384 tokens x 16 sequences");
    run<12, 3, 4>(16, 384, "384 x 16, one stream per SIMD");
    run<12, 3, 8>(16, 384, "384 x 16, two streams per SIMD");
    return 0;
}
