// Phase timing of the temporal attention kernel (csrc/uu3d_attn.h) by s_memtime stamps of wave 0 of every workgroup.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops -DUU3D_ATTN_STAMP -o tools/attn_stamp_exp tools/attn_stamp_exp.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include "../uplift-upsample-3dhpe_amd/csrc/uu3d_attn.h"
using namespace uu3d;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 128, L = 71, H = 8, D = 384;
    std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> q((size_t)B * L * 3 * D); for (auto& v : q) v = nd(rng);
    float *dq, *dout; CK(hipMalloc(&dq, q.size() * 4)); CK(hipMalloc(&dout, (size_t)(B * L + 32) * D * 4));
    CK(hipMemcpy(dq, q.data(), q.size() * 4, hipMemcpyHostToDevice));
    auto kern = attn_f32_kernel<5, 48, true>;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&] { hipLaunchKernelGGL(kern, dim3(B * H), dim3(320), 0, 0, dq, 3 * D, D, L, H, (const uint8_t*)nullptr, dout, D, ATTN_FRAG_ORDER, 0); };
    for (int i = 0; i < 3; ++i) launch();
    unsigned long long z[8] = {0}, h[8];
    CK(hipMemcpyToSymbol(HIP_SYMBOL(attn_clk), z, 64));
    CK(hipEventRecord(e0)); for (int i = 0; i < 20; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(attn_clk), 64));
    const double n = (double)h[4];
    printf("workgroup per item, B=%d: %.1f us per launch; per workgroup (s_memtime ticks, 100 MHz = 10 ns each): load+stage %.0f, QK^T %.0f, softmax %.0f, PV+store %.0f\n",
           B, ms * 1e3 / 20, h[0] / n, h[1] / n, h[2] / n, h[3] / n);
    {
        auto k2 = attn_head_wave_kernel<5, 48, true>;
        constexpr size_t lds = attn_head_wave_lds_bytes<5, 48>();
        CK(hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        auto launch2 = [&] { hipLaunchKernelGGL(k2, dim3(B * H / 4), dim3(256), lds, 0, dq, 3 * D, D, L, H, (const uint8_t*)nullptr, dout, D, ATTN_FRAG_ORDER, B * H); };
        for (int i = 0; i < 3; ++i) launch2();
        CK(hipMemcpyToSymbol(HIP_SYMBOL(attn_clk), z, 64));
        CK(hipEventRecord(e0)); for (int i = 0; i < 20; ++i) launch2(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(attn_clk), 64));
        const double n2 = (double)h[4];
        printf("wave per item, B=%d: %.1f us per launch; per workgroup (wave 0): stage K/V %.0f, all tiles: QK^T %.0f, softmax %.0f, PV+store %.0f\n",
               B, ms * 1e3 / 20, h[0] / n2, h[1] / n2, h[2] / n2, h[3] / n2);
    }
    return 0;
}
