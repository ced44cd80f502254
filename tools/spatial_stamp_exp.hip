// Phase timing of the spatial stack (csrc/uu3d_spatial_h3.h) by s_memtime stamps of every wave.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops -DUU3D_SPATIAL_STAMP -o tools/spatial_stamp_exp tools/spatial_stamp_exp.hip
//   tools/spatial_stamp_exp [frames = 9088] [lds bytes]
// Without -DUU3D_SPATIAL_STAMP and with -DUU3D_SP_SKIP=n (see uu3d_spatial_h3.h): the launch time with one phase left out.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include "../uplift-upsample-3dhpe_amd/csrc/uu3d_spatial_h3.h"
using namespace uu3d;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
template <class T> T* upload(const std::vector<T>& v) { T* d; CK(hipMalloc(&d, v.size() * sizeof(T))); CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 9088, J = 17, DS = 32, depth = 4;
    const size_t lds = argc > 2 ? (size_t)atoi(argv[2]) : sh3::lds_bytes();
    std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f);
    auto rnd = [&](size_t n, float sc) { std::vector<float> v(n); for (auto& x : v) x = sc * nd(rng); return v; };
    using LY = SpatialBlockLayoutV2<32, 64>;
    std::vector<float> blocks = rnd((size_t)depth * LY::size, 0.1f);
    for (int b = 0; b < depth; ++b) for (int i = 0; i < 32; ++i) { blocks[(size_t)b * LY::size + LY::ln1_g + i] = 1.f; blocks[(size_t)b * LY::size + LY::ln2_g + i] = 1.f; }
    std::vector<_Float16> frag((size_t)depth * SpatialFragLayoutH3::size);
    for (size_t i = 0; i < frag.size(); ++i) frag[i] = (_Float16)(0.1f * nd(rng));
    std::vector<float> g(32, 1.f), z32(32, 0.f);
    SpatialParams p{};
    p.embed_w = upload(rnd(2 * DS, 0.5f)); p.embed_b = upload(rnd(DS, 0.1f)); p.pe = upload(rnd(J * DS, 0.1f)); p.blocks = upload(blocks);
    p.norm_g = upload(g); p.norm_b = upload(z32); p.depth = depth; p.total_frames = M; p.frame_list = nullptr;
    float* kp = upload(rnd((size_t)M * J * 2, 1.f));
    _Float16* dfrag = upload(frag);
    float* out; CK(hipMalloc(&out, (size_t)M * J * DS * 4));
#ifndef MTV
#define MTV 1
#endif
    auto kern = spatial_stack_h3_kernel<17, 3, MTV>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    auto launch = [&] { hipLaunchKernelGGL(kern, dim3((M + 2) / 3), dim3(64 * (2 / MTV)), lds, 0, kp, p, dfrag, out, (_Float16*)nullptr, (_Float16*)nullptr, SpatialTrainIO{}); };
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    unsigned long long zz[12] = {0}, h[12];
#ifdef UU3D_SPATIAL_STAMP
    CK(hipMemcpyToSymbol(HIP_SYMBOL(spatial_clk), zz, sizeof(zz)));
#endif
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0)); for (int i = 0; i < 10; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
#ifdef UU3D_SPATIAL_STAMP
    CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(spatial_clk), sizeof(h)));
#else
    printf("frames %d, LDS %zu B, skip %d: %.1f us per launch\n", M, lds, UU3D_SP_SKIP, ms * 1e3 / 10); return 0;
#endif
    const double n = (double)h[11] * depth;
    const char* names[8] = {"LN1 + split", "q k v products + K/V store", "attention", "split o + projection + residual", "LN2 + split", "fc1 product", "GELU + split", "fc2 + residual"};
    printf("frames %d, LDS %zu B per wave: %.1f us per launch (with stamps); per wave %.0f ns in the block loop\n", M, lds, ms * 1e3 / 10, 10.0 * h[10] / h[11]);
    std::vector<float> o((size_t)M * J * DS); CK(hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost));
    double cs = 0; for (float v : o) cs += v; printf("checksum %.6f\n", cs);
    for (int i = 0; i < 8; ++i) printf("  %-34s %8.0f ns per block and wave\n", names[i], 10.0 * h[i] / n);
    return 0;
}
