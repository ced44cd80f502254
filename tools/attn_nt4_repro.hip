// Reproducer for the 4-tile instantiation of attn_head_wave_kernel (gated off in uu3d_api.hip): runs it against the workgroup-per-item kernel.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops -o tools/attn_nt4_repro tools/attn_nt4_repro.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
#include "../uplift-upsample-3dhpe_amd/csrc/uu3d_attn.h"
using namespace uu3d;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
#ifndef NTT
#define NTT 4
#endif
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8, L = argc > 2 ? atoi(argv[2]) : 63, H = 8, D = 384;
    std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> q((size_t)B * L * 3 * D); for (auto& v : q) v = nd(rng);
    float *dq; _Float16 *o1, *o2; const size_t planes = (size_t)(B * L + 32) * D;
    CK(hipMalloc(&dq, q.size() * 4)); CK(hipMalloc(&o1, planes * 2 * 2)); CK(hipMalloc(&o2, planes * 2 * 2));
    CK(hipMemset(o1, 0, planes * 4)); CK(hipMemset(o2, 0, planes * 4));
    CK(hipMemcpy(dq, q.data(), q.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((attn_f32_kernel<NTT, 48, true>), dim3(B * H), dim3(64 * NTT), 0, 0, dq, 3 * D, D, L, H, (const uint8_t*)nullptr, (float*)o1, D, planes, B * H);
    CK(hipDeviceSynchronize());
    printf("workgroup-per-item kernel done\n"); fflush(stdout);
    auto k2 = attn_head_wave_kernel<NTT, 48, true>;
    constexpr size_t lds = attn_head_wave_lds_bytes<NTT, 48>();
    CK(hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k2, dim3(B * H / 4), dim3(256), lds, 0, dq, 3 * D, D, L, H, (const uint8_t*)nullptr, (float*)o2, D, planes, B * H);
    CK(hipDeviceSynchronize());
    std::vector<_Float16> a(planes * 2), b(planes * 2);
    CK(hipMemcpy(a.data(), o1, planes * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), o2, planes * 4, hipMemcpyDeviceToHost));
    size_t bad = 0; for (size_t i = 0; i < a.size(); ++i) if ((float)a[i] != (float)b[i]) ++bad;
    printf("NT=%d L=%d B=%d: wave-per-head kernel done, %zu of %zu halfs differ\n", NTT, L, B, bad, a.size());
    return 0;
}
