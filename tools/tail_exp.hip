// tools/tail_exp.hip -- standalone harness of the XCD-cooperative tail kernel (csrc/uu3d_tail.h): times the launch on the shapes
// of h36m_351's last strided block + head2 and prints where the time goes from s_memrealtime stamps (STAMP build of the same
// source).  Weights / activations are random (no parity check here: tests/test_tail_gpu.py does that through the C ABI).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops -I uplift-upsample-3dhpe_amd/csrc \
//         tools/tail_exp.hip -o tools/tail_exp && tools/tail_exp [batch] [L_in] [stride] [iters]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "uu3d_tail.h"   // (moved from csrc/ to tools/ in round 6: no longer part of the library)

using namespace uu3d;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <class T> static T* dalloc(size_t n) { T* p; CK(hipMalloc((void**)&p, n * sizeof(T))); return p; }
static float* drand(size_t n, float a) {
    std::vector<float> h(n); for (auto& v : h) v = a * ((float)rand() / RAND_MAX * 2.f - 1.f);
    float* d = dalloc<float>(n); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice)); return d;
}
static _Float16* hrand(size_t n, float a) {
    std::vector<_Float16> h(n); for (auto& v : h) v = (_Float16)(a * ((float)rand() / RAND_MAX * 2.f - 1.f));
    _Float16* d = dalloc<_Float16>(n); CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice)); return d;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 128, L = argc > 2 ? atoi(argv[2]) : 3, stride = argc > 3 ? atoi(argv[3]) : 3;
    const int iters = argc > 4 ? atoi(argv[4]) : 200;
    const int Lo = (L - 3) / stride + 1, M = B * L, Mo = B * Lo;
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    TailParams p{};
    p.B = B; p.G = (B + 7) / 8; p.L_in = L; p.L_out = Lo; p.stride = stride; p.pad_left = 0; p.res_lo = stride > 1 ? 1 : 0; p.n_out = 51;
    p.x = drand((size_t)M * 384, 1.f); p.qkv = dalloc<float>((size_t)M * 1152); p.o = dalloc<float>((size_t)M * 384);
    p.hb = dalloc<float>((size_t)M * 768); p.part = dalloc<float>((size_t)2 * Mo * 384); p.out = dalloc<float>((size_t)Mo * 51);
    p.ln1_g = drand(384, 1.f); p.ln1_b = drand(384, .1f); p.bqkv = drand(1152, .1f); p.bp = drand(384, .1f);
    p.ln2_g = drand(384, 1.f); p.ln2_b = drand(384, .1f); p.b1 = drand(768, .1f); p.b2 = drand(384, .1f); p.bh = drand(64, .1f);
    p.wqkv_f = hrand((size_t)36 * 24 * 1024, .05f); p.wp_f = hrand((size_t)12 * 24 * 1024, .05f); p.w1_f = hrand((size_t)24 * 24 * 1024, .05f);
    p.wc_f = hrand((size_t)12 * 144 * 1024, .03f); p.wh_f = hrand((size_t)2 * 24 * 1024, .05f);
    p.ctl = dalloc<TailCtl>(1);
    p.dbg = dalloc<unsigned long long>((size_t)cus * 64);
    float* xsave = dalloc<float>((size_t)M * 384); CK(hipMemcpy(xsave, p.x, (size_t)M * 384 * 4, hipMemcpyDeviceToDevice));
    char* flush = dalloc<char>((size_t)512 << 20);
    hipStream_t st; CK(hipStreamCreate(&st));
    CK(hipFuncSetAttribute((const void*)strided_tail_kernel_t<true>, hipFuncAttributeMaxDynamicSharedMemorySize, tail::LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)strided_tail_kernel_t<false>, hipFuncAttributeMaxDynamicSharedMemorySize, tail::LDS_BYTES));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    auto launch = [&](bool stamp) {
        CK(hipMemsetAsync(p.ctl, 0, sizeof(TailCtl), st));
        if (stamp) hipLaunchKernelGGL(strided_tail_kernel_t<true>, dim3(cus), dim3(256), tail::LDS_BYTES, st, p);
        else hipLaunchKernelGGL(strided_tail_kernel_t<false>, dim3(cus), dim3(256), tail::LDS_BYTES, st, p);
    };
    for (int i = 0; i < 5; ++i) launch(false);
    CK(hipStreamSynchronize(st));
    // back to back (weights warm in L2 / MALL)
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) launch(false);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("B %d L_in %d stride %d: %d CUs, %.2f us per memset + launch (back to back)\n", B, L, stride, cus, 1e3f * ms / iters);
    // behind a 512 MiB memset (caches cold, as behind the rest of a forward)
    float cold = 0.f;
    for (int i = 0; i < 20; ++i) {
        CK(hipMemsetAsync(flush, i, (size_t)512 << 20, st));
        CK(hipMemcpyAsync(p.x, xsave, (size_t)M * 384 * 4, hipMemcpyDeviceToDevice, st));
        CK(hipEventRecord(e0, st)); launch(false); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); cold += ms;
    }
    printf("cold caches: %.2f us per memset + launch\n", 1e3f * cold / 20);
    TailCtl h; CK(hipMemcpy(&h, p.ctl, sizeof h, hipMemcpyDeviceToHost));
    printf("err %u owner", h.err); for (int i = 0; i < 8; ++i) printf(" %u", h.owner[i]);
    printf(" census"); for (int i = 0; i < 8; ++i) printf(" %u", h.census[i]); printf("\n");
    // stamps: slot ids = 0 start, 1 group claimed, 2 tickets; 8 + 8 phase + {0 weights issued, 1 previous phase complete, 2 operand
    // rows loaded + normalised (proj: K | V staged), 3 fragments in LDS, 4 MFMAs done, 5 stores issued, 7 published}
    CK(hipMemsetAsync(p.dbg, 0, (size_t)cus * 64 * 8, st));
    launch(true); CK(hipStreamSynchronize(st));
    std::vector<unsigned long long> d((size_t)cus * 64);
    CK(hipMemcpy(d.data(), p.dbg, d.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (int w = 0; w < cus; ++w) if (d[(size_t)w * 64]) t0 = std::min(t0, d[(size_t)w * 64]);
    printf("stamps (us after the first workgroup started; n, min / median / max over workgroups that reached the stamp)\n");
    const char* phn[6] = {"qkv", "attn", "proj", "fc1", "conv", "head"};
    for (int k = 0; k < 56; ++k) {
        std::vector<double> v;
        for (int w = 0; w < cus; ++w) if (d[(size_t)w * 64 + k]) v.push_back((double)(d[(size_t)w * 64 + k] - t0) * 0.01);
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        if (k < 8) printf("  %-8s %d: n %3zu  %7.2f %7.2f %7.2f\n", "start", k, v.size(), v.front(), v[v.size() / 2], v.back());
        else printf("  %-8s %d: n %3zu  %7.2f %7.2f %7.2f\n", phn[(k - 8) / 8], (k - 8) % 8, v.size(), v.front(), v[v.size() / 2], v.back());
    }
    return 0;
}
