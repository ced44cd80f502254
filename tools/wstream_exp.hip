// Round 6: how fast can every CU pull the SAME weight stream (4.6 MB, the temporal chain's 96 x 48 KiB chunks) from L2 into LDS by LDS-DMA?
// The question behind a 64-row chain tile (twice the weight bytes per token row): 256 workgroups, one per CU, 3 x 48 KiB ring, one s_barrier per
// chunk, NO compute.   hipcc -O3 --offload-arch=gfx950 -o tools/wstream_exp tools/wstream_exp.hip ;  tools/wstream_exp [workgroups] [passes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) void glb_void;
static constexpr int CHUNK = 49152, NCH = 96;

template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64) stream_kernel(const unsigned char* __restrict__ W, int passes, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int PIECES = CHUNK / 1024 / WAVES;               // 1 KiB pieces per wave and chunk
    auto issue = [&](int g) __attribute__((always_inline)) {
        const unsigned char* s = W + (size_t)(g % NCH) * CHUNK + wave * (PIECES * 1024) + lane * 16;
        unsigned char* d = sm + (g % 3) * CHUNK + wave * (PIECES * 1024);
#pragma unroll
        for (int i = 0; i < PIECES; ++i)
            __builtin_amdgcn_global_load_lds((glb_void*)(s + i * 1024), (lds_void*)(d + i * 1024), 16, 0, 0);
    };
    const int total = NCH * passes;
    issue(0); issue(1);
    unsigned acc = 0;
    for (int g = 0; g < total; ++g) {
        issue(g + 2);
        if constexpr (PIECES == 12) asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        acc += *reinterpret_cast<unsigned*>(sm + (g % 3) * CHUNK + threadIdx.x * 4);      // touch the landed chunk
        __builtin_amdgcn_s_barrier();                          // (everybody has read it before it is refilled two iterations later)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 256, passes = argc > 2 ? atoi(argv[2]) : 4;
    std::vector<unsigned char> h((size_t)NCH * CHUNK);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned char)(i * 2654435761u >> 13);
    unsigned char* W; CK(hipMalloc(&W, h.size())); CK(hipMemcpy(W, h.data(), h.size(), hipMemcpyHostToDevice));
    unsigned* sink; CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](auto kern, int threads, const char* tag) {
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * CHUNK));
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), 3 * CHUNK, 0, W, passes, sink);
        CK(hipEventRecord(e0));
        const int it = 10;
        for (int i = 0; i < it; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), 3 * CHUNK, 0, W, passes, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
        const double bytes = (double)NCH * CHUNK * passes;
        printf("%s: %d workgroups, %d passes of 4.6 MB: %.1f us per launch = %.1f GB/s per workgroup, %.2f TB/s chip-wide, %.0f cycles@2.4GHz per 48 KiB chunk\n",
               tag, wgs, passes, ms * 1e3, bytes / (ms * 1e-3) * 1e-9, bytes * wgs / (ms * 1e-3) * 1e-12, ms * 1e-3 / (NCH * passes) * 2.4e9);
    };
    run(stream_kernel<4>, 256, "4 waves");
    run(stream_kernel<8>, 512, "8 waves");
    return 0;
}
