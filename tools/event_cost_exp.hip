// What does a cross-stream dependency cost the RECORDING stream?  A chain of short kernels on stream A, optionally
// with an event record after each (and stream B waiting for it and running a short kernel of its own).
//   hipcc -O2 --offload-arch=gfx950 -o tools/event_cost_exp tools/event_cost_exp.hip && tools/event_cost_exp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(float* p, int n) { float v = p[threadIdx.x]; for (int i = 0; i < n; ++i) v = v * 1.0001f + 0.5f; p[threadIdx.x] = v; }

int main() {
    float* d; CK(hipMalloc(&d, 1 << 20)); CK(hipMemset(d, 0, 1 << 20));
    float* d2; CK(hipMalloc(&d2, 1 << 20)); CK(hipMemset(d2, 0, 1 << 20));
    hipStream_t A, B; CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    const int N = 400;
    std::vector<hipEvent_t> ev(N), evd(N);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : evd) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
    uint32_t* flag = nullptr;
    const bool have_sig = hipExtMallocWithFlags((void**)&flag, 64, hipMallocSignalMemory) == hipSuccess;
    if (have_sig) CK(hipMemset(flag, 0, 64));
    auto run = [&](const char* name, int mode, int spin_n) -> int {
        double best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            if (have_sig) { CK(hipMemset(flag, 0, 64)); }
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) {
                hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, A, d, spin_n);
                if (mode == 1) CK(hipEventRecord(ev[i], A));
                if (mode == 2) { CK(hipEventRecord(ev[i], A)); CK(hipStreamWaitEvent(B, ev[i], 0)); hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, B, d2, spin_n); }
                if (mode == 3) { CK(hipEventRecord(evd[i], A)); CK(hipStreamWaitEvent(B, evd[i], 0)); hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, B, d2, spin_n); }
                if (mode == 4) { CK(hipStreamWriteValue32(A, flag, (uint32_t)(i + 1), 0)); CK(hipStreamWaitValue32(B, flag, (uint32_t)(i + 1), hipStreamWaitValueGte, 0xffffffffu)); hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, B, d2, spin_n); }
                if (mode == 5) { hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, B, d2, spin_n); }            // no dependency at all
                if (mode == 6 && (i & 3) == 3) { CK(hipEventRecord(ev[i], A)); CK(hipStreamWaitEvent(B, ev[i], 0)); }
                if (mode == 6) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, B, d2, spin_n);
            }
            auto t1 = std::chrono::steady_clock::now();
            CK(hipDeviceSynchronize());
            auto t2 = std::chrono::steady_clock::now();
            const double tot = std::chrono::duration<double, std::micro>(t2 - t0).count() / N;
            if (tot < best) best = tot;
            if (rep == 4) printf("%-58s %7.2f us per link (host enqueue %.2f)\n", name, best, std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
        }
        return 0;
    };
    for (int spin_n : {200, 4000}) {
        printf("---- kernel body %d iterations\n", spin_n);
        run("A: kernels only", 0, spin_n);
        run("A: kernel + event record", 1, spin_n);
        run("A: kernel + record; B: wait + kernel", 2, spin_n);
        run("same, events with hipEventDisableSystemFence", 3, spin_n);
        if (have_sig) run("A: kernel + WriteValue32; B: WaitValue32 + kernel", 4, spin_n);
        run("A: kernel; B: kernel (independent)", 5, spin_n);
        run("A: kernel, record every 4th; B: kernel", 6, spin_n);
    }
    return 0;
}
