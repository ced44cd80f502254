// Attempt at a standalone reproducer of the packed-fp32 hazard of docs/HISTORY.md E.12: v_pk_mul_f32 / v_pk_fma_f32 with
// op_sel broadcasts from a VGPR pair that a global_load_dwordx2 has just written, next to waves that keep the MFMA pipe and
// the LDS busy (the situation of the f16x3 GEMM prologue).  Every lane checks the packed results against scalar arithmetic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// MODE 0: op_sel / op_sel_hi broadcasts from the (mean, rstd) pair; MODE 1: the same arithmetic with pre-splatted pairs and
// no op_sel (what hipcc emits in the GEMM loop body, which never failed).  BUSY: waves 2-3 run MFMA + LDS traffic.
template <int MODE, bool BUSY>
__global__ void __launch_bounds__(256)
repro(const float2* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ x,
      int rows, int K, int iters, unsigned* __restrict__ bad, float* __restrict__ sink, float* __restrict__ sample)
{
    __shared__ float lds[4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned nbad = 0;
    float keep = 0.f;
    if (wave >= 2 && !BUSY) return;
    if (wave >= 2) {       // two waves per workgroup keep the matrix pipe and the LDS busy, like co-resident GEMM waves
        h16x8 a, b; for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.01f * (lane + e)); b[e] = (_Float16)(0.5f - 0.01f * e); }
        f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int it = 0; it < iters * 8; ++it) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
            lds[(tid * 7 + it) & 4095] = acc[it & 15];
        }
        keep = acc[0] + lds[tid];
    } else {
        for (int it = 0; it < iters; ++it) {
            const int row = (blockIdx.x * 131 + it * 17 + (tid >> 3)) % rows;
            const int k = ((tid & 7) * 4 + it * 32) % K;
            const float2 s = stats[row];                                                  // global_load_dwordx2 -> VGPR pair (mean, rstd)
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + k);
            const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + k);
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)row * K + k);
            f32x2 st = {s.x, s.y};
            f32x2 g01 = {g[0], g[1]}, g23 = {g[2], g[3]}, b01 = {bt[0], bt[1]}, b23 = {bt[2], bt[3]}, x01 = {xv[0], xv[1]}, x23 = {xv[2], xv[3]};
            f32x2 inv01, inv23, t01, t23, y01, y23;
            // inv = gamma * rstd (both halves read st.y); t = beta - mean * inv (both halves read st.x); y = x * inv + t
            if (MODE == 0) {
                asm volatile("v_pk_mul_f32 %0, %2, %4 op_sel:[0,1]\n\t"
                             "v_pk_mul_f32 %1, %3, %4 op_sel:[0,1]"
                             : "=&v"(inv01), "=&v"(inv23) : "v"(g01), "v"(g23), "v"(st));
                asm volatile("v_pk_fma_f32 %0, %4, %2, %5 op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
                             "v_pk_fma_f32 %1, %4, %3, %6 op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]"
                             : "=&v"(t01), "=&v"(t23) : "v"(inv01), "v"(inv23), "v"(st), "v"(b01), "v"(b23));
            } else {
                f32x2 rr = {s.y, s.y}, mm = {s.x, s.x};
                asm volatile("v_pk_mul_f32 %0, %2, %4\n\t"
                             "v_pk_mul_f32 %1, %3, %4"
                             : "=&v"(inv01), "=&v"(inv23) : "v"(g01), "v"(g23), "v"(rr));
                asm volatile("v_pk_fma_f32 %0, %4, %2, %5 neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
                             "v_pk_fma_f32 %1, %4, %3, %6 neg_lo:[1,0,0] neg_hi:[1,0,0]"
                             : "=&v"(t01), "=&v"(t23) : "v"(inv01), "v"(inv23), "v"(mm), "v"(b01), "v"(b23));
            }
            asm volatile("v_pk_fma_f32 %0, %2, %4, %6\n\tv_pk_fma_f32 %1, %3, %5, %7"
                         : "=&v"(y01), "=&v"(y23) : "v"(x01), "v"(x23), "v"(inv01), "v"(inv23), "v"(t01), "v"(t23));
            const float got[4] = {y01[0], y01[1], y23[0], y23[1]};
            for (int e = 0; e < 4; ++e) {
                const float inv = __fmul_rn(g[e], s.y);                  // one rounding, like the packed multiply
                const float ref = fmaf(xv[e], inv, fmaf(-s.x, inv, bt[e]));
                if (got[e] != ref) { if (nbad == 0 && sample != nullptr && atomicAdd(bad + 1, 1u) < 8) { unsigned k2 = atomicAdd(bad + 2, 1u); if (k2 < 8) { sample[k2 * 6 + 0] = got[e]; sample[k2 * 6 + 1] = ref; sample[k2 * 6 + 2] = bt[e]; sample[k2 * 6 + 3] = (float)e; sample[k2 * 6 + 4] = (float)lane; sample[k2 * 6 + 5] = inv; } } ++nbad; }
            }
            keep += got[0];
        }
    }
    if (nbad) atomicAdd(bad, nbad);
    sink[blockIdx.x * 256 + tid] = keep;
}

int main() {
    const int rows = 9088, K = 384, wgs = 2048, iters = 400;
    std::vector<float> h((size_t)rows * K); srand(1); for (auto& v : h) v = rand() / (float)RAND_MAX * 2.f - 1.f;
    float *dx, *dg, *db, *sink; float2* ds; unsigned* dbad;
    CK(hipMalloc(&dx, h.size() * 4)); CK(hipMemcpy(dx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dg, K * 4)); CK(hipMemcpy(dg, h.data(), K * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&db, K * 4)); CK(hipMemcpy(db, h.data() + 5000, K * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&ds, rows * 8)); CK(hipMemcpy(ds, h.data() + 10000, rows * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&dbad, 12)); CK(hipMalloc(&sink, (size_t)wgs * 256 * 4));
    float* dsample; CK(hipMalloc(&dsample, 48 * 4));
    auto run = [&](auto kern, const char* what) {
        unsigned total = 0; float smp[48] = {0};
        CK(hipMemset(dsample, 0, 48 * 4));
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemset(dbad, 0, 12));
            hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, ds, dg, db, dx, rows, K, iters, dbad, sink, dsample);
            unsigned b; CK(hipMemcpy(&b, dbad, 4, hipMemcpyDeviceToHost)); total += b;
        }
        CK(hipMemcpy(smp, dsample, 48 * 4, hipMemcpyDeviceToHost));
        printf("%-62s %9u mismatching elements of %.0f M\n", what, total, 5.0 * wgs * 128 * iters * 4 / 1e6);
        if (total) for (int k = 0; k < 3; ++k) printf("      got %.9g  expected %.9g  (beta %.9g, element %d, lane %d, inv %.9g)\n", smp[k * 6], smp[k * 6 + 1], smp[k * 6 + 2], (int)smp[k * 6 + 3], (int)smp[k * 6 + 4], smp[k * 6 + 5]);
    };
    run(repro<0, true>,  "op_sel broadcasts from the VGPR pair, MFMA waves alongside:");
    run(repro<0, false>, "op_sel broadcasts from the VGPR pair, no MFMA waves:");
    run(repro<1, true>,  "pre-splatted pairs, no op_sel, MFMA waves alongside:");
    run(repro<1, false>, "pre-splatted pairs, no op_sel, no MFMA waves:");
    return 0;
}
