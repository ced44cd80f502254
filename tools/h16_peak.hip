// Microbenchmarks behind the f16x3 GEMM design: f16 MFMA issue rate, and the LDS-read + MFMA inner loop of
// gemm_h3p_kernel with no global traffic at all (the ceiling of that loop structure).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int NACC>
__global__ void __launch_bounds__(256) mfma_only(float* out, int iters) {
    h16x8 a, b; for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(1.0f - e * 0.01f); }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// The inner loop of gemm_h3p_kernel<TM,TN>: per 16-deep k-step 2*(TM+TN) ds_read_b128 and 3*TM*TN MFMAs; LDS is
// laid out exactly as in the kernel; `sync` adds the per-k-tile barrier.
template <int TM, int TN, bool SYNC>
__global__ void __launch_bounds__(256) lds_mfma(float* out, int ktiles) {
    constexpr int BM = 64 * TM, BN = 64 * TN, LD = 40, STAGE = 2 * (BM + BN) * LD;
    extern __shared__ __attribute__((aligned(16))) _Float16 hsm[];
    for (int i = threadIdx.x; i < 2 * STAGE; i += 256) hsm[i] = (_Float16)((i % 97) * 0.01f);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 31, fk = (lane >> 5) * 8;
    f32x16 acc0[TM][TN], acc1[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }
    for (int kt = 0; kt < ktiles; ++kt) {
        const _Float16* S = hsm + (kt & 1) * STAGE;
        const _Float16* Ahp = S + (wm * (BM / 2) + fr) * LD + fk;
        const _Float16* Alp = Ahp + BM * LD;
        const _Float16* Bhp = S + 2 * BM * LD + (wn * (BN / 2) + fr) * LD + fk;
        const _Float16* Blp = Bhp + BN * LD;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            h16x8 ah[TM], alo[TM], bh[TN], blo[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) { ah[i] = *reinterpret_cast<const h16x8*>(Ahp + i * 32 * LD + kk * 16); alo[i] = *reinterpret_cast<const h16x8*>(Alp + i * 32 * LD + kk * 16); }
#pragma unroll
            for (int j = 0; j < TN; ++j) { bh[j] = *reinterpret_cast<const h16x8*>(Bhp + j * 32 * LD + kk * 16); blo[j] = *reinterpret_cast<const h16x8*>(Blp + j * 32 * LD + kk * 16); }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc0[i][j], 0, 0, 0);
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], blo[j], acc1[i][j], 0, 0, 0);
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[i], bh[j], acc1[i][j], 0, 0, 0);
                }
        }
        if (SYNC) __syncthreads();
    }
    float s = 0; for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) s += acc0[i][j][r] + acc1[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static float* out;
static _Float16* gsrc;
template <class F> float timeit(F launch) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 5;
}
template <int NACC> void run_mfma(int wgs_per_cu, int iters) {
    const int grid = 256 * wgs_per_cu;
    float ms = timeit([&] { hipLaunchKernelGGL(mfma_only<NACC>, dim3(grid), dim3(256), 0, 0, out, iters); });
    double fl = (double)grid * 4 * iters * 4 * NACC * 32768.0;
    printf("mfma_f32_32x32x16_f16 only: %d acc/wave, %d WG/CU: %7.3f ms  %7.1f TFLOP/s\n", NACC, wgs_per_cu, ms, fl / ms / 1e9);
}
template <int TM, int TN, bool SYNC> void run_loop(int wgs_per_cu, int ktiles) {
    auto kern = lds_mfma<TM, TN, SYNC>;
    const size_t lds = 2 * 2 * (64 * TM + 64 * TN) * 40 * 2;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = 256 * wgs_per_cu;
    float ms = timeit([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, out, ktiles); });
    double alg = (double)grid * ktiles * 2.0 * 64 * TM * 64 * TN * 32;
    printf("LDS+MFMA loop %3dx%-3d %s %d WG/CU: %7.3f ms  algorithmic %6.1f TF (f16 pipe %6.1f TF)\n", 64 * TM, 64 * TN, SYNC ? "barrier" : "free   ", wgs_per_cu, ms, alg / ms / 1e9, 3 * alg / ms / 1e9);
}

// The same loop with the other two phases of the real kernel switched on one at a time: WR = the per-k-tile
// ds_write_b128 staging (from registers), LD = the global loads feeding it (planes of 384-half rows, L2 resident).
template <int TM, int TN, bool WR, bool LD>
__global__ void __launch_bounds__(256) loop_ablate(float* out, int ktiles, const _Float16* __restrict__ src) {
    constexpr int BM = 64 * TM, BN = 64 * TN, LDS = 40, STAGE = 2 * (BM + BN) * LDS, NI = 2 * (TM + TN);
    extern __shared__ __attribute__((aligned(16))) _Float16 hsm[];
    for (int i = threadIdx.x; i < 2 * STAGE; i += 256) hsm[i] = (_Float16)((i % 97) * 0.01f);
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 31, fk = (lane >> 5) * 8;
    const int srow = tid >> 2, scol = (tid & 3) * 8;
    h16x8 st[NI];
    for (int i = 0; i < NI; ++i) for (int e = 0; e < 8; ++e) st[i][e] = (_Float16)(0.001f * (tid + i + e));
    const _Float16* gp = src + ((size_t)(blockIdx.x % 128) * 64 + srow) * 384 + scol;
    f32x16 acc0[TM][TN], acc1[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }
    for (int kt = 0; kt < ktiles; ++kt) {
        if (LD) {
#pragma unroll
            for (int i = 0; i < NI; ++i) st[i] = *reinterpret_cast<const h16x8*>(gp + (size_t)i * 8192 * 384 + (kt % 12) * 32);
        }
        const _Float16* S = hsm + (kt & 1) * STAGE;
        const _Float16* Ahp = S + (wm * (BM / 2) + fr) * LDS + fk;
        const _Float16* Alp = Ahp + BM * LDS;
        const _Float16* Bhp = S + 2 * BM * LDS + (wn * (BN / 2) + fr) * LDS + fk;
        const _Float16* Blp = Bhp + BN * LDS;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            h16x8 ah[TM], alo[TM], bh[TN], blo[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) { ah[i] = *reinterpret_cast<const h16x8*>(Ahp + i * 32 * LDS + kk * 16); alo[i] = *reinterpret_cast<const h16x8*>(Alp + i * 32 * LDS + kk * 16); }
#pragma unroll
            for (int j = 0; j < TN; ++j) { bh[j] = *reinterpret_cast<const h16x8*>(Bhp + j * 32 * LDS + kk * 16); blo[j] = *reinterpret_cast<const h16x8*>(Blp + j * 32 * LDS + kk * 16); }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc0[i][j], 0, 0, 0);
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], blo[j], acc1[i][j], 0, 0, 0);
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[i], bh[j], acc1[i][j], 0, 0, 0);
                }
        }
        if (WR) {
            _Float16* W = hsm + ((kt & 1) ^ 1) * STAGE;
#pragma unroll
            for (int i = 0; i < NI; ++i) *reinterpret_cast<h16x8*>(&W[(srow + 64 * i) * LDS + scol]) = st[i];
        }
        __syncthreads();
    }
    float s = 0; for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) s += acc0[i][j][r] + acc1[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int TM, int TN, bool WR, bool LD> void run_ablate(int wgs_per_cu, int ktiles) {
    auto kern = loop_ablate<TM, TN, WR, LD>;
    const size_t lds = 2 * 2 * (64 * TM + 64 * TN) * 40 * 2;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = 256 * wgs_per_cu;
    float ms = timeit([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, out, ktiles, gsrc); });
    double alg = (double)grid * ktiles * 2.0 * 64 * TM * 64 * TN * 32;
    printf("loop %3dx%-3d + barrier%s%s, %d WG/CU: %7.3f ms  algorithmic %6.1f TF\n", 64 * TM, 64 * TN, WR ? " + LDS staging writes" : "", LD ? " + global loads" : "", wgs_per_cu, ms, alg / ms / 1e9);
}

// LDS-DMA staging (global_load_lds_dwordx4): unpadded 64-byte rows, 16-byte chunks XOR-swizzled by (row >> 2) & 3 so
// that fragment reads (16 rows, one chunk) and the lane-linear DMA image (4 rows x 4 chunks per 16 lanes) are both
// bank-conflict free.  NBUF = 2: wait for everything each k-tile; NBUF = 3: tile kt+2 stays in flight across the barrier.
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;
template <int TM, int TN, int NBUF>
__global__ void __launch_bounds__(256) loop_glds(float* out, int ktiles, const _Float16* __restrict__ src) {
    constexpr int BM = 64 * TM, BN = 64 * TN, ROWS = 2 * (BM + BN), STAGE = ROWS * 32, NP = ROWS / 64;
    extern __shared__ __attribute__((aligned(16))) _Float16 hsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 31, fkc = lane >> 5;
    // DMA source: lane -> (row, physical chunk) of its wave's 16-row piece; logical chunk = physical ^ swizzle(row)
    const int drow = 16 * wave + (lane >> 2);
    const int dchunk = (lane & 3) ^ ((drow >> 2) & 3);
    const _Float16* gp = src + ((size_t)(blockIdx.x % 128) * 64 + drow) * 384 + dchunk * 8;
    auto dma = [&](int kt, int buf) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const _Float16* g = gp + (size_t)p * 8192 * 384 + (kt % 12) * 32;
            _Float16* d = hsm + buf * STAGE + (p * 64 + 16 * wave) * 32;      // wave-uniform
            __builtin_amdgcn_global_load_lds((glb_void*)g, (lds_void*)d, 16, 0, 0);
        }
    };
    f32x16 acc0[TM][TN], acc1[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }
    dma(0, 0);
    if (NBUF == 3) dma(1, 1);
    if (NBUF == 3) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(NP) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < ktiles; ++kt) {
        const int cur = kt % NBUF;
        dma(kt + NBUF - 1, (kt + NBUF - 1) % NBUF);
        const _Float16* S = hsm + cur * STAGE;
        const int ra = wm * (BM / 2) + fr, rb = wn * (BN / 2) + fr;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            h16x8 ah[TM], alo[TM], bh[TN], blo[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = ra + 32 * i, c = (2 * kk + fkc) ^ ((r >> 2) & 3);
                ah[i] = *reinterpret_cast<const h16x8*>(S + r * 32 + c * 8);
                alo[i] = *reinterpret_cast<const h16x8*>(S + (BM + r) * 32 + c * 8);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = rb + 32 * j, c = (2 * kk + fkc) ^ ((r >> 2) & 3);
                bh[j] = *reinterpret_cast<const h16x8*>(S + (2 * BM + r) * 32 + c * 8);
                blo[j] = *reinterpret_cast<const h16x8*>(S + (2 * BM + BN + r) * 32 + c * 8);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc0[i][j], 0, 0, 0);
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], blo[j], acc1[i][j], 0, 0, 0);
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[i], bh[j], acc1[i][j], 0, 0, 0);
                }
        }
        if (NBUF == 3) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(NP) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0; for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) s += acc0[i][j][r] + acc1[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int TM, int TN, int NBUF> void run_glds(int wgs_per_cu, int ktiles) {
    auto kern = loop_glds<TM, TN, NBUF>;
    const size_t lds = (size_t)NBUF * 2 * (64 * TM + 64 * TN) * 32 * 2;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = 256 * wgs_per_cu;
    float ms = timeit([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, out, ktiles, gsrc); });
    double alg = (double)grid * ktiles * 2.0 * 64 * TM * 64 * TN * 32;
    printf("loop %3dx%-3d LDS-DMA %d buffers (%3zu KB), %d WG/CU: %7.3f ms  algorithmic %6.1f TF\n", 64 * TM, 64 * TN, NBUF, lds / 1024, wgs_per_cu, ms, alg / ms / 1e9);
}
int main() {
    CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    CK(hipMalloc(&gsrc, (size_t)8 * 8192 * 384 * 2)); CK(hipMemset(gsrc, 0, (size_t)8 * 8192 * 384 * 2));
    run_mfma<1>(1, 4000); run_mfma<4>(1, 1000); run_mfma<4>(2, 1000); run_mfma<2>(4, 1000);
    run_loop<1, 1, false>(4, 2000); run_loop<1, 1, true>(4, 2000);
    run_loop<2, 1, false>(2, 2000); run_loop<2, 1, true>(2, 2000);
    run_loop<1, 2, false>(2, 2000); run_loop<1, 2, true>(2, 2000);
    run_loop<2, 2, false>(2, 1000); run_loop<2, 2, true>(2, 1000); run_loop<2, 2, true>(1, 1000);
    run_loop<1, 2, true>(2, 12);
    run_ablate<1, 2, false, false>(2, 2000); run_ablate<1, 2, true, false>(2, 2000); run_ablate<1, 2, true, true>(2, 2000);
    run_ablate<2, 2, false, false>(2, 1000); run_ablate<2, 2, true, false>(2, 1000); run_ablate<2, 2, true, true>(2, 1000);
    run_glds<1, 2, 2>(2, 2000); run_glds<1, 2, 2>(3, 2000); run_glds<1, 2, 3>(2, 2000); run_glds<2, 2, 2>(2, 1000); run_glds<2, 2, 3>(1, 1000); run_glds<2, 1, 2>(3, 2000); run_glds<1, 1, 2>(4, 2000); run_glds<1, 1, 3>(3, 2000);
    run_ablate<1, 1, true, false>(4, 2000); run_ablate<1, 1, true, true>(4, 2000);   // K = 384 per workgroup, like the model
    return 0;
}
