// Round 6: does v_fma_mix_f32 / v_fma_mixlo_f16 / v_fma_mixhi_f16 give the f16x3 split bit for bit?   hi = v_cvt_f16_f32(y) (f16 denormal results flushed, MODE as the
// kernels set it), d = y - float(hi) by ONE v_fma_mix_f32 (hi read as f16), lo = f16(d * 2048) by ONE v_fma_mixlo/hi_f16 that writes its half of a packed register.
//   hipcc -O3 --offload-arch=gfx950 -o tools/fma_mix_check tools/fma_mix_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
__global__ void k(const float* in, unsigned* out, int n) {
    __builtin_amdgcn_s_setreg((1 /*MODE*/) | (6 << 6) | ((2 - 1) << 11), 0);      // f16 denormals flushed, as h3_flush_f16_denormals()
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float sc = 2048.0f;
    unsigned ref_h = 0, ref_l = 0, new_h = 0, new_l = 0, mix_h = 0;
    for (int e = 0; e < 2; ++e) {
        const float y = in[2 * i + e];
        _Float16 h; float hf;
        asm("v_cvt_f16_f32 %0, %2\n\tv_cvt_f32_f16 %1, %0" : "=&v"(h), "=v"(hf) : "v"(y));
        const _Float16 l = (_Float16)((y - hf) * 2048.0f);
        unsigned short hb, lb; memcpy(&hb, &h, 2); memcpy(&lb, &l, 2);
        ref_h |= (unsigned)hb << (16 * e); ref_l |= (unsigned)lb << (16 * e);
        _Float16 h2; float d;
        asm("v_cvt_f16_f32 %0, %2\n\tv_fma_mix_f32 %1, %2, 1.0, -%0 op_sel_hi:[0,0,1]" : "=&v"(h2), "=v"(d) : "v"(y));
        unsigned short hb2; memcpy(&hb2, &h2, 2);
        new_h |= (unsigned)hb2 << (16 * e);
        if (e == 0) { asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(new_l) : "v"(d), "s"(sc)); asm("v_fma_mixlo_f16 %0, %1, 1.0, 0 op_sel_hi:[0,0,0]" : "+v"(mix_h) : "v"(y)); }
        else { asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(new_l) : "v"(d), "s"(sc)); asm("v_fma_mixhi_f16 %0, %1, 1.0, 0 op_sel_hi:[0,0,0]" : "+v"(mix_h) : "v"(y)); }
    }
    out[5 * i + 0] = ref_h; out[5 * i + 1] = ref_l; out[5 * i + 2] = new_h; out[5 * i + 3] = new_l; out[5 * i + 4] = mix_h;
}
int main() {
    const int n = 1 << 22;
    std::vector<float> in(n); std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f); std::uniform_real_distribution<float> ex(-30.f, 6.f);
    for (int i = 0; i < n; ++i) in[i] = (i % 3 == 0) ? nd(rng) : ((i & 1) ? 1.f : -1.f) * exp2f(ex(rng)) * (1.f + 0.3f * nd(rng));      // normals, and magnitudes from 2^-30 to 2^6
    in[0] = 0.f; in[1] = -0.f; in[2] = 6.0e-5f; in[3] = 6.2e-5f; in[4] = 65504.f; in[5] = 3.0e-8f;
    float* di; unsigned* dout; CK(hipMalloc(&di, n * 4)); CK(hipMalloc(&dout, (size_t)n / 2 * 5 * 4)); CK(hipMemcpy(di, in.data(), n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, di, dout, n); CK(hipDeviceSynchronize());
    std::vector<unsigned> out((size_t)n / 2 * 5); CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
    size_t bad_h = 0, bad_l = 0, bad_mixh = 0;
    for (size_t i = 0; i < (size_t)n / 2; ++i) { bad_h += out[5 * i] != out[5 * i + 2]; bad_l += out[5 * i + 1] != out[5 * i + 3]; bad_mixh += out[5 * i] != out[5 * i + 4]; }
    printf("%d values: hi (v_cvt_f16_f32 either way) %zu pairs differ; lo by v_fma_mix_f32 + v_fma_mixlo/hi_f16 against cvt / sub / mul / cvt: %zu pairs differ; hi by v_fma_mixlo/hi_f16(y, 1.0, 0) against v_cvt_f16_f32: %zu pairs differ\n", n, bad_h, bad_l, bad_mixh);
    if (bad_mixh) for (size_t i = 0, s = 0; i < (size_t)n / 2 && s < 5; ++i) if (out[5 * i] != out[5 * i + 4]) { printf("   y = %g, %g: cvt %08x mix %08x\n", in[2 * i], in[2 * i + 1], out[5 * i], out[5 * i + 4]); ++s; }
    return 0;
}
