// uu3d_ops.hip -- C ABI of the backward building blocks (include/uu3d_ops.h).
#include <hip/hip_runtime.h>
#include <algorithm>
#include "../../include/uu3d.h"
#include "../../include/uu3d_ops.h"
#include "uu3d_gemm.h"
#include "uu3d_misc.h"
#include "uu3d_bwd.h"
#include "uu3d_launch.h"
#include "uu3d_gemm_panel.h"
#include <vector>

using namespace uu3d;

size_t uu3d_op_scratch_floats(void) { return kOpScratchFloats; }

int uu3d_op_gemm_tn(const float* a, int32_t lda, const float* b, int32_t ldb, int32_t R, int32_t P, int32_t Q, float* c,
                    int32_t ldc, float* scratch, size_t scratch_floats, void* stream) {
    if (!a || !b || !c || !scratch || R < 1 || P < 1 || Q < 1 || (lda & 3) || (ldb & 3) || (P & 3) || (Q & 3)) return UU3D_ERR_INVALID_ARGUMENT;
    TnLoadPlain al{a, lda, R, P};
    EpStore ep{c, ldc};
    return launch_gemm_tn(al, b, ldb, R, P, Q, ep, scratch, scratch_floats, (hipStream_t)stream);
}
int uu3d_op_gemm_tn_h3(const float* a, int32_t lda, const float* b, int32_t ldb, int32_t R, int32_t P, int32_t Q, float* c,
                       int32_t ldc, float* scratch, size_t scratch_floats, void* stream) {
    if (!a || !b || !c || !scratch || R < 1 || P < 1 || Q < 1 || (lda & 3) || (ldb & 3) || (P & 3) || (Q & 3)) return UU3D_ERR_INVALID_ARGUMENT;
    TnLoadPlain al{a, lda, R, P};
    EpStore ep{c, ldc};
    return launch_gemm_tn(al, b, ldb, R, P, Q, ep, scratch, scratch_floats, (hipStream_t)stream, true);
}

int uu3d_op_gemm_nt(const float* a, int32_t lda, const float* w, int32_t ldw, int32_t M, int32_t N, int32_t K, float* c,
                    int32_t ldc, float* scratch, size_t scratch_floats, void* stream) {
    if (!a || !w || !c || M < 1 || (N & 63) || (K & 31) || ldw != K) return UU3D_ERR_INVALID_ARGUMENT;
    ALoadPlain al{a, lda, M, K};
    EpStore ep{c, ldc};
    return launch_gemm(al, w, M, N, K, ep, scratch, scratch_floats, (hipStream_t)stream);
}

int uu3d_op_colsum(const float* x, int32_t ldx, int32_t R, int32_t C, int32_t period, const uint8_t* mask, int32_t want,
                   float* out, int32_t accumulate, float* scratch, size_t scratch_floats, void* stream) {
    if (!x || !out || !scratch || R < 1 || C < 1 || period < 0) return UU3D_ERR_INVALID_ARGUMENT;
    return launch_colsum(x, ldx, R, C, period, mask, want, out, accumulate, scratch, scratch_floats, (hipStream_t)stream);
}

int uu3d_op_row_stats(const float* x, int32_t ld, int32_t D, int32_t M, float eps, float* stats, void* stream) {
    if (!x || !stats || D < 4 || (D & 3) || D > 1024 || M < 1) return UU3D_ERR_INVALID_ARGUMENT;
    launch_row_stats(x, ld, D, M, eps, (float2*)stats, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? UU3D_OK : UU3D_ERR_HIP;
}

int uu3d_op_ln_bwd(const float* x, const float* dy, const float* stats, const float* gamma, int32_t ld, int32_t D, int32_t M,
                   float* dx, int32_t accumulate, float* dgamma, float* dbeta, float* scratch, size_t scratch_floats, void* stream) {
    if (!x || !dy || !stats || !gamma || !dx || !dgamma || !dbeta || !scratch || (D & 3) || D > 1024) return UU3D_ERR_INVALID_ARGUMENT;
    return launch_ln_bwd(x, dy, (const float2*)stats, gamma, ld, D, M, dx, accumulate, dgamma, dbeta, 0, scratch, scratch_floats, (hipStream_t)stream);
}

int uu3d_op_attn_fwd(const float* qkv, int32_t ld, int32_t D, int32_t B, int32_t L, int32_t H, int32_t dh, const uint8_t* mask,
                     float* out, int32_t ldo, void* stream) {
    if (!qkv || !out || L < 1 || L > 128 || (dh != 4 && dh != 48)) return UU3D_ERR_INVALID_ARGUMENT;
    return launch_attn_generic(false, qkv, nullptr, ld, D, B, L, H, dh, mask, out, ldo, (hipStream_t)stream);
}
int uu3d_op_attn_bwd(const float* qkv, const float* dout, int32_t ld, int32_t D, int32_t B, int32_t L, int32_t H, int32_t dh,
                     const uint8_t* mask, float* dqkv, int32_t ldo, void* stream) {
    if (!qkv || !dout || !dqkv || L < 1 || (dh != 4 && dh != 48)) return UU3D_ERR_INVALID_ARGUMENT;
    if (L > 96) return UU3D_ERR_UNSUPPORTED;      // P and dS matrices of one head must fit the 160 KiB LDS
    return launch_attn_generic(true, qkv, dout, ld, D, B, L, H, dh, mask, dqkv, ldo, (hipStream_t)stream);
}

// ---- row-panel path of a LayerNorm-fed Dense layer (uu3d_gemm_panel.h) --------------------------------------------
size_t uu3d_op_panel_operand_bytes(int32_t N) { return (N < 32 || (N & 31)) ? 0 : panel_b_halfs(N, 384) * sizeof(_Float16); }
size_t uu3d_op_panel_a_bytes(int32_t M) { return M < 1 ? 0 : panel_a_halfs(M, 384) * sizeof(_Float16); }

int uu3d_op_panel_pack(const float* w, int32_t N, void* operand_dev, void* stream) {
    if (!w || !operand_dev || N < 32 || (N & 31)) return UU3D_ERR_INVALID_ARGUMENT;
    const int K = 384;
    std::vector<_Float16> bh((size_t)N * K), bl((size_t)N * K), out(panel_b_halfs(N, K));
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            const float x = w[(size_t)k * N + n];                     // Keras (in, out) -> Bt[n][k]
            const _Float16 h = h3_hi(x);
            bh[(size_t)n * K + k] = h; bl[(size_t)n * K + k] = (_Float16)((x - (float)h) * H3_SCALE);
        }
    panel_pack_operand(bh.data(), bl.data(), N, K, K, out.data());
    if (hipMemcpyAsync(operand_dev, out.data(), out.size() * sizeof(_Float16), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) return UU3D_ERR_HIP;
    return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? UU3D_OK : UU3D_ERR_HIP;
}

int uu3d_op_ln_dense_panel(const float* x, int32_t ldx, int32_t M, const float* gamma, const float* beta, float eps, const void* operand,
                           const float* bias, int32_t N, int32_t relu, void* a_scratch, void* out, int32_t ldo, void* stream_) {
    if (!x || !gamma || !beta || !operand || !bias || !a_scratch || !out || M < 1 || N < 32 || (N & 31) || ldx < 384 || (ldx & 3)) return UU3D_ERR_INVALID_ARGUMENT;
    if ((double)M * (relu ? N : ldo) * 4.0 >= 4.0e9) return UU3D_ERR_UNSUPPORTED;          // 32-bit byte offsets in the epilogue stores
    hipStream_t stream = (hipStream_t)stream_;
    _Float16* Af = reinterpret_cast<_Float16*>(a_scratch);
    hipLaunchKernelGGL((ln_split_frag_kernel<24, 8>), dim3((M + 7) / 8), dim3(128), 0, stream, x, ldx, M, eps, gamma, beta, Af);
    const int mt = (M + 127) / 128, chunks = N / 32;
    int S = 0;                                                        // any divisor that keeps <= 32 chunks per workgroup; prefer ~one round
    for (int s = 1; s <= chunks; ++s) if (chunks % s == 0 && chunks / s <= 32 && (S == 0 || (mt * s + 7) / 8 <= 32)) S = s;
    if (S == 0) return UU3D_ERR_UNSUPPORTED;
    const dim3 grid(8 * S, ((mt * S + 7) / 8 + S - 1) / S);
    const _Float16* Bf = reinterpret_cast<const _Float16*>(operand);
    if (relu) {
        auto kern = gemm_h3_panel_kernel<24, PanelEpBiasReluSplit>;
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PANEL_LDS_TOTAL);
        _Float16* oh = reinterpret_cast<_Float16*>(out);
        hipLaunchKernelGGL(kern, grid, dim3(256), PANEL_LDS_TOTAL, stream, Af, Bf, bias, M, mt, S, chunks / S, PanelEpBiasReluSplit{oh, oh + (size_t)M * N, N});
    } else {
        auto kern = gemm_h3_panel_kernel<24, PanelEpBias>;
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PANEL_LDS_TOTAL);
        hipLaunchKernelGGL(kern, grid, dim3(256), PANEL_LDS_TOTAL, stream, Af, Bf, bias, M, mt, S, chunks / S, PanelEpBias{reinterpret_cast<float*>(out), ldo});
    }
    return hipGetLastError() == hipSuccess ? UU3D_OK : UU3D_ERR_HIP;
}
