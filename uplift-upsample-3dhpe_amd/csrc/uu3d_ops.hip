// uu3d_ops.hip -- C ABI of the backward building blocks (include/uu3d_ops.h).
#include <hip/hip_runtime.h>
#include <algorithm>
#include "../../include/uu3d.h"
#include "../../include/uu3d_ops.h"
#include "uu3d_gemm.h"
#include "uu3d_misc.h"
#include "uu3d_bwd.h"
#include "uu3d_launch.h"

using namespace uu3d;

size_t uu3d_op_scratch_floats(void) { return kOpScratchFloats; }

int uu3d_op_gemm_tn(const float* a, int32_t lda, const float* b, int32_t ldb, int32_t R, int32_t P, int32_t Q, float* c,
                    int32_t ldc, float* scratch, size_t scratch_floats, void* stream) {
    if (!a || !b || !c || !scratch || R < 1 || P < 1 || Q < 1 || (lda & 3) || (ldb & 3) || (P & 3) || (Q & 3)) return UU3D_ERR_INVALID_ARGUMENT;
    TnLoadPlain al{a, lda, R, P};
    EpStore ep{c, ldc};
    return launch_gemm_tn(al, b, ldb, R, P, Q, ep, scratch, scratch_floats, (hipStream_t)stream);
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
    hipLaunchKernelGGL(row_stats_kernel<4>, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ld, D, M, eps, (float2*)stats);
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
