/*
 * uu3d_ops.h -- building-block operators of the training step's backward pass,
 * exported so that each one can be checked on its own against autograd of the CPU oracle.
 * They replace what TensorFlow's tape.gradient (train.py:477,498) dispatches for the layers of
 * common/net/vision_transformer.py and common/net/uplift_upsample_transformer.py:
 *   Dense / Conv1D kernel gradients (X^T dY), bias and positional-encoding gradients (column sums),
 *   LayerNormalization backward, softmax-attention forward/backward.
 * Same conventions as uu3d.h: device pointers, stream-ordered, int status, no host sync.
 */
#ifndef UU3D_OPS_H_
#define UU3D_OPS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* C[P][Q] = A[R][P]^T B[R][Q]  (dW = X^T dY).  scratch_dev holds split-K slabs (>= uu3d_op_scratch_floats()). */
int uu3d_op_gemm_tn(const float* a_dev, int32_t lda, const float* b_dev, int32_t ldb, int32_t R, int32_t P, int32_t Q,
                    float* c_dev, int32_t ldc, float* scratch_dev, size_t scratch_floats, void* stream);
/* The same product in the f16x3 arithmetic of the training step (gemm_tn_h3_kernel: f16 MFMA on hi / lo planes, fragments
 * read transposed from LDS) when P, Q >= 128; narrower results fall back to the exact-f32 kernel. */
int uu3d_op_gemm_tn_h3(const float* a_dev, int32_t lda, const float* b_dev, int32_t ldb, int32_t R, int32_t P, int32_t Q,
                       float* c_dev, int32_t ldc, float* scratch_dev, size_t scratch_floats, void* stream);
/* C[M][N] = A[M][K] W[N][K]^T (dX = dY W^T with W in Keras (in,out) layout: N = in, K = out).
 * N % 64 == 0 and K % 32 == 0 (the training path keeps padded copies). */
int uu3d_op_gemm_nt(const float* a_dev, int32_t lda, const float* w_dev, int32_t ldw, int32_t M, int32_t N, int32_t K,
                    float* c_dev, int32_t ldc, float* scratch_dev, size_t scratch_floats, void* stream);
/* out[(r % period)][c] (+)= sum_r X[r][c] over rows with mask[r] == want (mask NULL: all rows). period 0 = 1. */
int uu3d_op_colsum(const float* x_dev, int32_t ldx, int32_t R, int32_t C, int32_t period, const uint8_t* mask_dev,
                   int32_t want, float* out_dev, int32_t accumulate, float* scratch_dev, size_t scratch_floats, void* stream);
/* (mean, 1/sqrt(var+eps)) per row. */
int uu3d_op_row_stats(const float* x_dev, int32_t ld, int32_t D, int32_t M, float eps, float* stats_dev, void* stream);
/* LayerNormalization backward: dx (written or accumulated), dgamma, dbeta (written). */
int uu3d_op_ln_bwd(const float* x_dev, const float* dy_dev, const float* stats_dev, const float* gamma_dev, int32_t ld,
                   int32_t D, int32_t M, float* dx_dev, int32_t accumulate, float* dgamma_dev, float* dbeta_dev,
                   float* scratch_dev, size_t scratch_floats, void* stream);
/* softmax attention over rows [q|k|v] (ld floats per row, head h at channels h*dh..), head dim 4 or 48,
 * L <= 128 forward, L <= 96 backward (UU3D_ERR_UNSUPPORTED beyond: P and dS of a head live in LDS) */
int uu3d_op_attn_fwd(const float* qkv_dev, int32_t ld, int32_t D, int32_t B, int32_t L, int32_t H, int32_t head_dim,
                     const uint8_t* key_mask_dev, float* out_dev, int32_t ldo, void* stream);
int uu3d_op_attn_bwd(const float* qkv_dev, const float* dout_dev, int32_t ld, int32_t D, int32_t B, int32_t L, int32_t H,
                     int32_t head_dim, const uint8_t* key_mask_dev, float* dqkv_dev, int32_t ldo, void* stream);
size_t uu3d_op_scratch_floats(void);

/* The forward's row-panel path for a LayerNorm-fed Dense layer on its own (csrc/uu3d_gemm_panel.h; in the model:
 * kl.LayerNormalization + kl.Dense of vit.TransformerBlock / vit.MLP, vision_transformer.py:46-68,135-137,183,188):
 *   out[M][N] = LayerNorm(x[M][384]; gamma, beta, eps) W[384][N] + bias          (relu = 0: f32, row stride ldo)
 *   relu = 1: ReLU of that, written as two f16 planes hi[M][N] | lo[M][N] at out_dev (value = hi + lo / 2048)
 * W comes as the fragment-ordered f16 planes made by uu3d_op_panel_pack from the Keras (in, out) kernel on the HOST;
 * a_scratch_dev holds the fragment-ordered LayerNorm output (uu3d_op_panel_a_bytes(M)).  N % 32 == 0, N <= 1024 * 32 / 32. */
size_t uu3d_op_panel_operand_bytes(int32_t N);
size_t uu3d_op_panel_a_bytes(int32_t M);
int uu3d_op_panel_pack(const float* w_host, int32_t N, void* operand_dev, void* stream);
int uu3d_op_ln_dense_panel(const float* x_dev, int32_t ldx, int32_t M, const float* gamma_dev, const float* beta_dev, float eps,
                           const void* operand_dev, const float* bias_dev, int32_t N, int32_t relu, void* a_scratch_dev,
                           void* out_dev, int32_t ldo, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UU3D_OPS_H_ */
