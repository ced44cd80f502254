/*
 * uu3d.h -- C ABI of the MI355X-native uplift/upsample 3D-HPE transformer forward path.
 *
 * This is the drop-in boundary of the hot path.  The reference has
 * no FFI of its own: the path sits behind a Keras model object,
 *
 *     model = build_uplift_upsample_transformer(config)
 *         (/root/reference/common/net/uplift_upsample_transformer_constructor.py:14-50)
 *     full, central = model([x, stride_mask], training=False)
 *         (/root/reference/common/net/uplift_upsample_transformer.py:388-421)
 *
 * so every entry point below cites the reference call it replaces.  The host-side mirror
 * of the Keras interface (uplift-upsample-3dhpe_amd/net/...) binds exactly these symbols
 * through ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions: plain C, no exceptions cross the ABI, every function returns a uu3d_status
 * (0 = ok).  Pointers named *_dev are device (HBM) pointers on the model's device; all
 * other pointers are host pointers.  `stream` is a hipStream_t passed as void* (NULL =
 * the null stream).  Calls are stream-ordered and perform no host synchronisation unless
 * stated.  The caller owns every buffer it passes in; the model owns its weights.
 */
#ifndef UU3D_H_
#define UU3D_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UU3D_MAX_STRIDED 8

typedef enum uu3d_status {
    UU3D_OK = 0,
    UU3D_ERR_INVALID_ARGUMENT = 1, /* NULL pointer, bad size, unknown weight name ...        */
    UU3D_ERR_UNSUPPORTED = 2,      /* config is schema-legal but outside the compiled / generic kernels */
    UU3D_ERR_SHAPE = 3,            /* tensor element count does not match the model           */
    UU3D_ERR_NOT_READY = 4,        /* forward before every weight was set and committed       */
    UU3D_ERR_WORKSPACE = 5,        /* workspace too small / misaligned                        */
    UU3D_ERR_HIP = 6,              /* a HIP runtime call failed (see uu3d_last_error)          */
    UU3D_ERR_NO_DEVICE = 7,        /* no usable gfx950 device                                 */
    UU3D_ERR_RANGE = 8             /* uu3d_range_status: a forward produced non-finite outputs -- activations left the f16 range of the
                                      f16x3 products (|x| >= 65504), or its inputs were not finite                                    */
} uu3d_status;

/* Arithmetic the GEMM-shaped work is carried out in. */
typedef enum uu3d_precision {
    UU3D_PREC_F32 = 0,     /* f32-input MFMA (v_mfma_f32_32x32x2_f32 / 16x16x4_f32), exact f32 */
    UU3D_PREC_F16X3 = 1    /* forward GEMMs as 3 f16 MFMA passes on hi/lo-split operands: f32-grade error,
                              f16-rate matrix pipe (csrc/uu3d_gemm_h3.h).  Attention, spatial stack, LayerNorm,
                              softmax and every epilogue stay f32.  The training step runs its forward, input-gradient
                              and weight-gradient GEMMs the same way (loss-scaled, see uu3d_train_forward_backward);
                              UU3D_TRAIN_F32=1 / UU3D_TN_F32=1 in the environment keep them on the exact-f32 kernels. */
} uu3d_precision;

/*
 * Architecture hyper-parameters: the keyword arguments the reference constructor derives
 * from the config (uplift_upsample_transformer_constructor.py:15-43) and hands to
 * UpliftUpsampleTransformer.__init__ (uplift_upsample_transformer.py:165-177).
 */
typedef struct uu3d_config {
    int32_t num_frames;          /* SEQUENCE_LENGTH (token count N)                     */
    int32_t num_keypoints;       /* NUM_KEYPOINTS (J)                                   */
    int32_t d_spatial;           /* SPATIAL_EMBED_DIM                                   */
    int32_t d_temporal;          /* TEMPORAL_EMBED_DIM                                  */
    int32_t h_spatial;           /* int(d_spatial  * MLP_RATIO)                         */
    int32_t h_temporal;          /* int(d_temporal * MLP_RATIO)                         */
    int32_t spatial_depth;       /* SPATIAL_TRANSFORMER_BLOCKS                          */
    int32_t temporal_depth;      /* TEMPORAL_TRANSFORMER_BLOCKS                         */
    int32_t num_strided;         /* len(STRIDES)                                        */
    int32_t strides[UU3D_MAX_STRIDED];
    int32_t pad_left[UU3D_MAX_STRIDED];   /* PADDINGS[i][0] ([1,1] when PADDINGS is null) */
    int32_t pad_right[UU3D_MAX_STRIDED];  /* PADDINGS[i][1]                               */
    int32_t num_heads;           /* NUM_HEADS                                           */
    int32_t qkv_bias;            /* QKV_BIAS                                            */
    int32_t has_strided_input;   /* constructor.py:16-21                                */
    int32_t first_strided_token_attention_layer; /* FIRST_STRIDED_TOKEN_ATTENTION_LAYER */
    int32_t full_output;         /* not USE_REFINE                                      */
    int32_t precision;           /* uu3d_precision                                      */
    int32_t output_bn;           /* OUTPUT_BN: BatchNormalization(momentum 0.1, eps 1e-5) in front of both heads
                                    (uplift_upsample_transformer.py:275-285).  uu3d_forward: moving statistics, folded into
                                    the head operands at commit time; the training step: batch statistics (see there)   */
    int32_t learnable_masked_token; /* TOKEN_MASK_RATE > 0 and LEARNABLE_MASKED_TOKEN: the model owns one more weight,
                                    "learnable_masked_token_layer/learnable_masked_token" (d_temporal,), the value random token
                                    masking writes in training mode (uplift_upsample_transformer.py:38-50,219-220,337); unused
                                    at inference                                                                        */
} uu3d_config;

typedef struct uu3d_model uu3d_model;

/* Library / build information: "uu3d <version> gfx950 ...". Never NULL. */
const char* uu3d_version(void);
const char* uu3d_status_string(int status);
/* Human-readable detail of the last failure on this model (or of the last failed
 * uu3d_create when model == NULL).  Never NULL. */
const char* uu3d_last_error(const uu3d_model* model);

/*
 * Replaces: build_uplift_upsample_transformer(config) -> model
 * (uplift_upsample_transformer_constructor.py:14-50).  Creates the model on HIP device
 * `device` with all weight storage allocated but unset.
 * The specialised kernels are compiled for J = 17, SPATIAL_EMBED_DIM 32, TEMPORAL_EMBED_DIM 384, NUM_HEADS 8, MLP_RATIO 2 (every
 * shipped config).  Other dims the constructor accepts (:26-32) give a handle whose forward (uu3d_forward_ex) and training step run on
 * generic, untuned kernels: embed dims and MLP widths multiples of 4, head dims in {2, 4, 8, 12, 16, 24, 32, 48, 64}, <= 128 keypoints, <= 128 frames, >= 1 temporal and strided
 * block, the full-sequence head.  The generic attention keeps a head in LDS: the FORWARD holds 2 L (d_h + 4) floats (fits every legal shape), the training
 * step's BACKWARD 4 L (d_h + 4) + 2 L (L + 1) floats <= 160 KiB -- i.e. up to 128 frames only for head dims <= 8 (and 48, which has an MFMA backward;
 * 96 with ATTENTION_DROP_RATE > 0), 127 at 12, 124 at 16, 117 at 24, 111 at 32, 90 at 64; uu3d_train_forward_backward refuses longer sequences up front.  Outside that: UU3D_ERR_UNSUPPORTED.
 */
int uu3d_create(const uu3d_config* config, int device, uu3d_model** out_model);
void uu3d_destroy(uu3d_model* model);

/*
 * Weight inventory in the reference's creation order, Keras layouts
 * (Dense kernel (in,out), Conv1D kernel (k,in,out)); names follow the Keras layer names
 * given at uplift_upsample_transformer.py:198-285, e.g.
 * "temporal_block_1/attn/wq/kernel".  Replaces model.weights / get_weights / set_weights
 * (train.py:400,503) and the by-name h5 loader's target (common/utils/weight_io.py:172-235).
 */
int uu3d_num_weights(const uu3d_model* model);
int uu3d_weight_info(const uu3d_model* model, int index, const char** out_name,
                     int32_t* out_ndim, int64_t out_dims[4]);
/* Copy `numel` host floats into the named weight.  Stages on the host; nothing reaches
 * the device until uu3d_commit_weights. */
int uu3d_set_weight(uu3d_model* model, const char* name, const float* host_data, int64_t numel);
int uu3d_get_weight(const uu3d_model* model, const char* name, float* host_out, int64_t numel);
/* Repack every weight into the device layouts the kernels read ([N][K]-transposed,
 * padded GEMM operands, fused QKV) and upload.  Fails with UU3D_ERR_NOT_READY (and names
 * the first missing tensor in uu3d_last_error) if any weight was never set.
 * Synchronises `stream` before returning. */
int uu3d_commit_weights(uu3d_model* model, void* stream);

/* Bytes of device scratch uu3d_forward needs for `batch` sequences (256-byte aligned). */
size_t uu3d_workspace_bytes(const uu3d_model* model, int32_t batch);

/*
 * Replaces: model([x, stride_mask], training=False) -> (full_output, central_output)
 * (uplift_upsample_transformer.py:388-421; called at eval.py:70, train.py:520).
 *   kp2d_dev        (B, N, J, 2) f32.  The CALLER zeroes masked frames (eval.py:67).
 *   stride_mask_dev (B, N) uint8, 1 = real 2D input present; must be NULL iff the model
 *                   has no strided input.
 *   full_out_dev    (B, N, J, 3) f32, or NULL to skip head1 (must be NULL-able only when
 *                   full_output == 0 / temporal_depth == 0, where the reference returns None).
 *   central_out_dev (B, J, 3) f32.
 */
int uu3d_forward(uu3d_model* model, const float* kp2d_dev, const uint8_t* stride_mask_dev,
                 int32_t batch, float* full_out_dev, float* central_out_dev,
                 void* workspace_dev, size_t workspace_bytes, void* stream);
/* The same forward with return_attention=True (uplift_upsample_transformer.py:365,418-419: `att_list`, the softmax weights every temporal
 * block's MHA returns, vision_transformer.py:117-130): attn_out_dev_ptrs is a HOST array of temporal_depth device pointers, each
 * (B, num_heads, N, N) f32 (or NULL to skip that block); NULL = uu3d_forward.  The maps are recomputed from the block's q | k by a
 * separate kernel -- the attention kernels of the hot path never materialise them.  Sequences of up to ~800 tokens (K of one head in LDS). */
int uu3d_forward_attention(uu3d_model* model, const float* kp2d_dev, const uint8_t* stride_mask_dev, int32_t batch,
                           float* full_out_dev, float* central_out_dev, float* const* attn_out_dev_ptrs,
                           void* workspace, size_t workspace_bytes, void* stream);


/*
 * Replaces: metrics.mpjpe(pred, gt, root_index, normalize=False)
 * (common/dataset/metrics.py:13-37) on device, float64 like the reference's numpy call:
 * root-align pred and gt, per-joint L2 distance; -1 where gt valid flag <= 0.
 *   pred_dev (B, J, 3) f32; gt_dev (B, J, 4) f32 (x, y, z, valid); out_dev (B, J) f64.
 * This (B_local, J) block is the payload of the multi-GPU all-gather.
 */
int uu3d_mpjpe(const float* pred_dev, const float* gt_dev, int32_t batch, int32_t num_keypoints,
               int32_t root_index, double* out_dev, void* stream);

/*
 * "Next" row 3 of the scope table: the window / stride-mask generator as a gather over a RESIDENT pose table.
 * Replaces the per-sample slicing, padding, stride mask and flip of H36mSequenceGenerator
 * (common/dataset/uplifiting_dataset.py:322-407) plus the harness's `x * stride_mask` (eval.py:67, train.py:474).
 *   poses_dev (F, J, C) f32: all videos back to back (C = 2 for 2D keypoints, 3 for 3D ground truth);
 *   video_start_dev (V) i64 first row of each video, video_len_dev (V) i32;
 *   windows_dev (B) uu3d_window: video, centre frame, sampling stride, ABSOLUTE mask stride, mask shift (centre frame
 *   for globally aligned masks, rand_shift * stride in training, 0 otherwise), flip flag;
 *   flip_order_dev (J) i32 or NULL (AUGM_FLIP_KEYPOINT_ORDER); pad_edge: 1 = "copy" padding, 0 = zeros.
 *   out_dev (B, N, J, C); stride_mask_dev (B, N) u8 (1 = real input); pad_mask_dev (B, N) u8 or NULL (1 = frame exists).
 */
typedef struct uu3d_window { int32_t video, center, stride, mask_stride, mask_shift, flip; } uu3d_window;
int uu3d_gather_windows(const float* poses_dev, const int64_t* video_start_dev, const int32_t* video_len_dev,
                        const uu3d_window* windows_dev, const int32_t* flip_order_dev,
                        int32_t batch, int32_t num_frames, int32_t num_keypoints, int32_t channels,
                        int32_t pad_edge, int32_t zero_masked,
                        float* out_dev, uint8_t* stride_mask_dev, uint8_t* pad_mask_dev, void* stream);

/*
 * World -> camera coordinates -> 2D projection with the Human3.6M camera model, one camera per window: replaces
 * tf_world_to_cam_and_2d (common/dataset/uplifiting_dataset.py:669-761), the on-the-fly AMASS projection of training.
 *   world_dev (B, N, J, 3) f32; cams_dev (B, 18) f32 = quaternion wxyz | translation | 11 intrinsics (res, focal,
 *   centre, 3 radial, 2 tangential coefficients at [7..18) as the reference stores them);
 *   cam3d_dev (B, N, J, 3) or NULL; kp2d_dev (B, N, J, 2) or NULL.
 */
int uu3d_world_to_cam_2d(const float* world_dev, const float* cams_dev, int32_t batch, int32_t num_frames, int32_t num_keypoints,
                         float* cam3d_dev, float* kp2d_dev, void* stream);

/*
 * The schedule of a forward: what its launch shapes are chosen for.  Below 1024 token rows (B * N) the results are bit-identical either
 * way (the same products per element, computed by other workgroups); from 1024 rows on the throughput schedule takes the temporal chain
 * (csrc/uu3d_tchain64.h, rounds 5-6: one launch per temporal block for every row-local stage, residual stream and relu(fc1) on chip -- other summation orders and LayerNorm's affine part
 * folded into the next layer's weights: within 3e-5 of the latency schedule (measured 4e-6), same 1e-4 bar against the oracle; deterministic run
 * to run; return_attention keeps the round-4 launches.  UU3D_TCHAIN=0 in the environment of uu3d_create: never; UU3D_TCHAIN_MIN_TILES=n: from n
 * row tiles of 128 tokens on).
 *   UU3D_SCHEDULE_LATENCY: one batch at a time -- every launch spreads over as many CUs as pays for ITS duration;
 *   UU3D_SCHEDULE_THROUGHPUT: several independent batches in flight on different streams (pipeline.ForwardPipeline) -- the chip is
 *     shared between forwards, so a launch is shaped for the fewest CU-microseconds instead: the attention projection runs as 71
 *     workgroups x 12 column chunks instead of 213 x 4 (27 instead of 16 us alone, +2.4 % sequences/s with four batches in flight;
 *     DESIGN.md section 5).
 * uu3d_forward_ex takes it as an ARGUMENT of the call (round 4): it is a property of the enqueued / captured forward, not of the
 * model, so a model(...) call on one thread and a pipeline on another never see each other's choice.  attention_out as in
 * uu3d_forward_attention (NULL: none).  uu3d_forward / uu3d_forward_attention = uu3d_forward_ex with the model's DEFAULT schedule,
 * which uu3d_set_schedule changes (UU3D_SCHEDULE_LATENCY unless set; kept for callers that cannot pass the argument).
 */
#define UU3D_SCHEDULE_LATENCY 0
#define UU3D_SCHEDULE_THROUGHPUT 1
/*
 * RANGE CONTRACT of precision f16x3 (round 5).  The reference computes in float32 end to end; the f16x3 products split
 * every operand into two f16 planes, so an ACTIVATION of magnitude >= 65504 (LayerNorm outputs, q | k | v, attention context, ReLU(fc1),
 * the residual stream in front of the full-sequence head, the spatial stack's operands) becomes Inf in its hi plane and the sequence it
 * belongs to comes out NaN.  (Weights are checked when they are committed: uu3d_commit_weights fails with UU3D_ERR_RANGE.)  Keras-default
 * and trained weights keep activations at O(10); nothing in the format guarantees it.  Therefore:
 *   - every f16x3 forward ends with a check of its outputs (one small launch): non-finite values set a STICKY device word of the model;
 *   - uu3d_range_status(model, stream, &flag) synchronises `stream`, returns the word (flag 0 / 1; may be NULL) and clears it; its return
 *     value is UU3D_ERR_RANGE when the word was set, UU3D_OK otherwise -- the forward itself stays asynchronous and keeps returning
 *     launch errors only;
 *   - OR-ing UU3D_SCHEDULE_EXACT_F32 into `schedule` runs THIS call on the exact-f32 kernels whatever the handle's precision (f32-input
 *     MFMA: the whole float32 range; sequences of <= 128 tokens): the fallback for a batch that overflowed.
 * The Python model(...) does all three: it checks after the call and repeats an overflowed batch in exact f32 (or raises Uu3dRangeError
 * where that path does not exist); pipelines check once, at ForwardPipeline.check_range() / the end of run_eval.
 */
#define UU3D_SCHEDULE_EXACT_F32 0x100
int uu3d_range_status(uu3d_model* model, void* stream, int32_t* out_flag);
int uu3d_forward_ex(uu3d_model* model, const float* kp2d_dev, const uint8_t* stride_mask_dev, int32_t batch, float* full_out_dev,
                    float* central_out_dev, float* const* attention_out, void* workspace_dev, size_t workspace_bytes,
                    int32_t schedule, void* stream);
int uu3d_set_schedule(uu3d_model* model, int32_t schedule);

/*
 * Per-kernel timing of the next uu3d_forward calls with HIP events on the launch stream.
 * When enabled, uu3d_forward records an event pair around every launch; uu3d_profile_read
 * synchronises those events and returns the per-launch records of the LAST forward.
 */
typedef struct uu3d_profile_entry {
    char name[48];        /* launch label, e.g. "t1.ln_qkv"              */
    char kernel[32];      /* kernel family, e.g. "gemm_f32"              */
    float ms;             /* event-measured duration                      */
    double flops;         /* algorithmic FLOPs (2*MAC) of this launch     */
    double bytes;         /* algorithmic HBM bytes (operands in + out)    */
} uu3d_profile_entry;
int uu3d_set_profiling(uu3d_model* model, int32_t enabled);
int uu3d_profile_read(uu3d_model* model, uu3d_profile_entry* out_entries, int32_t capacity,
                      int32_t* out_count);

/* ------------------------------------------------------------------------------------------
 * The training step: loss, optimizer and EMA kernels first, then the
 * training-mode forward + backward pass (T2, uu3d_train_forward_backward).
 * ------------------------------------------------------------------------------------------ */

/*
 * T1 -- replaces the loss of train_step (train.py:464-494, common/utils/losses_3d.py:13-14):
 *   gt      = gt3d - gt3d[:, :, root]                 (root shift, :467)
 *   central = sum ||pred_central - gt[:, N/2]||_2 / (batch_size_norm * J)
 *   seq     = sum ||pred_full    - gt        ||_2 / (batch_size_norm * N * J)
 *   loss    = w_center * central + w_seq * seq        (pred_full_dev == NULL: (w_center + w_seq) * central, :491-494)
 * and returns d loss / d pred (what tape.gradient feeds into the network's backward).
 *   pred_full_dev (B,N,J,3) or NULL, pred_central_dev (B,J,3), gt3d_dev (B,N,J,3) absolute 3D poses.
 *   batch_size_norm is config.BATCH_SIZE (the GLOBAL batch: per-rank sums simply add up).
 *   loss_out_dev[3] = {loss, central, seq}; grad_* may be NULL; scratch_dev >= 4096 floats.
 * Deterministic (fixed-order two-stage reduction).
 */
int uu3d_mpjpe_loss(const float* pred_full_dev, const float* pred_central_dev, const float* gt3d_dev,
                    int32_t batch, int32_t num_frames, int32_t num_keypoints, int32_t root_index,
                    float w_center, float w_seq, int32_t batch_size_norm,
                    float* loss_out_dev, float* grad_full_dev, float* grad_central_dev,
                    float* scratch_dev, void* stream);

/*
 * T3 -- replaces optimizer.apply_gradients for tfa.optimizers.AdamW (train.py:404-415,499) on
 * one flat parameter buffer: decoupled decay first, then the Keras/TF ApplyAdam update
 *   var -= wd * var
 *   alpha = lr * sqrt(1 - beta2^t) / (1 - beta1^t)              (t = step = iterations + 1)
 *   m += (g - m) * (1 - beta1);  v += (g*g - v) * (1 - beta2)
 *   var -= (m * alpha) / (sqrt(v) + epsilon)
 * vhat_dev != NULL selects Keras Adam's amsgrad=True form (TF ApplyAdamWithAmsgrad; the config class default
 * OPTIMIZER_PARAMS {"amsgrad": True}, uplift_upsample_transformer_config.py:88):
 *   vhat = max(vhat, v);  var -= (m * alpha) / (sqrt(vhat) + epsilon)
 * lr and wd are the schedule values at `iterations` (both ExponentialDecay for the shipped configs).
 * HBM-bound: 28 bytes per parameter (read var, g, m, v; write var, m, v), 36 with vhat.
 */
int uu3d_adamw_update(float* var_dev, float* m_dev, float* v_dev, float* vhat_dev, const float* grad_dev, int64_t n,
                      float lr, float wd, float beta1, float beta2, float epsilon, int64_t step,
                      void* stream);
/* The same update, skipped ON THE DEVICE (weights and moments untouched) when *skip_flag_dev != 0 -- no host synchronisation.
 * skip_flag_dev = uu3d_train_nonfinite_flag(model): uu3d_train_forward_backward clears it and its loss-scaled backward pass
 * (f16x3 gradient GEMMs, train.py:477,498 replaced) raises it when a finished gradient range holds a non-finite value; NULL =
 * uu3d_adamw_update.  A skipped step keeps TF's semantics of "no update" only approximately: the host-side iteration counter
 * (bias correction, schedules) still advances. */
int uu3d_adamw_update_guarded(float* var_dev, float* m_dev, float* v_dev, float* vhat_dev, const float* grad_dev, int64_t n,
                              float lr, float wd, float beta1, float beta2, float epsilon, int64_t step,
                              const uint32_t* skip_flag_dev, void* stream);
/* Device address of the model's non-finite-gradient word (valid after uu3d_train_init, until the next uu3d_train_init /
 * uu3d_destroy); NULL without training state. */
uint32_t* uu3d_train_nonfinite_flag(const uu3d_model* model);
/* Host-side read of that word (synchronises the device): *out = 1 when the last backward pass flagged non-finite gradients. */
int uu3d_train_nonfinite(uu3d_model* model, int32_t* out);


/* T4 -- replaces the EMA update of train_step (train.py:502-504): ema -= (1 - decay) * (ema - w). */
int uu3d_ema_update(float* ema_dev, const float* w_dev, int64_t n, float decay, void* stream);

/*
 * T2 -- replaces `with tf.GradientTape(): model(..., training=True)` + `tape.gradient(loss, model.trainable_variables)`
 * (train.py:477-498).  The optimizer owns ONE flat float32 master buffer of all parameters in inventory order,
 * Keras layouts (uu3d_num_params floats); gradients are returned in a buffer of the same layout.
 *   uu3d_train_init     uploads the model's current weights into params_dev and builds the device-side operand packs
 *   uu3d_train_repack   regenerates the packs from params_dev (call after every optimizer step)
 *   uu3d_train_export   params_dev -> host weights -> inference operands (so uu3d_forward evaluates the trained model)
 *   uu3d_train_forward_backward: training-mode forward (DropPath, vision_transformer.py:16-43), the loss of
 *       uu3d_mpjpe_loss, and the full backward pass.
 *       drop_path_rates[3] = DROP_PATH_RATE (spatial, temporal, strided); per block rate linspace(0, rate, depth)
 *       drop_path_uniform_dev: U[0,1) draws, layout [spatial blocks][2][B*N] then [temporal blocks][2][B]
 *                              (two draws per block: attention branch, MLP branch); NULL disables DropPath.
 *       full_out_dev / central_out_dev may be NULL.  loss_out_dev[3] = {loss, central, sequence}.
 *       gt3d_dev == NULL: training-mode FORWARD ONLY (model(inputs, training=True) outside a tape, train.py:478);
 *       loss_out_dev / grads_dev may then be NULL.  drop_path_rates[2] != 0: DropPath inside the strided blocks, its draws behind the
 *       temporal stack's ([strided blocks][2][B]).
 *       uu3d_config.output_bn: the heads' BatchNormalization runs in TRAINING mode (uplift_upsample_transformer.py:275-285: batch mean
 *       and biased variance of the head's input over all rows) and -- the one exception to `const` -- the moving statistics, the last
 *       tensors of params_dev, are updated in place (moving = 0.1 moving + 0.9 batch), as Keras does inside the training-mode call; they are
 *       not trainable: their slots of grads_dev are zeros and the optimizer must leave them alone (trainer.Trainer steps the prefix in
 *       front of them).
 *   uu3d_train_set_grad_callback: `fn(user, first, count, stream)` is called on the calling host thread from inside
 *       uu3d_train_forward_backward each time the range [first, first + count) of grads_dev is final; every kernel that
 *       writes it has been enqueued on `stream` before the call (a bucketed all-reduce makes its communication stream wait
 *       for `stream` and overlaps with the rest of the backward pass).  Ranges are disjoint and cover the buffer.
 */
typedef void (*uu3d_grad_ready_fn)(void* user, int64_t first, int64_t count, void* stream);
int uu3d_train_set_grad_callback(uu3d_model* model, uu3d_grad_ready_fn fn, void* user);
/*
 * The Dropout layers of the model in TRAINING mode (config DROP_RATE / ATTENTION_DROP_RATE: kl.Dropout in
 * common/net/vision_transformer.py:57-58,63-67 (MLP: behind the activation and behind fc2), :87-90,127-128 (attention weights),
 * :153-154 (projection output); common/net/uplift_upsample_transformer.py:78-79,84-89 (StridedMLP), :201,324 (token_dropout behind
 * the keypoint embedding + positional encoding)).  Applies to the following uu3d_train_forward_backward* calls of this model
 * (forward-only calls included) until changed; rates 0 (the default, and every shipped config) = no layer.  An element is kept iff
 * u >= rate, kept elements are scaled by 1 / (1 - rate) (Keras); u is a counter-based function of (seed, layer site, element
 * index) -- csrc/uu3d_dropout.h lists the sites -- so the backward pass recomputes the masks instead of storing them and a CPU
 * restatement can evaluate the same masks (oracle/dropout_oracle.py).  Give every step a fresh seed.  With a rate > 0 the
 * spatial stack runs as its unfused chain of launches and attention on the generic kernels (the fused / MFMA kernels have no
 * Dropout sites): correct, slower.
 */
int uu3d_train_set_dropout(uu3d_model* model, float drop_rate, float attention_drop_rate, uint64_t seed);
int64_t uu3d_num_params(const uu3d_model* model);
int uu3d_train_init(uu3d_model* model, float* params_dev, void* stream);
int uu3d_train_repack(uu3d_model* model, const float* params_dev, void* stream);
int uu3d_train_export(uu3d_model* model, const float* params_dev, void* stream);
size_t uu3d_train_workspace_bytes(const uu3d_model* model, int32_t batch);
int uu3d_train_forward_backward(uu3d_model* model, const float* params_dev, const float* kp2d_dev,
                                const uint8_t* stride_mask_dev, const float* gt3d_dev, int32_t batch,
                                int32_t batch_size_norm, float w_center, float w_seq, int32_t root_index,
                                const float* drop_path_rates, const float* drop_path_uniform_dev,
                                float* loss_out_dev, float* full_out_dev, float* central_out_dev,
                                float* grads_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

/* The same step with random token masking (uplift_upsample_transformer.py:287-311, 336-338; TOKEN_MASK_RATE > 0): token (b, n) entering the temporal transformer -- spatial_to_temporal_fc's output, before the
 * strided-input token blend and the positional encoding -- is replaced by 0 (by the learnable masked token when the model was
 * created with uu3d_config.learnable_masked_token, whose gradient is the sum of d x over the replaced rows) where
 * token_mask_uniform_dev[b * N + n] < token_mask_rate, never at the central frame n = N / 2.  token_mask_uniform_dev: B * N draws of U[0, 1) (the caller's generator,
 * like drop_path_uniform_dev); NULL or rate 0: no masking (= uu3d_train_forward_backward). */
int uu3d_train_forward_backward_masked(uu3d_model* model, const float* params_dev, const float* kp2d_dev,
                                       const uint8_t* stride_mask_dev, const float* gt3d_dev, int32_t batch,
                                       int32_t batch_size_norm, float w_center, float w_seq, int32_t root_index,
                                       const float* drop_path_rates, const float* drop_path_uniform_dev,
                                       const float* token_mask_uniform_dev, float token_mask_rate,
                                       float* loss_out_dev, float* full_out_dev, float* central_out_dev,
                                       float* grads_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UU3D_H_ */
