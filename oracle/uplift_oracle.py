"""CPU ORACLE (test infrastructure, NOT the product path) -- PyTorch-CPU restatement.

A plain restatement, op for op, of the reference's uplifting-transformer forward
(``common/net/vision_transformer.py``, ``common/net/uplift_upsample_transformer.py``) and of
the harness arithmetic around it (``eval.py:63-71,154-193``, ``common/dataset/metrics.py:13-37``,
``common/dataset/uplifiting_dataset.py:377-394``).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; the
shipped path (``uplift-upsample-3dhpe_amd/``) never does.

PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures, and its
arithmetic lives in TensorFlow 2.4.3 (``requirements.txt:3``), which is neither vendored
under ``/root/reference`` nor installable here.  This file therefore restates the
*published semantics* of the TF/Keras ops at the reference's call sites:

* ``Dense``: kernel ``(in, out)``, ``y = x @ W + b``.
* ``LayerNormalization`` (eps 1e-5 / 1e-6 < 1.001e-5 -> Keras' non-fused path):
  ``mean, var = moments(x, -1)`` (biased), ``inv = rsqrt(var + eps) * gamma``,
  ``y = x * inv + (beta - mean * inv)``.
* ``keras.activations.gelu`` (approximate=False): ``0.5 * x * (1 + erf(x / sqrt(2)))``.
* ``Conv1D`` ``valid``: kernel ``(k, in, out)``, ``y[t] = sum_j xpad[t*s + j] @ W[j] + b``.
* ``MaxPool1D(pool_size=1, strides=s)``: rows ``0, s, 2s, ...``.
* ``tf.nn.softmax`` over the last axis after adding ``mask * -1e9`` (finite, not -inf).

It is cross-checked against an independently written numpy restatement
(``oracle/uplift_oracle_np.py``) and against semantic invariants in ``tests/``.

The oracle is deliberately self-contained: hyper-parameters arrive as a plain dict and
weights as a ``{name: ndarray}`` dict in Keras layouts (names: see ``weight_names`` below).
"""
import math

import numpy as np
import torch


# --------------------------------------------------------------------------------------
# primitive ops
# --------------------------------------------------------------------------------------
def _t(a, dtype):
    return torch.as_tensor(np.asarray(a)).to(dtype)


def dense(x, w, b=None):
    y = torch.matmul(x, w)
    if b is not None:
        y = y + b
    return y


def layer_norm(x, gamma, beta, eps):
    # Keras non-fused path: tf.nn.moments + tf.nn.batch_normalization
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    inv = torch.rsqrt(var + eps) * gamma
    return x * inv + (beta - mean * inv)


def gelu_exact(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def _dropout(t, cfg, site, attention=False):
    """kl.Dropout in training mode with the masks of csrc/uu3d_dropout.h (oracle/dropout_oracle.py); cfg = dict(rate=DROP_RATE,
    attn_rate=ATTENTION_DROP_RATE, seed=...) or None (inference / no Dropout layers)."""
    if cfg is None:
        return t
    rate = cfg["attn_rate"] if attention else cfg["rate"]
    if rate <= 0:
        return t                                                       # Keras builds no layer for rate 0
    from oracle.dropout_oracle import drop_factor
    return t * torch.as_tensor(drop_factor(tuple(t.shape), rate, cfg["seed"], site)).to(t.dtype)


def mha(p, prefix, x, num_heads, mask=None, drop=None, site=0):
    """vision_transformer.py:132-156 (self-attention: v = k = q = x).  drop / site: Dropout on the attention weights (:127-128, site)
    and on the projection output (:153-154, site + 1)."""
    b, L, d = x.shape
    depth = d // num_heads
    q = dense(x, p[f"{prefix}/wq/kernel"], p.get(f"{prefix}/wq/bias"))
    k = dense(x, p[f"{prefix}/wk/kernel"], p.get(f"{prefix}/wk/bias"))
    v = dense(x, p[f"{prefix}/wv/kernel"], p.get(f"{prefix}/wv/bias"))

    def split(t):  # :92-97
        return t.reshape(b, L, num_heads, depth).permute(0, 2, 1, 3)

    q, k, v = split(q), split(k), split(v)
    logits = torch.matmul(q, k.transpose(-1, -2))                       # :117
    logits = logits / torch.sqrt(torch.tensor(float(depth), dtype=x.dtype))  # :119-120
    if mask is not None:
        logits = logits + mask * torch.tensor(-1e9, dtype=x.dtype)      # :122-123
    attn = torch.softmax(logits, dim=-1)                                # :126
    attn = _dropout(attn, drop, site, attention=True)                  # :127-128 (the returned weights are the dropped ones)
    out = torch.matmul(attn, v)                                         # :129
    out = out.permute(0, 2, 1, 3).reshape(b, L, d)                      # :147-150
    out = dense(out, p[f"{prefix}/projection/kernel"], p[f"{prefix}/projection/bias"])  # :151
    out = _dropout(out, drop, site + 1)                                 # :153-154
    return out, attn


def drop_path(x, rate, u):
    """vision_transformer.py:16-28 with the uniform draw `u` (one per leading-dim sample) given explicitly."""
    keep = 1.0 - rate
    gate = torch.floor(u + keep).reshape((-1,) + (1,) * (x.dim() - 1))
    return (x / keep) * gate


def transformer_block(p, prefix, x, num_heads, activation, mask=None, dp=None, drop=None, site=0, inner=False):
    """vision_transformer.py:176-195.  dp = (rate, u_attn, u_mlp) in training mode with rate > 0, else None.
    drop / site: the block's Dropout layers (site + 0 .. 3); inner: the MLP has an inner Dropout behind its activation (:63-64; the
    temporal blocks are built with inner_dropout = drop_rate, the spatial ones without, u_u_t.py:233-235,247-248)."""
    y = layer_norm(x, p[f"{prefix}/norm1/gamma"], p[f"{prefix}/norm1/beta"], 1e-5)
    y, attn = mha(p, f"{prefix}/attn", y, num_heads, mask, drop, site)
    if dp is not None:
        y = drop_path(y, dp[0], dp[1])
    x = x + y
    z = layer_norm(x, p[f"{prefix}/norm2/gamma"], p[f"{prefix}/norm2/beta"], 1e-5)
    z = dense(z, p[f"{prefix}/mlp/fc1/kernel"], p[f"{prefix}/mlp/fc1/bias"])
    z = activation(z)
    if inner:
        z = _dropout(z, drop, site + 2)                                 # :63-64
    z = dense(z, p[f"{prefix}/mlp/fc2/kernel"], p[f"{prefix}/mlp/fc2/bias"])
    z = _dropout(z, drop, site + 3)                                     # :65-66
    if dp is not None:
        z = drop_path(z, dp[0], dp[2])
    x = x + z
    return x, attn


def strided_transformer_block(p, prefix, x, pe, num_heads, stride, pad, dp=None, mask=None, drop=None, site=0):
    """uplift_upsample_transformer.py:122-160 with StridedMLP :81-90.  dp = (rate, u_attn, u_mlp): DropPath on both branches (:132-137).
    drop / site: Dropout on attention weights, projection output, hidden activations (:84-85) and convolution output (:88-89)."""
    assert x.shape[1] == pe.shape[0]                                    # :127
    x = x + pe                                                          # :128
    y = layer_norm(x, p[f"{prefix}/norm1/gamma"], p[f"{prefix}/norm1/beta"], 1e-5)
    y, attn = mha(p, f"{prefix}/attn", y, num_heads, mask, drop, site)
    if dp is not None:
        y = drop_path(y, dp[0], dp[1])                                  # :132-133
    x = x + y
    z = layer_norm(x, p[f"{prefix}/norm2/gamma"], p[f"{prefix}/norm2/beta"], 1e-5)
    # fc1: Conv1D k=1
    z = dense(z, p[f"{prefix}/mlp/fc1/kernel"][0], p[f"{prefix}/mlp/fc1/bias"])
    z = torch.relu(z)
    z = _dropout(z, drop, site + 2)                                     # :84-85
    # ZeroPadding1D(pad) + Conv1D(k=3, stride, 'valid')
    b, L, h = z.shape
    zp = torch.zeros(b, L + pad[0] + pad[1], h, dtype=z.dtype)
    zp[:, pad[0]:pad[0] + L] = z
    wk = p[f"{prefix}/mlp/strided_conv/kernel"]                         # (3, h, d)
    Lout = (L + pad[0] + pad[1] - 3) // stride + 1
    taps = []
    for j in range(3):
        rows = zp[:, j:j + (Lout - 1) * stride + 1:stride]              # (b, Lout, h)
        taps.append(torch.matmul(rows, wk[j]))
    z = taps[0] + taps[1] + taps[2] + p[f"{prefix}/mlp/strided_conv/bias"]
    z = _dropout(z, drop, site + 3)                                     # :88-89
    if dp is not None:
        z = drop_path(z, dp[0], dp[2])                                  # :136-137
    # residual path :138-156
    if stride > 1:
        identity = x
        if pad[0] == 0:
            identity = identity[:, 1:]
        if pad[1] == 0:
            identity = identity[:, :-1]
        identity = identity[:, ::stride]                                # MaxPool1D(pool 1, strides s)
    else:
        identity = x
    return identity + z, attn


# --------------------------------------------------------------------------------------
# the model forward
# --------------------------------------------------------------------------------------
def hp_from_arch(a):
    """The hyper-parameter dict of this oracle from the product's UpliftArch (plain attribute reads; test-side glue)."""
    return dict(num_frames=a.num_frames, num_keypoints=a.num_keypoints, d_spatial=a.d_spatial,
                d_temporal=a.d_temporal, spatial_depth=a.spatial_depth, temporal_depth=a.temporal_depth,
                strides=tuple(a.strides), paddings=tuple(a.paddings), num_heads=a.num_heads,
                has_strided_input=a.has_strided_input,
                first_strided_token_attention_layer=a.first_strided_token_attention_layer,
                full_output=a.full_output, output_bn=bool(getattr(a, "output_bn", False)))


def forward(hp, weights, x, stride_mask=None, dtype=torch.float32, return_attention=False):
    """``UpliftUpsampleTransformer.call`` (u_u_t.py:388-421), ``training=False``; numpy in, numpy out."""
    p = {k: _t(v, dtype) for k, v in weights.items()}
    full, central, att_list = forward_torch(hp, p, _t(x, dtype), stride_mask, dtype)
    full_np = None if full is None else full.numpy()
    if return_attention:
        return full_np, central.numpy(), [a.numpy() for a in att_list]
    return full_np, central.numpy()


def forward_torch(hp, p, x, stride_mask=None, dtype=torch.float32, drop_path_cfg=None, token_mask_cfg=None, bn_train=None, dropout_cfg=None):
    """Differentiable core on torch tensors (weights `p` may require grad).

    dropout_cfg (training mode with DROP_RATE / ATTENTION_DROP_RATE > 0): dict(rate, attn_rate, seed) -- every kl.Dropout layer of the
    reference with the counter-based masks of csrc/uu3d_dropout.h (oracle/dropout_oracle.py lists the sites); None = inference.

    bn_train (training mode with OUTPUT_BN, u_u_t.py:275-285): a dict that receives the updated moving statistics
    ("<layer>/moving_mean" / "/moving_variance"); the heads then normalise with the BATCH mean and biased variance (Keras' non-fused
    BatchNormalization on rank-3 inputs: tf.nn.moments over every axis but the last), moving = moving * 0.1 + batch * 0.9.  None = inference.
    token_mask_cfg (training mode, u_u_t.py:287-311,336-338; masked-token value: the learnable token when `p` holds one, else 0): dict(rate=TOKEN_MASK_RATE, u=(B, N) explicit U[0,1) draws).

    drop_path_cfg (training mode, vision_transformer.py:31-43): dict(rates=(spatial, temporal, strided),
    u_spatial (Ls, 2, B*N), u_temporal (Lt, 2, B), u_strided (len(strides), 2, B) -- the last only when rates[2] > 0) with explicit
    U[0,1) draws; None = inference.

    hp: dict with num_frames, num_keypoints, d_spatial, d_temporal, spatial_depth,
        temporal_depth, strides, paddings, num_heads, has_strided_input,
        first_strided_token_attention_layer, full_output.
    x: (B, N, J, 2); stride_mask: (B, N) bool, 1 = real input present.
    Returns (full (B,N,J,3) or None, central (B,J,3)) as numpy arrays of ``dtype``.
    """
    B, N, J, _ = x.shape
    H = hp["num_heads"]
    att_list = []

    def dp_for(stack, i, depth):
        if drop_path_cfg is None:
            return None
        rate = float(np.linspace(0, drop_path_cfg["rates"][stack], depth)[i])
        if rate == 0:
            return None                                               # no DropPath layer (vision_transformer.py:170)
        u = drop_path_cfg[("u_spatial", "u_temporal", "u_strided")[stack]]
        return (rate, torch.as_tensor(u[i, 0]).to(dtype), torch.as_tensor(u[i, 1]).to(dtype))

    # spatial_transformation :313-333
    if hp["spatial_depth"] == 0:
        x = x.reshape(B, N, J * 2)
    else:
        x = x.reshape(B * N, J, 2)
        x = dense(x, p["keypoint_embedding/kernel"], p["keypoint_embedding/bias"])
        x = x + p["spatial_pe/positional_encoding_weights"]
        x = _dropout(x, dropout_cfg, 1)                                 # token_dropout :324
        for i in range(hp["spatial_depth"]):
            x, _ = transformer_block(p, f"spatial_block_{i + 1}", x, H, gelu_exact, None, dp_for(0, i, hp["spatial_depth"]),
                                     dropout_cfg, 10 + 4 * i, inner=False)
        x = layer_norm(x, p["spatial_norm/gamma"], p["spatial_norm/beta"], 1e-6)
        x = x.reshape(B, N, J * hp["d_spatial"])                        # "(b n) p c -> b n (p c)"
    x = dense(x, p["spatial_to_temporal_fc/kernel"], p["spatial_to_temporal_fc/bias"])

    # temporal_transformation :335-367
    if token_mask_cfg is not None and token_mask_cfg["rate"] > 0:      # random_token_masking :287-311
        tmask = torch.as_tensor(np.asarray(token_mask_cfg["u"], np.float32) < np.float32(token_mask_cfg["rate"]))
        tmask = tmask & (torch.arange(N) != N // 2)[None, :]            # the central frame is never masked :291-294,303
        tmask = tmask.to(dtype)[..., None]
        value = p["learnable_masked_token_layer/learnable_masked_token"] if "learnable_masked_token_layer/learnable_masked_token" in p else 0.0   # :337
        x = x * (1.0 - tmask) + value * tmask                            # :310
    pe = p["temporal_pe/positional_encoding_weights"]
    inv = None
    if hp["has_strided_input"]:
        m = _t(np.asarray(stride_mask).astype(np.float32), dtype)
        inv = 1.0 - m
        tok = p["strided_input_token_layer/learnable_masked_token"]
        x = m[..., None] * x + inv[..., None] * tok                     # :350
    x = x + pe                                                          # :352
    for i in range(hp["temporal_depth"]):
        mask = None
        if hp["has_strided_input"] and i < hp["first_strided_token_attention_layer"]:
            mask = inv[:, None, None, :]                                # :361
        x, att = transformer_block(p, f"temporal_block_{i + 1}", x, H, torch.relu, mask, dp_for(1, i, hp["temporal_depth"]),
                                   dropout_cfg, 100 + 4 * i, inner=True)
        att_list.append(att)

    def batch_norm_training(t, name):
        red = tuple(range(t.dim() - 1))
        mean = t.mean(dim=red)
        var = ((t - mean) ** 2).mean(dim=red)                           # biased, like tf.nn.moments
        bn_train[f"{name}/moving_mean"] = (p[f"{name}/moving_mean"] * 0.1 + mean * 0.9).detach()
        bn_train[f"{name}/moving_variance"] = (p[f"{name}/moving_variance"] * 0.1 + var * 0.9).detach()
        return (t - mean) * torch.rsqrt(var + 1e-5) * p[f"{name}/gamma"] + p[f"{name}/beta"]

    def batch_norm_inference(t, name):
        if bn_train is not None:
            return batch_norm_training(t, name)
        # kl.BatchNormalization(momentum=0.1, epsilon=1e-5, axis=-1) with training=False (u_u_t.py:275-285,400-404,414-416):
        # tf.nn.batch_normalization with the moving statistics: x * (gamma * rsqrt(var + eps)) + (beta - mean * gamma * rsqrt(var + eps))
        inv = p[f"{name}/gamma"] * torch.rsqrt(p[f"{name}/moving_variance"] + 1e-5)
        return t * inv + (p[f"{name}/beta"] - p[f"{name}/moving_mean"] * inv)

    full = None
    if hp["full_output"] and hp["temporal_depth"] > 0:
        full = batch_norm_inference(x, "temporal_norm") if hp.get("output_bn") else x
        full = dense(full, p["temporal_fc/kernel"], p["temporal_fc/bias"])
        full = full.reshape(B, N, J, 3)                                 # "b n (p c) -> b n p c"

    # strided_temporal_transformation :369-386
    if len(hp["strides"]) > 0:
        for i, s in enumerate(hp["strides"]):
            smask = None
            if hp["temporal_depth"] == 0 and hp["has_strided_input"] and \
                    i < hp["first_strided_token_attention_layer"]:
                smask = inv[:, None, None, :]                           # :372-377 (broadcasts against the keys: right for the first block only)
            pe_i = p[f"strided_temporal_pe_{i + 1}/positional_encoding_weights"]
            x, _ = strided_transformer_block(p, f"strided_temporal_block_{i + 1}", x, pe_i, H,
                                             s, hp["paddings"][i], dp_for(2, i, len(hp["strides"])), smask, dropout_cfg, 200 + 4 * i)
        central = x
    else:
        central = x[:, N // 2: N // 2 + 1, :]
    if hp.get("output_bn"):
        central = batch_norm_inference(central, "strided_temporal_norm")
    central = dense(central, p["strided_temporal_fc/kernel"], p["strided_temporal_fc/bias"])
    assert central.shape[1] == 1                                        # einops n=1 at :416
    central = central.reshape(B, J, 3)

    return full, central, att_list


# --------------------------------------------------------------------------------------
# harness arithmetic around the model call
# --------------------------------------------------------------------------------------
def stride_mask_eval(num_frames, seq_stride, mask_stride, frame_index):
    """Global-aligned stride mask (uplifiting_dataset.py:377-384,394). 1 = real input."""
    mid = num_frames // 2
    idx = (np.arange(num_frames) - mid) * seq_stride + frame_index
    return np.equal(idx % mask_stride, 0)


def test_step(hp, weights, keypoints2d, stride_masks, dtype=torch.float32):
    """eval.py:63-71: the CALLER zeroes masked frames, then calls the model."""
    if hp["has_strided_input"]:
        masked = np.asarray(keypoints2d, np.float32) * \
            np.asarray(stride_masks).astype(np.float32)[:, :, None, None]
        return forward(hp, weights, masked, stride_masks, dtype)
    return forward(hp, weights, keypoints2d, None, dtype)


def eval_step_with_flip(hp, weights, keypoints2d, stride_masks, flip_order, dtype=torch.float32):
    """eval.py:152-180: test-time flip augmentation, averaged."""
    seq, cen = test_step(hp, weights, keypoints2d, stride_masks, dtype)
    kp = np.asarray(keypoints2d, np.float32)
    fl = np.concatenate([kp[..., :1] * -1.0, kp[..., 1:]], axis=-1)[:, :, flip_order]
    fseq, fcen = test_step(hp, weights, fl, stride_masks, dtype)
    fcen = np.concatenate([fcen[..., :1] * -1.0, fcen[..., 1:]], axis=-1)[:, flip_order]
    cen = (cen + fcen) / 2.0
    if seq is not None:
        fseq = np.concatenate([fseq[..., :1] * -1.0, fseq[..., 1:]], axis=-1)[:, :, flip_order]
        seq = (seq + fseq) / 2.0
    return seq, cen


def mpjpe(pred, gt, root_index, normalize=True):
    """metrics.py:13-37. pred (B,K,3), gt (B,K,4) with valid flag; float64 like the caller."""
    pred = np.asarray(pred, np.float64)
    gt = np.asarray(gt, np.float64)
    gt3d = gt[:, :, :3]
    valid = gt[:, :, 3] > 0
    gt3d = gt3d - gt3d[:, root_index, None, :]
    pred3d = pred - pred[:, root_index, None, :]
    dist = np.linalg.norm(pred3d - gt3d, ord=2, axis=-1)
    if normalize is False:
        return np.where(valid, dist, -1.0)
    dist = np.where(valid, dist, 0.0)
    return np.sum(dist) / float(np.sum(valid > 0.0))


def frame_mpjpe_mm(pred, gt3, root_index):
    """action_wise_eval.py:25-26,43 on root-shifted GT with the dummy valid flag (eval.py:185-196)."""
    gt3 = np.asarray(gt3, np.float64)
    gt3 = gt3 - gt3[:, root_index:root_index + 1, :]
    gt4 = np.concatenate([gt3, np.ones(gt3.shape[:-1] + (1,))], axis=-1)
    per_joint = mpjpe(pred, gt4, root_index, normalize=False) * 1000.0
    return per_joint, float(np.mean(per_joint[per_joint >= 0]))
