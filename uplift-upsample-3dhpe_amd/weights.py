"""Weight inventory, Keras-default initialisation and interchange order.

The interchange contract follows the reference's Keras layer naming
(``common/net/uplift_upsample_transformer.py:198-285``) and the positional order inside
each top-level layer that its by-name ``.h5`` loader relies on
(``common/utils/weight_io.py:172-201,235``): norm1 gamma,beta; wq W,b; wk W,b; wv W,b;
projection W,b; norm2 gamma,beta; fc1 W,b; fc2 (or strided_conv) W,b.

Layouts are Keras layouts: Dense kernel ``(in, out)``, Conv1D kernel ``(k, in, out)``.
Initialisers are the Keras defaults: ``glorot_uniform`` kernels, zero biases, LayerNorm
gamma=1 / beta=0, positional encodings and the strided-input token
``TruncatedNormal(stddev=0.02)`` (u_u_t.py:30,45; truncation at two sigma by resampling).
"""
from collections import OrderedDict

import numpy as np

from .arch import UpliftArch


def _block_spec(prefix, d, h, strided, qkv_bias):
    s = [(f"{prefix}/norm1/gamma", (d,)), (f"{prefix}/norm1/beta", (d,))]
    for nm in ("wq", "wk", "wv"):
        s.append((f"{prefix}/attn/{nm}/kernel", (d, d)))
        if qkv_bias:
            s.append((f"{prefix}/attn/{nm}/bias", (d,)))
    s += [(f"{prefix}/attn/projection/kernel", (d, d)), (f"{prefix}/attn/projection/bias", (d,))]
    s += [(f"{prefix}/norm2/gamma", (d,)), (f"{prefix}/norm2/beta", (d,))]
    if strided:
        s += [(f"{prefix}/mlp/fc1/kernel", (1, d, h)), (f"{prefix}/mlp/fc1/bias", (h,)),
              (f"{prefix}/mlp/strided_conv/kernel", (3, h, d)), (f"{prefix}/mlp/strided_conv/bias", (d,))]
    else:
        s += [(f"{prefix}/mlp/fc1/kernel", (d, h)), (f"{prefix}/mlp/fc1/bias", (h,)),
              (f"{prefix}/mlp/fc2/kernel", (h, d)), (f"{prefix}/mlp/fc2/bias", (d,))]
    return s


def weight_spec(a: UpliftArch):
    """Ordered ``[(name, shape)]`` of every trainable tensor (``model.weights`` analogue).

    Order = layer creation order in the reference's ``__init__`` (u_u_t.py:196-285).
    """
    J, N, ds, dt = a.num_keypoints, a.num_frames, a.d_spatial, a.d_temporal
    spec = []
    if a.spatial_depth > 0:
        spec += [("keypoint_embedding/kernel", (2, ds)), ("keypoint_embedding/bias", (ds,))]
        spec += [("spatial_pe/positional_encoding_weights", (J, ds))]
    spec += [("temporal_pe/positional_encoding_weights", (N, dt))]
    for i in range(len(a.strides)):
        spec += [(f"strided_temporal_pe_{i + 1}/positional_encoding_weights", (a.strided_lengths[i], dt))]
    if getattr(a, "learnable_masked_token", False):       # LearnableMaskedTokenLayer without a name argument (u_u_t.py:219-220): Keras' default layer name
        spec += [("learnable_masked_token_layer/learnable_masked_token", (dt,))]
    if a.has_strided_input:
        spec += [("strided_input_token_layer/learnable_masked_token", (dt,))]
    if a.spatial_depth > 0:
        for i in range(a.spatial_depth):
            spec += _block_spec(f"spatial_block_{i + 1}", ds, a.h_spatial, False, a.qkv_bias)
        spec += [("spatial_norm/gamma", (ds,)), ("spatial_norm/beta", (ds,))]
    s2t_in = J * ds if a.spatial_depth > 0 else J * 2
    spec += [("spatial_to_temporal_fc/kernel", (s2t_in, dt)), ("spatial_to_temporal_fc/bias", (dt,))]
    for i in range(a.temporal_depth):
        spec += _block_spec(f"temporal_block_{i + 1}", dt, a.h_temporal, False, a.qkv_bias)
    for i in range(len(a.strides)):
        spec += _block_spec(f"strided_temporal_block_{i + 1}", dt, a.h_temporal, True, a.qkv_bias)
    bn = bool(getattr(a, "output_bn", False))
    h1 = a.full_output and a.temporal_depth > 0
    if h1:
        if bn:
            spec += [("temporal_norm/gamma", (dt,)), ("temporal_norm/beta", (dt,))]
        spec += [("temporal_fc/kernel", (dt, a.out_dim)), ("temporal_fc/bias", (a.out_dim,))]
    if bn:
        spec += [("strided_temporal_norm/gamma", (dt,)), ("strided_temporal_norm/beta", (dt,))]
    spec += [("strided_temporal_fc/kernel", (dt, a.out_dim)), ("strided_temporal_fc/bias", (a.out_dim,))]
    if bn:      # Keras' model.weights lists the non-trainable weights (BatchNorm moving statistics) after all trainable ones
        if h1:
            spec += [("temporal_norm/moving_mean", (dt,)), ("temporal_norm/moving_variance", (dt,))]
        spec += [("strided_temporal_norm/moving_mean", (dt,)), ("strided_temporal_norm/moving_variance", (dt,))]
    return spec


def count_params(a: UpliftArch) -> int:
    return int(sum(int(np.prod(shape)) for _, shape in weight_spec(a)))


def _glorot_uniform(rng, shape):
    if len(shape) == 2:
        fan_in, fan_out = shape
    else:  # Conv1D (k, in, out): receptive field multiplies both fans
        rf = int(np.prod(shape[:-2]))
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    limit = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=shape).astype(np.float32)


def _truncated_normal(rng, shape, stddev):
    out = rng.standard_normal(size=shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.standard_normal(size=int(bad.sum()))
        bad = np.abs(out) > 2.0
    return (out * stddev).astype(np.float32)


def init_weights(a: UpliftArch, seed: int = 0, perturb: float = 0.0) -> "OrderedDict[str, np.ndarray]":
    """Seeded Keras-default initialisation.

    ``perturb`` > 0 additionally randomises the tensors Keras would initialise to
    constants (biases, LayerNorm gamma/beta) so that tests exercise every term:
    bias ~ N(0, perturb), gamma ~ 1 + N(0, perturb), beta ~ N(0, perturb).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    w = OrderedDict()
    for name, shape in weight_spec(a):
        leaf = name.rsplit("/", 1)[1]
        if leaf == "kernel":
            w[name] = _glorot_uniform(rng, shape)
        elif leaf in ("positional_encoding_weights", "learnable_masked_token"):
            w[name] = _truncated_normal(rng, shape, 0.02)
        elif leaf == "gamma":
            w[name] = np.ones(shape, np.float32)
            if perturb > 0:
                w[name] += (rng.standard_normal(size=shape) * perturb).astype(np.float32)
        elif leaf in ("beta", "bias", "moving_mean"):
            w[name] = np.zeros(shape, np.float32)
            if perturb > 0:
                w[name] += (rng.standard_normal(size=shape) * perturb).astype(np.float32)
        elif leaf == "moving_variance":                    # Keras: ones; perturbed upwards only (a variance)
            w[name] = np.ones(shape, np.float32)
            if perturb > 0:
                w[name] += np.abs(rng.standard_normal(size=shape) * perturb).astype(np.float32)
        else:
            raise KeyError(name)
    return w
