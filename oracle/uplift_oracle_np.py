"""CPU ORACLE TWIN (test infrastructure, NOT the product path) -- numpy restatement.

An independently written, loop-heavy numpy restatement of the same reference forward
(``common/net/uplift_upsample_transformer.py:388-421`` and the layers it calls).  It exists
only to cross-check ``oracle/uplift_oracle.py``: the two share no code, use different
formulations (per-head einsum loops, textbook LayerNorm, per-output-position strided
conv), and must agree to rounding in float64.  PARITY UNPINNED against TensorFlow (see
``uplift_oracle.py`` header).  Slow: use at small batch only.
"""
import math

import numpy as np

try:  # exact erf without scipy dependency at import time
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover
    _erf = np.vectorize(math.erf)


def _ln(x, g, b, eps):
    mu = x.mean(-1, keepdims=True)
    var = x.var(-1, keepdims=True)  # biased
    return (x - mu) / np.sqrt(var + eps) * g + b


def _softmax(z):
    z = z - z.max(-1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(-1, keepdims=True)


def _attention(w, pre, x, H, key_penalty=None):
    B, L, d = x.shape
    dh = d // H
    q = x @ w[f"{pre}/wq/kernel"] + w.get(f"{pre}/wq/bias", 0.0)
    k = x @ w[f"{pre}/wk/kernel"] + w.get(f"{pre}/wk/bias", 0.0)
    v = x @ w[f"{pre}/wv/kernel"] + w.get(f"{pre}/wv/bias", 0.0)
    out = np.empty_like(q)
    for h in range(H):  # head h owns the contiguous channel chunk [h*dh, (h+1)*dh)
        sl = slice(h * dh, (h + 1) * dh)
        logits = np.einsum("bqc,bkc->bqk", q[..., sl], k[..., sl]) / math.sqrt(dh)
        if key_penalty is not None:  # (B, L) additive term per key
            logits = logits + key_penalty[:, None, :]
        out[..., sl] = np.einsum("bqk,bkc->bqc", _softmax(logits), v[..., sl])
    return out @ w[f"{pre}/projection/kernel"] + w[f"{pre}/projection/bias"]


def _block(w, pre, x, H, act, key_penalty=None):
    x = x + _attention(w, f"{pre}/attn", _ln(x, w[f"{pre}/norm1/gamma"], w[f"{pre}/norm1/beta"], 1e-5),
                       H, key_penalty)
    z = _ln(x, w[f"{pre}/norm2/gamma"], w[f"{pre}/norm2/beta"], 1e-5)
    z = act(z @ w[f"{pre}/mlp/fc1/kernel"] + w[f"{pre}/mlp/fc1/bias"])
    return x + (z @ w[f"{pre}/mlp/fc2/kernel"] + w[f"{pre}/mlp/fc2/bias"])


def _strided_block(w, pre, x, pe, H, s, pad, key_penalty=None):
    x = x + pe[None]
    x = x + _attention(w, f"{pre}/attn", _ln(x, w[f"{pre}/norm1/gamma"], w[f"{pre}/norm1/beta"], 1e-5), H, key_penalty)
    z = _ln(x, w[f"{pre}/norm2/gamma"], w[f"{pre}/norm2/beta"], 1e-5)
    z = np.maximum(z @ w[f"{pre}/mlp/fc1/kernel"][0] + w[f"{pre}/mlp/fc1/bias"], 0.0)
    B, L, hdim = z.shape
    Lp = L + pad[0] + pad[1]
    Lout = (Lp - 3) // s + 1
    wk, bk = w[f"{pre}/mlp/strided_conv/kernel"], w[f"{pre}/mlp/strided_conv/bias"]
    out = np.zeros((B, Lout, wk.shape[2]), x.dtype)
    for t in range(Lout):
        acc = np.broadcast_to(bk, (B, wk.shape[2])).astype(x.dtype).copy()
        for j in range(3):
            src = t * s + j - pad[0]          # index into the unpadded sequence
            if 0 <= src < L:
                acc += z[:, src] @ wk[j]
        # residual: trimmed sequence, every s-th row
        lo = 1 if pad[0] == 0 else 0
        out[:, t] = acc + x[:, lo + t * s]
    # the trimmed/pooled identity must have exactly Lout rows
    hi = L - (1 if pad[1] == 0 else 0)
    assert len(range(lo, hi, s)) == Lout
    return out


def forward(hp, weights, x, stride_mask=None, dtype=np.float64):
    w = {k: np.asarray(v, dtype) for k, v in weights.items()}
    x = np.asarray(x, dtype)
    B, N, J, _ = x.shape
    H = hp["num_heads"]
    gelu = lambda t: 0.5 * t * (1.0 + _erf(t / math.sqrt(2.0)))
    relu = lambda t: np.maximum(t, 0.0)

    if hp["spatial_depth"] > 0:
        tok = np.empty((B, N, J, hp["d_spatial"]), dtype)
        for n in range(N):  # frames are independent
            f = x[:, n] @ w["keypoint_embedding/kernel"] + w["keypoint_embedding/bias"]
            f = f + w["spatial_pe/positional_encoding_weights"]
            for i in range(hp["spatial_depth"]):
                f = _block(w, f"spatial_block_{i + 1}", f, H, gelu)
            tok[:, n] = _ln(f, w["spatial_norm/gamma"], w["spatial_norm/beta"], 1e-6)
        feat = tok.reshape(B, N, J * hp["d_spatial"])  # joint-major flatten
    else:
        feat = x.reshape(B, N, J * 2)
    t = feat @ w["spatial_to_temporal_fc/kernel"] + w["spatial_to_temporal_fc/bias"]

    penalty = None
    if hp["has_strided_input"]:
        m = np.asarray(stride_mask).astype(dtype)
        t = np.where(m[..., None] > 0, t, w["strided_input_token_layer/learnable_masked_token"])
        penalty = (1.0 - m) * np.asarray(-1e9, dtype)
    t = t + w["temporal_pe/positional_encoding_weights"]
    for i in range(hp["temporal_depth"]):
        kp = penalty if (hp["has_strided_input"] and i < hp["first_strided_token_attention_layer"]) else None
        t = _block(w, f"temporal_block_{i + 1}", t, H, relu, kp)

    def bn(v, name):          # BatchNormalization at inference (u_u_t.py:275-285): moving statistics, eps 1e-5
        return (v - w[f"{name}/moving_mean"]) / np.sqrt(w[f"{name}/moving_variance"] + 1e-5) * w[f"{name}/gamma"] + w[f"{name}/beta"]

    full = None
    if hp["full_output"] and hp["temporal_depth"] > 0:
        f = bn(t, "temporal_norm") if hp.get("output_bn") else t
        full = (f @ w["temporal_fc/kernel"] + w["temporal_fc/bias"]).reshape(B, N, J, 3)

    if len(hp["strides"]) > 0:
        for i, s in enumerate(hp["strides"]):
            kp = penalty if (hp["temporal_depth"] == 0 and hp["has_strided_input"] and i < hp["first_strided_token_attention_layer"]) else None
            t = _strided_block(w, f"strided_temporal_block_{i + 1}", t,
                               w[f"strided_temporal_pe_{i + 1}/positional_encoding_weights"], H, s,
                               hp["paddings"][i], kp)
        c = t[:, 0]
    else:
        c = t[:, N // 2]
    if hp.get("output_bn"):
        c = bn(c, "strided_temporal_norm")
    central = (c @ w["strided_temporal_fc/kernel"] + w["strided_temporal_fc/bias"]).reshape(B, J, 3)
    return full, central
