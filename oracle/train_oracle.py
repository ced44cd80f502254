"""CPU ORACLE (test infrastructure, NOT the product path) -- training-step arithmetic without
back-propagation: the loss of train_step (train.py:464-494, common/utils/losses_3d.py:13-14), the
learning-rate schedules (keras ExponentialDecay; common/utils/schedules.py:36-99), the
tensorflow-addons AdamW dense update (train.py:404-415,499) and the EMA update (train.py:502-504,554-556).

PARITY UNPINNED against TensorFlow: tensorflow==2.4.3 and tensorflow-addons==0.13.0
(requirements.txt:3-4) are not vendored and cannot be installed here.  The optimizer restates the
published algorithm of the pinned versions:
  tfa DecoupledWeightDecayExtension.apply_gradients -> `var.assign_sub(wd_t * var)` before the base
  optimizer's update; Keras Adam (non-amsgrad) -> TF's ApplyAdam functor
      alpha = lr * sqrt(1 - beta2^t) / (1 - beta1^t),  t = iterations + 1
      m += (g - m) * (1 - beta1);  v += (g*g - v) * (1 - beta2);  var -= (m * alpha) / (sqrt(v) + eps)
  all in float32, lr/wd = schedule(iterations).
"""
import numpy as np
import torch

f32 = np.float32


def exponential_decay(initial, decay_steps, decay_rate, step, staircase=False):
    """keras.optimizers.schedules.ExponentialDecay: initial * rate ** (step / decay_steps) (floored if staircase)."""
    p = f32(step) / f32(decay_steps)
    if staircase:
        p = np.floor(p)
    return f32(initial) * np.power(f32(decay_rate), f32(p), dtype=f32)


def exponential_decay_with_steps(initial, decay_steps, decay_rate, large_decay_steps, large_decay_rate, step):
    """common/utils/schedules.py:76-99."""
    p = np.floor(f32(step) / f32(decay_steps))
    large_p = np.floor(f32(step) / f32(large_decay_steps))
    p = p - large_p
    decayed = f32(initial) * np.power(f32(decay_rate), f32(p), dtype=f32)
    return f32(decayed * np.power(f32(large_decay_rate), f32(large_p), dtype=f32))


def ema_decay_value(ema_decay, global_step):
    """train.py:554-556."""
    return f32(min(ema_decay, (1.0 + global_step) / (10.0 + global_step)))


def train_loss(pred_full, pred_central, gt3d, root, w_center, w_seq, batch_size_norm, with_grad=True):
    """train.py:467-494 in float32; gradients w.r.t. the predictions via autograd (tape.gradient)."""
    gt = torch.as_tensor(np.asarray(gt3d, f32))
    gt = gt - gt[:, :, root:root + 1, :]
    N, J = gt.shape[1], gt.shape[2]
    pc = torch.as_tensor(np.asarray(pred_central, f32)).clone().requires_grad_(with_grad)
    central = torch.linalg.norm(gt[:, N // 2] - pc, dim=-1).sum() / (batch_size_norm * J)
    pf = None
    if pred_full is not None:
        pf = torch.as_tensor(np.asarray(pred_full, f32)).clone().requires_grad_(with_grad)
        seq = torch.linalg.norm(gt - pf, dim=-1).sum() / (batch_size_norm * N * J)
        loss = (w_center * central) + (w_seq * seq)
    else:
        seq = torch.zeros(())
        loss = (w_center + w_seq) * central
    out = dict(loss=float(loss.detach()), central=float(central.detach()), seq=float(seq.detach()))
    if with_grad:
        loss.backward()
        out["grad_central"] = pc.grad.numpy()
        out["grad_full"] = None if pf is None else pf.grad.numpy()
    return out


def adamw_update(var, m, v, g, lr, wd, beta1, beta2, eps, step, vhat=None):
    """One tfa-AdamW dense update in float32 numpy (separately rounded ops). Returns (var, m, v), or (var, m, v, vhat)
    when `vhat` is given: Keras Adam(amsgrad=True) -> TF's ApplyAdamWithAmsgrad functor, vhat = max(vhat, v) and
    sqrt(vhat) in the denominator."""
    var, m, v, g = (np.asarray(a, f32).copy() for a in (var, m, v, g))
    lr, wd, beta1, beta2, eps = f32(lr), f32(wd), f32(beta1), f32(beta2), f32(eps)
    var = var - wd * var
    b1p = f32(float(beta1) ** float(step))          # beta^t rounded once from double (numpy's float32 power is an ulp off at some t)
    b2p = f32(float(beta2) ** float(step))
    alpha = f32(lr * np.sqrt(f32(1) - b2p, dtype=f32) / (f32(1) - b1p))
    m = m + (g - m) * (f32(1) - beta1)
    v = v + (g * g - v) * (f32(1) - beta2)
    if vhat is not None:
        vhat = np.maximum(np.asarray(vhat, f32), v)
        var = var - (m * alpha) / (np.sqrt(vhat, dtype=f32) + eps)
        return var, m, v, vhat
    var = var - (m * alpha) / (np.sqrt(v, dtype=f32) + eps)
    return var, m, v


def ema_update(ema, w, decay):
    ema, w = np.asarray(ema, f32), np.asarray(w, f32)
    return ema - (f32(1) - f32(decay)) * (ema - w)


def train_step_grads(hp, weights, keypoints2d, stride_masks, keypoints3d, root, w_center, w_seq, batch_size_norm,
                     drop_path_cfg=None, dtype=torch.float64, token_mask_cfg=None, bn_train=None, dropout_cfg=None):
    """Loss and d loss / d weights of train_step (train.py:464-498) by autograd through the forward oracle.

    keypoints2d (B,N,J,2) raw, stride_masks (B,N) bool, keypoints3d (B,N,J,3) absolute.  Returns
    (dict loss/central/seq, {name: grad ndarray}, full, central).  bn_train: a dict (OUTPUT_BN in training mode) that receives the updated
    moving statistics, see uplift_oracle.forward_torch.  dropout_cfg: dict(rate, attn_rate, seed) -- the Dropout layers with the
    counter-based masks of the HIP library (oracle/dropout_oracle.py).
    """
    from oracle import uplift_oracle as O
    p = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True) for k, v in weights.items()}
    x = torch.tensor(np.asarray(keypoints2d), dtype=dtype)
    if hp["has_strided_input"]:
        x = x * torch.tensor(np.asarray(stride_masks).astype(np.float64), dtype=dtype)[:, :, None, None]   # train.py:474
    full, central, _ = O.forward_torch(hp, p, x, stride_masks if hp["has_strided_input"] else None, dtype, drop_path_cfg, token_mask_cfg, bn_train, dropout_cfg)
    gt = torch.tensor(np.asarray(keypoints3d), dtype=dtype)
    gt = gt - gt[:, :, root:root + 1, :]
    N, J = gt.shape[1], gt.shape[2]
    cen = torch.linalg.norm(gt[:, N // 2] - central, dim=-1).sum() / (batch_size_norm * J)
    seq = torch.linalg.norm(gt - full, dim=-1).sum() / (batch_size_norm * N * J)
    loss = (w_center * cen) + (w_seq * seq)
    loss.backward()
    grads = {k: (v.grad.numpy() if v.grad is not None else np.zeros(v.shape)) for k, v in p.items()}
    return dict(loss=float(loss.detach()), central=float(cen.detach()), seq=float(seq.detach())), grads, \
        full.detach().numpy(), central.detach().numpy()
