"""Host-side mirrors of the reference's optimizer / schedule / loss objects for the training step
(train.py:396-415,464-506), over the C-ABI kernels uu3d_mpjpe_loss / uu3d_adamw_update / uu3d_ema_update.

All state lives in flat float32 device buffers in the model's weight order (Keras layouts), which is
what the fused AdamW kernel updates in one launch; back-propagation itself is trainer.Trainer.
"""
import ctypes as C
import math

import numpy as np

from . import _capi


class ExponentialDecay(object):
    """keras.optimizers.schedules.ExponentialDecay (config SCHEDULE "ExponentialDecay", train.py:404)."""

    def __init__(self, initial_learning_rate, decay_steps, decay_rate, staircase=False, name=None):
        self.initial_learning_rate, self.decay_steps = initial_learning_rate, decay_steps
        self.decay_rate, self.staircase = decay_rate, staircase

    def __call__(self, step):
        p = np.float32(step) / np.float32(self.decay_steps)
        if self.staircase:
            p = np.floor(p)
        return float(np.float32(self.initial_learning_rate) * np.power(np.float32(self.decay_rate), np.float32(p), dtype=np.float32))


class ExponentialDecayWithSteps(object):
    """common/utils/schedules.py:36-99."""

    def __init__(self, initial_learning_rate, decay_steps, decay_rate, large_decay_steps, large_decay_rate, name=None):
        self.initial_learning_rate, self.decay_steps, self.decay_rate = initial_learning_rate, decay_steps, decay_rate
        self.large_decay_steps, self.large_decay_rate = large_decay_steps, large_decay_rate

    def __call__(self, step):
        f = np.float32
        p = np.floor(f(step) / f(self.decay_steps))
        large_p = np.floor(f(step) / f(self.large_decay_steps))
        decayed = f(self.initial_learning_rate) * np.power(f(self.decay_rate), f(p - large_p), dtype=f)
        return float(f(decayed * np.power(f(self.large_decay_rate), f(large_p), dtype=f)))


def scheduler_by_name(name):
    """common/utils/schedules.py:17-33 (the two schedules the shipped configs use)."""
    if name == "ExponentialDecay":
        return ExponentialDecay
    if name == "ExponentialDecayWithSteps":
        return ExponentialDecayWithSteps
    raise NotImplementedError(name)


def ema_decay_value(ema_decay, global_step):
    """train.py:554-556."""
    return min(ema_decay, (1.0 + global_step) / (10.0 + global_step))


class AdamW(object):
    """tfa.optimizers.AdamW(weight_decay, learning_rate, beta_1=.9, beta_2=.999, epsilon) on ONE flat
    float32 device tensor of parameters (train.py:404-415).  ``weight_decay`` / ``learning_rate`` may be
    numbers or schedules called with ``iterations`` (0-based), like the Keras optimizer does."""

    def __init__(self, params, weight_decay, learning_rate, beta_1=0.9, beta_2=0.999, epsilon=1e-7, amsgrad=False):
        import torch
        if params.dtype != torch.float32 or not params.is_cuda or not params.is_contiguous():
            raise ValueError("params must be a contiguous float32 tensor on the ROCm device (no CPU fallback)")
        self._torch = torch
        self._lib = _capi.load_library()
        self.params = params.view(-1)
        self.m = torch.zeros_like(self.params)
        self.v = torch.zeros_like(self.params)
        # Keras Adam(amsgrad=True): one more slot, vhat = max(vhat, v) (the config class default, config.py:88)
        self.amsgrad = bool(amsgrad)
        self.vhat = torch.zeros_like(self.params) if self.amsgrad else None
        self.weight_decay, self.learning_rate = weight_decay, learning_rate
        self.beta_1, self.beta_2, self.epsilon = beta_1, beta_2, epsilon
        self.iterations = 0

    def _value(self, x):
        return float(x(self.iterations)) if callable(x) else float(x)

    def apply_gradients(self, grads, skip_flag_ptr=None):
        """skip_flag_ptr: device address of a word that, when non-zero, makes the update a no-op on the device
        (uu3d_train_nonfinite_flag: the backward pass found non-finite gradients)."""
        torch = self._torch
        g = grads.view(-1)
        if g.shape != self.params.shape or g.dtype != torch.float32 or g.device != self.params.device:
            raise ValueError("grads must match params (flat float32, same device)")
        lr, wd = self._value(self.learning_rate), self._value(self.weight_decay)
        stream = torch.cuda.current_stream(self.params.device).cuda_stream
        st = self._lib.uu3d_adamw_update_guarded(C.c_void_p(self.params.data_ptr()), C.c_void_p(self.m.data_ptr()),
                                                 C.c_void_p(self.v.data_ptr()),
                                                 C.c_void_p(self.vhat.data_ptr()) if self.amsgrad else None, C.c_void_p(g.contiguous().data_ptr()),
                                                 self.params.numel(), lr, wd, self.beta_1, self.beta_2, self.epsilon,
                                                 self.iterations + 1, C.c_void_p(skip_flag_ptr) if skip_flag_ptr else None, C.c_void_p(stream))
        _capi.check(self._lib, st, None)
        self.iterations += 1


def ema_update(ema, weights, decay):
    """ema_w.assign_sub((1 - ema_decay) * (ema_w - w)) (train.py:502-504) on flat device tensors."""
    import torch
    lib = _capi.load_library()
    if ema.shape != weights.shape or ema.dtype != torch.float32 or not ema.is_cuda:
        raise ValueError("ema / weights must be matching float32 device tensors")
    stream = torch.cuda.current_stream(ema.device).cuda_stream
    st = lib.uu3d_ema_update(C.c_void_p(ema.data_ptr()), C.c_void_p(weights.contiguous().data_ptr()), ema.numel(),
                             float(decay), C.c_void_p(stream))
    _capi.check(lib, st, None)


def train_loss(pred_full, pred_central, gt3d, config, want_grads=True):
    """Loss of train_step and d loss / d predictions (train.py:467-494).

    pred_full (B,N,J,3) or None, pred_central (B,J,3), gt3d (B,N,J,3) ABSOLUTE poses (root shift is
    applied inside, :467).  Returns (loss[3] = {loss, central, sequence}, grad_full, grad_central)."""
    import torch
    lib = _capi.load_library()
    B, N, J = gt3d.shape[0], gt3d.shape[1], gt3d.shape[2]
    dev = pred_central.device
    gt3d = gt3d.to(torch.float32).contiguous()
    pc = pred_central.to(torch.float32).contiguous()
    pf = None if pred_full is None else pred_full.to(torch.float32).contiguous()
    loss = torch.empty(3, dtype=torch.float32, device=dev)
    scratch = torch.empty(4096, dtype=torch.float32, device=dev)
    gf = torch.empty_like(pf) if (want_grads and pf is not None) else None
    gc = torch.empty_like(pc) if want_grads else None
    stream = torch.cuda.current_stream(dev).cuda_stream
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = lib.uu3d_mpjpe_loss(p(pf), p(pc), p(gt3d), B, N, J, int(config.ROOT_KEYTPOINT),
                             float(config.LOSS_WEIGHT_CENTER), float(config.LOSS_WEIGHT_SEQUENCE),
                             int(config.BATCH_SIZE), p(loss), p(gf), p(gc), p(scratch), C.c_void_p(stream))
    _capi.check(lib, st, None)
    return loss, gf, gc
