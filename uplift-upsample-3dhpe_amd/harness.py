"""Device-side harness arithmetic around the model call (the reference's eval protocol).

* ``per_joint_error``  -- metrics.mpjpe(normalize=False) (common/dataset/metrics.py:13-37) as a
  HIP kernel in float64; its (B_local, J) output is the payload of the multi-GPU all-gather.
* ``test_step``        -- eval.py:63-71: zero masked frames, call the model.
* ``eval_step_with_flip`` -- eval.py:152-180: test-time flip augmentation, averaged.
* ``stride_mask_eval`` -- uplifiting_dataset.py:377-384,394 (global-aligned stride mask).
"""
import ctypes as C

import numpy as np

from . import _capi


def stride_mask_eval(num_frames, seq_stride, mask_stride, frame_index):
    idx = (np.arange(num_frames) - num_frames // 2) * seq_stride + frame_index
    return np.equal(idx % mask_stride, 0)


def stride_masks_train(num_frames, seq_stride, mask_strides, batch, rng, rand_shift=True):
    """Training-mode stride masks: per sample an absolute mask stride drawn from MASK_STRIDE
    (uplifiting_dataset.py:329-337) and a random shift of the mask grid (:386-392); 1 = real input."""
    if not isinstance(mask_strides, (list, tuple)):
        mask_strides = [mask_strides]
    mid = num_frames // 2
    out = np.zeros((batch, num_frames), bool)
    for b in range(batch):
        ams = int(mask_strides[rng.integers(0, len(mask_strides))]) if len(mask_strides) > 1 else int(mask_strides[0])
        r = ams // seq_stride
        idx = (np.arange(num_frames) - mid) * seq_stride
        if rand_shift:
            max_shift = int(np.ceil((r - 1) / 2))
            idx = idx + int(rng.integers(-max_shift, max_shift, endpoint=(r % 2 != 0))) * seq_stride
        out[b] = np.equal(idx % ams, 0)
    return out


def per_joint_error(pred, gt, root_index, out=None):
    """pred (B,J,3) f32, gt (B,J,4) f32 [x,y,z,valid] on the GPU -> (B,J) f64 metres, -1 = invalid."""
    import torch
    lib = _capi.load_library()
    B, J = pred.shape[0], pred.shape[1]
    pred = pred.to(torch.float32).contiguous()
    gt = gt.to(torch.float32).contiguous()
    if gt.shape != (B, J, 4):
        raise ValueError("gt must be (B, J, 4) with the valid flag last")
    if out is None:
        out = torch.empty((B, J), dtype=torch.float64, device=pred.device)
    stream = torch.cuda.current_stream(pred.device).cuda_stream
    st = lib.uu3d_mpjpe(C.c_void_p(pred.data_ptr()), C.c_void_p(gt.data_ptr()), B, J, int(root_index),
                        C.c_void_p(out.data_ptr()), C.c_void_p(stream))
    _capi.check(lib, st, None)
    return out


def test_step(model, keypoints2d, stride_masks):
    if model.has_strided_input:
        masked = keypoints2d * stride_masks[:, :, None, None].to(keypoints2d.dtype)
        return model([masked, stride_masks], training=False)
    return model(keypoints2d, training=False)


def _flip(t, order, joint_axis):
    import torch
    t = torch.cat([t[..., :1] * -1.0, t[..., 1:]], dim=-1)
    return t.index_select(joint_axis, order)


def eval_step_with_flip(model, keypoints2d, stride_masks, flip_order):
    import torch
    order = torch.as_tensor(flip_order, dtype=torch.long, device=keypoints2d.device)
    seq, cen = test_step(model, keypoints2d, stride_masks)
    fseq, fcen = test_step(model, _flip(keypoints2d, order, 2), stride_masks)
    cen = (cen + _flip(fcen, order, 1)) / 2.0
    if seq is not None:
        seq = (seq + _flip(fseq, order, 2)) / 2.0
    return seq, cen
