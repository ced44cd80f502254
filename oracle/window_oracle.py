"""TEST INFRASTRUCTURE ONLY: numpy restatement of the reference's per-sample window generator, one window at a time.

Follows /root/reference/common/dataset/uplifiting_dataset.py: H36mSequenceGenerator.__init__ :270-296 (sample list),
next_epoch_iterator :303-428 (frame-rate multiplier :317-321, mask-stride draw :323-334, window bounds and padding
:336-375, stride mask :377-394, flip :402-412).  PARITY UNPINNED (the reference has no tests or fixtures for it).
"""
import math

import numpy as np


def sample_list(video_lens, frame_rates, subsample, flip_augment, in_batch_augment):
    """(video, centre frame, do_flip, frame_rate) rows in the reference's order."""
    rows = []
    for s_i, n in enumerate(video_lens):
        pos = np.arange(0, n, subsample)
        blk = [(s_i, int(p), 0, int(frame_rates[s_i])) for p in pos]
        if not in_batch_augment and flip_augment:
            blk = blk + [(s_i, int(p), 1, int(frame_rates[s_i])) for p in pos]
        rows += blk
    return np.array(rows, dtype=np.int64).reshape(-1, 4)


def one_window(video, i, seq_len, stride, pad_type, abs_mask_stride, shift_mode, shift_value, do_flip, flip_order):
    """video (F, J, C) -> (window (N, J, C), pad mask (N,), stride mask (N,)); shift_mode in {'global', 'rand', None}."""
    left = (seq_len - 1) * stride // 2
    right = (seq_len - 1) * stride - left
    n = video.shape[0]
    begin, end = i - left, i + right + 1
    pad_l = pad_r = 0
    if begin < 0:
        pad_l = math.ceil(-begin / stride)
        begin = begin + pad_l * stride
    if end > n:
        pad_r = math.ceil((end - n) / stride)
        end = end - pad_r * stride
    seq = video[begin:end:stride]
    mask = np.ones(seq.shape[0], np.float32)
    if pad_l or pad_r:
        seq = np.pad(seq, ((pad_l, pad_r), (0, 0), (0, 0)), mode=pad_type)
        mask = np.pad(mask, (pad_l, pad_r), mode="constant")
    idx = (np.arange(seq_len) - seq_len // 2) * stride
    if shift_mode == "global":
        idx = idx + i
    elif shift_mode == "rand":
        idx = idx + shift_value * stride
    stride_mask = np.equal(idx % abs_mask_stride, 0)
    assert seq.shape[0] == seq_len
    if do_flip:
        seq = seq[:, flip_order].copy()
        seq[..., 0] *= -1
    return seq, mask, stride_mask


def world_to_cam_and_2d(seq3d, cam):
    """One window: seq3d (N, J, 3) world coordinates, cam (18,) = quaternion | translation | 11 intrinsics -> (camera-space 3D,
    2D), float64 (the reference slices cam[7:19]; its vectors hold 18 values, uplifiting_dataset.py:506-515).
    Follows uplifiting_dataset.py:669-761 (tf_world_to_cam :713-716 with tf_qrot :697-705 / tf_qinverse :707-711,
    tf_project_to_2d :735-761)."""
    q, t, intr = cam[:4], cam[4:7], cam[7:19]
    qi = np.concatenate([q[:1], -q[1:]])
    v = seq3d - t
    qv = np.broadcast_to(qi[1:], v.shape)
    uv = np.cross(qv, v)
    uuv = np.cross(qv, uv)
    xc = v + 2 * (qi[0] * uv + uuv)
    f, c, k, p = intr[2:4], intr[4:6], intr[6:9], intr[9:11]
    xx = np.clip(xc[..., :2] / xc[..., 2:], -1.0, 1.0)
    r2 = (xx ** 2).sum(-1, keepdims=True)
    radial = 1 + (k * np.concatenate([r2, r2 ** 2, r2 ** 3], -1)).sum(-1, keepdims=True)
    tan = (p * xx).sum(-1, keepdims=True)
    return xc, f * (xx * (radial + tan) + p * r2) + c
