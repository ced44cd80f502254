"""Synthetic workloads of BASELINE.md section 3 / SURVEY.md section 8(d): the shipped configs by name, seeded 2D
keypoints ~ U(-1, 1) and eval-style (globally aligned) stride masks.  Used by bench.py, __graft_entry__.smoke() and the
tests; no dataset or checkpoint exists offline."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = {"h36m_351": "config/h36m_351.json", "h36m_81": "config/h36m_81.json",
           "h36m_351_pt": "config/h36m_351_pt.json", "amass_351": "config/amass_351.json"}

# max-abs tolerance on fp32 3D joints and |delta MPJPE| budget (mm) stated by BASELINE.json's north_star
TOL_MAX_ABS = 1e-4
TOL_MPJPE_MM = 0.05


def load_config(name):
    """One of the reference's unmodified config files.  "dense_351" is SURVEY.md 8(d)'s stress shape -- NOT a shipped
    config: h36m_351 with SEQUENCE_LENGTH 351, SEQUENCE_STRIDE 1, STRIDES [3, 9, 13], PADDINGS [[0, 0]] x 3
    (351 -> 117 -> 13 -> 1), i.e. temporal attention over 351 tokens."""
    from .net.uplift_upsample_transformer_config import UpliftUpsampleConfig
    if name == "dense_351":
        cfg = UpliftUpsampleConfig(os.path.join(ROOT, CONFIGS["h36m_351"]))
        cfg.SEQUENCE_LENGTH, cfg.SEQUENCE_STRIDE = 351, 1
        cfg.STRIDES, cfg.PADDINGS = [3, 9, 13], [[0, 0], [0, 0], [0, 0]]
        return cfg
    return UpliftUpsampleConfig(os.path.join(ROOT, CONFIGS[name]))


def eval_stride_mask(num_frames, seq_stride, mask_stride, frame_index):
    """Global-aligned stride mask (uplifiting_dataset.py:377-384,394); 1 = real input present."""
    idx = (np.arange(num_frames) - num_frames // 2) * seq_stride + frame_index
    return np.equal(idx % mask_stride, 0)


def synthetic_batch(cfg, batch, seed=0, mask_specs=None):
    """2D keypoints ~ U(-1,1) and eval-style stride masks.

    mask_specs: list of (mask_stride, frame_index) cycled over the batch; default exercises
    keyframe-aligned, centre-masked and all-masked rows.
    """
    rng = np.random.default_rng(seed)
    N, J = cfg.SEQUENCE_LENGTH, cfg.NUM_KEYPOINTS
    x = rng.uniform(-1.0, 1.0, size=(batch, N, J, 2)).astype(np.float32)
    s_out = cfg.SEQUENCE_STRIDE
    strides = cfg.MASK_STRIDE if isinstance(cfg.MASK_STRIDE, list) else [cfg.MASK_STRIDE]
    if mask_specs is None:
        mask_specs = [(strides[0], 0), (strides[1], 0), (strides[2], s_out), (strides[1], s_out),
                      (strides[0], 1 if s_out > 1 else 0), (strides[2], 0)]
    m = np.stack([eval_stride_mask(N, s_out, *mask_specs[i % len(mask_specs)]) for i in range(batch)])
    return x, m
