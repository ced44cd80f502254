"""Builds the TensorFlow cross-check kit: everything a machine WITH TensorFlow 2.4.3 + tensorflow-addons 0.13.0 and a checkout
of goldbricklemon/uplift-upsample-3dhpe needs to pin this repository's CPU oracle (and through it the HIP path) against
the reference's own arithmetic.  Needs only numpy + torch-CPU + this repository (no GPU, no TensorFlow):

    python tools/tf_crosscheck/make_kit.py --out /tmp/uu3d_tf_kit [--configs h36m_351 h36m_81] [--batch 4]

Per config it writes into <out>/:
    <cfg>.h5          seeded Keras-default weights (+ a perturbation so that biases / LayerNorm parameters are not 0 / 1), in the
                      Keras weight-file layout the reference's loader reads (common/utils/weight_io.py:76-263)
    <cfg>_io.npz      inputs (keypoints2d, stride masks: keyframe-aligned, centre-masked and all-masked rows), the oracle's
                      outputs in float32 and float64 (full, central), a training batch (3D targets, DropPath disabled) with the
                      oracle's loss and a few gradient tensors, one AdamW step of a small vector (tfa semantics, restated)
    config/<cfg>.json the unmodified config file
and copies check_with_tf.py + README.md next to them.  The files are NOT committed (41 MB of weights per config): the
script is deterministic, anyone can regenerate them.
"""
import argparse
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def build(cfgname, out, batch):
    import torch
    import uplift_upsample_3dhpe_amd as pkg
    from uplift_upsample_3dhpe_amd import synthetic
    from uplift_upsample_3dhpe_amd.utils import weight_io
    from oracle import uplift_oracle as O, train_oracle as T
    cfg = synthetic.load_config(cfgname)
    arch = pkg.arch_from_config(cfg)
    w = pkg.init_weights(arch, seed=0, perturb=0.1)
    spec = pkg.weight_spec(arch)
    weight_io.save_keras_h5(os.path.join(out, f"{cfgname}.h5"), w, spec)
    x, m = synthetic.synthetic_batch(cfg, batch, seed=0)
    xm = x * m[:, :, None, None].astype(np.float32)
    hp = O.hp_from_arch(arch)
    f32, c32 = O.forward(hp, w, xm, m, torch.float32)
    f64, c64 = O.forward(hp, w, xm, m, torch.float64)
    # training: masks without all-masked rows, DropPath off (TF's random draws cannot be matched)
    ms = cfg.MASK_STRIDE if isinstance(cfg.MASK_STRIDE, list) else [cfg.MASK_STRIDE]
    mt = np.stack([synthetic.eval_stride_mask(arch.num_frames, cfg.SEQUENCE_STRIDE, ms[i % len(ms)], 0) for i in range(batch)])
    gt = np.random.default_rng(50).normal(0, 0.3, size=(batch, arch.num_frames, arch.num_keypoints, 3)).astype(np.float32)
    loss, grads, _, _ = T.train_step_grads(hp, w, x, mt, gt, cfg.ROOT_KEYTPOINT, cfg.LOSS_WEIGHT_CENTER, cfg.LOSS_WEIGHT_SEQUENCE, batch, None)
    keep = ["temporal_block_1/attn/wq/kernel", "temporal_pe/positional_encoding_weights", "strided_temporal_block_1/mlp/strided_conv/kernel",
            "spatial_block_1/mlp/fc1/kernel", "strided_temporal_fc/bias", "strided_input_token_layer/learnable_masked_token"]
    rng = np.random.default_rng(3)
    v0 = rng.normal(0, 0.05, 1000).astype(np.float32); g0 = rng.normal(0, 1e-2, 1000).astype(np.float32)
    v1, m1, s1 = T.adamw_update(v0, np.zeros_like(v0), np.zeros_like(v0), g0, 2e-5, 2e-6, 0.9, 0.999, 1e-8, 1)
    np.savez(os.path.join(out, f"{cfgname}_io.npz"), keypoints2d=x, stride_masks=m, full_f32=f32, central_f32=c32, full_f64=f64, central_f64=c64,
             train_masks=mt, train_gt3d=gt, train_loss=np.float64(loss["loss"]), train_batch_size=np.int64(batch),
             adamw_var0=v0, adamw_grad=g0, adamw_var1=v1, adamw_m1=m1, adamw_v1=s1,
             **{"grad/" + k: grads[k] for k in keep if k in grads})
    os.makedirs(os.path.join(out, "config"), exist_ok=True)
    shutil.copy(os.path.join(ROOT, synthetic.CONFIGS[cfgname]), os.path.join(out, "config", f"{cfgname}.json"))
    return float(np.abs(f32 - f64).max())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--configs", nargs="+", default=["h36m_351", "h36m_81"])
    ap.add_argument("--batch", type=int, default=4)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    for c in a.configs:
        d = build(c, a.out, a.batch)
        print(f"{c}: kit written, oracle f32 vs f64 max-abs {d:.2e}")
    here = os.path.dirname(os.path.abspath(__file__))
    for f in ("check_with_tf.py", "README.md"):
        shutil.copy(os.path.join(here, f), os.path.join(a.out, f))


if __name__ == "__main__":
    main()
