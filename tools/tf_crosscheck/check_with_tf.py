"""Runs on a machine with TensorFlow 2.4.3 (+ tensorflow-addons 0.13.0, einops 0.3.2: the reference's requirements.txt) and a
checkout of goldbricklemon/uplift-upsample-3dhpe.  Compares the REFERENCE model with what this repository's CPU oracle produced
for the same weights and inputs (files written by make_kit.py):

    python check_with_tf.py --reference /path/to/uplift-upsample-3dhpe --kit . [--configs h36m_351 h36m_81]

Checks, per config:
  1. weight loading: common/utils/weight_io.load_weights_with_callback accepts <cfg>.h5 (names, order, shapes);
  2. forward: model([x * mask, mask], training=False) vs the oracle's float32 outputs -- bar 1e-4 max-abs (north_star), and
     vs the float64 twin on rows with at least one real token;
  3. training: loss and a few gradient tensors of train.py:464-498's arithmetic (DropPath layers see training=False here:
     TF's draws cannot be reproduced; the kit's gradients were made with DropPath off) -- bar 1e-4 of each tensor's scale;
  4. tfa.optimizers.AdamW single step on a 1000-vector vs the restated update (bit-exact expected).
Prints one PASS / FAIL line per check and exits non-zero on any FAIL.  Please send the output back to the repository's
maintainers: a PASS on (2) lifts "parity unpinned" for SURVEY rows A1-A9, (3)-(4) for T1-T3.
"""
import argparse
import os
import sys

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", required=True)
    ap.add_argument("--kit", default=".")
    ap.add_argument("--configs", nargs="+", default=["h36m_351", "h36m_81"])
    a = ap.parse_args()
    sys.path.insert(0, a.reference)
    import tensorflow as tf
    from common.net.uplift_upsample_transformer_config import UpliftUpsampleConfig
    from common.net.uplift_upsample_transformer_constructor import build_uplift_upsample_transformer
    from common.utils import weight_io
    ok = True

    def report(name, passed, detail):
        nonlocal ok
        ok = ok and passed
        print(("PASS" if passed else "FAIL"), name, detail, flush=True)

    for cfgname in a.configs:
        z = np.load(os.path.join(a.kit, f"{cfgname}_io.npz"))
        cfg = UpliftUpsampleConfig(config_file=os.path.join(a.kit, "config", f"{cfgname}.json"))
        B = int(z["keypoints2d"].shape[0])
        cfg.BATCH_SIZE = B                                  # the model is built for a static batch (constructor.py:44-49)
        model = build_uplift_upsample_transformer(config=cfg)
        weight_io.load_weights_with_callback(model, filepath=os.path.join(a.kit, f"{cfgname}.h5"), skip_mismatch=False, verbose=False)
        report(f"{cfgname} weights", True, f"{len(model.weights)} tensors loaded by name")
        x = tf.constant(z["keypoints2d"]); m = tf.constant(z["stride_masks"])
        xm = x * tf.cast(m[:, :, tf.newaxis, tf.newaxis], tf.float32)
        full, central = model([xm, m], training=False)
        e32 = max(float(np.abs(full.numpy() - z["full_f32"]).max()), float(np.abs(central.numpy() - z["central_f32"]).max()))
        rows = z["stride_masks"].any(axis=1)
        e64 = max(float(np.abs(full.numpy() - z["full_f64"])[rows].max()), float(np.abs(central.numpy() - z["central_f64"])[rows].max()))
        report(f"{cfgname} forward", e32 <= 1e-4 and e64 <= 1e-4, f"max-abs vs oracle f32 {e32:.3e}, vs f64 (rows with a real token) {e64:.3e}")
        # training arithmetic of train.py:464-498 with DropPath inactive
        mt = tf.constant(z["train_masks"]); gt = tf.constant(z["train_gt3d"])
        with tf.GradientTape() as tape:
            g3 = gt - gt[:, :, cfg.ROOT_KEYTPOINT:cfg.ROOT_KEYTPOINT + 1, :]
            pf, pc = model([x * tf.cast(mt[:, :, tf.newaxis, tf.newaxis], tf.float32), mt], training=False)
            N = cfg.SEQUENCE_LENGTH
            cen = tf.reduce_sum(tf.norm(g3[:, N // 2] - pc, axis=-1)) / (B * cfg.NUM_KEYPOINTS)
            seq = tf.reduce_sum(tf.norm(g3 - pf, axis=-1)) / (B * N * cfg.NUM_KEYPOINTS)
            loss = cfg.LOSS_WEIGHT_CENTER * cen + cfg.LOSS_WEIGHT_SEQUENCE * seq
        grads = tape.gradient(loss, model.trainable_variables)
        byname = {v.name.split(":")[0]: g for v, g in zip(model.trainable_variables, grads)}
        rel = abs(float(loss) - float(z["train_loss"])) / abs(float(z["train_loss"]))
        worst = 0.0
        for k in [k for k in z.files if k.startswith("grad/")]:
            name = k[5:]
            cand = [n for n in byname if n.endswith(name) or name.endswith(n)]
            if not cand:
                report(f"{cfgname} gradient {name}", False, "variable not found by name")
                continue
            g = byname[cand[0]].numpy()
            worst = max(worst, float(np.abs(g - z[k]).max() / max(np.abs(z[k]).max(), 1e-12)))
        report(f"{cfgname} training", rel <= 2e-5 and worst <= 1e-4, f"loss rel. diff {rel:.2e}, worst gradient error / scale {worst:.2e}")
    try:
        import tensorflow_addons as tfa
        z = np.load(os.path.join(a.kit, f"{a.configs[0]}_io.npz"))
        v = tf.Variable(z["adamw_var0"])
        opt = tfa.optimizers.AdamW(weight_decay=2e-6, learning_rate=2e-5, epsilon=1e-8)
        opt.apply_gradients([(tf.constant(z["adamw_grad"]), v)])
        d = int((v.numpy() != z["adamw_var1"]).sum())
        report("tfa AdamW step", d == 0, f"{d} of {v.shape[0]} elements differ from the restated update "
               f"(max {np.abs(v.numpy() - z['adamw_var1']).max():.2e})")
    except ImportError:
        print("SKIP tfa AdamW step: tensorflow_addons not installed")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
