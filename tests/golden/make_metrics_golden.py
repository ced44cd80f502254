"""Generates metric fixtures from the REFERENCE's own (TensorFlow-free) modules.  Run in the build container:

    python tests/golden/make_metrics_golden.py          # needs /root/reference

common/dataset/metrics.py (mpjpe, nmpjpe root / mean alignment, pmpjpe) and common/dataset/action_wise_eval.py
(frame_wise_eval, h36_action_wise_eval, interpolate_between_keyframes) on seeded poses with invalid joints, every H36M
action present, and several videos of frame indices -> tests/golden/metrics_expected.npz.
Checked by tests/test_evaluation_cpu.py (host functions) and tests/test_hotpath_gpu.py (uu3d_mpjpe)."""
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference")
from common.dataset import metrics as M                    # noqa: E402
from common.dataset import action_wise_eval as E           # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(21)
B, J, root = 120, 17, 6
gt3 = rng.normal(0, 0.35, size=(B, J, 3))
pred = gt3 * rng.uniform(0.8, 1.25, size=(B, 1, 1)) + rng.normal(0, 0.04, size=(B, J, 3)) + rng.normal(0, 0.5, size=(B, 1, 3))
valid = (rng.uniform(size=(B, J)) > 0.1).astype(np.float64)
valid[:, root] = 1.0
gt = np.concatenate([gt3, valid[..., None]], -1)
actions = np.arange(B) % 15
frame_indices = np.concatenate([np.arange(0, 40), np.arange(3, 43), np.arange(0, 40)])      # three videos, the second starts off-keyframe
out = {"pred": pred, "gt": gt, "actions": actions, "frame_indices": frame_indices, "root": np.int64(root)}
out["mpjpe_jp"] = M.mpjpe(pred, gt, root, normalize=False)
out["mpjpe"] = np.float64(M.mpjpe(pred, gt, root, normalize=True))
out["nmpjpe_root_jp"] = M.nmpjpe(pred, gt, root, alignment="root", normalize=False)
out["nmpjpe_root"] = np.float64(M.nmpjpe(pred, gt, root, alignment="root", normalize=True))
out["nmpjpe_mean_jp"] = M.nmpjpe(pred, gt, root, alignment="mean", normalize=False)
out["pmpjpe_jp"] = M.pmpjpe(pred, gt, normalize=False)
out["pmpjpe"] = np.float64(M.pmpjpe(pred, gt, normalize=True))
fr = E.frame_wise_eval(pred, gt, root)
out["frame_wise"] = np.array([fr["mpjpe"], fr["nmpjpe"], fr["pampjpe"]])
frame_results, average_results, per_action = E.h36_action_wise_eval(pred, gt, actions, root)
out["aw_frame"] = np.array([frame_results[k] for k in ("mpjpe", "nmpjpe", "pampjpe")])
out["aw_average"] = np.array([average_results[k] for k in ("mpjpe", "nmpjpe", "pampjpe")])
out["aw_actions"] = np.array(list(per_action.keys()))
out["aw_per_action"] = np.array([[d[k] for k in ("mpjpe", "nmpjpe", "pampjpe")] for d in per_action.values()])
for stride in (5, 10):
    # the reference indexes pred3d[None] for frames in front of a video's first keyframe; start every video on a keyframe for it
    fi = np.concatenate([np.arange(0, 40), np.arange(stride, stride + 40), np.arange(0, 40)])
    interp, keyframes = E.interpolate_between_keyframes(pred, fi, stride)
    out[f"interp_{stride}"] = interp
    out[f"keyframes_{stride}"] = keyframes
    out[f"frame_indices_{stride}"] = fi
np.savez_compressed(os.path.join(HERE, "metrics_expected.npz"), **out)
print("wrote", len(out), "arrays; mpjpe", out["mpjpe"], "pmpjpe", out["pmpjpe"])
