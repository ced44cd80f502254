"""Generates the H36M-ingestion fixtures from the REFERENCE's own (TensorFlow-free) modules.  Run in the build container:

    python tests/golden/make_h36m_golden.py          # needs /root/reference

It imports common/dataset/{h36m_dataset,camera,keypoint_order,h36m_splits}.py -- pure numpy -- and
  1. dumps the Human3.6M camera calibration tables (dataset constants, h36m_dataset.py:33-218) as data:
        uplift-upsample-3dhpe_amd/utils/h36m_cameras.json
  2. writes a tiny synthetic dataset in the VideoPose3D .npz format (data_3d_h36m.npz / data_2d_h36m_*.npz:
     dict subject -> action -> (F, 32, 3) world positions / list of 4 (F', 17, 2) pixel keypoints)
        tests/golden/h36m_tiny_3d.npz, tests/golden/h36m_tiny_2d.npz
  3. runs the reference on it -- Human36mDataset(path) (h36m_dataset.py:225-272), then the steps of
     load_dataset_and_2d_poses (uplifiting_dataset.py:25-92: that function itself sits in a module that imports
     tensorflow, so its loop is applied here with the reference's own world_to_camera / normalize_screen_coordinates /
     H36MOrder17POriginalOrder) -- and stores what it produced:
        tests/golden/h36m_tiny_expected.npz
The package's own loader (uplift-upsample-3dhpe_amd/h36m.py) is checked against these in tests/test_h36m_cpu.py."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, "/root/reference")

from common.dataset import h36m_dataset as R                                     # noqa: E402
from common.dataset.camera import world_to_camera, normalize_screen_coordinates   # noqa: E402
from common.dataset.keypoint_order import H36MOrder17POriginalOrder, H36MOrderFull  # noqa: E402
from common.dataset import h36m_splits                                            # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# 1. calibration tables as data
cams = {"intrinsic": R.h36m_cameras_intrinsic_params,
        "extrinsic": R.h36m_cameras_extrinsic_params,
        "order_full_to_17": [int(i) for i in H36MOrderFull.to_17p_order()],
        "order_17_original_to_ours": [int(i) for i in H36MOrder17POriginalOrder.to_our_17p_order()],
        "all_subjects": list(h36m_splits.all_subjects), "subjects_by_split": h36m_splits.subjects_by_split,
        "renamed_actions": list(h36m_splits.renamed_actions), "camera_ids": list(h36m_splits.cameras)}
with open(os.path.join(ROOT, "uplift-upsample-3dhpe_amd", "utils", "h36m_cameras.json"), "w") as fh:
    json.dump(cams, fh, indent=1)

# 2. tiny synthetic dataset
rng = np.random.default_rng(7)
frames = {("S1", "Walking 1"): 9, ("S1", "Photo"): 7, ("S9", "WalkDog 1"): 8, ("S9", "Sitting"): 6}
pos3d, pos2d = {}, {}
for (s, a), F in frames.items():
    p = rng.normal(0.0, 0.4, size=(F, 32, 3)).astype(np.float32)
    p[..., 2] += 1.0                                          # metres, roughly a standing person
    p += np.array([0.3, 4.0, 0.0], np.float32)                # in front of the cameras
    pos3d.setdefault(s, {})[a] = p
    seqs = []
    for c in range(4):
        extra = 2 if (c == 1 and a == "Photo") else 0         # some H3.6M videos hold extra frames: truncated to the mocap length
        seqs.append(rng.uniform(0.0, 1000.0, size=(F + extra, 17, 2)).astype(np.float32))
    pos2d.setdefault(s, {})[a] = seqs
np.savez_compressed(os.path.join(HERE, "h36m_tiny_3d.npz"), positions_3d=np.array(pos3d, dtype=object))
np.savez_compressed(os.path.join(HERE, "h36m_tiny_2d.npz"), positions_2d=np.array(pos2d, dtype=object),
                    metadata=np.array({"layout_name": "h36m", "num_joints": 17,
                                       "keypoints_symmetry": [[4, 5, 6, 11, 12, 13], [1, 2, 3, 14, 15, 16]]}, dtype=object))

# 3. the reference on it
dataset = R.Human36mDataset(os.path.join(HERE, "h36m_tiny_3d.npz"))
out = {}
for subject in dataset.subjects():
    for action in dataset[subject].keys():
        anim = dataset[subject][action]
        for ci, cam in enumerate(anim["cameras"]):
            out[f"p3d/{subject}/{action}/{ci}"] = world_to_camera(anim["positions"], R=cam["orientation"], t=cam["translation"])
keypoints = np.load(os.path.join(HERE, "h36m_tiny_2d.npz"), allow_pickle=True)["positions_2d"].item()
for subject in dataset.subjects():
    for action in dataset[subject].keys():
        for ci in range(len(keypoints[subject][action])):
            n = out[f"p3d/{subject}/{action}/{ci}"].shape[0]
            kps = keypoints[subject][action][ci][:n]
            cam = dataset.cameras()[subject][ci]
            kps = kps[:, H36MOrder17POriginalOrder.to_our_17p_order()].copy()
            kps[..., :2] = normalize_screen_coordinates(kps[..., :2], w=cam["res_w"], h=cam["res_h"])
            out[f"p2d/{subject}/{action}/{ci}"] = kps
            out[f"intrinsic/{subject}/{ci}"] = cam["intrinsic"]
np.savez_compressed(os.path.join(HERE, "h36m_tiny_expected.npz"), **out)
print("wrote", len(out), "arrays")
