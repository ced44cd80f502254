"""Human3.6M ingestion in the VideoPose3D ``.npz`` format: what the reference does between the
files on disk and the per-video pose lists its generators consume.

    dataset, keypoints = load_dataset_and_2d_poses("data_3d_h36m.npz", "data_2d_h36m_cpn_ft_h36m_dbb.npz")
    cams, p3d, p2d, names, subj, act, fps = filter_and_subsample_dataset(dataset, keypoints, ["S9", "S11"], "*")
    table = pose_table(p2d, p3d, subj, act, fps)                      # uplift_upsample_3dhpe_amd.data.PoseTable

Mirrors (same names, arguments, return values):
  * ``Human36mDataset`` -- ``common/dataset/h36m_dataset.py:225-272``: calibration merge, screen-normalised intrinsics,
    translations in metres, the 11-value ``intrinsic`` vector, 32 -> 17 joints in the repository's own order;
  * ``world_to_camera`` / ``normalize_screen_coordinates`` -- ``common/dataset/camera.py:15-32``;
  * ``load_dataset_and_2d_poses`` -- ``common/dataset/uplifiting_dataset.py:25-92``;
  * ``filter_and_subsample_dataset`` -- ``common/dataset/uplifiting_dataset.py:95-210``.
The camera calibration tables, joint orders and split lists are DATA extracted from the reference's modules into
``utils/h36m_cameras.json`` by ``tests/golden/make_h36m_golden.py``; the same script runs the reference's (TensorFlow-free)
modules on a tiny synthetic dataset and stores their output, against which ``tests/test_h36m_cpu.py`` checks this file.
Host-side numpy throughout: ingestion happens once per run; the windows are then cut on the device (``data.py``).
"""
import copy
import json
import os

import numpy as np

_TABLES = None


def tables():
    """Calibration tables, joint orders, subject / action lists (utils/h36m_cameras.json)."""
    global _TABLES
    if _TABLES is None:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "utils", "h36m_cameras.json")) as fh:
            _TABLES = json.load(fh)
    return _TABLES


def normalize_screen_coordinates(X, w, h):
    """[0, w] -> [-1, 1], aspect ratio preserved (camera.py:15-19)."""
    assert X.shape[-1] == 2
    return X / w * 2 - [1, h / w]


def image_coordinates(X, w, h):
    """Inverse of normalize_screen_coordinates (camera.py:22-26)."""
    assert X.shape[-1] == 2
    return (X + [1, h / w]) * w / 2


def _qrot(q, v):
    """Rotate v by the unit quaternion q = (w, x, y, z), broadcast over leading axes: v + 2 (w (u x v) + u x (u x v))."""
    u = q[..., 1:]
    uv = np.cross(u, v, axis=-1)
    uuv = np.cross(u, uv, axis=-1)
    return v + 2 * (q[..., :1] * uv + uuv)


def world_to_camera(X, R, t):
    """Camera-frame coordinates of world points X (..., 3) for a camera with orientation quaternion R and position t
    (camera.py:29-31): rotate X - t by the inverse of R."""
    Rt = np.concatenate([R[..., :1], -R[..., 1:]], axis=-1)
    return _qrot(np.tile(Rt, (*X.shape[:-1], 1)), X - t)


def camera_to_world(X, R, t):
    return _qrot(np.tile(R, (*X.shape[:-1], 1)), X) + t


class Human36mDataset(object):
    """``dataset[subject][action] = {'positions': (F, 17, 3) world, 'cameras': [...], 'frame_rate': 50}``."""

    def __init__(self, path):
        T = tables()
        self._fps = 50
        self._cameras = copy.deepcopy(T["extrinsic"])
        for cameras in self._cameras.values():
            for i, cam in enumerate(cameras):
                cam.update(copy.deepcopy(T["intrinsic"][i]))
                for k, v in cam.items():
                    if k not in ("id", "res_w", "res_h"):
                        cam[k] = np.array(v, dtype="float32")
                cam["center"] = normalize_screen_coordinates(cam["center"], w=cam["res_w"], h=cam["res_h"]).astype("float32")
                cam["focal_length"] = cam["focal_length"] / cam["res_w"] * 2
                if "translation" in cam:
                    cam["translation"] = cam["translation"] / 1000                 # mm -> m
                cam["intrinsic"] = np.concatenate(([cam["res_w"], cam["res_h"]], cam["focal_length"], cam["center"],
                                                   cam["radial_distortion"], cam["tangential_distortion"]))
        data = np.load(path, allow_pickle=True)["positions_3d"].item()
        order = T["order_full_to_17"]
        self._data = {}
        for subject, actions in data.items():
            self._data[subject] = {}
            for action_name, positions in actions.items():
                self._data[subject][action_name] = {"positions": positions[:, order].copy(),       # x right, y forward, z up
                                                    "cameras": self._cameras[subject], "frame_rate": 50}

    def __getitem__(self, key):
        return self._data[key]

    def subjects(self):
        return self._data.keys()

    def fps(self):
        return self._fps

    def cameras(self):
        return self._cameras


def load_dataset_and_2d_poses(dataset_path, poses_2d_path, dataset_name="h36m", verbose=True):
    """3D dataset + matching 2D detections: 3D to every camera's frame, 2D truncated to the mocap length, reordered to the
    repository's 17-point order and screen-normalised.  Returns (dataset, keypoints[subject][action][camera])."""
    if dataset_name != "h36m":
        raise KeyError("Invalid dataset")
    if verbose:
        print(f"Loading 3D dataset from {dataset_path}")
    dataset = Human36mDataset(dataset_path)
    if verbose:
        print(f"Loading 2D poses from {poses_2d_path}")
    keypoints = np.load(poses_2d_path, allow_pickle=True)["positions_2d"].item()
    for subject in list(dataset.subjects()):
        if subject not in keypoints:
            raise AssertionError(f"2D detections hold no subject {subject}")
        for action, anim in dataset[subject].items():
            if action not in keypoints[subject]:
                raise AssertionError(f"2D detections hold no action {action} for subject {subject}")
            if "positions" not in anim:
                continue
            # every camera sees the same world-frame mocap: one camera-frame copy per camera
            anim["positions_3d"] = [world_to_camera(anim["positions"], R=cam["orientation"], t=cam["translation"])
                                    for cam in anim["cameras"]]
            views = keypoints[subject][action]
            assert len(views) == len(anim["positions_3d"]), "camera count of the 2D detections differs from the 3D data"
            for ci, p3 in enumerate(anim["positions_3d"]):
                assert views[ci].shape[0] >= p3.shape[0], "2D detections shorter than the mocap sequence"
                views[ci] = views[ci][:p3.shape[0]]                    # some H3.6M videos hold extra frames
    to_ours = tables()["order_17_original_to_ours"]
    for subject, by_action in keypoints.items():                       # every subject of the detections file, as the reference
        for action, views in by_action.items():
            for ci in range(len(views)):
                cam = dataset.cameras()[subject][ci]
                kps = views[ci][:, to_ours].copy()
                kps[..., :2] = normalize_screen_coordinates(kps[..., :2], w=cam["res_w"], h=cam["res_h"])
                views[ci] = kps
    return dataset, keypoints


def create_image_paths(base_path, subject, action, cam_id, frame_nums):
    """0-based frame file names (h36m_splits.py:96-101)."""
    d = os.path.join(base_path, "frames", subject, f"{action}.{cam_id}")
    return [os.path.join(d, f"img_{k:06d}.jpg") for k in frame_nums]


def filter_and_subsample_dataset(dataset, poses_2d, subjects, action_filter, downsample=1, image_base_path=None, verbose=True):
    """One list entry per (subject, action, camera) video.  Returns (camera_params (11 values each), poses_3d, poses_2d,
    frame_names, subjects, actions, frame_rates); entries that cannot be produced are None (as in the reference)."""
    T = tables()
    action_filter = None if action_filter == "*" else action_filter
    if verbose:
        print(f"Filtering subjects: {subjects}")
        if action_filter is not None:
            print(f"Filtering actions: {action_filter}")
    translated = {"Photo": "TakingPhoto", "WalkDog": "WalkingDog"}
    subject_dict = {name: i for i, name in enumerate(T["all_subjects"])}
    action_dict = {name: i for i, name in enumerate(T["renamed_actions"])}
    out_cam, out_3d, out_2d, out_names, out_subj, out_act, out_fps = [], [], [], [], [], [], []
    for subject in subjects:
        for action in poses_2d[subject].keys():
            action_name = action.split(" ")[0]
            if action_filter is not None and action_name not in action_filter:
                continue
            seqs_2d = poses_2d[subject][action]
            for seq in seqs_2d:
                out_2d.append(seq.copy())
                out_subj.append(subject_dict[subject])
                out_act.append(action_dict[action_name])
            if subject in dataset.cameras():
                cams = dataset.cameras()[subject]
                assert len(cams) == len(seqs_2d), "Camera count mismatch"
                for cam in cams:
                    if "intrinsic" in cam:
                        out_cam.append(cam["intrinsic"].copy())
            if "positions_3d" in dataset[subject][action]:
                for seq in dataset[subject][action]["positions_3d"]:
                    out_3d.append(seq.copy())
                    out_fps.append(dataset[subject][action].get("frame_rate", 50))
            if image_base_path is not None:
                for i, seq in enumerate(seqs_2d):
                    cam_id = dataset.cameras()[subject][i]["id"]
                    names = create_image_paths(image_base_path, subject, action, cam_id, range(seq.shape[0]))
                    for new_name, original in translated.items():          # the canonical renaming is reverted when the files say so
                        if new_name in action and not os.path.exists(names[0]):
                            names = create_image_paths(image_base_path, subject, action.replace(new_name, original), cam_id,
                                                       range(seq.shape[0]))
                    out_names.append(names)
    out_cam = out_cam or None
    out_3d = out_3d or None
    out_names = out_names or None
    out_fps = out_fps or None
    if downsample > 1:
        for i in range(len(out_2d)):
            out_2d[i] = out_2d[i][::downsample]
            if out_3d is not None:
                out_3d[i] = out_3d[i][::downsample]
            if out_names is not None:
                out_names[i] = out_names[i][::downsample]
    return out_cam, out_3d, out_2d, out_names, out_subj, out_act, out_fps


def subjects_of_split(split):
    """'train' / 'val' / 'trainval' / 'test' / a single subject -> subject names (h36m_splits.py:25-58)."""
    return list(tables()["subjects_by_split"][split])


def pose_table(poses_2d, poses_3d=None, subjects=None, actions=None, frame_rates=None, device=None):
    """The filtered videos as one device-resident table for ``data.SequenceGenerator``."""
    from .data import PoseTable
    return PoseTable(poses_2d, poses_3d, subjects, actions, frame_rates, device=device)
