"""AMASS ingestion: the serialised per-dataset ``.npz`` files -> the 3D sequences and the camera
table the on-the-fly projection (``data.world_to_cam_and_2d`` / ``uu3d_world_to_cam_2d``) draws from.

    amass = AMASSDataset("amass_dir", "data_3d_h36m.npz", "train")            # or a list of (dataset, subject, action) regexes
    seqs, fps = sequences(amass)                                               # list of (F, 17, 3) float32, list of int
    cams = camera_table(amass)                                                 # (n_cameras, 18) f32: orientation | translation | intrinsic

Mirrors ``AMASSDataset`` (``common/dataset/amass_dataset.py:71-118``: split patterns applied with ``re.fullmatch`` to
dataset / subject / action, the regressor's 17 joints reordered to the repository's order, optional down-sampling, the
Human3.6M cameras) and the sequence / camera collection of ``AMASSSequenceGenerator.__init__``
(``common/dataset/uplifiting_dataset.py:488-515``).  The joint reorder and the split patterns are DATA extracted into
``utils/amass_tables.json`` by ``tests/golden/make_amass_golden.py``, which also runs the reference module on tiny files;
``tests/test_h36m_cpu.py`` checks this loader against what it produced.
"""
import copy
import json
import os
import re

import numpy as np

from . import h36m

_TABLES = None


def tables():
    global _TABLES
    if _TABLES is None:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "utils", "amass_tables.json")) as fh:
            _TABLES = json.load(fh)
    return _TABLES


class AMASSDataset(object):
    """``amass._data[dataset][subject][action] = {'dataset', 'subject', 'action', 'positions' (F, 17, 3) f32, 'frame_rate'}``."""

    def __init__(self, path, h36m_path, split, downsample=1, h36m_cameras=None):
        self._fps = 50
        self._cameras = copy.deepcopy(h36m.Human36mDataset(h36m_path).cameras() if h36m_cameras is None else h36m_cameras)
        self.split = split
        patterns = tables()["amass_splits"][split] if isinstance(split, str) else split
        order = tables()["amass_reorder"]
        self._data = {}
        for fname in sorted(os.listdir(path)):
            dataset, ext = os.path.splitext(fname)
            if ext != ".npz":
                continue
            for_dataset = [p for p in patterns if re.fullmatch(p[0], dataset) is not None]
            if not for_dataset:
                continue
            content = np.load(os.path.join(path, fname), allow_pickle=True)["positions_3d"].item()
            self._data[dataset] = {}
            for subject, actions in content.items():
                for_subject = [p for p in for_dataset if re.fullmatch(p[1], subject) is not None]
                if not for_subject:
                    continue
                self._data[dataset][subject] = {}
                for action, rec in actions.items():
                    if not any(re.fullmatch(p[2], action) is not None for p in for_subject):
                        continue
                    if rec["frame_rate"] != 50.0:
                        raise AssertionError(f"{dataset}/{subject}/{action}: frame rate {rec['frame_rate']}, the serialised AMASS data is 50 Hz")
                    pos = rec["positions_3d"].astype(np.float32)[:, order]
                    if downsample > 1:
                        pos = pos[::downsample]
                    self._data[dataset][subject][action] = {"dataset": dataset, "subject": subject, "action": action,
                                                            "positions": pos.copy(), "frame_rate": int(rec["frame_rate"])}

    def __getitem__(self, key):
        return self._data[key]

    def subjects(self):
        return self._data.keys()

    def fps(self):
        return self._fps

    def cameras(self):
        return self._cameras

    def supports_semi_supervised(self):
        return False


def sequences(amass):
    """All selected sequences in the reference's iteration order, and their frame rates."""
    seqs, rates = [], []
    for subjects in amass._data.values():
        for actions in subjects.values():
            for rec in actions.values():
                seqs.append(rec["positions"])
                rates.append(rec.get("frame_rate", 50))
    return seqs, rates


def camera_table(amass):
    """(n, 18) float32: orientation quaternion (4) | translation in metres (3) | intrinsic (11) of every extrinsic camera."""
    rows = [np.concatenate([cam["orientation"], cam["translation"], cam["intrinsic"]], axis=0).astype(np.float32)
            for cams in amass.cameras().values() for cam in cams if "orientation" in cam]
    return np.stack(rows)
