"""Generates the AMASS-ingestion fixtures from the REFERENCE's own (TensorFlow-free) module.  Run in the build container:

    python tests/golden/make_amass_golden.py          # needs /root/reference and tests/golden/h36m_tiny_3d.npz

  1. dumps the joint reorder and the split patterns (common/dataset/amass_dataset.py:20-68) as data:
        uplift-upsample-3dhpe_amd/utils/amass_tables.json
  2. writes three tiny per-dataset files in the reference's serialisation (dict subject -> action ->
     {"positions_3d": (F, 17, 3), "frame_rate": 50.0}) under tests/golden/amass_tiny/
  3. runs the reference's AMASSDataset (amass_dataset.py:71-118) on them for the "train" and "val" splits, a custom
     regex split and downsample = 2, and stores the selected sequences and the 18-value camera vectors
     (orientation | translation | intrinsic, the concatenation of uplifiting_dataset.py:506-515 applied to the
     reference dataset's own cameras()):   tests/golden/amass_tiny_expected.npz"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, "/root/reference")
from common.dataset import amass_dataset as R      # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
TINY = os.path.join(HERE, "amass_tiny")
os.makedirs(TINY, exist_ok=True)

with open(os.path.join(ROOT, "uplift-upsample-3dhpe_amd", "utils", "amass_tables.json"), "w") as fh:
    json.dump({"amass_reorder": [int(i) for i in R.amass_reorder],
               "amass_splits": {k: [list(p) for p in v] for k, v in R.amass_splits.items()}}, fh, indent=1)

rng = np.random.default_rng(11)
layout = {"CMU": {"01": {"01_01_poses": 9, "01_02_poses": 6}, "02": {"02_01_poses": 7}},
          "SFU": {"0005": {"0005_Walking001_poses": 8}},
          "ACCAD": {"Female1General_c3d": {"A1 - Stand_poses": 5, "A2 - Sway_poses": 10}}}
for ds, subjects in layout.items():
    d = {}
    for s, actions in subjects.items():
        d[s] = {a: {"positions_3d": rng.normal(0, 0.5, size=(F, 17, 3)), "frame_rate": 50.0} for a, F in actions.items()}
    np.savez_compressed(os.path.join(TINY, ds + ".npz"), positions_3d=np.array(d, dtype=object))

out = {}
h36m = os.path.join(HERE, "h36m_tiny_3d.npz")
cases = {"train": ("train", 1), "val": ("val", 1), "train_ds2": ("train", 2),
         "custom": ([("CMU", "0[12]", ".*_01_poses"), ("ACCAD", ".*", "A2.*")], 1)}
for tag, (split, ds) in cases.items():
    a = R.AMASSDataset(TINY, h36m, split, downsample=ds)
    for dataset, subjects in a._data.items():
        for subject, actions in subjects.items():
            for action, seq in actions.items():
                out[f"{tag}/{dataset}/{subject}/{action}"] = seq["positions"]
    cams = []
    for subject, cs in a.cameras().items():
        for cam in cs:
            if "orientation" in cam.keys():
                cams.append(np.concatenate([cam["orientation"], cam["translation"], cam["intrinsic"]], axis=0).astype(np.float32))
    out[f"{tag}/__cameras__"] = np.stack(cams)
np.savez_compressed(os.path.join(HERE, "amass_tiny_expected.npz"), **out)
print("wrote", len(out), "arrays;", [k for k in out if k.startswith("custom/")])
