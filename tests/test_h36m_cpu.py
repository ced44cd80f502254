"""H36M ingestion (uplift-upsample-3dhpe_amd/h36m.py) against outputs of the REFERENCE's own TensorFlow-free modules on a
tiny synthetic dataset (tests/golden/make_h36m_golden.py ran common/dataset/h36m_dataset.py, camera.py, keypoint_order.py
in the build container and committed what they produced).  This path's parity is therefore pinned to the reference."""
import copy
import importlib
import os

import numpy as np
import pytest

from tests import util

G = os.path.join(util.ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def H():
    return importlib.import_module("uplift_upsample_3dhpe_amd.h36m")


@pytest.fixture(scope="module")
def loaded(H):
    return H.load_dataset_and_2d_poses(os.path.join(G, "h36m_tiny_3d.npz"), os.path.join(G, "h36m_tiny_2d.npz"), verbose=False)


def test_matches_reference_outputs(H, loaded):
    dataset, keypoints = loaded
    exp = np.load(os.path.join(G, "h36m_tiny_expected.npz"))
    seen = 0
    for key in exp.files:
        kind, rest = key.split("/", 1)
        if kind == "p3d":
            subject, action, ci = rest.rsplit("/", 2)[0], rest.rsplit("/", 2)[1], int(rest.rsplit("/", 1)[1])
            got = dataset[subject][action]["positions_3d"][ci]
        elif kind == "p2d":
            subject, action, ci = rest.rsplit("/", 2)[0], rest.rsplit("/", 2)[1], int(rest.rsplit("/", 1)[1])
            got = keypoints[subject][action][ci]
        else:
            subject, ci = rest.split("/")
            got = dataset.cameras()[subject][int(ci)]["intrinsic"]
        want = exp[key]
        assert got.shape == want.shape and got.dtype == want.dtype, (key, got.shape, want.shape, got.dtype, want.dtype)
        assert np.array_equal(got, want), (key, np.abs(got - want).max())            # same arithmetic: bit exact
        seen += 1
    assert seen == 40


def test_shapes_orders_and_truncation(H, loaded):
    dataset, keypoints = loaded
    assert dataset["S1"]["Photo"]["positions"].shape == (7, 17, 3)
    raw2d = np.load(os.path.join(G, "h36m_tiny_2d.npz"), allow_pickle=True)["positions_2d"].item()
    assert raw2d["S1"]["Photo"][1].shape[0] == 9 and keypoints["S1"]["Photo"][1].shape[0] == 7       # extra frames dropped
    # joint 6 of "our" order is the pelvis = joint 0 of the file order; screen normalisation maps [0, w] to [-1, 1]
    cam = dataset.cameras()["S1"][0]
    px = raw2d["S1"]["Walking 1"][0][:, 0]
    back = H.image_coordinates(keypoints["S1"]["Walking 1"][0][:, 6].astype(np.float64), w=cam["res_w"], h=cam["res_h"])
    assert np.abs(back - px).max() < 1e-3
    assert abs(float(cam["translation"][0])) < 10.0                                                    # metres, not millimetres


def test_filter_and_subsample(H, loaded):
    dataset, keypoints = copy.deepcopy(loaded)
    cams, p3d, p2d, names, subj, act, fps = H.filter_and_subsample_dataset(dataset, keypoints, ["S9"], "*", verbose=False)
    assert len(p2d) == len(p3d) == len(cams) == 8 and names is None                                    # 2 actions x 4 cameras
    assert set(subj) == {H.tables()["all_subjects"].index("S9")}
    ra = H.tables()["renamed_actions"]
    assert sorted(set(act)) == sorted({ra.index("WalkDog"), ra.index("Sitting")})
    assert all(f == 50 for f in fps) and all(c.shape == (11,) for c in cams)
    cams, p3d, p2d, names, subj, act, fps = H.filter_and_subsample_dataset(dataset, keypoints, ["S1", "S9"], ["Walking"],
                                                                           downsample=2, image_base_path="/data/h36m", verbose=False)
    assert len(p2d) == 4 and all(a == ra.index("Walking") for a in act)                               # "Walking 1" of S1 only
    assert p2d[0].shape[0] == 5 and p3d[0].shape[0] == 5 and len(names[0]) == 5                       # 9 frames, every 2nd
    assert names[0][1].endswith(os.path.join("frames", "S1", "Walking 1.54138969", "img_000002.jpg"))
    assert H.subjects_of_split("test") == ["S9", "S11"]


def test_world_camera_round_trip(H):
    rng = np.random.default_rng(0)
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    t = rng.normal(size=3)
    X = rng.normal(size=(5, 17, 3))
    back = H.camera_to_world(H.world_to_camera(X, q, t), q, t)
    assert np.abs(back - X).max() < 1e-12


def test_amass_matches_reference_outputs():
    """amass.py against the reference's AMASSDataset run on tests/golden/amass_tiny (make_amass_golden.py): split patterns,
    regex filters, joint reorder, down-sampling, and the 18-value camera vectors -- bit exact."""
    A = importlib.import_module("uplift_upsample_3dhpe_amd.amass")
    exp = np.load(os.path.join(G, "amass_tiny_expected.npz"))
    cases = {"train": ("train", 1), "val": ("val", 1), "train_ds2": ("train", 2),
             "custom": ([("CMU", "0[12]", ".*_01_poses"), ("ACCAD", ".*", "A2.*")], 1)}
    for tag, (split, ds) in cases.items():
        a = A.AMASSDataset(os.path.join(G, "amass_tiny"), os.path.join(G, "h36m_tiny_3d.npz"), split, downsample=ds)
        want = {k[len(tag) + 1:]: exp[k] for k in exp.files if k.startswith(tag + "/") and not k.endswith("__cameras__")}
        got = {f"{d}/{s}/{n}": rec["positions"] for d, ss in a._data.items() for s, aa in ss.items() for n, rec in aa.items()}
        assert sorted(got) == sorted(want), (tag, sorted(got), sorted(want))
        for k in want:
            assert got[k].dtype == want[k].dtype and np.array_equal(got[k], want[k]), (tag, k)
        cams = A.camera_table(a)
        assert cams.dtype == np.float32 and np.array_equal(cams, exp[f"{tag}/__cameras__"])
        seqs, rates = A.sequences(a)
        assert len(seqs) == len(want) and all(r == 50 for r in rates)
    assert sorted(a._data) == ["ACCAD", "CMU"] and len(cams) == 28 and cams.shape[1] == 18       # 7 subjects x 4 cameras
