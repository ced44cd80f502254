"""Evaluation protocol, host side: batched metrics against the per-pose oracle restatement, the
invariances the metrics are defined by, keyframe interpolation and the action-wise bookkeeping."""
import numpy as np
import pytest

from uplift_upsample_3dhpe_amd import evaluation as E
from oracle import metrics_oracle as MO
from oracle import uplift_oracle as O
from tests import util


def _data(B=40, K=17, seed=0, invalid=True):
    rng = np.random.default_rng(seed)
    gt = rng.normal(0, 0.3, size=(B, K, 3))
    pred = gt + rng.normal(0, 0.05, size=(B, K, 3))
    v = np.ones((B, K, 1))
    if invalid:
        v[rng.random((B, K, 1)) < 0.1] = 0.0
        v[:, 6] = 1.0                          # keep the root valid
    return pred, np.concatenate([gt, v], -1)


def _rot(rng):
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    return q * np.sign(np.linalg.det(q))


def test_batched_metrics_match_the_per_pose_oracle():
    pred, gt = _data()
    assert np.allclose(E.pmpjpe(pred, gt, normalize=False), MO.pmpjpe(pred, gt, normalize=False), atol=1e-12)
    assert abs(E.pmpjpe(pred, gt) - MO.pmpjpe(pred, gt)) < 1e-12
    for al in ("root", "mean"):
        assert np.allclose(E.nmpjpe(pred, gt, 6, al, normalize=False), MO.nmpjpe(pred, gt, 6, al, normalize=False), atol=1e-12)
    assert np.allclose(E.mpjpe(pred, gt, 6, normalize=False), O.mpjpe(pred, gt, 6, normalize=False), atol=1e-12)
    # a reflected prediction exercises the det < 0 branch of the Procrustes fit (metrics.py:178-182)
    refl = pred * np.array([-1.0, 1.0, 1.0])
    assert np.allclose(E.pmpjpe(refl, gt, normalize=False), MO.pmpjpe(refl, gt, normalize=False), atol=1e-12)
    assert E.pmpjpe(refl, gt) > 2 * E.pmpjpe(pred, gt)            # a rotation cannot undo a reflection


def test_metric_invariances():
    rng = np.random.default_rng(1)
    pred, gt = _data(invalid=False, seed=1)
    base_p, base_n = E.pmpjpe(pred, gt, normalize=False), E.nmpjpe(pred, gt, 6, normalize=False)
    moved = np.stack([2.5 * p @ _rot(rng) + rng.normal(size=3) for p in pred])
    assert np.allclose(E.pmpjpe(moved, gt, normalize=False), base_p, atol=1e-10)       # similarity transforms of the prediction
    assert np.allclose(E.nmpjpe(pred * 3.0, gt, 6, normalize=False), base_n, atol=1e-10)   # scale
    assert np.allclose(E.nmpjpe(pred + 0.7, gt, 6, normalize=False), base_n, atol=1e-10)   # root alignment removes translation
    assert np.all(base_p <= E.mpjpe(pred, gt, 6, normalize=False).mean(1, keepdims=True) * 17)  # sanity: finite, bounded
    exact = np.concatenate([gt[:, :, :3]], -1)
    assert E.pmpjpe(exact, gt) < 1e-12 and E.nmpjpe(exact, gt, 6) < 1e-12


def test_invalid_joints_are_flagged_and_excluded():
    pred, gt = _data(seed=2)
    for m in (E.mpjpe(pred, gt, 6, normalize=False), E.nmpjpe(pred, gt, 6, normalize=False), E.pmpjpe(pred, gt, normalize=False)):
        assert np.all((m == -1.0) == (gt[:, :, 3] == 0))
    v = gt[:, :, 3] > 0
    per = E.mpjpe(pred, gt, 6, normalize=False)
    assert abs(E.mpjpe(pred, gt, 6) - per[v].mean()) < 1e-12


def test_keyframe_interpolation():
    # two videos (frame index restarts), stride 4: keyframes 0, 4, 8 ... ; tail after the last keyframe repeats it
    idx = np.concatenate([np.arange(0, 11), np.arange(0, 6)])
    pred = np.arange(len(idx), dtype=np.float64)[:, None, None] * np.ones((1, 2, 3))
    out, key = E.interpolate_between_keyframes(pred, idx, 4)
    assert key.tolist() == [i % 4 == 0 for i in idx]
    assert np.allclose(out[:9, 0, 0], np.arange(9))               # linear data interpolates to itself
    assert np.allclose(out[9:11, 0, 0], 8)                        # after the last keyframe of video 1
    assert np.allclose(out[11:16, 0, 0], [11, 12, 13, 14, 15]) and np.allclose(out[16, 0, 0], 15)
    wiggle = pred.copy(); wiggle[1:4] += 100.0                   # non-keyframes are ignored between keyframes
    out2, _ = E.interpolate_between_keyframes(wiggle, idx, 4)
    assert np.allclose(out2[1:4, 0, 0], [1, 2, 3])
    out3, _ = E.interpolate_between_keyframes(pred, idx, np.full(len(idx), 4))   # per-frame stride array, as eval.py passes
    assert np.array_equal(out3, out)


def test_action_wise_eval_and_run_bookkeeping():
    pred, gt = _data(B=90, seed=3, invalid=False)
    actions = np.arange(90) % 15
    frame, avg, per = E.h36_action_wise_eval(pred, gt, actions, 6)
    assert list(per) == E.H36M_ACTIONS and set(frame) == set(E.METRICS)
    fm = E.mpjpe(pred, gt, 6, normalize=False) * 1000.0
    assert abs(per["Eating"]["mpjpe"] - fm[actions == 2].mean()) < 1e-9
    assert abs(avg["mpjpe"] - np.mean([fm[actions == a].mean() for a in range(15)])) < 1e-9
    assert abs(frame["mpjpe"] - fm.mean()) < 1e-9 and frame["pampjpe"] <= frame["nmpjpe"] + 1e-9 <= frame["mpjpe"] + 2e-9
    assert E.frame_wise_eval(pred, gt, 6)["nmpjpe"] == pytest.approx(frame["nmpjpe"])
    cfg = util.load_config("h36m_351")                            # SEQUENCE_STRIDE 5, MASK_STRIDE 5, TEST_STRIDED_EVAL
    idx = np.arange(90)
    res = E.evaluate_predictions(pred, gt[:, :, :3], actions, idx, cfg)
    assert res["keyframes"] is not None
    key = idx % 5 == 0
    assert res["keyframes"][0]["mpjpe"] == pytest.approx(fm[key].mean())
    interp, _ = E.interpolate_between_keyframes(pred, idx, 5)
    assert res["all_frames"][0]["mpjpe"] == pytest.approx((E.mpjpe(interp, gt, 6, normalize=False) * 1000).mean())


def test_window_oracle_shapes_and_padding():
    """oracle/window_oracle.py itself: edge / zero padding at both ends, stride-mask alignment."""
    from oracle import window_oracle as WO
    v = np.arange(10, dtype=np.float32)[:, None, None] * np.ones((1, 2, 2), np.float32)
    w, m, s = WO.one_window(v, 1, 9, 2, "edge", 4, "global", 0, False, None)       # frames -7 -5 -3 -1 1 3 5 7 9
    assert w[:, 0, 0].tolist() == [1, 1, 1, 1, 1, 3, 5, 7, 9] and m.tolist() == [0, 0, 0, 0, 1, 1, 1, 1, 1]
    assert s.tolist() == [(f % 4 == 0) for f in range(-7, 11, 2)]
    w, m, _ = WO.one_window(v, 8, 5, 3, "constant", 3, None, 0, False, None)        # frames 2 5 8 11 14
    assert w[:, 0, 0].tolist() == [2, 5, 8, 0, 0] and m.tolist() == [1, 1, 1, 0, 0]
    w, _, _ = WO.one_window(v, 8, 5, 3, "edge", 3, None, 0, True, [1, 0])
    assert w[:, 0, 0].tolist() == [-2, -5, -8, -8, -8] and w[:, 0, 1].tolist() == [2, 5, 8, 8, 8]
    rows = WO.sample_list([5, 3], [50, 100], 2, True, False)
    assert rows.tolist() == [[0, 0, 0, 50], [0, 2, 0, 50], [0, 4, 0, 50], [0, 0, 1, 50], [0, 2, 1, 50], [0, 4, 1, 50],
                             [1, 0, 0, 100], [1, 2, 0, 100], [1, 0, 1, 100], [1, 2, 1, 100]]


def test_metrics_match_the_reference_modules():
    """evaluation.py against outputs of the reference's own common/dataset/metrics.py and action_wise_eval.py (both numpy
    only; tests/golden/make_metrics_golden.py ran them in the build container).  The batched SVD / vectorised code paths
    differ from the reference's per-pose loops, so agreement is to rounding (1e-9 m), not bitwise."""
    import os
    g = np.load(os.path.join(util.ROOT, "tests", "golden", "metrics_expected.npz"))
    pred, gt, root, actions = g["pred"], g["gt"], int(g["root"]), g["actions"]
    tol = 1e-9
    assert np.abs(E.mpjpe(pred, gt, root, normalize=False) - g["mpjpe_jp"]).max() < tol
    assert abs(E.mpjpe(pred, gt, root) - float(g["mpjpe"])) < tol
    assert np.abs(E.nmpjpe(pred, gt, root, alignment="root", normalize=False) - g["nmpjpe_root_jp"]).max() < tol
    assert abs(E.nmpjpe(pred, gt, root) - float(g["nmpjpe_root"])) < tol
    assert np.abs(E.nmpjpe(pred, gt, root, alignment="mean", normalize=False) - g["nmpjpe_mean_jp"]).max() < tol
    assert np.abs(E.pmpjpe(pred, gt, normalize=False) - g["pmpjpe_jp"]).max() < 1e-8
    assert abs(E.pmpjpe(pred, gt) - float(g["pmpjpe"])) < 1e-8
    fr = E.frame_wise_eval(pred, gt, root)
    assert np.abs(np.array([fr["mpjpe"], fr["nmpjpe"], fr["pampjpe"]]) - g["frame_wise"]).max() < 1e-5        # millimetres
    frame_results, average_results, per_action = E.h36_action_wise_eval(pred, gt, actions, root)
    keys = ("mpjpe", "nmpjpe", "pampjpe")
    assert np.abs(np.array([frame_results[k] for k in keys]) - g["aw_frame"]).max() < 1e-5
    assert np.abs(np.array([average_results[k] for k in keys]) - g["aw_average"]).max() < 1e-5
    assert list(per_action.keys()) == [str(a) for a in g["aw_actions"]]
    assert np.abs(np.array([[d[k] for k in keys] for d in per_action.values()]) - g["aw_per_action"]).max() < 1e-5
    for stride in (5, 10):
        interp, keyframes = E.interpolate_between_keyframes(pred, g[f"frame_indices_{stride}"], stride)
        assert np.array_equal(keyframes, g[f"keyframes_{stride}"])
        assert np.abs(interp - g[f"interp_{stride}"]).max() < 1e-12
