"""Host side of the evaluation protocol: the metrics and the bookkeeping that eval.py runs on
the collected predictions, in float64 numpy like the reference, but batched (no per-pose Python loop).

* ``mpjpe / nmpjpe / pmpjpe``          -- common/dataset/metrics.py:13-37, 40-84, 87-118 (+ optimal_scaling :121-134,
                                          compute_similarity_transform :137-201 as one batched SVD)
* ``frame_wise_eval / h36_action_wise_eval`` -- common/dataset/action_wise_eval.py:17-74
* ``interpolate_between_keyframes``     -- action_wise_eval.py:77-100
* ``evaluate_predictions``              -- eval.py:214-251 (ALL FRAMES / KEYFRAMES evaluation of a finished run)

Inputs: ``pred`` (B, K, 3) root-relative predictions in metres, ``gt`` (B, K, 4) = (x, y, z, valid).  Per-joint
results use -1 for invalid ground truth, averages run over entries >= 0 (action_wise_eval.py:22).
"""
import numpy as np

# common/dataset/h36m_splits.py:73-76
H36M_ACTIONS = ["Directions", "Discussion", "Eating", "Greeting", "Phoning", "Photo", "Posing", "Purchases",
                "Sitting", "SittingDown", "Smoking", "Waiting", "WalkDog", "Walking", "WalkTogether"]
METRICS = ["mpjpe", "nmpjpe", "pampjpe"]


def _finish(dist, valid, normalize):
    if normalize is False:
        return np.where(valid, dist, -1.)
    return np.sum(np.where(valid, dist, 0.)) / float(np.sum(valid > 0.))


def mpjpe(pred, gt, root_index, normalize=True):
    gt3d, valid = gt[:, :, :3], gt[:, :, 3] > 0
    p = pred - pred[:, [root_index], :]
    g = gt3d - gt3d[:, [root_index], :]
    return _finish(np.linalg.norm(p - g, ord=2, axis=-1), valid, normalize)


def nmpjpe(pred, gt, root_index, alignment="root", normalize=True):
    gt3d, valid = gt[:, :, :3], gt[:, :, 3] > 0
    if alignment == "mean":
        n = np.sum(valid, axis=1)
        g = gt3d - (np.sum(gt3d * valid[:, :, None], axis=1) / n[:, None])[:, None, :]
        p = pred - (np.sum(pred * valid[:, :, None], axis=1) / n[:, None])[:, None, :]
    else:
        g = gt3d - gt3d[:, [root_index], :]
        p = pred - pred[:, [root_index], :]
    mp, mg = p * valid[:, :, None], g * valid[:, :, None]
    s_opt = np.sum(mp * mg, axis=(1, 2)) / np.sum(mp * mp, axis=(1, 2))          # optimal_scaling, metrics.py:121-134
    return _finish(np.linalg.norm(p * s_opt[:, None, None] - g, ord=2, axis=-1), valid, normalize)


def procrustes_align(pred, gt3d):
    """Similarity transform (scale, rotation, translation) of every pose of ``pred`` onto ``gt3d``: the batched form of
    compute_similarity_transform(X=gt, Y=pred, compute_optimal_scale=True) (metrics.py:137-201)."""
    muX, muY = gt3d.mean(axis=1, keepdims=True), pred.mean(axis=1, keepdims=True)
    X0, Y0 = gt3d - muX, pred - muY
    normX = np.sqrt(np.square(X0).sum(axis=(1, 2)))
    normY = np.sqrt(np.square(Y0).sum(axis=(1, 2)))
    X0, Y0 = X0 / normX[:, None, None], Y0 / normY[:, None, None]
    A = np.einsum("bki,bkj->bij", X0, Y0)                       # X0^T Y0
    U, s, Vt = np.linalg.svd(A, full_matrices=False)
    V = np.transpose(Vt, (0, 2, 1))
    det = np.linalg.det(np.einsum("bij,bkj->bik", V, U))       # det(V U^T)
    sign = np.sign(det)
    V = V.copy(); s = s.copy()
    V[:, :, -1] *= sign[:, None]
    s[:, -1] *= sign
    T = np.einsum("bij,bkj->bik", V, U)
    trace = s.sum(axis=1)
    return (normX * trace)[:, None, None] * np.einsum("bki,bij->bkj", Y0, T) + muX


def pmpjpe(pred, gt, normalize=True):
    gt3d, valid = gt[:, :, :3], gt[:, :, 3] > 0
    aligned = procrustes_align(np.asarray(pred, np.float64), np.asarray(gt3d, np.float64))
    bad = ~np.isfinite(aligned).all(axis=(1, 2))               # the reference keeps the raw prediction when the SVD fails (:107-109)
    if bad.any():
        aligned[bad] = pred[bad]
    return _finish(np.linalg.norm(aligned - gt3d, ord=2, axis=-1), valid, normalize)


def _average(a):
    return np.mean(a[a >= 0])


def _frame_metrics(pred_3d, gt_3d, root_index):
    return {"mpjpe": mpjpe(pred_3d, gt_3d, root_index, normalize=False) * 1000.,
            "nmpjpe": nmpjpe(pred_3d, gt_3d, root_index, alignment="root", normalize=False) * 1000.,
            "pampjpe": pmpjpe(pred_3d, gt_3d, normalize=False) * 1000.}


def frame_wise_eval(pred_3d, gt_3d, root_index):
    fm = _frame_metrics(pred_3d, gt_3d, root_index)
    return {k: _average(v) for k, v in fm.items()}


def h36_action_wise_eval(pred_3d, gt_3d, actions, root_index, action_set=None):
    """-> (frame_results, average_results, per_action_results), millimetres (action_wise_eval.py:17-54)."""
    action_set = H36M_ACTIONS if action_set is None else action_set
    fm = _frame_metrics(pred_3d, gt_3d, root_index)
    per_action = {}
    for a_i, name in enumerate(action_set):
        sel = np.where(actions == a_i)
        per_action[name] = {k: _average(fm[k][sel]) for k in METRICS}
    frame_results = {k: _average(fm[k]) for k in METRICS}
    average_results = {k: np.mean([d[k] for d in per_action.values()]) for k in METRICS}
    return frame_results, average_results, per_action


def interpolate_between_keyframes(pred3d, frame_indices, keyframe_stride):
    """Linear interpolation of the predictions between keyframes (frame index % stride == 0) of each video; frames after
    the last keyframe repeat it; a drop of the frame index starts a new video (action_wise_eval.py:77-100).  Frames before
    the first keyframe of a video keep their own prediction (the reference indexes with ``None`` there)."""
    interp = np.copy(pred3d)
    frame_indices = np.asarray(frame_indices)
    keyframes = np.equal(np.mod(frame_indices, keyframe_stride), 0)
    last = None
    for i, (f, is_key) in enumerate(zip(frame_indices, keyframes)):
        if i > 0 and f <= frame_indices[i - 1]:
            last = None
        if is_key:
            if last is not None and i - last > 1:
                k = np.arange(last + 1, i)
                w_right = ((k - last) / float(i - last)).reshape((-1,) + (1,) * (pred3d.ndim - 1))
                interp[k] = pred3d[last] * (1.0 - w_right) + pred3d[i] * w_right
            last = i
        elif last is not None:
            interp[i] = pred3d[last]
    return interp, keyframes


def evaluate_predictions(pred3d, gt3d, actions, frame_indices, config, action_wise=True):
    """The bookkeeping of eval.py:195-251 on a finished run: pred3d (B,K,3), gt3d (B,K,3) root-relative, actions (B,),
    frame_indices (B,).  -> {"all_frames": ..., "keyframes": ... or None}; each entry is
    (frame_results, average_results, per_action_results) when ``action_wise`` else frame_results."""
    gt = np.concatenate([np.asarray(gt3d, np.float64), np.ones(np.shape(gt3d)[:-1] + (1,))], axis=-1)   # dummy valid flag
    pred = np.asarray(pred3d, np.float64)
    full_pred = pred
    mask_stride = config.MASK_STRIDE[0] if isinstance(config.MASK_STRIDE, (list, tuple)) else config.MASK_STRIDE
    if config.SEQUENCE_STRIDE > 1 and config.TEST_STRIDED_EVAL is True:
        strides = np.tile([config.SEQUENCE_STRIDE], reps=(len(frame_indices)))
        if getattr(config, "EVAL_DISABLE_LEARNED_UPSAMPLING", False) and mask_stride is not None:
            strides[:] = mask_stride
        pred, _ = interpolate_between_keyframes(pred, frame_indices, strides)

    def run(p, g, a):
        if action_wise:
            return h36_action_wise_eval(p, g, a, config.ROOT_KEYTPOINT)
        return frame_wise_eval(p, g, config.ROOT_KEYTPOINT)

    out = {"all_frames": run(pred, gt, np.asarray(actions)), "keyframes": None}
    if (config.SEQUENCE_STRIDE > 1 or (mask_stride is not None and mask_stride > 1)) and config.TEST_STRIDED_EVAL is True:
        input_stride = config.SEQUENCE_STRIDE if mask_stride is None else mask_stride
        key = np.equal(np.mod(frame_indices, input_stride), 0)
        out["keyframes"] = run(full_pred[key], gt[key], np.asarray(actions)[key])
    return out
