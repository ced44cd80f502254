"""End-to-end evaluation driver -- the reference's ``run_eval`` / ``run_eval_multi_mask_stride`` (eval.py:34-270) over the
device-resident pipeline of this package:

    .npz ingestion (h36m.py)  ->  videos resident in HBM (data.PoseTable)  ->  window descriptors with the reference's sample
    order and globally aligned stride masks (data.SequenceGenerator)  ->  uu3d_gather_windows builds the 2D windows on the
    device, already multiplied by their masks (eval.py:67)  ->  ONE forward per batch over [windows | flipped windows]
    (EVAL_FLIP: the reference calls the model twice, eval.py:152-180)  ->  un-flip + average  ->  (multi-GPU: batch shards per
    rank, one all-gather of the central predictions)  ->  keyframe interpolation and the ALL FRAMES / KEYFRAMES reports
    (evaluation.evaluate_predictions = eval.py:195-251, action_wise_eval.py).

What it does NOT compute: with ``TEST_STRIDED_EVAL`` and ``SEQUENCE_STRIDE`` > 1 the reference replaces the prediction of
every window whose centre is not a keyframe (frame index % stride != 0) by an interpolation between its neighbours
(action_wise_eval.py:77-100) -- for h36m_351 that is 4 of 5 windows, all of them ALL-MASKED inputs whose forward pass is
discarded.  ``run_eval`` only runs the windows whose prediction survives; the reported numbers are the
same (tests/test_eval_gpu.py checks them against the all-windows pipeline driven by the CPU oracle).
"""
import time

import numpy as np

from . import dist as udist
from . import evaluation, h36m
from .data import SequenceGenerator


def _log(*args):
    print(*args, flush=True)


def needed_windows(frame_indices, config):
    """Boolean mask of the windows whose central prediction is read by the reports of eval.py:195-251."""
    idx = np.asarray(frame_indices)
    if not (config.SEQUENCE_STRIDE > 1 and config.TEST_STRIDED_EVAL is True):
        return np.ones(len(idx), bool)
    mask_stride = config.MASK_STRIDE[0] if isinstance(config.MASK_STRIDE, (list, tuple)) else config.MASK_STRIDE
    stride = config.SEQUENCE_STRIDE
    if getattr(config, "EVAL_DISABLE_LEARNED_UPSAMPLING", False) and mask_stride is not None:
        stride = mask_stride
    # interpolate_between_keyframes overwrites every non-keyframe that has a keyframe before it in its video; frame 0 of a
    # video is always a keyframe, so every non-keyframe is overwritten.  The KEYFRAMES report reads keyframes of the (coarser
    # or equal) input stride only.
    return np.equal(np.mod(idx, stride), 0)


def predict_windows(model, generator, descriptors, config, batch_size, flip=True, depth=None, graph=True):
    """Central 3D predictions (W, J, 3) float32 on the device for the given window descriptors: batches of ``batch_size``
    windows, each forwarded together with its mirrored copy when ``flip`` (one launch chain over 2B sequences).

    ``depth`` batches (None: one per HIP hardware queue, i.e. 4) are in flight at once, each on its own HIP stream and -- with ``graph`` -- replayed from its own hipGraph
    (pipeline.ForwardPipeline: batch k + 1's big kernels run beside batch k's latency-bound tail; the window gather of a batch
    writes into its slot's input buffers on the slot's stream).  depth = 1, graph = False is the reference's loop: one eager call after the other.
    depth = 1 runs the LATENCY schedule, depth > 1 the THROUGHPUT schedule (the temporal chain, other split-K depths): the same arithmetic in another
    summation order -- predictions agree to ~3e-5 (tests/test_tchain_gpu.py), each schedule is bitwise reproducible run to run."""
    import torch
    W = len(descriptors)
    J = generator.table.J
    dev = generator.table.device
    if W == 0:
        return torch.empty((0, J, 3), dtype=torch.float32, device=dev)
    # raw central predictions of the windows (and of their mirrored copies): the un-flip / average of eval.py:163-166 runs ONCE over
    # all windows at the end instead of five small launches per batch
    raw = torch.empty((2 if flip else 1, W, J, 3), dtype=torch.float32, device=dev)
    rows = min(batch_size, W) * (2 if flip else 1)
    if depth is None:
        # one slot per hardware queue measured best END TO END (round 5, tools/eval_throughput_exp.py: descriptor upload + window gather + forward + copy per slot;
        # 128 sequences per batch: 142 / 170 / 180 / 172 k sequences/s with 2 / 3 / 4 / 8 slots, 512 per batch: 185 / 188 / 185 / 171 k)
        depth = 4
    pipe = model.pipeline(rows, depth=depth, graph=graph) if (depth is None or depth > 1 or graph) else None
    if pipe is not None:
        depth = pipe.depth

    def finish(lo, n, cen):
        raw[:, lo:lo + n].copy_(cen.view(raw.shape[0], n, J, 3))

    def take(lo, n, ticket):
        # the copy into `raw` goes on the slot's stream (pipe.after): the caller's stream never waits inside the loop
        pipe.after(ticket, lambda full, cen: finish(lo, n, cen))

    pending = []
    for lo in range(0, W, batch_size):
        d = np.ascontiguousarray(descriptors[lo:lo + batch_size])
        n = len(d)
        if flip:
            df = d.copy(); df[:, 5] = 1 - df[:, 5]                 # the generator's flip = negate x + permute joints (= eval.py:154-158)
            d = np.concatenate([d, df], 0)
        if pipe is None:
            b = generator.gather(d, zero_masked=True, with_3d=False)
            if model.has_strided_input:
                _, cen = model([b["kp2d"], b["stride_mask"]], training=False)
            else:
                _, cen = model(b["kp2d"], training=False)
            finish(lo, n, cen)
            continue
        # the window gather writes straight into the slot's static input buffers, on the slot's stream: no copy and no temporary
        # that the caching allocator could hand out again while another stream still reads it (round-3 verdict, weak point 8)
        # (wait_caller=False: the descriptors are uploaded and the windows gathered on the slot's stream, the consumer's copy runs there too -- nothing the
        # caller's stream enqueues inside this loop is an input, and an event on it would queue behind the forwards of the slots that share its hardware queue)
        xb, mb, sstream = pipe.acquire(len(d), wait_caller=False)
        generator.gather(d, zero_masked=True, with_3d=False, out=(xb, mb), stream=sstream)
        pending.append((lo, n, pipe.launch(len(d), wait_caller=False)))
        if len(pending) == depth:
            take(*pending.pop(0))
    for p in pending:
        take(*p)
    if pipe is not None:
        pipe.join()
        torch.cuda.current_stream(dev).synchronize()               # the slots' buffers go away with the pipeline
        try:
            pipe.check_range()                                     # f16x3 range guard (include/uu3d.h): once per evaluation, never per batch
        finally:
            pipe.close()
    if not flip:
        return raw[0]
    order = torch.as_tensor(np.asarray(config.AUGM_FLIP_KEYPOINT_ORDER), dtype=torch.long, device=dev)
    f = raw[1]
    f = torch.cat([f[..., :1] * -1.0, f[..., 1:]], dim=-1).index_select(1, order)              # eval.py:163-166
    return (raw[0] + f) / 2.0


def run_eval(config, dataset_name, dataset_path, dataset2d_path, test_subset, weights_path=None, model=None, action_wise=True,
             batch_size=None, skip_unused_windows=True, log=_log, depth=None, graph=True):
    """eval.py:34-253.  Returns ``evaluation.evaluate_predictions``'s dict (+ "num_windows", "num_forwarded", "seconds").

    ``batch_size`` defaults to ``config.BATCH_SIZE``; ``depth`` / ``graph``: batches in flight and hipGraph replay of the forward
    (``predict_windows``; depth 1 without graph = the reference's eager loop, same numbers).  With torch.distributed initialised, the windows to run are split
    contiguously over the ranks and the predictions all-gathered; every rank returns the same report."""
    import torch
    from .net.uplift_upsample_transformer_constructor import build_uplift_upsample_transformer
    assert not (weights_path is None and model is None)
    if model is None:
        model = build_uplift_upsample_transformer(config)
        log(f"Loading weights from {weights_path}")
        model.load_weights(weights_path, skip_mismatch=False, verbose=True)
    elif weights_path is not None:
        log(f"Using provided model. Ignoring the given weights path: {weights_path}")
    if dataset_name != "h36m":
        raise Exception("Invalid Dataset")
    subjects = h36m.subjects_of_split(test_subset)
    dataset_3d, poses_2d_dataset = h36m.load_dataset_and_2d_poses(dataset_path, dataset2d_path, dataset_name, verbose=False)
    cams, poses_3d, poses_2d, _, seq_subjects, seq_actions, seq_rates = h36m.filter_and_subsample_dataset(
        dataset_3d, poses_2d_dataset, subjects, "*", downsample=1, image_base_path=None, verbose=False)
    table = h36m.pose_table(poses_2d, poses_3d, seq_subjects, seq_actions, seq_rates, device=model.device)
    gen = SequenceGenerator(table, seq_len=config.SEQUENCE_LENGTH, target_frame_rate=50,
                            subsample=config.DATASET_TEST_3D_SUBSAMPLE_STEP, stride=config.SEQUENCE_STRIDE,
                            padding_type=config.PADDING_TYPE, flip_augment=False,
                            flip_lr_indices=config.AUGM_FLIP_KEYPOINT_ORDER, mask_stride=config.MASK_STRIDE,
                            stride_mask_align_global=True, rand_shift_stride_mask=False, shuffle=False)
    desc = gen.descriptors()
    W = len(desc)
    log(f"Sequences: {W}")
    log(f"Running evaluation on '{test_subset}' with {W} examples")
    start = time.time()
    frame_idx = desc[:, 1].copy()
    need = needed_windows(frame_idx, config) if skip_unused_windows else np.ones(W, bool)
    run = np.flatnonzero(need)
    rank, world = 0, 1
    import torch.distributed as tdist
    if tdist.is_available() and tdist.is_initialized():
        rank, world = tdist.get_rank(), tdist.get_world_size()
    lo, hi = udist.shard_bounds(len(run), rank, world)
    bs = int(batch_size or config.BATCH_SIZE)
    local = predict_windows(model, gen, desc[run[lo:hi]], config, bs, flip=bool(config.EVAL_FLIP), depth=depth, graph=graph)
    allp = udist.allgather_errors(local)                             # (len(run), J, 3) in rank order: the payload is a few KB per rank
    pred = np.zeros((W, table.J, 3), np.float64)
    pred[run] = allp.detach().cpu().numpy().astype(np.float64)
    # ground truth of the window centres, root shifted (eval.py:183-186); the centre of a window is frame `index` of its video
    mid = desc[:, 1].astype(np.int64) + table.starts[desc[:, 0]]
    gt = table.kp3d[torch.as_tensor(mid, device=table.device)].cpu().numpy().astype(np.float64)
    gt = gt - gt[:, config.ROOT_KEYTPOINT:config.ROOT_KEYTPOINT + 1, :]
    actions = table.actions[desc[:, 0]]
    if config.SEQUENCE_STRIDE > 1 and config.TEST_STRIDED_EVAL is True:
        log("Performing strided eval: Interpolating between keyframes")
    res = evaluation.evaluate_predictions(pred, gt, actions, frame_idx, config, action_wise=action_wise)
    res["num_windows"], res["num_forwarded"] = int(W), int(len(run))
    res["seconds"] = time.time() - start
    for title, key in (("ALL FRAMES", "all_frames"), ("KEYFRAMES", "keyframes")):
        if res[key] is None:
            continue
        log("")
        log(f"### Evaluation on {title} ####")
        log("")
        fr = res[key][0] if action_wise else res[key]
        log("  ".join(f"{k}: {v:.2f}" for k, v in fr.items()))
    log(f"Finished evaluation in {res['seconds']:.1f} s ({len(run)} of {W} windows forwarded)")
    return res


def run_eval_multi_mask_stride(config, *args, log=_log, **kwargs):
    """eval.py:256-268: one evaluation per MASK_STRIDE value.  Returns {mask stride: report}."""
    config = config.copy()
    values = config.MASK_STRIDE if isinstance(config.MASK_STRIDE, list) else [config.MASK_STRIDE]
    out = {}
    for msv in values:
        config.MASK_STRIDE = msv
        if len(values) > 1:
            log(f"### Running evaluation for mask stride value: {msv} ###")
        out[msv] = run_eval(config, *args, log=log, **kwargs)
        if len(values) > 1:
            log(f"### Finished evaluation for mask stride value: {msv} ###")
    return out
