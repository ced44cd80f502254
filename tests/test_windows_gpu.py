"""uu3d_gather_windows + data.SequenceGenerator against the per-window numpy restatement (oracle/window_oracle.py):
padding of both kinds at both ends, frame-rate multiplier, globally aligned and randomly shifted stride masks, flips,
several mask strides, shuffling -- bit exact (byte / index work)."""
import numpy as np
import pytest

from uplift_upsample_3dhpe_amd import data as D
from oracle import window_oracle as WO
from tests import util

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
FLIP = [5, 4, 3, 2, 1, 0, 6, 7, 8, 9, 10, 16, 15, 14, 13, 12, 11]


def _videos(seed, lens=(3, 40, 97, 260, 7), J=17):
    rng = np.random.default_rng(seed)
    p2 = [rng.uniform(-1, 1, size=(n, J, 2)).astype(np.float32) for n in lens]
    p3 = [rng.normal(0, 0.4, size=(n, J, 3)).astype(np.float32) for n in lens]
    return p2, p3


@pytest.mark.parametrize("mode", [
    dict(seq_len=41, stride=2, padding_type="copy", mask_stride=4, stride_mask_align_global=True, flip_augment=False, shuffle=False),
    dict(seq_len=71, stride=5, padding_type="copy", mask_stride=[5, 10, 20], rand_shift_stride_mask=True, flip_augment=True, shuffle=True, subsample=3),
    dict(seq_len=9, stride=1, padding_type="zeros", mask_stride=None, flip_augment=True, in_batch_augment=True, shuffle=True),
    dict(seq_len=27, stride=3, padding_type="zeros", mask_stride=[3, 9], stride_mask_align_global=True, flip_augment=False, shuffle=False, subsample=2),
])
def test_batches_match_the_per_window_restatement(mode):
    p2, p3 = _videos(0)
    rates = [50, 100, 50, 100, 50]                     # 100 Hz videos double the stride (uplifiting_dataset.py:317-321)
    table = D.PoseTable(p2, p3, subjects=[1, 5, 6, 7, 8], actions=[0, 3, 3, 14, 2], frame_rates=rates)
    gen = D.SequenceGenerator(table, flip_lr_indices=FLIP, seed=3, **mode)
    desc = gen.descriptors()
    assert len(desc) == len(gen)
    # the same epoch from an identically seeded generator: descriptors are reproducible
    assert np.array_equal(desc, D.SequenceGenerator(table, flip_lr_indices=FLIP, seed=3, **mode).descriptors())
    take = desc[np.linspace(0, len(desc) - 1, 300).astype(int)] if len(desc) > 300 else desc
    out = gen.gather(take, zero_masked=True)
    k2, k3 = out["kp2d"].cpu().numpy(), out["kp3d"].cpu().numpy()
    sm, pm = out["stride_mask"].cpu().numpy(), out["mask"].cpu().numpy()
    pad = "edge" if mode["padding_type"] == "copy" else "constant"
    for b, (v, i, stride, ams, shift, flip) in enumerate(take):
        shift_mode = "global" if mode.get("stride_mask_align_global") else ("rand" if mode.get("rand_shift_stride_mask") else None)
        sv = shift // stride if shift_mode == "rand" else 0
        w2, m, s = WO.one_window(p2[v], int(i), mode["seq_len"], int(stride), pad, int(ams), shift_mode, sv, bool(flip), FLIP)
        w3, _, _ = WO.one_window(p3[v], int(i), mode["seq_len"], int(stride), pad, int(ams), shift_mode, sv, bool(flip), FLIP)
        assert np.array_equal(sm[b].astype(bool), s) and np.array_equal(pm[b].astype(np.float32), m)
        assert np.array_equal(k2[b], w2 * s[:, None, None].astype(np.float32))          # x * stride_mask, eval.py:67
        assert np.array_equal(k3[b], w3)
    assert np.array_equal(out["index"], take[:, 1]) and np.array_equal(out["actions"], np.array([0, 3, 3, 14, 2])[take[:, 0]])


def test_sample_list_and_random_draw_order_follow_the_reference():
    p2, _ = _videos(1, lens=(11, 30))
    table = D.PoseTable(p2, frame_rates=[50, 50])
    gen = D.SequenceGenerator(table, seq_len=9, stride=2, subsample=4, flip_augment=True, flip_lr_indices=FLIP,
                              mask_stride=[2, 4, 8], rand_shift_stride_mask=True, shuffle=True, seed=7)
    ref_rows = WO.sample_list([11, 30], [50, 50], 4, True, False)
    assert np.array_equal(gen.sequence_locations, ref_rows)
    # replay the reference's three generators by hand
    rng, srng, mrng = np.random.default_rng(7), np.random.default_rng(7), np.random.default_rng(7)
    locs = ref_rows.copy(); rng.shuffle(locs)
    exp = []
    for s_i, i, fl, fr in locs:
        ams = [2, 4, 8][mrng.integers(low=0, high=3, endpoint=False)]
        r = ams // 2
        ms = int(np.ceil((r - 1) / 2))
        sh = int(srng.integers(low=-ms, high=ms, endpoint=(r % 2 != 0))) * 2
        exp.append((s_i, i, 2, ams, sh, fl))
    assert np.array_equal(gen.descriptors(), np.array(exp, np.int32))


def test_model_consumes_gathered_batches():
    import uplift_upsample_3dhpe_amd as pkg
    cfg = util.load_config("h36m_81")
    p2, p3 = _videos(2, lens=(120, 64))
    table = D.PoseTable(p2, p3, frame_rates=[50, 50])
    gen = D.SequenceGenerator(table, seq_len=cfg.SEQUENCE_LENGTH, stride=cfg.SEQUENCE_STRIDE, padding_type="copy",
                              mask_stride=cfg.MASK_STRIDE[0], stride_mask_align_global=True, flip_augment=False, shuffle=False)
    model = pkg.build_uplift_upsample_transformer(cfg, seed=1)
    batch = next(gen.batches(16))
    full, central = model([batch["kp2d"], batch["stride_mask"]], training=False)
    assert tuple(central.shape) == (16, 17, 3) and bool(torch.isfinite(full).all())
    # masked frames arrive zeroed: the result equals the one for the explicitly masked numpy windows
    x = torch.stack([torch.from_numpy(WO.one_window(p2[v], int(i), cfg.SEQUENCE_LENGTH, int(s), "edge", int(a), "global", 0, False, FLIP)[0])
                     for v, i, s, a, sh, fl in gen.descriptors()[:16]]).cuda()
    f2, c2 = model([x * batch["stride_mask"][:, :, None, None].float(), batch["stride_mask"]], training=False)
    assert torch.equal(central, c2) and torch.equal(full, f2)


def test_world_to_cam_and_2d_matches_the_restatement():
    """uu3d_world_to_cam_2d vs oracle/window_oracle.world_to_cam_and_2d (float64): H36M-like cameras incl. distortion."""
    rng = np.random.default_rng(5)
    B, N, J = 7, 13, 17
    world = rng.normal(0, 0.5, size=(B, N, J, 3)) + np.array([0.0, 0.0, 1.0])
    cams = np.zeros((B, 18))
    q = rng.normal(size=(B, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    cams[:, :4] = q
    cams[:, 4:7] = rng.normal(0, 0.3, size=(B, 3)) + np.array([0.0, 0.0, -4.5])
    cams[:, 7:9] = [1000, 1002]
    cams[:, 9:11] = rng.uniform(2.2, 2.4, size=(B, 2)); cams[:, 11:13] = rng.uniform(-0.05, 0.05, size=(B, 2))
    cams[:, 13:16] = rng.normal(0, 0.1, size=(B, 3)); cams[:, 16:18] = rng.normal(0, 0.01, size=(B, 2))
    c3, k2 = D.world_to_cam_and_2d(torch.from_numpy(world.astype(np.float32)).cuda(), torch.from_numpy(cams.astype(np.float32)))
    c3, k2 = c3.cpu().numpy(), k2.cpu().numpy()
    w32, c32 = world.astype(np.float32).astype(np.float64), cams.astype(np.float32).astype(np.float64)
    for b in range(B):
        xc, x2 = WO.world_to_cam_and_2d(w32[b], c32[b])
        assert np.abs(c3[b] - xc).max() < 5e-6 * max(1.0, np.abs(xc).max())          # float32 arithmetic vs float64 restatement
        assert np.abs(k2[b] - x2).max() < 2e-5
    assert np.isfinite(k2).all()


def test_h36m_npz_to_model_end_to_end():
    """Files on disk -> Human3.6M ingestion (h36m.py, pinned against the reference in tests/test_h36m_cpu.py) -> device
    pose table -> windows + stride masks -> forward -> root-relative MPJPE: the eval.py data path on the tiny fixture."""
    import os
    import uplift_upsample_3dhpe_amd as pkg
    from uplift_upsample_3dhpe_amd import h36m, harness
    g = os.path.join(util.ROOT, "tests", "golden")
    dataset, keypoints = h36m.load_dataset_and_2d_poses(os.path.join(g, "h36m_tiny_3d.npz"), os.path.join(g, "h36m_tiny_2d.npz"), verbose=False)
    cams, p3d, p2d, names, subj, act, fps = h36m.filter_and_subsample_dataset(dataset, keypoints, ["S9"], "*", verbose=False)           # the tiny fixture holds S1 and S9
    table = h36m.pose_table(p2d, p3d, subj, act, fps)
    cfg = util.load_config("h36m_81")
    gen = D.SequenceGenerator(table, seq_len=cfg.SEQUENCE_LENGTH, stride=cfg.SEQUENCE_STRIDE, padding_type="copy",
                              mask_stride=cfg.MASK_STRIDE[0], stride_mask_align_global=True, flip_augment=False, shuffle=False)
    assert len(gen) == sum(len(v) for v in p2d)                       # one window per frame of the 8 test videos (S9: 8 + 6 frames x 4 cameras)
    model = pkg.build_uplift_upsample_transformer(cfg, seed=1)
    batch = next(gen.batches(16))
    full, central = model([batch["kp2d"], batch["stride_mask"]], training=False)
    mid = cfg.SEQUENCE_LENGTH // 2
    gt = torch.cat([batch["kp3d"][:, mid], torch.ones_like(batch["kp3d"][:, mid, :, :1])], -1)
    err = harness.per_joint_error(central, gt, cfg.ROOT_KEYTPOINT)
    assert tuple(err.shape) == (16, 17) and bool(torch.isfinite(err).all())
    # the window centres are the videos' own frames: the generator's 3D target equals the ingested camera-frame pose
    v, i = gen.descriptors()[0][:2]
    assert np.allclose(batch["kp3d"][0, mid].cpu().numpy(), p3d[int(v)][int(i)], atol=1e-6)


@pytest.mark.parametrize("tag,mode", [
    ("eval41", dict(seq_len=41, stride=2, padding_type="copy", mask_stride=4, stride_mask_align_global=True, flip_augment=False, shuffle=False)),
    ("train71", dict(seq_len=71, stride=5, padding_type="copy", mask_stride=[5, 10, 20], rand_shift_stride_mask=True, flip_augment=True, shuffle=True, subsample=3)),
    ("inbatch9", dict(seq_len=9, stride=1, padding_type="zeros", mask_stride=None, flip_augment=True, in_batch_augment=True, shuffle=True)),
    ("zeros27", dict(seq_len=27, stride=3, padding_type="zeros", mask_stride=[3, 9], stride_mask_align_global=True, flip_augment=False, shuffle=False, subsample=2)),
])
def test_windows_match_the_reference_generator(tag, mode):
    """data.SequenceGenerator + uu3d_gather_windows against what the reference's own H36mSequenceGenerator yields
    (tests/golden/make_windows_golden.py executed the class from uplifiting_dataset.py:213-428 in the build container):
    sample order incl. shuffling and random draws, windows, padding masks, stride masks, flips -- bit exact; the first 24
    windows of an epoch are stored, the rest enters weighted float64 checksums."""
    import os
    g = np.load(os.path.join(util.ROOT, "tests", "golden", "windows_expected.npz"))
    n_videos = len(g["lens"])
    p2 = [g[f"video2d_{v}"] for v in range(n_videos)]
    p3 = [g[f"video3d_{v}"] for v in range(n_videos)]
    table = D.PoseTable(p2, p3, subjects=g["subjects"], actions=g["actions"], frame_rates=g["rates"])
    gen = D.SequenceGenerator(table, flip_lr_indices=FLIP, seed=3, **mode)
    desc = gen.descriptors()
    assert len(desc) == len(gen) == int(g[f"{tag}/count"])
    out = gen.gather(desc, zero_masked=False)
    k2, k3 = out["kp2d"].cpu().numpy(), out["kp3d"].cpu().numpy()
    sm, pm = out["stride_mask"].cpu().numpy(), out["mask"].cpu().numpy()
    assert np.array_equal(out["index"], g[f"{tag}/index"])
    assert np.array_equal(out["subjects"], g[f"{tag}/subject"]) and np.array_equal(out["actions"], g[f"{tag}/action"])
    k = len(g[f"{tag}/seq2d"])
    assert np.array_equal(k2[:k], g[f"{tag}/seq2d"]) and np.array_equal(k3[:k], g[f"{tag}/seq3d"])
    assert np.array_equal(pm[:k].astype(np.float32), g[f"{tag}/mask"]) and np.array_equal(sm[:k].astype(bool), g[f"{tag}/stride_mask"])
    w = 1.0 + (np.arange(len(desc)) % 7)
    chk = np.array([(w * k3.astype(np.float64).sum(axis=(1, 2, 3))).sum(), (w * k2.astype(np.float64).sum(axis=(1, 2, 3))).sum(),
                    (w * pm.astype(np.float64).sum(axis=1)).sum(), (w * sm.astype(np.float64).sum(axis=1)).sum()])
    assert np.allclose(chk, g[f"{tag}/checksum"], rtol=1e-9, atol=1e-6), (chk, g[f"{tag}/checksum"])


@pytest.mark.parametrize("tag,mode", [
    ("train9", dict(seq_len=9, stride=2, padding_type="copy", mask_stride=[2, 4, 8], rand_shift_stride_mask=True, flip_augment=True, shuffle=True)),
    ("eval27", dict(seq_len=27, stride=1, padding_type="zeros", mask_stride=5, stride_mask_align_global=True, flip_augment=False, shuffle=False)),
    ("inbatch5", dict(seq_len=5, stride=1, padding_type="copy", mask_stride=None, flip_augment=True, in_batch_augment=True, shuffle=True, subsample=2)),
])
def test_amass_windows_match_the_reference_generator(tag, mode):
    """amass.AMASSDataset -> data.AmassSequenceGenerator against what the reference's own AMASSSequenceGenerator yields on the
    same tiny files (tests/golden/make_amass_windows_golden.py): sample order, per-sample camera draws, 3D windows, masks,
    flips -- bit exact -- and the camera projection kernel runs on the result."""
    import os
    from uplift_upsample_3dhpe_amd import amass
    gdir = os.path.join(util.ROOT, "tests", "golden")
    g = np.load(os.path.join(gdir, "amass_windows_expected.npz"))
    a = amass.AMASSDataset(os.path.join(gdir, "amass_tiny"), os.path.join(gdir, "h36m_tiny_3d.npz"), "train")
    seqs, rates = amass.sequences(a)
    table = D.PoseTable(None, seqs, frame_rates=rates)
    gen = D.AmassSequenceGenerator(table, amass.camera_table(a), flip_lr_indices=FLIP, seed=4, **mode)
    desc = gen.descriptors()
    assert len(desc) == len(gen) == len(g[f"{tag}/index"])
    out = gen.gather(desc, gen.camera_indices)
    assert np.array_equal(out["index"], g[f"{tag}/index"])
    assert np.array_equal(out["kp3d"].cpu().numpy(), g[f"{tag}/seq3d"])
    assert np.array_equal(out["cams"].cpu().numpy(), g[f"{tag}/cams"])
    assert np.array_equal(out["mask"].cpu().numpy().astype(np.float32), g[f"{tag}/mask"])
    assert np.array_equal(out["stride_mask"].cpu().numpy().astype(bool), g[f"{tag}/stride_mask"])
    cam3d, kp2d = D.world_to_cam_and_2d(out["kp3d"], out["cams"])[:2]
    assert tuple(kp2d.shape) == tuple(out["kp3d"].shape[:3]) + (2,) and bool(torch.isfinite(cam3d).all())
