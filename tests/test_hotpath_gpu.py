"""GPU tests of the HIP path through the C ABI: golden fixtures, semantic invariants, the
harness protocol, error behaviour, and size-independent properties at BASELINE's full sizes."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import uplift_upsample_3dhpe_amd as pkg
from tests import util

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

GOLDEN = sorted(glob.glob(os.path.join(util.ROOT, "tests", "golden", "*_seed*.npz")))      # the forward fixtures (make_golden.py); h36m_tiny_* belong to test_h36m_cpu.py


def _model(cfgname, seed=0, perturb=0.1, precision="f16x3"):
    cfg = util.load_config(cfgname)
    arch = pkg.arch_from_config(cfg)
    w = pkg.init_weights(arch, seed=seed, perturb=perturb)
    return cfg, arch, w, pkg.build_uplift_upsample_transformer(cfg, weights=w, precision=precision)


def _call(model, x, m):
    full, cen = model([torch.as_tensor(x).cuda(), torch.as_tensor(m).cuda()], training=False)
    torch.cuda.synchronize()
    return full.cpu().numpy(), cen.cpu().numpy()


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_golden_fixtures(path, precision):
    """Committed oracle vectors (tests/golden/make_golden.py); tolerance = north_star's 1e-4 max-abs."""
    g = np.load(path)
    cfg, arch, w, model = _model(str(g["config"]), int(g["seed"]), float(g["perturb"]), precision)
    full, cen = _call(model, g["x"], g["mask"])
    assert np.abs(full - g["full_f32"]).max() <= util.TOL_MAX_ABS
    assert np.abs(cen - g["central_f32"]).max() <= util.TOL_MAX_ABS
    rows = g["mask"].any(axis=1)
    assert np.abs(cen - g["central_f64"])[rows].max() <= util.TOL_MAX_ABS
    mid = arch.num_frames // 2
    assert np.abs(full[:, mid] - g["full_f64_center"])[rows].max() <= util.TOL_MAX_ABS
    # MPJPE budget: 0.05 mm against synthetic GT, oracle vs HIP
    from oracle import uplift_oracle as O
    gt = np.random.default_rng(0).normal(0, 0.3, size=cen.shape)
    _, a = O.frame_mpjpe_mm(cen, gt, cfg.ROOT_KEYTPOINT)
    _, b = O.frame_mpjpe_mm(g["central_f32"], gt, cfg.ROOT_KEYTPOINT)
    assert abs(a - b) <= util.TOL_MPJPE_MM


def test_masked_frames_do_not_matter_bitwise():
    cfg, arch, w, model = _model("h36m_351", seed=1)
    x, m = util.synthetic_batch(cfg, 5, seed=1)
    a = _call(model, x * m[:, :, None, None], m)
    b = _call(model, x, m)                                  # garbage in masked frames
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_batch_rows_are_independent_and_deterministic():
    cfg, arch, w, model = _model("h36m_81", seed=2)
    x, m = util.synthetic_batch(cfg, 9, seed=2)
    x = x * m[:, :, None, None]
    f, c = _call(model, x, m)
    f2, c2 = _call(model, x, m)
    assert np.array_equal(f, f2) and np.array_equal(c, c2)          # run-to-run bit identical
    perm = np.random.default_rng(0).permutation(9)
    fp, cp = _call(model, x[perm], m[perm])
    assert np.array_equal(fp, f[perm]) and np.array_equal(cp, c[perm])   # permutation equivariance
    f1, c1 = _call(model, x[4:5], m[4:5])                            # ragged: batch of one
    assert np.array_equal(f1[0], f[4]) and np.array_equal(c1[0], c[4])


@pytest.mark.parametrize("cfgname,batch", [("h36m_351", 128), ("h36m_81", 256)])
def test_full_size_properties(cfgname, batch):
    """BASELINE.json sizes: finite, equivariant under batch permutation, and the first rows agree
    with the oracle (which only has to run a handful of sequences)."""
    from oracle import uplift_oracle as O
    cfg, arch, w, model = _model(cfgname, seed=3)
    x, m = util.synthetic_batch(cfg, batch, seed=3)
    x = x * m[:, :, None, None]
    f, c = _call(model, x, m)
    assert f.shape == (batch, arch.num_frames, 17, 3) and c.shape == (batch, 17, 3)
    assert np.isfinite(f).all() and np.isfinite(c).all()
    perm = np.random.default_rng(1).permutation(batch)
    fp, cp = _call(model, x[perm], m[perm])
    assert np.array_equal(fp, f[perm]) and np.array_equal(cp, c[perm])
    idx = np.array([0, 1, 2, 3, batch // 2, batch - 1])
    fo, co = O.forward(util.hp_from_arch(arch), w, x[idx], m[idx], torch.float32)
    assert np.abs(f[idx] - fo).max() <= util.TOL_MAX_ABS and np.abs(c[idx] - co).max() <= util.TOL_MAX_ABS


def test_flip_protocol_and_mpjpe_kernel():
    from oracle import uplift_oracle as O
    from uplift_upsample_3dhpe_amd import harness
    cfg, arch, w, model = _model("h36m_81", seed=4)
    x, m = util.synthetic_batch(cfg, 4, seed=4)
    xt, mt = torch.as_tensor(x).cuda(), torch.as_tensor(m).cuda()
    seq, cen = harness.eval_step_with_flip(model, xt, mt, cfg.AUGM_FLIP_KEYPOINT_ORDER)   # masks inside
    so, co = O.eval_step_with_flip(util.hp_from_arch(arch), w, x, m, cfg.AUGM_FLIP_KEYPOINT_ORDER)
    assert np.abs(seq.cpu().numpy() - so).max() <= util.TOL_MAX_ABS
    assert np.abs(cen.cpu().numpy() - co).max() <= util.TOL_MAX_ABS
    rng = np.random.default_rng(4)
    gt = np.concatenate([rng.normal(0, 0.3, size=(4, 17, 3)), np.ones((4, 17, 1))], -1).astype(np.float32)
    gt[1, 2, 3] = 0.0
    err = harness.per_joint_error(cen, torch.as_tensor(gt).cuda(), cfg.ROOT_KEYTPOINT).cpu().numpy()
    ref = O.mpjpe(cen.cpu().numpy(), gt, cfg.ROOT_KEYTPOINT, normalize=False)
    assert err[1, 2] == -1.0 and np.abs(err - ref).max() < 1e-12          # float64, like numpy


def test_weights_roundtrip_and_errors():
    from uplift_upsample_3dhpe_amd import _capi
    cfg, arch, w, model = _model("h36m_81", seed=5)
    back = model.get_weights_dict()
    assert all(np.array_equal(back[k], w[k]) for k in w)
    assert [n for n, _ in pkg.weight_spec(arch)] == model.weight_names
    x, m = util.synthetic_batch(cfg, 2, seed=5)
    with pytest.raises(ValueError):
        model([torch.zeros(2, 40, 17, 2).cuda(), torch.as_tensor(m).cuda()])
    with pytest.raises(ValueError):
        model([torch.as_tensor(x).cuda(), torch.as_tensor(m[:, :-1]).cuda()])
    with pytest.raises(ValueError):
        model([torch.as_tensor(x), torch.as_tensor(m)])                  # host tensors: no CPU fallback
    ft, ct = model([torch.as_tensor(x).cuda(), torch.as_tensor(m).cuda()], training=True)      # train.py:478 (tests/test_train_step_gpu.py)
    assert ft.shape == (2, arch.num_frames, 17, 3) and ct.shape == (2, 17, 3)
    bad = dict(w); bad["temporal_fc/bias"] = np.zeros(50, np.float32)
    with pytest.raises(ValueError):
        model.set_weights_dict(bad)
    st = model._lib.uu3d_forward(model._h, None, None, 2, None, None, None, 0, None)
    assert st == _capi.UU3D_ERR_NOT_READY            # weights were touched but never re-committed
    model.set_weights_dict(w)
    # C ABI level: wrong element count, unknown name, workspace too small
    lib = model._lib
    buf = (C.c_float * 4)()
    assert lib.uu3d_set_weight(model._h, b"temporal_fc/bias", buf, 4) == _capi.UU3D_ERR_SHAPE
    assert lib.uu3d_set_weight(model._h, b"nope", buf, 4) == _capi.UU3D_ERR_INVALID_ARGUMENT
    xt = torch.as_tensor(x).cuda(); mt = torch.as_tensor(m).cuda().to(torch.uint8)
    out = torch.empty(2, 17, 3, device="cuda"); full = torch.empty(2, arch.num_frames, 17, 3, device="cuda")
    ws = torch.empty(1024, dtype=torch.uint8, device="cuda")
    st = lib.uu3d_forward(model._h, xt.data_ptr(), mt.data_ptr(), 2, full.data_ptr(), out.data_ptr(),
                          ws.data_ptr(), 1024, None)
    assert st == _capi.UU3D_ERR_WORKSPACE
    st = lib.uu3d_forward(model._h, xt.data_ptr(), None, 2, full.data_ptr(), out.data_ptr(), ws.data_ptr(), 1024, None)
    assert st == _capi.UU3D_ERR_INVALID_ARGUMENT                        # mask required for strided-input models


def test_mask_stride_one_config_takes_plain_input():
    """MASK_STRIDE == 1 -> has_strided_input False (constructor.py:16-21): x alone, no token blend."""
    from oracle import uplift_oracle as O
    cfg = util.load_config("h36m_81")
    cfg.MASK_STRIDE = 1
    arch = pkg.arch_from_config(cfg)
    assert not arch.has_strided_input
    w = pkg.init_weights(arch, seed=6, perturb=0.1)
    assert "strided_input_token_layer/learnable_masked_token" not in w
    model = pkg.build_uplift_upsample_transformer(cfg, weights=w)
    x, _ = util.synthetic_batch(util.load_config("h36m_81"), 3, seed=6)
    full, cen = model(torch.as_tensor(x).cuda(), training=False)
    fo, co = O.forward(util.hp_from_arch(arch), w, x, None, torch.float32)
    assert np.abs(full.cpu().numpy() - fo).max() <= util.TOL_MAX_ABS
    assert np.abs(cen.cpu().numpy() - co).max() <= util.TOL_MAX_ABS


def test_h5_weight_file_roundtrip_through_the_model(tmp_path):
    """save_weights / load_weights (train.py:706,719; weight_io.py:76-122): a second model fed only by the .h5 file
    computes bit-identical outputs."""
    cfg, arch, w, model = _model("h36m_81", seed=7)
    path = str(tmp_path / "uplift.h5")
    model.save_weights(path)
    other = pkg.build_uplift_upsample_transformer(cfg, seed=123)             # different weights
    rep = other.load_weights(path)
    assert not rep["unassigned_layers"] and not rep["unconsumed_layers"]
    x, m = util.synthetic_batch(cfg, 3, seed=7)
    a, b = _call(model, x * m[:, :, None, None], m), _call(other, x * m[:, :, None, None], m)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("cfgname,batch", [("h36m_351", 65), ("h36m_81", 129)])
def test_concurrent_half_batches_option(cfgname, batch):
    """concurrent_halves=True (two half-batch chains on two streams; the default until the row-panel GEMM made a
    single chain faster) must give the single chain's results: the halves see different M, hence different tile
    shapes and split-K choices, so equality is to rounding, not bitwise."""
    cfg = util.load_config(cfgname)
    arch = pkg.arch_from_config(cfg)
    w = pkg.init_weights(arch, seed=4, perturb=0.1)
    x, m = util.synthetic_batch(cfg, batch, seed=4)
    x = x * m[:, :, None, None]
    one = pkg.build_uplift_upsample_transformer(cfg, weights=w)
    two = pkg.build_uplift_upsample_transformer(cfg, weights=w, concurrent_halves=True)
    f1, c1 = _call(one, x, m)
    f2, c2 = _call(two, x, m)
    assert np.isfinite(f2).all() and np.isfinite(c2).all()
    assert np.abs(f1 - f2).max() <= 2e-5 and np.abs(c1 - c2).max() <= 2e-5


@pytest.mark.parametrize("env", ["UU3D_NO_PANEL", "UU3D_NO_PLANES", "UU3D_ATTN_WG", "UU3D_NO_WT", "UU3D_ATTN_F32", "UU3D_NO_MLPF", "UU3D_NO_PANEL_PROJ"])
def test_optional_kernel_paths_agree(env, monkeypatch):
    """The opt-out switches kept for A/B measurements at the full h36m_351 batch, where every one of them
    changes the kernels that run: same results as the product path to rounding.  (The round-1 experiments that measured
    neutral or slower -- LNFUSE, PANEL_ACC, LNFOLD, LN_PLANES, ATTN_PIPE, S2T_PLANES, G_TILE22 -- left the library for
    the git history (round 1).)"""
    cfg = util.load_config("h36m_351")
    arch = pkg.arch_from_config(cfg)
    w = pkg.init_weights(arch, seed=5, perturb=0.1)
    x, m = util.synthetic_batch(cfg, 128, seed=5)
    x = x * m[:, :, None, None]
    f0, c0 = _call(pkg.build_uplift_upsample_transformer(cfg, weights=w), x, m)
    monkeypatch.setenv(env, "1")
    f1, c1 = _call(pkg.build_uplift_upsample_transformer(cfg, weights=w), x, m)
    monkeypatch.delenv(env)
    d = max(np.abs(f0 - f1).max(), np.abs(c0 - c1).max())
    print(f"{env}=1: max deviation from the product path {d:.3e}")
    assert np.isfinite(f1).all() and np.isfinite(c1).all()
    assert d <= 3e-5, d
    # (the two attention kernels schedule the same exact-f32 products, the two projection kernels the same f16x3 products in the
    # same k order with the same epilogue sum: bit-identical is right for both)
    if env not in ("UU3D_ATTN_WG", "UU3D_NO_PANEL_PROJ"):
        assert d > 0.0, "the switch did not change the path"


def test_throughput_schedule_matches_and_is_an_argument():
    """The schedules of include/uu3d.h.  Below 1024 token rows, and with the temporal chain switched off, the throughput schedule only reshapes
    launches (the projection as 71 workgroups x 12 column chunks with LayerNorm 2 in the same launch): bit-identical outputs.  From 1024 rows
    on it runs the temporal chain (csrc/uu3d_tchain.h): other summation orders -- within 3e-5 of the latency schedule, bit-identical run to
    run.  Either way the schedule may arrive as the argument of uu3d_forward_ex (what the pipeline does) or as the model's default
    (uu3d_set_schedule + uu3d_forward); an unknown schedule is refused."""
    import ctypes as C
    import os
    cfg = util.load_config("h36m_351")
    arch = pkg.arch_from_config(cfg)
    w = pkg.init_weights(arch, seed=9, perturb=0.1)

    def build(tchain):
        old = os.environ.get("UU3D_TCHAIN")
        os.environ["UU3D_TCHAIN"] = tchain
        try:
            return pkg.build_uplift_upsample_transformer(cfg, weights=w)
        finally:
            if old is None:
                del os.environ["UU3D_TCHAIN"]
            else:
                os.environ["UU3D_TCHAIN"] = old

    def forward(model, x, m, schedule=None, legacy=False):
        xt, mt = torch.as_tensor(x).cuda(), torch.as_tensor(m).cuda()
        B = xt.shape[0]
        full = torch.empty((B, arch.num_frames, arch.num_keypoints, 3), dtype=torch.float32, device="cuda")
        cen = torch.empty((B, arch.num_keypoints, 3), dtype=torch.float32, device="cuda")
        mu = model._mask_u8(mt)
        stream = torch.cuda.current_stream()
        if legacy:                                       # uu3d_forward: the model's default schedule
            ws = model._workspace(B, 0)
            st = model._lib.uu3d_forward(model._h, C.c_void_p(xt.data_ptr()), C.c_void_p(mu.data_ptr()), B, C.c_void_p(full.data_ptr()),
                                         C.c_void_p(cen.data_ptr()), C.c_void_p(ws.data_ptr()), C.c_size_t(ws.numel()), C.c_void_p(stream.cuda_stream))
            assert st == 0
        else:
            model._forward(xt, mu, full, cen, 0, stream, schedule=schedule)
        torch.cuda.synchronize()
        return full.cpu().numpy(), cen.cpu().numpy()

    for tchain in ("0", "1"):
        model = build(tchain)
        for batch in (128, 17, 9):                     # 9088 rows (whole panels) / 1207 rows (a ragged last panel) / 639 rows (below the chain's 1024)
            x, m = util.synthetic_batch(cfg, batch, seed=9)
            x = x * m[:, :, None, None]
            f0, c0 = forward(model, x, m, schedule=0)
            f1, c1 = forward(model, x, m, schedule=1)
            if tchain == "0" or batch == 9:
                assert np.array_equal(f0, f1) and np.array_equal(c0, c1)
            else:
                assert 0 < max(np.abs(f0 - f1).max(), np.abs(c0 - c1).max()) <= 3e-5
                f1b, c1b = forward(model, x, m, schedule=1)
                assert np.array_equal(f1, f1b) and np.array_equal(c1, c1b)
            assert model._lib.uu3d_set_schedule(model._h, 1) == 0
            try:
                f2, c2 = forward(model, x, m, legacy=True)
            finally:
                assert model._lib.uu3d_set_schedule(model._h, 0) == 0
            assert np.array_equal(f1, f2) and np.array_equal(c1, c2)
        assert model._lib.uu3d_set_schedule(model._h, 7) != 0
        # the throughput schedule really is another set of launches
        model.set_profiling(True)
        forward(model, x, m, schedule=1)           # (639 rows: the reshaped round-4 launches in both models)
        thr_small = model.read_profile()
        x, m = util.synthetic_batch(cfg, 17, seed=9)
        forward(model, x * m[:, :, None, None], m, schedule=1)
        thr = model.read_profile()
        forward(model, x * m[:, :, None, None], m, schedule=0)
        lat = model.read_profile()
        model.set_profiling(False)
        assert any(r["name"].endswith("proj_res") and r["kernel"].startswith("gemm_panel") for r in lat)
        assert not any(r["kernel"] == "tchain" for r in lat) and not any(r["kernel"] == "tchain" for r in thr_small)
        if tchain == "1":
            assert sum(r["kernel"] == "tchain" for r in thr) == arch.temporal_depth + 2 and len(thr) < len(lat) - 15
        else:
            # (the projection's throughput shape -- one workgroup per row tile, LayerNorm 2 inside -- from 64 row tiles on: 1207 rows keep the latency shape)
            assert not any(r["kernel"] == "gemm_panel8<BiasResidualLn>" for r in thr) and not any(r["kernel"] == "gemm_panel8<BiasResidualLn>" for r in lat)
            x, m = util.synthetic_batch(cfg, 128, seed=9)
            model.set_profiling(True)
            forward(model, x * m[:, :, None, None], m, schedule=1)
            thr = model.read_profile()
            forward(model, x * m[:, :, None, None], m, schedule=0)
            lat = model.read_profile()
            model.set_profiling(False)
            assert any(r["kernel"] == "gemm_panel8<BiasResidualLn>" for r in thr) and not any(r["kernel"] == "gemm_panel8<BiasResidualLn>" for r in lat)
            assert sum(r["name"].endswith("ln2_split") for r in thr) < sum(r["name"].endswith("ln2_split") for r in lat)


def test_mpjpe_kernel_matches_the_reference_metric():
    """uu3d_mpjpe against the reference's own metrics.mpjpe(normalize=False) (tests/golden/make_metrics_golden.py
    ran common/dataset/metrics.py in the build container): per-joint errors incl. the -1 flags of invalid joints.  The
    kernel takes float32 poses, the fixture holds float64: agreement to float32 input rounding (2e-7 m)."""
    from uplift_upsample_3dhpe_amd import harness
    g = np.load(os.path.join(util.ROOT, "tests", "golden", "metrics_expected.npz"))
    pred = torch.from_numpy(g["pred"].astype(np.float32)).cuda()
    gt = torch.from_numpy(g["gt"].astype(np.float32)).cuda()
    err = harness.per_joint_error(pred, gt, int(g["root"])).cpu().numpy()
    want = g["mpjpe_jp"]
    assert err.dtype == np.float64 and err.shape == want.shape
    assert np.array_equal(err < 0, want < 0) and np.all(err[want < 0] == -1.0)
    assert np.abs(err - want)[want >= 0].max() < 2e-7
    mean_mm = err[err >= 0].mean() * 1000.0
    assert abs(mean_mm - float(g["mpjpe"]) * 1000.0) < 1e-3
