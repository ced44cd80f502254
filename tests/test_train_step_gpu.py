"""Training step: forward in training mode, loss and every weight gradient against autograd through
the float64 oracle (train.py:464-498), with and without DropPath; then one optimizer step."""
import numpy as np
import pytest

import uplift_upsample_3dhpe_amd as pkg
from tests import util

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _setup(cfgname, B, seed, batch_norm):
    cfg = util.load_config(cfgname)
    cfg.BATCH_SIZE = batch_norm
    arch = pkg.arch_from_config(cfg)
    w = pkg.init_weights(arch, seed=seed, perturb=0.1)
    model = pkg.build_uplift_upsample_transformer(cfg, weights=w)
    x, m = util.synthetic_batch(cfg, B, seed=seed)
    gt = np.random.default_rng(seed + 50).normal(0, 0.3, size=(B, arch.num_frames, 17, 3)).astype(np.float32)
    return cfg, arch, w, model, x, m, gt


# masks without all-masked rows (those differ between fp32 and the float64 oracle by design): the gradient
# bound below is then asserted for every case
_MASKS = [(0, 0), (1, 0), (2, 0)]


@pytest.mark.parametrize("cfgname,droppath,batch_norm", [("h36m_81", False, 4), ("h36m_351", False, 4), ("h36m_81", True, 4), ("h36m_351", True, 4),
                                                         ("h36m_351", True, 512), ("h36m_81", False, 512), ("h36m_351", "strided", 4), ("h36m_81", "strided", 512),
                                                         ("h36m_351", "tokenmask", 4), ("h36m_81", "tokenmask", 4), ("h36m_351", "tokenmask_learnable", 4),
                                                         ("h36m_351", "bn", 4), ("h36m_81", "bn", 512),
                                                         ("h36m_351", "dropout", 4), ("h36m_81", "dropout", 512), ("h36m_351", "dropout_all", 4)])
def test_gradients_match_autograd(cfgname, droppath, batch_norm):
    """Every gradient tensor against float64 autograd through the oracle, <= 1e-4 of its scale.  batch_norm = 512 is the
    PRODUCTION loss normaliser (config BATCH_SIZE): d loss / d joint is 8e-7 there, which the f16x3 gradient GEMMs only
    resolve because the backward pass runs loss-scaled (uu3d_train_step.inc, gscale)."""
    from oracle import train_oracle as T
    from uplift_upsample_3dhpe_amd.trainer import Trainer
    B = 12 if droppath == "bn" else 3      # (BatchNorm over 3 samples: 1 / sqrt(var + eps) reaches 300 and multiplies every f32 rounding of the backward pass)
    cfg, arch, w, model, x, m, gt = _setup(cfgname, B, seed=7, batch_norm=batch_norm)
    if droppath == "strided":                     # DropPath inside the strided blocks too (DROP_PATH_RATE[2] > 0, u_u_t.py:110,132-137; round 3)
        cfg.DROP_PATH_RATE = [0.1, 0.1, 0.4]      # (spatial / temporal rates as shipped; with 0.2 / 0.2 and this draw one ReLU of strided block 1 sits within rounding of 0 and flips against the float64 oracle, with or without strided DropPath)
        arch = pkg.arch_from_config(cfg)
        model = pkg.build_uplift_upsample_transformer(cfg, weights=w)
    tokenmask = droppath in ("tokenmask", "tokenmask_learnable")
    if tokenmask:                                 # random token masking (TOKEN_MASK_RATE > 0: u_u_t.py:287-311,336-338; round 3): masked-token value 0, or
        cfg.TOKEN_MASK_RATE = 0.3                 # the extra trainable vector of LEARNABLE_MASKED_TOKEN (:38-50,219-220,337)
        cfg.LEARNABLE_MASKED_TOKEN = droppath == "tokenmask_learnable"
        arch = pkg.arch_from_config(cfg)
        if arch.learnable_masked_token:
            w = pkg.init_weights(arch, seed=7, perturb=0.1)
            assert "learnable_masked_token_layer/learnable_masked_token" in w
        model = pkg.build_uplift_upsample_transformer(cfg, weights=w)
    if droppath == "bn":                          # OUTPUT_BN in training mode (batch statistics + moving-average update, u_u_t.py:275-285; round 3)
        cfg.OUTPUT_BN = True
        arch = pkg.arch_from_config(cfg)
        w = pkg.init_weights(arch, seed=7, perturb=0.1)
        model = pkg.build_uplift_upsample_transformer(cfg, weights=w)
    dropout = None
    if droppath in ("dropout", "dropout_all"):    # the Dropout layers (DROP_RATE / ATTENTION_DROP_RATE > 0: vit.py:57-67,87-90,127-128,153-154; u_u_t.py:78-89,201,324; round 4),
        cfg.DROP_RATE, cfg.ATTENTION_DROP_RATE = 0.1, 0.15        # alone and ("dropout_all") together with DropPath in all three stacks and token masking
        if droppath == "dropout_all":
            cfg.DROP_PATH_RATE = [0.1, 0.1, 0.4]
            cfg.TOKEN_MASK_RATE = 0.3
        arch = pkg.arch_from_config(cfg)
        from uplift_upsample_3dhpe_amd.arch import training_unsupported
        assert training_unsupported(arch) == []
        model = pkg.build_uplift_upsample_transformer(cfg, weights=w)
        dropout = dict(rate=0.1, attn_rate=0.15, seed=0x1234567890ABCDE + batch_norm)
        tokenmask = droppath == "dropout_all"
        droppath = "strided" if droppath == "dropout_all" else False
    ms = cfg.MASK_STRIDE if isinstance(cfg.MASK_STRIDE, list) else [cfg.MASK_STRIDE]
    m = np.stack([util.eval_stride_mask(arch.num_frames, cfg.SEQUENCE_STRIDE, ms[_MASKS[b % len(_MASKS)][0]], 0) for b in range(B)])
    tr = Trainer(model, cfg)
    rng = np.random.default_rng(11)
    u = rng.random(tr.drop_path_size(B)).astype(np.float32) if droppath else None
    tmu = None
    if tokenmask:
        tmu = rng.random((B, arch.num_frames)).astype(np.float32)
        tmu[:, arch.num_frames // 2] = 0.0        # a draw that WOULD mask the central frame: it must stay
        hit = (tmu < 0.3) & (m != 0)
        hit[:, arch.num_frames // 2] = False
        assert hit.any() and ((tmu >= 0.3) & (m != 0)).any()      # real tokens both masked and kept
    loss, full, central = tr.forward_backward(torch.from_numpy(x).cuda(), torch.from_numpy(gt).cuda(), torch.from_numpy(m).cuda(),
                                              drop_path_uniform=None if u is None else torch.from_numpy(u).cuda(),
                                              token_mask_uniform=None if tmu is None else torch.from_numpy(tmu).cuda(),
                                              dropout_seed=None if dropout is None else dropout["seed"])
    torch.cuda.synchronize()
    if dropout is not None:
        assert tr.last_dropout_seed == dropout["seed"]
    dp = None
    if droppath:
        ns = arch.spatial_depth * 2 * B * arch.num_frames
        nt = arch.temporal_depth * 2 * B
        dp = dict(rates=tuple(cfg.DROP_PATH_RATE), u_spatial=u[:ns].reshape(arch.spatial_depth, 2, B * arch.num_frames),
                  u_temporal=u[ns:ns + nt].reshape(arch.temporal_depth, 2, B))
        if droppath == "strided":
            dp["u_strided"] = u[ns + nt:].reshape(len(arch.strides), 2, B)
            # make sure the test sees dropped AND kept branches in the strided blocks
            assert (np.floor(dp["u_strided"][1:] + 1 - np.linspace(0, 0.4, len(arch.strides))[1:, None, None]) == 0).any()
    ref, gref, fref, cref = T.train_step_grads(util.hp_from_arch(arch), w, x, m, gt, cfg.ROOT_KEYTPOINT, cfg.LOSS_WEIGHT_CENTER,
                                               cfg.LOSS_WEIGHT_SEQUENCE, cfg.BATCH_SIZE, dp,
                                               token_mask_cfg=None if tmu is None else dict(rate=0.3, u=tmu),
                                               bn_train=(moving := {}) if droppath == "bn" else None, dropout_cfg=dropout)
    if droppath == "bn":
        # the training-mode forward has updated the moving statistics inside the master buffer (Keras: non-trainable weights); their
        # gradient slots are zeros and the optimizer never sees them
        live = tr.params_dict()
        assert set(moving) == {"temporal_norm/moving_mean", "temporal_norm/moving_variance", "strided_temporal_norm/moving_mean", "strided_temporal_norm/moving_variance"}
        for name, want in moving.items():
            got = live[name]
            assert np.abs(got - want.numpy()).max() <= 2e-5 * max(1.0, np.abs(want.numpy()).max()), name
            assert np.abs(got - w[name]).max() > 1e-3, name          # ... and they did move
            assert not tr.grads_dict()[name].any(), name
        assert tr.n_trainable == tr.n_params - 4 * arch.d_temporal
        gref = {k: v for k, v in gref.items() if "/moving_" not in k}
    rows = m.any(axis=1)        # all-masked rows: fp32 uniform attention vs float64
    assert np.abs(full.cpu().numpy() - fref)[rows].max() <= util.TOL_MAX_ABS
    assert np.abs(central.cpu().numpy() - cref)[rows].max() <= util.TOL_MAX_ABS
    g = tr.grads_dict()
    worst = ("", 0.0)
    if rows.all():
        assert loss.cpu().numpy()[0] == pytest.approx(ref["loss"], rel=2e-5)
    gmax = max(np.abs(v).max() for v in gref.values())
    errs = []
    for name in gref:
        # scale floor: the key-bias gradients are identically zero (softmax shift invariance), so a purely
        # relative measure would compare rounding noise with rounding noise
        # OUTPUT_BN: the gradient entering a BatchNorm input sums to zero over the batch (and d x . xhat too), and the residual stream
        # carries that component unchanged through every block below the heads -- each bias / positional-encoding gradient there is a
        # column sum in which it cancels (the last strided block's conv bias: to exactly zero).  Their rounding error is that of the
        # summands (elements ~gmax), not of the much smaller sums, so the floor of the scale is 1e-3 gmax for that case (1e-4 elsewhere).
        scale = max(np.abs(gref[name]).max(), (1e-3 if droppath == "bn" else 1e-4) * gmax)
        if name.endswith("/attn/wk/bias") and np.abs(gref[name]).max() < 1e-12 * gmax:
            # structurally zero: what the HIP path holds there is the rounding residue of column sums of d K, so its natural
            # scale is the key kernel's gradient (the same d K) -- every other tensor keeps the bound above
            scale = max(scale, np.abs(gref[name.replace("/bias", "/kernel")]).max())
        err = np.abs(g[name] - gref[name]).max() / scale
        errs.append((err, name, np.abs(gref[name]).max()))
        if err > worst[1]:
            worst = (name, err)
    top = sorted(errs, reverse=True)[:12]
    for e in top:
        print("   %.2e  %-60s |g|max %.2e" % e)
    print(f"{cfgname} droppath={droppath}: worst relative gradient error {worst[1]:.2e} at {worst[0]} (largest gradient element {gmax:.2e})")
    assert rows.all()
    assert worst[1] <= 1e-4, (worst, [(n, float("%.2e" % e)) for e, n, _ in top])


def test_train_step_updates_weights_and_exports():
    from oracle import train_oracle as T
    from uplift_upsample_3dhpe_amd.trainer import Trainer
    cfg, arch, w, model, x, m, gt = _setup("h36m_81", 2, seed=9, batch_norm=2)
    m[:] = util.eval_stride_mask(arch.num_frames, cfg.SEQUENCE_STRIDE, 4, 0)      # no all-masked rows
    tr = Trainer(model, cfg)
    p0 = tr.params.clone()
    loss = tr.train_step(torch.from_numpy(x).cuda(), torch.from_numpy(gt).cuda(), torch.from_numpy(m).cuda(), drop_path_uniform=None)
    torch.cuda.synchronize()
    g = tr.grads.cpu().numpy()
    lr = T.exponential_decay(cfg.SCHEDULE_PARAMS["initial_learning_rate"], cfg.SCHEDULE_PARAMS["decay_steps"], cfg.SCHEDULE_PARAMS["decay_rate"], 0, True)
    wd = T.exponential_decay(cfg.WEIGHT_DECAY, cfg.SCHEDULE_PARAMS["decay_steps"], cfg.SCHEDULE_PARAMS["decay_rate"], 0, True)
    ref, _, _ = T.adamw_update(p0.cpu().numpy(), np.zeros_like(g), np.zeros_like(g), g, lr, wd, 0.9, 0.999, 1e-8, 1)
    assert np.array_equal(tr.params.cpu().numpy(), ref)
    assert tr.ema is not None and tr.global_step == 1                  # h36m_81: EMA_ENABLED
    # the packs were refreshed: a second forward/backward uses the new weights and matches a fresh model with them
    tr.export_to_model()
    xm = x * m[:, :, None, None]
    full, central = model([torch.from_numpy(xm).cuda(), torch.from_numpy(m).cuda()], training=False)
    _, f2, c2 = tr.forward_backward(torch.from_numpy(x).cuda(), torch.from_numpy(gt).cuda(), torch.from_numpy(m).cuda(), drop_path_uniform=None)
    assert np.abs(full.cpu().numpy() - f2.cpu().numpy()).max() <= util.TOL_MAX_ABS
    assert np.abs(central.cpu().numpy() - c2.cpu().numpy()).max() <= util.TOL_MAX_ABS


def test_shard_gradients_add_up_to_full_batch_gradient():
    """Data parallelism: with the loss normalised by the GLOBAL batch size, the gradients of two half
    batches sum to the gradient of the whole batch (what the RCCL all-reduce computes)."""
    from uplift_upsample_3dhpe_amd.trainer import Trainer
    cfg, arch, w, model, x, m, gt = _setup("h36m_81", 4, seed=13, batch_norm=4)
    tr = Trainer(model, cfg)
    T = lambda a: torch.from_numpy(a).cuda()
    tr.forward_backward(T(x), T(gt), T(m), drop_path_uniform=None)
    g_full = tr.grads.clone()
    tr.forward_backward(T(x[:2]), T(gt[:2]), T(m[:2]), drop_path_uniform=None)
    g_a = tr.grads.clone()
    tr.forward_backward(T(x[2:]), T(gt[2:]), T(m[2:]), drop_path_uniform=None)
    g_sum = (g_a + tr.grads).cpu().numpy()
    ref = g_full.cpu().numpy()
    assert np.abs(g_sum - ref).max() <= 2e-5 * np.abs(ref).max()


def test_checkpoint_resume_is_bit_identical(tmp_path):
    """Trainer.save_checkpoint / load_checkpoint (the tf.train.Checkpoint of train.py:420-436): two steps, save, two more
    steps == load into a fresh trainer and run the same two steps (weights, moments, schedules, DropPath draws)."""
    import uplift_upsample_3dhpe_amd as pkg
    from uplift_upsample_3dhpe_amd.trainer import Trainer
    cfg = util.load_config("h36m_81")
    cfg.BATCH_SIZE = 6
    arch = pkg.arch_from_config(cfg)
    rng = np.random.default_rng(9)
    x = torch.from_numpy(rng.uniform(-1, 1, size=(6, arch.num_frames, 17, 2)).astype(np.float32)).cuda()
    gt = torch.from_numpy(rng.normal(0, 0.3, size=(6, arch.num_frames, 17, 3)).astype(np.float32)).cuda()
    m = torch.from_numpy(rng.random((6, arch.num_frames)) < 0.5).cuda()
    w = pkg.init_weights(arch, seed=4)
    t1 = Trainer(pkg.build_uplift_upsample_transformer(cfg, weights=w), cfg, seed=11)
    for _ in range(2):
        t1.train_step(x, gt, m)
    ck = str(tmp_path / "ck.npz")
    t1.save_checkpoint(ck)
    ref = [t1.train_step(x, gt, m).cpu().numpy().copy() for _ in range(2)]
    t2 = Trainer(pkg.build_uplift_upsample_transformer(cfg, weights=pkg.init_weights(arch, seed=99)), cfg, seed=0)
    t2.load_checkpoint(ck)
    got = [t2.train_step(x, gt, m).cpu().numpy().copy() for _ in range(2)]
    assert all(np.array_equal(a, b) for a, b in zip(ref, got))
    assert torch.equal(t1.params, t2.params) and torch.equal(t1.optimizer.v, t2.optimizer.v)
    assert t2.global_step == 4 and t2.optimizer.iterations == 4
    if t1.ema is not None:
        assert torch.equal(t1.ema, t2.ema)


@pytest.mark.parametrize("cfgname,B", [("h36m_351", 16), ("h36m_81", 5)])
def test_side_stream_schedule_is_bit_identical_to_in_order(cfgname, B, monkeypatch):
    """The backward pass runs the parameter-gradient work on a second stream (uu3d_train_step.inc).  Both schedules launch
    the same kernels on the same data, so every gradient must agree BIT FOR BIT with the in-order run -- a missing
    dependency between the streams shows up here as a mismatch (repeated: a race need not lose every time)."""
    from uplift_upsample_3dhpe_amd.trainer import Trainer
    cfg, arch, w, model, x, m, gt = _setup(cfgname, B, seed=21, batch_norm=B)
    xs, gts, ms = torch.from_numpy(x).cuda(), torch.from_numpy(gt).cuda(), torch.from_numpy(m).cuda()
    u = torch.from_numpy(np.random.default_rng(5).random(Trainer(model, cfg).drop_path_size(B)).astype(np.float32)).cuda()

    def grads(in_order):
        if in_order:
            monkeypatch.setenv("UU3D_TRAIN_1STREAM", "1")
        else:
            monkeypatch.delenv("UU3D_TRAIN_1STREAM", raising=False)
        tr = Trainer(model, cfg)                                  # the switch is read by uu3d_train_init
        out = []
        for _ in range(4):
            loss, _, _ = tr.forward_backward(xs, gts, ms, drop_path_uniform=u)
            torch.cuda.synchronize()
            out.append((tr.grads.clone(), loss.clone()))
        return out
    ref = grads(True)
    two = grads(False)
    for (g1, l1), (g2, l2) in zip(ref, two):
        assert torch.equal(l1, l2)
        assert torch.equal(g1, g2), int((g1 != g2).sum())
    assert torch.equal(ref[0][0], ref[-1][0])


def test_training_call_of_the_model_object():
    """model([x, m], training=True) (train.py:478): the training-mode forward with DropPath -- the same outputs as the
    Trainer's forward for the same draws, different from the inference call; without DropPath draws mattering
    (rate 0) equal to inference.  model.weights / trainable_variables list name -> array views in inventory order."""
    from uplift_upsample_3dhpe_amd.trainer import Trainer
    cfg, arch, w, model, x, m, gt = _setup("h36m_81", 4, seed=3, batch_norm=4)
    xm = torch.from_numpy(x * m[:, :, None, None]).cuda(); mt = torch.from_numpy(m).cuda()
    f_inf, c_inf = model([xm, mt], training=False)
    f_tr, c_tr = model([xm, mt], training=True)                       # the model's own generator (seed 0)
    assert f_tr.shape == f_inf.shape and torch.isfinite(f_tr).all() and torch.isfinite(c_tr).all()
    assert (f_tr - f_inf).abs().max() > 1e-3                            # DropPath dropped / rescaled branches
    model2 = pkg.build_uplift_upsample_transformer(cfg, weights=w)
    f_tr2, c_tr2 = model2([xm, mt], training=True)
    assert torch.equal(f_tr, f_tr2) and torch.equal(c_tr, c_tr2)        # same seed, same draws
    cfg0 = util.load_config("h36m_81"); cfg0.DROP_PATH_RATE = [0.0, 0.0, 0.0]
    model0 = pkg.build_uplift_upsample_transformer(cfg0, weights=w)
    f0, c0 = model0([xm, mt], training=True)
    assert (f0 - f_inf).abs().max() <= 3e-5 and (c0 - c_inf).abs().max() <= 3e-5
    cfg0.DROP_PATH_RATE = [0.0, 0.0, 0.9]                               # DropPath in the strided blocks only (round 3): central changes, full does not
    f9, c9 = pkg.build_uplift_upsample_transformer(cfg0, weights=w)([xm, mt], training=True)
    assert (f9 - f0).abs().max() == 0 and (c9 - c0).abs().max() > 1e-3 and torch.isfinite(c9).all()
    cfg0.DROP_PATH_RATE = [0.0, 0.0, 0.0]; cfg0.TOKEN_MASK_RATE = 0.5   # random token masking only (round 3): training differs, inference does not
    modelm = pkg.build_uplift_upsample_transformer(cfg0, weights=w)
    fm, cm = modelm([xm, mt], training=True)
    fi, ci = modelm([xm, mt], training=False)
    assert torch.equal(fi, f_inf) and torch.equal(ci, c_inf)
    assert (fm - f0).abs().max() > 1e-3 and torch.isfinite(fm).all() and torch.isfinite(cm).all()
    cfg0.LEARNABLE_MASKED_TOKEN = True                                  # one more weight in the inventory, unused at inference
    archl = pkg.arch_from_config(cfg0)
    wl = dict(w); wl["learnable_masked_token_layer/learnable_masked_token"] = np.full((archl.d_temporal,), 0.5, np.float32)
    modell = pkg.build_uplift_upsample_transformer(cfg0, weights=wl)
    assert "learnable_masked_token_layer/learnable_masked_token" in modell.weight_names
    fli, cli = modell([xm, mt], training=False)
    assert torch.equal(fli, f_inf) and torch.equal(cli, c_inf)
    fl, cl = modell([xm, mt], training=True)                            # same draws as modelm (same seed): only the masked rows' value differs
    assert (fl - fm).abs().max() > 1e-3 and torch.isfinite(fl).all()
    names = [v.name for v in model.weights]
    assert names == model.weight_names and [v.name for v in model.trainable_variables] == names
    v = model.weights[1]
    assert np.array_equal(v.numpy(), w[v.name]) and v.shape == w[v.name].shape
    # with a Trainer attached the call uses the trainer's live weights and generator
    tr = Trainer(model, cfg, seed=5)
    tr2 = Trainer(model2, cfg, seed=5)
    a = model([xm, mt], training=True)
    u = torch.rand(tr2.drop_path_size(4), generator=tr2._rng, device="cuda", dtype=torch.float32)
    _, fb, cb = tr2.forward_backward(torch.from_numpy(x).cuda(), torch.from_numpy(gt).cuda(), mt, drop_path_uniform=u)
    assert torch.equal(a[0], fb) and torch.equal(a[1], cb)


def test_training_call_with_dropout_layers():
    """model(inputs, training=True) with DROP_RATE / ATTENTION_DROP_RATE > 0 (round 4): the Dropout layers act in the training-mode
    call only, every call draws a new mask stream from the model's generator, and the output equals the oracle's forward with the
    masks of the seed the call used (fp32, <= 1e-4)."""
    from oracle import uplift_oracle as O
    cfg, arch, w, model, x, m, gt = _setup("h36m_351", 3, seed=13, batch_norm=4)
    cfg.DROP_PATH_RATE = [0.0, 0.0, 0.0]
    cfg.DROP_RATE, cfg.ATTENTION_DROP_RATE = 0.2, 0.1
    arch = pkg.arch_from_config(cfg)
    model = pkg.build_uplift_upsample_transformer(cfg, weights=w)
    ms = cfg.MASK_STRIDE if isinstance(cfg.MASK_STRIDE, list) else [cfg.MASK_STRIDE]
    m = np.stack([util.eval_stride_mask(arch.num_frames, cfg.SEQUENCE_STRIDE, ms[0], 0) for _ in range(3)])
    xm = torch.from_numpy(x * m[:, :, None, None].astype(np.float32)).cuda()
    mt = torch.from_numpy(m).cuda()
    f_inf, c_inf = model([xm, mt], training=False)
    f1, c1 = model([xm, mt], training=True)
    s1 = model.last_dropout_seed
    f2, c2 = model([xm, mt], training=True)
    assert model.last_dropout_seed != s1 and (f1 - f2).abs().max() > 1e-3       # a fresh mask stream per call
    assert (f1 - f_inf).abs().max() > 1e-3 and torch.isfinite(f1).all() and torch.isfinite(c1).all()
    assert torch.equal(model([xm, mt], training=False)[0], f_inf)                # inference: no Dropout layer acts
    p = {k: torch.tensor(v, dtype=torch.float32) for k, v in w.items()}
    fr, cr, _ = O.forward_torch(util.hp_from_arch(arch), p, torch.tensor(x * m[:, :, None, None].astype(np.float32)), m, torch.float32,
                                dropout_cfg=dict(rate=0.2, attn_rate=0.1, seed=s1))
    assert (f1.cpu() - fr).abs().max() <= util.TOL_MAX_ABS and (c1.cpu() - cr).abs().max() <= util.TOL_MAX_ABS


def test_model_sees_trained_weights_without_explicit_export(tmp_path):
    """train.py:393,706,719 validate and checkpoint with the live model: after Trainer.train_step, get_weights / save_weights /
    model(..., training=False) must use the UPDATED weights (round-1 returned the stale host copy until export_to_model)."""
    from uplift_upsample_3dhpe_amd.trainer import Trainer
    cfg, arch, w, model, x, m, gt = _setup("h36m_81", 2, seed=9, batch_norm=2)
    cfg.EMA_ENABLED = False
    tr = Trainer(model, cfg)
    T_ = lambda a: torch.from_numpy(a).cuda()
    tr.train_step(T_(x), T_(gt), T_(m), drop_path_uniform=None)
    flat = tr.params.cpu().numpy()
    got = np.concatenate([a.ravel() for a in model.get_weights()])
    assert np.array_equal(got, flat) and not np.array_equal(got, np.concatenate([w[k].ravel() for k in w]))
    tr.train_step(T_(x), T_(gt), T_(m), drop_path_uniform=None)
    xm = T_(x * m[:, :, None, None])
    full, central = model([xm, T_(m)], training=False)                  # syncs by itself
    fresh = pkg.build_uplift_upsample_transformer(cfg, weights=dict(zip(model.weight_names, model.get_weights())))
    f2, c2 = fresh([xm, T_(m)], training=False)
    assert torch.equal(full, f2) and torch.equal(central, c2)
    assert np.array_equal(np.concatenate([a.ravel() for a in fresh.get_weights()]), tr.params.cpu().numpy())


def test_train_step_at_the_benchmarked_batch():
    """BASELINE config 5's per-GPU shape (h36m_351_pt, 64 sequences, normaliser 512, DropPath on): finite, run-to-run
    bit-identical, the two half batches' gradients add up to the full one, and the bucket callback tiles the buffer."""
    from uplift_upsample_3dhpe_amd.trainer import Trainer
    from uplift_upsample_3dhpe_amd import harness
    cfg = util.load_config("h36m_351_pt")
    assert cfg.BATCH_SIZE == 512
    arch = pkg.arch_from_config(cfg)
    model = pkg.build_uplift_upsample_transformer(cfg, weights=pkg.init_weights(arch, seed=0))
    rng = np.random.default_rng(6)
    B, N = 64, arch.num_frames
    x = torch.from_numpy(rng.uniform(-1, 1, size=(B, N, 17, 2)).astype(np.float32)).cuda()
    gt = torch.from_numpy(rng.normal(0, 0.3, size=(B, N, 17, 3)).astype(np.float32)).cuda()
    m = torch.from_numpy(harness.stride_masks_train(N, cfg.SEQUENCE_STRIDE, cfg.MASK_STRIDE, B, rng, cfg.STRIDE_MASK_RAND_SHIFT)).cuda()
    tr = Trainer(model, cfg, seed=1)
    seen = []
    orig = tr._buckets.ready
    tr._buckets.ready = lambda first, count, stream=None: (seen.append((first, count)), orig(first, count, stream))[1]
    u = torch.rand(tr.drop_path_size(B), generator=tr._rng, device="cuda", dtype=torch.float32)
    loss, _, _ = tr.forward_backward(x, gt, m, drop_path_uniform=u)
    g1, l1 = tr.grads.clone(), loss.clone()
    tr._buckets.wait()                                                    # checks that the ranges tile the buffer
    assert len(seen) == 4 and sorted(seen)[0][0] == 0 and sum(c for _, c in seen) == tr.n_params
    assert seen[0][0] + seen[0][1] == tr.n_params                         # the tail (strided blocks + heads) finishes first
    assert torch.isfinite(g1).all() and torch.isfinite(l1).all() and float(g1.abs().max()) > 0
    loss, _, _ = tr.forward_backward(x, gt, m, drop_path_uniform=u)
    assert torch.equal(tr.grads, g1) and torch.equal(loss, l1)
    tr._buckets.wait()
    # halves: DropPath draws are laid out [spatial blocks][2][B*N] then [temporal blocks][2][B]; without DropPath the halves need no re-indexing
    tr.forward_backward(x, gt, m, drop_path_uniform=None); tr._buckets.wait()
    gfull = tr.grads.clone()
    tr.forward_backward(x[:32], gt[:32], m[:32], drop_path_uniform=None); tr._buckets.wait()
    ga = tr.grads.clone()
    tr.forward_backward(x[32:], gt[32:], m[32:], drop_path_uniform=None); tr._buckets.wait()
    assert (ga + tr.grads - gfull).abs().max() <= 2e-5 * gfull.abs().max()
    tr.apply_gradients()
    assert torch.isfinite(tr.params).all()


@pytest.mark.parametrize("B,alt", [(5, {"UU3D_TRAIN_SPATIAL_UNFUSED": "1", "UU3D_ATTN_BWD_GENERIC": "1"}), (16, {"UU3D_TRAIN_NO_PANEL": "1"})])
def test_fused_and_unfused_spatial_training_forward_agree(monkeypatch, B, alt):
    """The training-mode forward of the spatial stack: ONE launch of spatial_stack_h3_kernel<.., TRAIN> (saved activations, row
    statistics, DropPath gates written by the kernel) against the chain of generic kernels it replaces
    (UU3D_TRAIN_SPATIAL_UNFUSED=1, read by uu3d_train_init), and the spatial attention backward on attn_small_bwd_kernel<17>
    against the generic kernel (UU3D_ATTN_BWD_GENERIC=1): outputs, loss and every gradient tensor, with DropPath."""
    from uplift_upsample_3dhpe_amd.trainer import Trainer
    # second case: 16 x 71 = 1136 token rows >= 1024: the LayerNorm-fed Dense layers of the temporal blocks and of strided block 1 run
    # on ln_split_frag_stats + the row-panel GEMM (UU3D_TRAIN_NO_PANEL=1: row_stats + the tiled GEMM with the LayerNorm loader)
    results = []
    for env in ({}, alt):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        cfg, arch, w, model, x, m, gt = _setup("h36m_351", B, seed=21, batch_norm=4)
        tr = Trainer(model, cfg)
        u = torch.from_numpy(np.random.default_rng(13).random(tr.drop_path_size(B)).astype(np.float32)).cuda()
        loss, full, central = tr.forward_backward(torch.from_numpy(x).cuda(), torch.from_numpy(gt).cuda(), torch.from_numpy(m).cuda(), drop_path_uniform=u)
        torch.cuda.synchronize()
        results.append((loss.cpu().numpy().copy(), full.cpu().numpy().copy(), central.cpu().numpy().copy(), {k_: v_.copy() for k_, v_ in tr.grads_dict().items()}))
        for k in env:
            monkeypatch.delenv(k)
    (l0, f0, c0, g0), (l1, f1, c1, g1) = results
    assert np.abs(f0 - f1).max() <= 2e-5 and np.abs(c0 - c1).max() <= 2e-5
    assert l0[0] == pytest.approx(l1[0], rel=1e-5)
    assert np.abs(f0 - f1).max() > 0.0, "the switch did not change the path"
    worst = ("", 0.0)
    gmax = max(np.abs(v).max() for v in g1.values())
    for name in g0:
        scale = max(np.abs(g1[name]).max(), 1e-4 * gmax)     # floor: the key-bias gradients are identically zero (see above)
        e = np.abs(g0[name] - g1[name]).max() / scale
        if e > worst[1]:
            worst = (name, e)
    print(f"fused vs unfused spatial training path: worst gradient deviation {worst[1]:.2e} of scale ({worst[0]})")
    assert worst[1] <= 1e-4, worst


def test_ema_export_survives_the_validation_forward():
    """train.py:398-401 validates on ema_model.  Trainer.export_to_model(use_ema=True) must leave the EMA weights in the model for the
    inference call that follows (a dirty flag made model(...) re-export the live weights over them), and the next
    train step must make the model follow the live weights again."""
    from uplift_upsample_3dhpe_amd.trainer import Trainer
    cfg, arch, w, model, x, m, gt = _setup("h36m_81", 2, seed=9, batch_norm=2)
    assert cfg.EMA_ENABLED                                               # h36m_81 ships with EMA on
    cfg.EMA_DECAY = 0.5                                                  # far from 1: EMA and live weights clearly differ after two steps
    tr = Trainer(model, cfg)
    T_ = lambda a: torch.from_numpy(a).cuda()
    for _ in range(2):
        tr.train_step(T_(x), T_(gt), T_(m), drop_path_uniform=None)
    ema, live = tr.ema.cpu().numpy(), tr.params.cpu().numpy()
    assert not np.array_equal(ema, live)
    tr.export_to_model(use_ema=True)
    xm = T_(x * m[:, :, None, None])
    full, central = model([xm, T_(m)], training=False)                   # must NOT resync from the trainer
    got = np.concatenate([a.ravel() for a in model.get_weights()])
    assert np.array_equal(got, ema) and not np.array_equal(got, live)
    names = model.weight_names
    o, wd = 0, {}
    for n, s in model._spec:
        k = int(np.prod(s)); wd[n] = ema[o:o + k].reshape(s); o += k
    fresh = pkg.build_uplift_upsample_transformer(cfg, weights=wd)
    f2, c2 = fresh([xm, T_(m)], training=False)
    assert torch.equal(full, f2) and torch.equal(central, c2)
    # the next step: live weights again, without an explicit export
    tr.train_step(T_(x), T_(gt), T_(m), drop_path_uniform=None)
    got = np.concatenate([a.ravel() for a in model.get_weights()])
    assert np.array_equal(got, tr.params.cpu().numpy())
    assert len(names) == len(model._spec)


def test_assign_after_an_ema_export_keeps_the_trained_weights():
    """after Trainer.export_to_model(use_ema=True) the model holds the EMA weights and is not "dirty"; a
    WeightView.assign then reloaded the trainer's master buffer from the model = EMA weights + the one assigned tensor, and the
    trained weights were gone without a word.  Now the live weights are exported first: every OTHER tensor of trainer.params is
    unchanged by the assign, and repeated reloads do not probe streams again (the training state is re-used)."""
    from uplift_upsample_3dhpe_amd.trainer import Trainer
    cfg, arch, w, model, x, m, gt = _setup("h36m_81", 2, seed=11, batch_norm=2)
    cfg.EMA_DECAY = 0.5
    tr = Trainer(model, cfg)
    T_ = lambda a: torch.from_numpy(a).cuda()
    for _ in range(2):
        tr.train_step(T_(x), T_(gt), T_(m), drop_path_uniform=None)
    live = tr.params.clone()
    assert not torch.equal(tr.ema, live)
    tr.export_to_model(use_ema=True)
    v0 = model.weights[0]
    n0 = int(np.prod(v0.shape))
    new0 = (live[:n0].cpu().numpy().reshape(v0.shape) * 0.5).astype(np.float32)
    v0.assign(new0)
    tr.forward_backward(T_(x), T_(gt), T_(m), drop_path_uniform=None)       # flushes the assign into the master buffer
    assert np.array_equal(tr.params[:n0].cpu().numpy(), new0.ravel())
    assert torch.equal(tr.params[n0:], live[n0:]), "the trained weights were replaced by the EMA weights"
    # assign once per step, tf.Variable style: uu3d_train_init re-uses its state (side streams, events, pack arenas) and only redoes
    # "master <- host weights" + the operand repack; the steps stay finite and the assigned tensor is what the step started from
    for k in range(4):
        v0.assign(new0 * (1.0 + 0.01 * k))
        tr.forward_backward(T_(x), T_(gt), T_(m), drop_path_uniform=None)
        assert np.array_equal(tr.params[:n0].cpu().numpy(), (new0 * (1.0 + 0.01 * k)).astype(np.float32).ravel())
        tr.apply_gradients()
    torch.cuda.synchronize()
    assert torch.isfinite(tr.params).all()


def test_two_backward_passes_without_an_optimizer_step():
    """forward_backward twice (gradient inspection / accumulation), then apply_gradients: the bucket bookkeeping of the first pass
    must not leak into the second ("gradient ranges do not tile the buffer")."""
    from uplift_upsample_3dhpe_amd.trainer import Trainer
    cfg, arch, w, model, x, m, gt = _setup("h36m_81", 2, seed=3, batch_norm=2)
    tr = Trainer(model, cfg)
    T_ = lambda a: torch.from_numpy(a).cuda()
    tr.forward_backward(T_(x), T_(gt), T_(m), drop_path_uniform=None)
    g1 = tr.grads.clone()
    tr.forward_backward(T_(x), T_(gt), T_(m), drop_path_uniform=None)
    assert torch.equal(g1, tr.grads)
    tr.apply_gradients()
    assert tr.global_step == 1


def test_weight_views_assign_like_tf_variables():
    """model.weights entries behave like the tf.Variables of train.py:503: assign / assign_sub of single variables (one
    uu3d_set_weight each, ONE deferred commit) reach the inference forward, get_weights, and an attached Trainer's master buffer --
    the reference's EMA loop `for w, ema_w in zip(model.weights, ema_model.weights): ema_w.assign_sub((1 - d) * (ema_w - w))`."""
    from uplift_upsample_3dhpe_amd.trainer import Trainer
    cfg, arch, w, model, x, m, gt = _setup("h36m_81", 2, seed=4, batch_norm=2)
    cfg.EMA_ENABLED = False
    ema_model = pkg.build_uplift_upsample_transformer(cfg, weights=w)
    w2 = pkg.init_weights(arch, seed=99, perturb=0.1)
    model.set_weights_dict(w2)
    d = 0.75
    for v, ev in zip(model.weights, ema_model.weights):
        ev.assign_sub((1.0 - d) * (ev.numpy() - v.numpy()))
    want = {k: (w[k] - np.float32(1.0 - d) * (w[k] - w2[k])).astype(np.float32) for k in w}
    got = ema_model.get_weights_dict()
    assert all(np.array_equal(got[k], want[k]) for k in want)
    T_ = lambda a: torch.from_numpy(a).cuda()
    xm = T_(x * m[:, :, None, None])
    fresh = pkg.build_uplift_upsample_transformer(cfg, weights=want)
    a, b = ema_model([xm, T_(m)], training=False), fresh([xm, T_(m)], training=False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    # a Trainer follows a single assigned variable on its next step
    tr = Trainer(ema_model, cfg)
    v0 = ema_model.weights[0]
    v0.assign(v0.numpy() * 0.5)
    tr.forward_backward(T_(x), T_(gt), T_(m), drop_path_uniform=None)
    n0 = int(np.prod(v0.shape))
    assert np.array_equal(tr.params[:n0].cpu().numpy(), (want[v0.name] * 0.5).ravel())


def test_batchnorm_heads_through_a_whole_training_step(tmp_path):
    """OUTPUT_BN = true through Trainer.train_step (round 3): the moving statistics are written by the training-mode forward only --
    AdamW (weight decay!) and the gradient buffer never touch them, the EMA copy follows them like every other weight
    (train.py:502-504 loops over model.weights) -- and the inference call after an export normalises with the UPDATED statistics."""
    from oracle import uplift_oracle as O
    from uplift_upsample_3dhpe_amd.trainer import Trainer
    cfg = util.load_config("h36m_81")
    cfg.OUTPUT_BN, cfg.BATCH_SIZE, cfg.EMA_ENABLED = True, 8, True
    arch = pkg.arch_from_config(cfg)
    w = pkg.init_weights(arch, seed=4, perturb=0.1)
    model = pkg.build_uplift_upsample_transformer(cfg, weights=w)
    x, m = util.synthetic_batch(cfg, 8, seed=4)
    gt = np.random.default_rng(54).normal(0, 0.3, size=(8, arch.num_frames, 17, 3)).astype(np.float32)
    xt, gtt, mt = torch.from_numpy(x).cuda(), torch.from_numpy(gt).cuda(), torch.from_numpy(m).cuda()
    tr = Trainer(model, cfg, seed=1)
    names = [n for n in tr.params_dict() if "/moving_" in n]
    assert len(names) == 4 and tr.optimizer.params.numel() == tr.n_trainable == tr.n_params - 4 * arch.d_temporal
    _, _, _ = tr.forward_backward(xt, gtt, mt, drop_path_uniform=None)
    after_fwd = {n: tr.params_dict()[n].copy() for n in names}
    tr.apply_gradients()
    p1 = tr.params_dict()
    for n in names:
        assert np.array_equal(p1[n], after_fwd[n]), n                   # the optimizer step left them alone
        assert np.abs(after_fwd[n] - w[n]).max() > 1e-3, n               # the forward moved them
    assert np.abs(p1["temporal_norm/gamma"] - w["temporal_norm/gamma"]).max() > 0     # trainable BN weights do step
    ema = tr.ema.cpu().numpy()[tr.n_trainable:]
    live = tr.params.cpu().numpy()[tr.n_trainable:]
    w0 = np.concatenate([w[n].ravel() for n, _ in model._spec if "/moving_" in n])
    d = 0.1                                                             # ema_decay_value(EMA_DECAY, 0) = min(EMA_DECAY, 1 / 10)
    assert np.allclose(ema, w0 - (1 - d) * (w0 - live), rtol=0, atol=1e-6)
    # inference with the trained weights = the oracle's inference on the exported weights (moving statistics included)
    tr.export_to_model(use_ema=False)
    xm = torch.from_numpy(x * m[:, :, None, None]).cuda()
    full, cen = model([xm, mt], training=False)
    fo, co = O.forward(util.hp_from_arch(arch), tr.params_dict(), x * m[:, :, None, None], m)
    assert np.abs(full.cpu().numpy() - fo).max() <= util.TOL_MAX_ABS and np.abs(cen.cpu().numpy() - co).max() <= util.TOL_MAX_ABS
    # checkpoint round trip with the shorter optimizer slots
    path = str(tmp_path / "bn.npz")
    tr.save_checkpoint(path)
    tr2 = Trainer(pkg.build_uplift_upsample_transformer(cfg, weights=w), cfg, seed=1)
    tr2.load_checkpoint(path)
    assert torch.equal(tr2.params, tr.params) and torch.equal(tr2.optimizer.m, tr.optimizer.m)
    # model(training=True) (train.py:478) takes the batch statistics too
    f_tr, c_tr = model([xm, mt], training=True)
    assert torch.isfinite(f_tr).all() and (f_tr - full).abs().max() > 1e-4


def test_pipeline_sees_the_trainers_weights():
    """A ForwardPipeline built before training keeps working while a Trainer changes the weights (validation inside a training loop,
    train.py:517-536): every submit after an optimizer step runs on the freshly exported weights -- the same numbers as model(...)."""
    from uplift_upsample_3dhpe_amd.trainer import Trainer
    cfg, arch, w, model, x, m, gt = _setup("h36m_81", 6, seed=12, batch_norm=6)
    xm = torch.from_numpy(x * m[:, :, None, None]).cuda(); mt = torch.from_numpy(m).cuda()
    pipe = model.pipeline(6)                                        # one slot per hardware queue, hipGraphs captured with the initial weights
    f0, c0 = [t.clone() for t in next(iter(pipe.run([(xm, mt)])))]
    tr = Trainer(model, cfg, seed=2)
    for step in range(2):
        tr.train_step(torch.from_numpy(x).cuda(), torch.from_numpy(gt).cuda(), mt)
        got = [(f.clone(), c.clone()) for f, c in pipe.run([(xm, mt)] * 5)]      # more batches than slots: every slot replays
        want = model([xm, mt], training=False)
        for f, c in got:
            assert torch.equal(f, want[0]) and torch.equal(c, want[1]), step
        assert (want[1] - c0).abs().max() > 1e-6                    # ... and they are not the initial weights' numbers
    pipe.close()


def test_free_running_pipeline_loop_sees_the_trainers_weights():
    """The step loop of bench.py / eval.predict_windows (round 5): inputs resident in the slots, ``launch(wait_caller=False)``, the consumer on
    the slot's stream (``after``), ``join`` at the end -- nothing in the loop touches the caller's stream.  A weight change between two such loops
    still reaches every slot: the packs are rewritten on the caller's stream and every slot's stream waits for that, whatever wait_caller says;
    ``drain`` waits for slots whose results were taken with ``after``.  Results = ``model.call_scheduled(..., "throughput")``."""
    from uplift_upsample_3dhpe_amd.trainer import Trainer
    cfg, arch, w, model, x, m, gt = _setup("h36m_351", 16, seed=13, batch_norm=16)
    xm = torch.from_numpy(x * m[:, :, None, None]).cuda(); mt = torch.from_numpy(m).cuda()
    pipe = model.pipeline(16)
    pipe.preload(xm, mt)
    table = torch.zeros((24, 16, arch.num_keypoints, 3), device="cuda")

    def loop():
        tickets = []
        for k in range(24):
            tickets.append((k, pipe.launch(wait_caller=False)))
            if len(tickets) == pipe.depth:
                kk, t = tickets.pop(0)
                pipe.after(t, lambda f, c, kk=kk: table[kk].copy_(c))
        for kk, t in tickets:
            pipe.after(t, lambda f, c, kk=kk: table[kk].copy_(c))
        pipe.join()
        torch.cuda.synchronize()
    loop()
    want0 = model.call_scheduled([xm, mt], "throughput")[1]
    assert all(torch.equal(table[k], want0) for k in range(24))
    assert (want0 - model([xm, mt], training=False)[1]).abs().max() <= 3e-5          # (1136 rows: the temporal chain)
    tr = Trainer(model, cfg, seed=2)
    for step in range(2):
        tr.train_step(torch.from_numpy(x).cuda(), torch.from_numpy(gt).cuda(), mt)
        table.zero_()
        loop()                                                                       # (no result() in between: every slot's last use was an after())
        want = model.call_scheduled([xm, mt], "throughput")[1]
        assert all(torch.equal(table[k], want) for k in range(24)), step
        assert (want - want0).abs().max() > 1e-6
    pipe.close()


@pytest.mark.parametrize("n,strides", [(125, [5, 5, 5]), (128, [4, 4, 8])])
def test_gradients_at_the_longest_sequences(n, strides):
    """The training step at 97 .. 128 tokens (round 4: the limit was 96, set by the Dropout path's LDS-resident attention backward; without
    ATTENTION_DROP_RATE the MFMA backward holds 8 x 16 tokens in registers): every gradient tensor against float64 autograd."""
    from oracle import train_oracle as T
    from uplift_upsample_3dhpe_amd.trainer import Trainer
    cfg = util.load_config("h36m_351")
    cfg.SEQUENCE_LENGTH, cfg.STRIDES, cfg.PADDINGS, cfg.BATCH_SIZE = n, strides, None, 4
    cfg.SPATIAL_TRANSFORMER_BLOCKS, cfg.TEMPORAL_TRANSFORMER_BLOCKS, cfg.DROP_PATH_RATE = 1, 2, [0.0, 0.0, 0.0]
    arch = pkg.arch_from_config(cfg)
    w = pkg.init_weights(arch, seed=3, perturb=0.1)
    model = pkg.build_uplift_upsample_transformer(cfg, weights=w)
    B = 2
    rng = np.random.default_rng(8)
    x = rng.uniform(-1, 1, size=(B, n, 17, 2)).astype(np.float32)
    gt = rng.normal(0, 0.3, size=(B, n, 17, 3)).astype(np.float32)
    ms = cfg.MASK_STRIDE if isinstance(cfg.MASK_STRIDE, list) else [cfg.MASK_STRIDE]
    m = np.stack([util.eval_stride_mask(n, cfg.SEQUENCE_STRIDE, ms[b % 2], 0) for b in range(B)])
    tr = Trainer(model, cfg)
    loss, full, central = tr.forward_backward(torch.from_numpy(x).cuda(), torch.from_numpy(gt).cuda(), torch.from_numpy(m).cuda(), drop_path_uniform=None)
    torch.cuda.synchronize()
    ref, gref, fref, cref = T.train_step_grads(util.hp_from_arch(arch), w, x, m, gt, cfg.ROOT_KEYTPOINT, cfg.LOSS_WEIGHT_CENTER, cfg.LOSS_WEIGHT_SEQUENCE, cfg.BATCH_SIZE, None)
    assert float(loss.cpu()[0]) == pytest.approx(ref["loss"], rel=2e-5)
    g = tr.grads_dict()
    gmax = max(np.abs(v).max() for v in gref.values())
    worst = max(((np.abs(g[k] - gref[k]).max() / max(np.abs(gref[k]).max(), 1e-4 * gmax), k) for k in gref if not k.endswith("/attn/wk/bias")), key=lambda t: t[0])
    print(f"{n} tokens: worst relative gradient error {worst[0]:.2e} at {worst[1]}")
    assert worst[0] <= 1e-4, worst
