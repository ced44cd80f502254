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
    cfg = _config(17, 40, 80, 8, 9, [3, 3])                   # head dims 5 / 10: no instantiation
    with pytest.raises(Exception, match="head dims"):
        pkg.build_uplift_upsample_transformer(cfg)
    cfg = _config(17, 16, 64, 4, 9, [3, 3])
    model = pkg.build_uplift_upsample_transformer(cfg)
    from uplift_upsample_3dhpe_amd.trainer import Trainer
    with pytest.raises(NotImplementedError, match="backward"):
        Trainer(model, cfg)
