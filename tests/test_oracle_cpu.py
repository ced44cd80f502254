"""Pins the CPU oracle: golden fixtures, an independently written numpy twin, and invariants
that follow from the reference's semantics.  PARITY UNPINNED against
TensorFlow itself (the reference ships no tests or vectors and TF cannot be installed here)."""
import glob
import os

import numpy as np
import pytest
import torch

import uplift_upsample_3dhpe_amd as pkg
from oracle import uplift_oracle as O
from oracle import uplift_oracle_np as ON
from tests import util
from tests.golden.make_golden import weights_checksum
from uplift_upsample_3dhpe_amd.weights import weight_spec

GOLDEN = sorted(glob.glob(os.path.join(util.ROOT, "tests", "golden", "*_seed*.npz")))


def _setup(cfgname, seed=0, perturb=0.1):
    cfg = util.load_config(cfgname)
    arch = pkg.arch_from_config(cfg)
    return cfg, arch, util.hp_from_arch(arch), pkg.init_weights(arch, seed=seed, perturb=perturb)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_reproduces_golden(path):
    g = np.load(path)
    cfg, arch, hp, w = _setup(str(g["config"]), int(g["seed"]), float(g["perturb"]))
    assert weights_checksum(w) == str(g["weights_sha256"])
    f32, c32 = O.forward(hp, w, g["x"], g["mask"], torch.float32)
    # same code, same machine class: tight; a different BLAS may reorder sums slightly
    assert np.abs(f32 - g["full_f32"]).max() <= 2e-5
    assert np.abs(c32 - g["central_f32"]).max() <= 2e-5
    rows = g["mask"].any(axis=1)
    assert np.abs(c32 - g["central_f64"])[rows].max() <= util.TOL_MAX_ABS


@pytest.mark.parametrize("cfgname", ["h36m_81", "h36m_351"])
def test_oracle_matches_numpy_twin(cfgname):
    cfg, arch, hp, w = _setup(cfgname, seed=5)
    x, m = util.synthetic_batch(cfg, 2, seed=5, mask_specs=[(cfg.MASK_STRIDE[1], 0), (cfg.MASK_STRIDE[2], cfg.SEQUENCE_STRIDE)])
    xm = x * m[:, :, None, None]
    f64, c64 = O.forward(hp, w, xm, m, torch.float64)
    fn, cn = ON.forward(hp, w, xm, m, np.float64)
    assert np.abs(f64 - fn).max() < 1e-11 and np.abs(c64 - cn).max() < 1e-11


def test_output_bn_heads_in_both_oracles():
    """OUTPUT_BN = true (u_u_t.py:275-285): BatchNormalization in front of both heads, inference form.  The torch restatement
    against the independently written numpy twin, and against the algebra the HIP library uses (the affine folded into the
    Dense head: W' = s W, b' = b + (beta - mean s) W)."""
    cfg = util.load_config("h36m_81")
    cfg.OUTPUT_BN = True
    arch = pkg.arch_from_config(cfg)
    hp = util.hp_from_arch(arch)
    assert hp["output_bn"] is True
    w = pkg.init_weights(arch, seed=8, perturb=0.2)
    names = [n for n, _ in pkg.weight_spec(arch)]
    assert names[-4:] == ["temporal_norm/moving_mean", "temporal_norm/moving_variance",
                          "strided_temporal_norm/moving_mean", "strided_temporal_norm/moving_variance"]        # non-trainable weights last (Keras model.weights)
    assert names.index("temporal_norm/gamma") + 2 == names.index("temporal_fc/kernel")
    x, m = util.synthetic_batch(cfg, 2, seed=8)
    xm = x * m[:, :, None, None]
    f64, c64 = O.forward(hp, w, xm, m, torch.float64)
    fn, cn = ON.forward(hp, w, xm, m, np.float64)
    assert np.abs(f64 - fn).max() < 1e-11 and np.abs(c64 - cn).max() < 1e-11
    # folded form: a model WITHOUT the BatchNorm layers and with rewritten head weights gives the same outputs
    cfg0 = util.load_config("h36m_81")
    hp0 = util.hp_from_arch(pkg.arch_from_config(cfg0))
    w0 = {k: v for k, v in w.items() if "_norm/" not in k or k.startswith("spatial_norm")}
    for fc, bn in (("temporal_fc", "temporal_norm"), ("strided_temporal_fc", "strided_temporal_norm")):
        s = w[f"{bn}/gamma"].astype(np.float64) / np.sqrt(w[f"{bn}/moving_variance"].astype(np.float64) + 1e-5)
        sh = w[f"{bn}/beta"] - w[f"{bn}/moving_mean"] * s
        w0[f"{fc}/kernel"] = s[:, None] * w[f"{fc}/kernel"]
        w0[f"{fc}/bias"] = w[f"{fc}/bias"] + sh @ w[f"{fc}/kernel"]
    f0, c0 = O.forward(hp0, w0, xm, m, torch.float64)
    assert np.abs(f64 - f0).max() < 1e-10 and np.abs(c64 - c0).max() < 1e-10
    assert np.abs(c64 - O.forward(hp0, {k: v for k, v in w.items() if k in w0}, xm, m, torch.float64)[1]).max() > 1e-3      # the layers do something


def test_masked_frame_content_is_irrelevant():
    cfg, arch, hp, w = _setup("h36m_81", seed=1)
    x, m = util.synthetic_batch(cfg, 2, seed=1, mask_specs=[(10, 0), (20, 2)])
    a = O.forward(hp, w, x * m[:, :, None, None], m, torch.float32)
    b = O.forward(hp, w, x, m, torch.float32)                 # garbage left in masked frames
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])   # u_u_t.py:350


def test_all_ones_mask_equals_unmasked_blend():
    cfg, arch, hp, w = _setup("h36m_81", seed=2)
    x, _ = util.synthetic_batch(cfg, 2, seed=2)
    ones = np.ones((2, arch.num_frames), bool)
    a = O.forward(hp, w, x, ones, torch.float64)
    hp2 = dict(hp, has_strided_input=False)
    b = O.forward(hp2, w, x, None, torch.float64)
    assert np.abs(a[0] - b[0]).max() < 1e-12 and np.abs(a[1] - b[1]).max() < 1e-12


def test_all_masked_row_is_finite_and_fp32_uniform_attention():
    """fp32 `logits + (-1e9)` rounds every logit to -1e9: block-1 attention becomes uniform.
    (float64 keeps the logits; the two differ -- fp32 is what the reference computes.)"""
    cfg, arch, hp, w = _setup("h36m_81", seed=3)
    x, _ = util.synthetic_batch(cfg, 1, seed=3)
    zeros = np.zeros((1, arch.num_frames), bool)
    f, c, att = O.forward(hp, w, x * 0, zeros, torch.float32, return_attention=True)
    assert np.isfinite(f).all() and np.isfinite(c).all()
    assert np.allclose(att[0], 1.0 / arch.num_frames, rtol=0, atol=1e-7)


@pytest.mark.parametrize("cfgname", ["h36m_81", "h36m_351"])
def test_strided_residual_rows(cfgname):
    """With zero conv/attention weights the strided block is x -> (x + pe)[residual rows]."""
    cfg, arch, hp, w = _setup(cfgname, seed=4, perturb=0.0)
    pre = "strided_temporal_block_1"
    for k in w:
        if k.startswith(pre) and ("kernel" in k or "bias" in k):
            w[k] = np.zeros_like(w[k])
    p = {k: torch.as_tensor(v) for k, v in w.items()}
    L = arch.num_frames
    x = torch.arange(L, dtype=torch.float32)[None, :, None].repeat(1, 1, arch.d_temporal)
    pe = torch.zeros(L, arch.d_temporal)
    y, _ = O.strided_transformer_block(p, pre, x, pe, arch.num_heads, arch.strides[0], arch.paddings[0])
    rows = y[0, :, 0].numpy().astype(int).tolist()
    s, pad = arch.strides[0], arch.paddings[0]
    expect = list(range(1 if pad[0] == 0 else 0, L - (1 if pad[1] == 0 else 0), s))
    assert rows == expect
    assert len(rows) == arch.strided_lengths[1]


def test_stride_mask_rule_table():
    # SURVEY.md appendix B
    f = O.stride_mask_eval
    assert f(41, 2, 4, 0).sum() == 21 and f(41, 2, 4, 0)[20]
    assert f(41, 2, 4, 2).sum() == 20 and not f(41, 2, 4, 2)[20]
    assert f(41, 2, 4, 1).sum() == 0
    assert f(71, 5, 5, 0).all() and f(71, 5, 5, 3).sum() == 0
    assert f(71, 5, 10, 0).sum() == 35 and f(71, 5, 10, 5).sum() == 36
    assert f(71, 5, 20, 0).sum() == 17 and f(71, 5, 20, 10).sum() == 18
    assert np.array_equal(f(71, 5, 20, 5), util.eval_stride_mask(71, 5, 20, 5))


def test_mpjpe_and_flip_protocol():
    rng = np.random.default_rng(0)
    pred = rng.normal(size=(5, 17, 3))
    gt = np.concatenate([rng.normal(size=(5, 17, 3)), np.ones((5, 17, 1))], -1)
    gt[0, 3, 3] = 0
    e = O.mpjpe(pred, gt, 6, normalize=False)
    assert e[0, 3] == -1 and np.all(e[:, 6] == 0)
    d = (pred - pred[:, 6:7]) - (gt[..., :3] - gt[:, 6:7, :3])
    assert np.allclose(e[1], np.sqrt((d[1] ** 2).sum(-1)))
    assert O.mpjpe(pred, gt, 6) == pytest.approx(np.where(gt[..., 3] > 0, np.sqrt((d ** 2).sum(-1)), 0).sum() / 84)
    # flip twice = identity on the input transform (eval.py:154-157)
    cfg, arch, hp, w = _setup("h36m_81", seed=6)
    order = cfg.AUGM_FLIP_KEYPOINT_ORDER
    assert sorted(order) == list(range(17)) and [order[i] for i in order] == list(range(17))
    x, m = util.synthetic_batch(cfg, 1, seed=6, mask_specs=[(4, 0)])
    seq, cen = O.eval_step_with_flip(hp, w, x, m, order, torch.float64)
    s1, c1 = O.test_step(hp, w, x, m, torch.float64)
    xf = np.concatenate([-x[..., :1], x[..., 1:]], -1)[:, :, order]
    s2, c2 = O.test_step(hp, w, xf, m, torch.float64)
    c2 = np.concatenate([-c2[..., :1], c2[..., 1:]], -1)[:, order]
    assert np.allclose(cen, (c1 + c2) / 2, atol=1e-12)


# ---- wiring pin against the REFERENCE's own model classes (tests/golden/make_model_wiring_golden.py) -------------------------------
_WIRING = os.path.join(util.ROOT, "tests", "golden", "model_wiring_expected.npz")


def _wiring_cases():
    g = np.load(_WIRING)
    return sorted({k.split("/")[0] for k in g.files})


@pytest.mark.parametrize("case", _wiring_cases())
def test_oracle_wiring_matches_the_reference_classes(case):
    """The reference's OWN MLP / MHA / TransformerBlock / StridedMLP / StridedTransformerBlock / UpliftUpsampleTransformer and constructor
    (common/net/vision_transformer.py:46-195, uplift_upsample_transformer.py:21-421, ..._constructor.py:14-50), executed from their ASTs in
    the build container with float64 numpy stand-ins for the Keras / TensorFlow ops they call, against the oracle in float64 on the same
    seeded weights and inputs: control flow, masks, positional encodings, head layout, residual trims, padding / pooling rules -- and the
    order of `model.weights` (Keras attribute tracking: what the by-name .h5 loader relies on, common/utils/weight_io.py:172-201,235)
    against weights.weight_spec.  Pins the WIRING; TensorFlow's float32 kernels stay unpinned (the stand-ins are numpy)."""
    import ast
    g = np.load(_WIRING)
    cfg = util.load_config(str(g[f"{case}/config"]).replace(".json", ""))
    for k, v in ast.literal_eval(str(g[f"{case}/overrides"])):
        setattr(cfg, k, v)
    arch = pkg.arch_from_config(cfg)
    w = pkg.init_weights(arch, seed=int(g[f"{case}/seed"]), perturb=0.1)
    # 1. the order of model.weights and the top-level layer names
    assert [n for n, _ in weight_spec(arch)] == [str(s) for s in g[f"{case}/weights_order"]]
    tops = [str(s) for s in g[f"{case}/top_level_layers"]]
    assert set(n.split("/")[0] for n, _ in weight_spec(arch)) <= set(tops)
    # 2. the outputs, float64 against float64
    x = g[f"{case}/x"]
    m = g[f"{case}/mask"] if f"{case}/mask" in g.files else None
    has_att = f"{case}/attention_0" in g.files
    res = O.forward(util.hp_from_arch(arch), w, x, m, torch.float64, return_attention=has_att)
    full, central = res[0], res[1]
    err = np.abs(central - g[f"{case}/central"]).max()
    assert (full is None) == (f"{case}/full" not in g.files)
    if full is not None:
        err = max(err, np.abs(full - g[f"{case}/full"]).max())
    print(f"{case}: oracle f64 vs the reference's classes {err:.3e}")
    assert err <= 1e-9
    if has_att:
        for i, a in enumerate(res[2]):
            assert np.abs(a - g[f"{case}/attention_{i}"]).max() <= 1e-6
