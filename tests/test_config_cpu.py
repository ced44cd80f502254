"""Config schema, architecture derivation and weight inventory (host logic, no GPU)."""
import json
import os

import numpy as np
import pytest

import uplift_upsample_3dhpe_amd as pkg
from tests import util


@pytest.mark.parametrize("name,N,strided,params,mflop", [
    ("h36m_351", 71, (71, 23, 3, 1), 10404902, 1051.241344),
    ("h36m_351_pt", 71, (71, 23, 3, 1), 10404902, 1051.241344),
    ("amass_351", 71, (71, 23, 3, 1), 10404902, 1051.241344),
    ("h36m_81", 41, (41, 11, 3, 1), 10377254, 594.506368),
])
def test_shipped_configs(name, N, strided, params, mflop):
    cfg = util.load_config(name)
    a = pkg.arch_from_config(cfg)
    assert a.num_frames == N and a.num_keypoints == 17
    assert (a.d_spatial, a.d_temporal, a.num_heads) == (32, 384, 8)
    assert a.strided_lengths == strided
    assert a.has_strided_input and a.full_output
    assert a.first_strided_token_attention_layer == 1
    assert pkg.count_params(a) == params
    assert pkg.flops_per_sequence(a)["total"] / 1e6 == pytest.approx(mflop, rel=1e-9)


def test_defaults_and_override(tmp_path):
    c = pkg.UpliftUpsampleConfig()
    assert c.TEMPORAL_EMBED_DIM == 348 and c.SEQUENCE_LENGTH == 27 and c.ROOT_KEYTPOINT == 6
    assert c.MASK_STRIDE is None and c.PADDINGS is None
    p = tmp_path / "c.json"
    p.write_text(json.dumps({"SEQUENCE_LENGTH": 9, "UNKNOWN_KEY": [1, 2], "MASK_STRIDE": 1}))
    c = pkg.UpliftUpsampleConfig(str(p))
    assert c.SEQUENCE_LENGTH == 9 and c.UNKNOWN_KEY == [1, 2]     # unknown keys tolerated
    assert c.TEMPORAL_EMBED_DIM == 348                             # missing keys keep defaults
    t = tmp_path / "c.txt"
    t.write_text("# comment\nSEQUENCE_LENGTH 11\nSTRIDES [3, 3]\nARCH 'x'\n")
    c = pkg.UpliftUpsampleConfig(str(t))
    assert c.SEQUENCE_LENGTH == 11 and c.STRIDES == [3, 3] and c.ARCH == "x"


def test_copy_dump_roundtrip(tmp_path):
    c = util.load_config("h36m_81")
    d = c.copy()
    d.STRIDES.append(99)
    assert c.STRIDES == [4, 4, 3]                                  # deep copy
    out = tmp_path / "dump.json"
    c.dump(str(out))
    e = pkg.UpliftUpsampleConfig(str(out))
    assert e.to_dict() == c.to_dict()


def test_has_strided_input_rule():
    c = pkg.UpliftUpsampleConfig()
    c.SEQUENCE_LENGTH, c.STRIDES, c.TEMPORAL_EMBED_DIM = 27, [3, 3, 3], 384
    for ms, expect in [(None, False), (1, False), ([1, 2], False), (5, True), ([5, 10], True)]:
        c.MASK_STRIDE = ms
        assert pkg.arch_from_config(c).has_strided_input is expect   # constructor.py:16-21
    assert pkg.arch_from_config(c).paddings == ((1, 1),) * 3          # PADDINGS None
    assert pkg.arch_from_config(c).strided_lengths == (27, 9, 3, 1)


def test_bad_configs():
    c = util.load_config("h36m_351")
    c.STRIDES = [3, 3, 3]            # 71 -> 23 -> 7 -> 2: does not reduce to one token
    with pytest.raises((ValueError, AssertionError)):
        pkg.arch_from_config(c)
    c = util.load_config("h36m_351")
    c.TEMPORAL_EMBED_DIM = 380        # not divisible by 8 heads (vit.py:79)
    with pytest.raises(AssertionError):
        pkg.arch_from_config(c)


def test_weight_spec_order_and_init():
    a = pkg.arch_from_config(util.load_config("h36m_351"))
    spec = pkg.weight_spec(a)
    names = [n for n, _ in spec]
    assert names[:4] == ["keypoint_embedding/kernel", "keypoint_embedding/bias",
                         "spatial_pe/positional_encoding_weights", "temporal_pe/positional_encoding_weights"]
    assert names[-2:] == ["strided_temporal_fc/kernel", "strided_temporal_fc/bias"]
    blk = [n for n in names if n.startswith("strided_temporal_block_1/")]
    assert [n.split("/", 1)[1] for n in blk] == [
        "norm1/gamma", "norm1/beta", "attn/wq/kernel", "attn/wq/bias", "attn/wk/kernel", "attn/wk/bias",
        "attn/wv/kernel", "attn/wv/bias", "attn/projection/kernel", "attn/projection/bias",
        "norm2/gamma", "norm2/beta", "mlp/fc1/kernel", "mlp/fc1/bias", "mlp/strided_conv/kernel",
        "mlp/strided_conv/bias"]
    shapes = dict(spec)
    assert shapes["strided_temporal_block_1/mlp/strided_conv/kernel"] == (3, 768, 384)
    assert shapes["strided_temporal_block_1/mlp/fc1/kernel"] == (1, 384, 768)
    assert shapes["strided_temporal_pe_2/positional_encoding_weights"] == (23, 384)
    w = pkg.init_weights(a, seed=3)
    w2 = pkg.init_weights(a, seed=3)
    assert all(np.array_equal(w[k], w2[k]) for k in w)               # deterministic
    k = w["temporal_block_1/attn/wq/kernel"]
    assert abs(k).max() <= np.sqrt(6.0 / 768) + 1e-7                 # glorot_uniform limit
    assert np.all(w["temporal_block_1/norm1/gamma"] == 1) and np.all(w["temporal_fc/bias"] == 0)
    pe = w["temporal_pe/positional_encoding_weights"]
    assert abs(pe).max() <= 0.04 + 1e-7 and 0.015 < pe.std() < 0.02  # truncated normal, sigma .02


def test_config_matches_the_reference_classes(tmp_path):
    """UpliftUpsampleConfig against what the reference's own Config / UpliftUpsampleConfig (pure Python, run in the build
    container by tests/golden/make_config_golden.py) make of the class defaults, of every shipped config/*.json and of a
    text-mode file: the same set of public attributes with the same values."""
    exp = json.load(open(os.path.join(util.ROOT, "tests", "golden", "config_expected.json")))

    def public(c):
        return {k: getattr(c, k) for k in dir(c) if not k.startswith("_") and not callable(getattr(c, k))}

    def same(got, want, tag):
        assert sorted(got) == sorted(want), (tag, sorted(set(got) ^ set(want)))
        for k, v in want.items():
            g = got[k]
            g = json.loads(json.dumps(g, default=str))            # tuples -> lists etc., as the fixture was serialised
            assert g == v, (tag, k, g, v)

    same(public(pkg.UpliftUpsampleConfig()), exp["__defaults__"], "defaults")
    for name in ("amass_351.json", "h36m_351.json", "h36m_351_pt.json", "h36m_81.json"):
        same(public(pkg.UpliftUpsampleConfig(os.path.join(util.ROOT, "config", name))), exp[name], name)
    t = tmp_path / "c.txt"
    t.write_text("# comment\nSEQUENCE_LENGTH 11\nSTRIDES [3, 3]\nARCH 'x'\nMASK_STRIDE [2, 4]\n")
    with pytest.warns(UserWarning):                   # single quotes: the reference warns and reads them as double quotes
        c = pkg.UpliftUpsampleConfig(str(t))
    same(public(c), exp["__text_mode__"], "text mode")


@pytest.mark.parametrize("key", ["DROP_RATE", "ATTENTION_DROP_RATE"])
def test_every_training_option_is_implemented_and_bad_rates_are_rejected(key):
    """Round 4: the Dropout layers (u_u_t.py:201,324, vit.py:57-67,87-90,127-128,153-154) were the last schema-legal options that a
    training entry point refused; ``arch.training_unsupported`` (what Trainer / model(training=True) raise from) is now empty for any
    legal rate.  Rates Keras itself rejects (outside [0, 1)) still fail loudly instead of running a different model."""
    from uplift_upsample_3dhpe_amd.arch import training_unsupported
    cfg = util.load_config("h36m_351")
    assert training_unsupported(pkg.arch_from_config(cfg)) == []
    cfg.TOKEN_MASK_RATE = 0.2
    cfg.OUTPUT_BN = True
    setattr(cfg, key, 0.1)
    a = pkg.arch_from_config(cfg)
    assert training_unsupported(a) == [] and getattr(a, "drop_rate" if key == "DROP_RATE" else "attention_drop_rate") == pytest.approx(0.1)
    for bad_rate in (1.0, -0.1):
        setattr(cfg, key, bad_rate)
        bad = training_unsupported(pkg.arch_from_config(cfg))      # inference is unaffected: the arch still builds
        assert len(bad) == 1 and key in bad[0]


def test_dropout_oracle_masks():
    """oracle/dropout_oracle.py (the numpy twin of csrc/uu3d_dropout.h): deterministic in (seed, site, index), different per site and
    seed, the kept fraction follows the rate, kept elements carry 1 / (1 - rate) in float32, rate 0 means no layer."""
    import torch
    from oracle import uplift_oracle as O
    from oracle.dropout_oracle import drop_factor, drop_hash
    f = drop_factor((64, 71, 384), 0.1, 987654321012345, 101)
    assert f.dtype == np.float32 and set(np.unique(f)) == {np.float32(0.0), np.float32(1.0) / (np.float32(1.0) - np.float32(0.1))}
    assert abs((f > 0).mean() - 0.9) < 2e-3
    assert np.array_equal(f, drop_factor((64, 71, 384), 0.1, 987654321012345, 101))
    assert not np.array_equal(f, drop_factor((64, 71, 384), 0.1, 987654321012345, 102))
    assert not np.array_equal(f, drop_factor((64, 71, 384), 0.1, 987654321012346, 101))
    # the hash sees all 64 bits of an index and of the seed
    assert drop_hash(5, 1, np.uint64(3)) != drop_hash(5, 1, np.uint64(3 + 2 ** 32)) and drop_hash(5, 1, np.uint64(3)) != drop_hash(5 + 2 ** 32, 1, np.uint64(3))
    x = torch.ones(4, 8)
    assert O._dropout(x, None, 1) is x and O._dropout(x, dict(rate=0.0, attn_rate=0.0, seed=1), 1) is x
    y = O._dropout(x, dict(rate=0.5, attn_rate=0.25, seed=1), 7)
    ya = O._dropout(x, dict(rate=0.5, attn_rate=0.25, seed=1), 7, attention=True)
    assert set(np.unique(y.numpy())) <= {0.0, 2.0} and set(np.unique(ya.numpy())) <= {0.0, np.float32(1.0) / np.float32(0.75)}


def test_learnable_masked_token_rule():
    """LEARNABLE_MASKED_TOKEN alone creates nothing in the reference (the layer exists only with TOKEN_MASK_RATE > 0,
    u_u_t.py:219-220); together with a token mask rate the model owns one more weight, created in front of the strided-input
    token (Keras' default layer name: the reference passes none)."""
    from uplift_upsample_3dhpe_amd.weights import weight_spec
    cfg = util.load_config("h36m_351")
    cfg.LEARNABLE_MASKED_TOKEN = True
    a = pkg.arch_from_config(cfg)
    assert not a.learnable_masked_token
    names0 = [n for n, _ in weight_spec(a)]
    cfg.TOKEN_MASK_RATE = 0.1
    a = pkg.arch_from_config(cfg)
    assert a.learnable_masked_token
    spec = weight_spec(a)
    names = [n for n, _ in spec]
    i = names.index("learnable_masked_token_layer/learnable_masked_token")
    assert names[i + 1] == "strided_input_token_layer/learnable_masked_token" and spec[i][1] == (a.d_temporal,)
    assert names[:i] + names[i + 1:] == names0
