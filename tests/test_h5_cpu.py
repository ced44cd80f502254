"""Keras .h5 weight I/O: the pure-Python HDF5 subset reader / writer against files written
by a real HDF5 library, the reference loader's by-name / by-position semantics, and the model-sized round trip."""
import os
import subprocess

import numpy as np
import pytest

import uplift_upsample_3dhpe_amd as pkg
from uplift_upsample_3dhpe_amd.utils import hdf5_min, weight_io
from tests import util

GOLD = os.path.join(util.ROOT, "tests", "golden")
SPEC = [("embed/kernel", (2, 8)), ("embed/bias", (8,)),
        ("block_1/norm1/gamma", (8,)), ("block_1/norm1/beta", (8,)), ("block_1/attn/wq/kernel", (8, 8)),
        ("block_1/mlp/strided_conv/kernel", (3, 16, 8)),
        ("token/learnable_masked_token", (8,))]
H5PY = "/opt/conda/bin/python3.9"


def _values():
    rng = np.random.RandomState(1234)          # same stream as tests/golden/make_h5_fixture.py
    return {n: rng.uniform(-1, 1, size=s).astype(np.float32) for n, s in SPEC}


@pytest.mark.parametrize("fname", ["keras_like_weights.h5", "keras_like_full_model_vlen.h5"])
def test_reads_files_written_by_libhdf5(fname):
    got, rep = weight_io.load_keras_h5(os.path.join(GOLD, fname), SPEC)
    exp = _values()
    assert list(got) == [n for n, _ in SPEC]
    assert all(np.array_equal(got[n], exp[n]) for n in exp)
    assert rep == {"unconsumed_layers": [], "unassigned_layers": [], "skipped": []}
    root = hdf5_min.read_hdf5(os.path.join(GOLD, fname))
    node = root["model_weights"] if "model_weights" in root else root
    assert node.attrs["backend"] in (b"tensorflow",) and node.attrs["keras_version"] == b"2.4.0"
    assert node["block_1"]["block_1/mlp/strided_conv/kernel:0"].value.shape == (3, 16, 8)


def test_loader_semantics_follow_the_reference(tmp_path):
    """weight_io.py:172-235: match layers by name, weights by position; mismatches raise unless skip_mismatch."""
    p = os.path.join(GOLD, "keras_like_weights.h5")
    # a model with an extra layer and without 'token': reported, not fatal
    spec2 = [s for s in SPEC if not s[0].startswith("token")] + [("head/kernel", (8, 3))]
    got, rep = weight_io.load_keras_h5(p, spec2)
    assert "head/kernel" not in got and rep["unassigned_layers"] == ["head"] and rep["unconsumed_layers"] == ["token"]
    # renamed weights inside a layer still load: position decides
    spec3 = [(n.replace("norm1", "layer_normalization"), s) for n, s in SPEC]
    got3, _ = weight_io.load_keras_h5(p, spec3)
    assert np.array_equal(got3["block_1/layer_normalization/gamma"], _values()["block_1/norm1/gamma"])
    # wrong count / wrong shape
    with pytest.raises(ValueError, match="expects 3 weight"):
        weight_io.load_keras_h5(p, [s for s in SPEC if s[0] != "block_1/norm1/beta"])
    bad = [(n, (8, 9) if n == "block_1/attn/wq/kernel" else s) for n, s in SPEC]
    with pytest.raises(ValueError, match="saved weight has shape"):
        weight_io.load_keras_h5(p, bad)
    got4, rep4 = weight_io.load_keras_h5(p, bad, skip_mismatch=True)
    assert "block_1/attn/wq/kernel" not in got4 and len(rep4["skipped"]) == 1 and "embed/kernel" in got4
    with pytest.raises(hdf5_min.HDF5Error):
        f = tmp_path / "not.h5"; f.write_bytes(b"hello world, definitely not hdf5"); weight_io.load_keras_h5(str(f), SPEC)


@pytest.mark.parametrize("cfgname", ["h36m_81", "h36m_351", "h36m_81+bn", "h36m_351+mtoken"])
def test_model_sized_roundtrip_and_h5py_reads_our_files(cfgname, tmp_path):
    cfg = util.load_config(cfgname.split("+")[0])
    if cfgname.endswith("+bn"):                        # OUTPUT_BN: two BatchNormalization layers, four weights each (gamma, beta, moving mean / variance)
        cfg.OUTPUT_BN = True
    if cfgname.endswith("+mtoken"):                    # TOKEN_MASK_RATE > 0 with LEARNABLE_MASKED_TOKEN: one more layer with one weight
        cfg.TOKEN_MASK_RATE, cfg.LEARNABLE_MASKED_TOKEN = 0.1, True
    arch = pkg.arch_from_config(cfg)
    spec = pkg.weight_spec(arch)
    assert ("learnable_masked_token_layer/learnable_masked_token" in dict(spec)) == cfgname.endswith("+mtoken")
    w = pkg.init_weights(arch, seed=5, perturb=0.1)
    path = str(tmp_path / "w.h5")
    weight_io.save_keras_h5(path, w, spec)
    back, rep = weight_io.load_keras_h5(path, spec)
    assert all(np.array_equal(back[n], w[n]) for n, _ in spec) and not rep["unassigned_layers"]
    layers = []
    for n, _ in spec:
        if n.split("/")[0] not in layers:
            layers.append(n.split("/")[0])
    assert [x.decode() for x in hdf5_min.read_hdf5(path).attrs["layer_names"]] == layers      # SURVEY 8(f)-1 names
    if not os.path.exists(H5PY):
        pytest.skip("no interpreter with h5py on this machine: libhdf5 cross-check skipped")
    code = ("import h5py, numpy as np, sys\n"
            "f = h5py.File(sys.argv[1], 'r'); n = 0; s = 0.0\n"
            "for l in f.attrs['layer_names']:\n"
            "    g = f[l.decode()]\n"
            "    for wn in g.attrs['weight_names']:\n"
            "        a = np.asarray(g[wn.decode()]); n += a.size; s += float(np.abs(a.astype(np.float64)).sum())\n"
            "print(n, repr(s))\n")
    out = subprocess.run([H5PY, "-c", code, path], capture_output=True, text=True, check=True).stdout.split()
    assert int(out[0]) == sum(int(np.prod(s)) for _, s in spec)
    assert abs(float(out[1]) - sum(float(np.abs(w[n].astype(np.float64)).sum()) for n, _ in spec)) < 1e-6


def test_callbacks_can_adjust_or_drop_weights():
    p = os.path.join(GOLD, "keras_like_weights.h5")

    class Scale(weight_io.KerasWeightLoadingCallback):
        def __call__(self, target_weight, weight_name, weight_value):
            if weight_name == "embed/bias:0":
                assert target_weight.name == "embed/bias" and target_weight.shape == (8,)
                return True, weight_value * 2.0
            if weight_name.startswith("token/"):
                return True, None
            return False, weight_value

    got, _ = weight_io.load_keras_h5(p, SPEC, callbacks=[Scale()])
    assert np.array_equal(got["embed/bias"], _values()["embed/bias"] * 2.0) and "token/learnable_masked_token" not in got
    with pytest.raises(AssertionError):
        weight_io.load_keras_h5(p, SPEC, callbacks=[Scale(), Scale()])
