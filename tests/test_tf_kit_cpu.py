"""tools/tf_crosscheck: the kit builder runs without GPU / TensorFlow and writes files the repository's own HDF5 reader and the
reference-side script's expectations agree with (names, shapes, arrays); the TF-side script at least parses."""
import ast
import os
import subprocess
import sys

import numpy as np

from tests import util


def test_kit_builds_and_is_self_consistent(tmp_path):
    out = str(tmp_path / "kit")
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "tools", "tf_crosscheck", "make_kit.py"), "--out", out,
                        "--configs", "h36m_81", "--batch", "2"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    for f in ("h36m_81.h5", "h36m_81_io.npz", "config/h36m_81.json", "check_with_tf.py", "README.md"):
        assert os.path.exists(os.path.join(out, f)), f
    z = np.load(os.path.join(out, "h36m_81_io.npz"))
    assert z["full_f32"].shape == (2, 41, 17, 3) and z["central_f32"].shape == (2, 17, 3)
    assert np.abs(z["full_f32"] - z["full_f64"]).max() < 1e-4
    assert any(k.startswith("grad/") for k in z.files) and np.isfinite(z["train_loss"])
    import uplift_upsample_3dhpe_amd as pkg
    from uplift_upsample_3dhpe_amd.utils import hdf5_min
    root = hdf5_min.read_hdf5(os.path.join(out, "h36m_81.h5"))
    names = [n.decode() if isinstance(n, bytes) else str(n) for n in root.attrs["layer_names"]]
    arch = pkg.arch_from_config(util.load_config("h36m_81"))
    tops = []
    for n, _ in pkg.weight_spec(arch):
        t = n.split("/", 1)[0]
        if t not in tops:
            tops.append(t)
    assert names == tops
    ast.parse(open(os.path.join(util.ROOT, "tools", "tf_crosscheck", "check_with_tf.py")).read())
