"""Writes the small Keras-style .h5 fixtures with a REAL HDF5 library (h5py 3.3.0 / libhdf5, the interpreter
/opt/conda/bin/python3.9 of the build container), following Keras 2.4 `save_weights_to_hdf5_group`:

    /opt/conda/bin/python3.9 tests/golden/make_h5_fixture.py

The main interpreter has no h5py; tests/test_h5_cpu.py reads these files with the package's own pure-Python
reader (uplift-upsample-3dhpe_amd/utils/hdf5_min.py).  The weights are a fixed pseudo-random sequence so the
expected values can be regenerated in the test without h5py.
"""
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# a toy "model": three top-level layers, nested weight names like the uplift model's blocks
SPEC = [("embed/kernel", (2, 8)), ("embed/bias", (8,)),
        ("block_1/norm1/gamma", (8,)), ("block_1/norm1/beta", (8,)), ("block_1/attn/wq/kernel", (8, 8)),
        ("block_1/mlp/strided_conv/kernel", (3, 16, 8)),
        ("token/learnable_masked_token", (8,))]


def values():
    rng = np.random.RandomState(1234)
    return {n: rng.uniform(-1, 1, size=s).astype(np.float32) for n, s in SPEC}


def save(group, vlen_attrs):
    w = values()
    layers = []
    for n, _ in SPEC:
        l = n.split("/")[0]
        if l not in layers:
            layers.append(l)
    if vlen_attrs:
        group.attrs.create("layer_names", [l for l in layers], dtype=h5py.string_dtype())
        group.attrs["backend"] = "tensorflow"
        group.attrs["keras_version"] = "2.4.0"
    else:
        group.attrs["layer_names"] = np.array([l.encode("utf8") for l in layers])       # Keras: fixed-length byte strings
        group.attrs["backend"] = b"tensorflow"
        group.attrs["keras_version"] = b"2.4.0"
    for l in layers:
        g = group.create_group(l)
        names = [n + ":0" for n, _ in SPEC if n.split("/")[0] == l]
        if vlen_attrs:
            g.attrs.create("weight_names", names, dtype=h5py.string_dtype())
        else:
            g.attrs["weight_names"] = np.array([n.encode("utf8") for n in names])
        for n in names:
            val = w[n[:-2]]
            d = g.create_dataset(n, val.shape, dtype=val.dtype)
            d[...] = val


with h5py.File(os.path.join(HERE, "keras_like_weights.h5"), "w") as f:
    save(f, vlen_attrs=False)
with h5py.File(os.path.join(HERE, "keras_like_full_model_vlen.h5"), "w") as f:      # model.save(): weights under /model_weights
    f.attrs["model_config"] = "{}"
    save(f.create_group("model_weights"), vlen_attrs=True)
    f.create_group("optimizer_weights")
print("written")
