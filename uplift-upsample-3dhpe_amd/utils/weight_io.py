"""Keras ``.h5`` weight files for the uplift model: what ``model.save_weights(path.h5)`` writes (train.py:706,719)
and what the reference's by-name loader reads (``common/utils/weight_io.py:76-263``).

File layout (Keras 2.4 ``save_weights_to_hdf5_group``): root attributes ``layer_names`` (array of byte strings),
``backend``, ``keras_version``; one group per top-level layer with attribute ``weight_names`` and one dataset per
weight at ``<layer>/<weight name>``.  A full-model file keeps the same under ``/model_weights``
(``weight_io.py:121-122``).

Loading follows the reference's loader: layers are matched by NAME, the weights inside a layer are assigned
by POSITION (``weight_io.py:172-201,235``), a layer whose count or a weight whose shape disagrees raises
``ValueError`` unless ``skip_mismatch``; layers of the file that the model does not have are reported, not fatal.
The top-level layer names and the order inside each layer are those of ``weights.weight_spec`` (SURVEY section 8(f)-1;
the order inside a block is the Keras attribute-tracking order and is unverified against a real checkpoint --
no ``.h5`` exists offline).
"""
from collections import OrderedDict

import numpy as np

from .hdf5_min import HDF5Error, read_hdf5, write_hdf5

KERAS_VERSION = b"2.4.0"          # tensorflow==2.4.3 (reference requirements.txt:3)


def _layers_of(spec):
    layers = OrderedDict()
    for name, shape in spec:
        layers.setdefault(name.split("/", 1)[0], []).append((name, tuple(shape)))
    return layers


def save_keras_h5(path, weights, spec):
    """Write ``weights`` (dict name -> array, names / shapes of ``spec``) as a Keras weight file."""
    layers = _layers_of(spec)
    root = {"attrs": OrderedDict(), "groups": OrderedDict()}
    root["attrs"]["layer_names"] = np.array([n.encode("utf8") for n in layers], dtype="S")
    root["attrs"]["backend"] = b"tensorflow"
    root["attrs"]["keras_version"] = KERAS_VERSION
    for lname, entries in layers.items():
        g = {"attrs": OrderedDict(), "groups": OrderedDict(), "datasets": OrderedDict()}
        wnames = []
        for name, shape in entries:
            a = np.asarray(weights[name], dtype=np.float32)
            if tuple(a.shape) != shape:
                raise ValueError(f"weight {name}: shape {a.shape} != {shape}")
            wname = name + ":0"                      # Keras variable names end in ':0'
            wnames.append(wname.encode("utf8"))
            node = g
            parts = wname.split("/")
            for part in parts[:-1]:
                node = node["groups"].setdefault(part, {"attrs": OrderedDict(), "groups": OrderedDict(), "datasets": OrderedDict()})
            node["datasets"][parts[-1]] = (a, None)
        g["attrs"]["weight_names"] = np.array(wnames, dtype="S")
        root["groups"][lname] = g
    write_hdf5(path, root)


class KerasWeightLoadingCallback(object):
    """Same hook as the reference's class of this name (weight_io.py:49-73): may transform or drop (return None as
    the value) a weight read from the file before it is assigned."""

    def __init__(self, verbose=True):
        self.verbose = verbose

    def __call__(self, target_weight, weight_name, weight_value):
        return False, weight_value


class TargetWeight(object):
    """What a callback sees as ``target_weight``: the name and shape of the model-side tensor."""

    def __init__(self, name, shape):
        self.name, self.shape = name, tuple(shape)


def load_weights_with_callback(model, filepath, skip_mismatch=False, callbacks=(), verbose=True):
    """Reference entry point (weight_io.py:76-122) for the HIP model object: read ``filepath`` (.h5), let the callbacks
    adjust each weight, assign by layer name / position, commit to the device.  Returns the loader's report."""
    if not str(filepath).endswith((".h5", ".hdf5", ".keras")):
        raise ValueError("load_weights_with_callback only reads HDF5 weight files")
    spec = list(zip(model.weight_names, [tuple(s) for _, s in model._spec]))
    values, report = load_keras_h5(filepath, spec, skip_mismatch=skip_mismatch, verbose=verbose, callbacks=callbacks)
    current = model.get_weights_dict()
    current.update(values)
    model.set_weights_dict(current)
    return report


def load_keras_h5(path, spec, skip_mismatch=False, verbose=False, callbacks=()):
    """-> (dict name -> float32 array for every weight that was assigned, report dict).

    ``report``: ``unconsumed_layers`` (in the file, not in the model), ``unassigned_layers`` (in the model, not in the
    file), ``skipped`` (mismatching layers / weights when ``skip_mismatch``).
    """
    root = read_hdf5(path)
    if "layer_names" not in root.attrs and "model_weights" in root:
        root = root["model_weights"]
    if "layer_names" not in root.attrs:
        raise HDF5Error("no 'layer_names' attribute: not a Keras weight file")
    dec = lambda x: x.decode("utf8") if isinstance(x, (bytes, np.bytes_)) else str(x)
    layer_names = [dec(x) for x in np.asarray(root.attrs["layer_names"]).reshape(-1)]
    layers = _layers_of(spec)
    out, skipped = OrderedDict(), []
    consumed = {n: False for n in layer_names}
    for k, lname in enumerate(layer_names):
        if lname not in layers:
            continue
        consumed[lname] = True
        g = root[lname]
        wnames = [dec(x) for x in np.asarray(g.attrs.get("weight_names", np.array([], dtype="S"))).reshape(-1)]
        values = [np.asarray(g[w].value) for w in wnames]
        target = layers[lname]
        if len(values) != len(target):
            if skip_mismatch:
                skipped.append(f"{lname}: {len(target)} weights expected, file has {len(values)}")
                continue
            raise ValueError(f'Layer #{k} (named "{lname}") expects {len(target)} weight(s), but the saved weights have '
                             f"{len(values)} element(s).")
        for (name, shape), wname, v in zip(target, wnames, values):
            adjusted = False
            for cb in callbacks:
                did, nv = cb(target_weight=TargetWeight(name, shape), weight_name=wname, weight_value=v)
                if adjusted and did:
                    raise AssertionError("Two (or more) callbacks tried to transform the same weights. This is not allowed")
                adjusted = adjusted or did
                if did:
                    v = nv
            if v is None:            # a callback removed the weight
                continue
            if tuple(v.shape) != shape:
                if skip_mismatch:
                    skipped.append(f"{lname}: {name} has shape {shape}, file weight {wname} has {tuple(v.shape)}")
                    continue
                raise ValueError(f'Layer #{k} (named "{lname}"), weight {name} has shape {shape}, but the saved weight '
                                 f"has shape {tuple(v.shape)}.")
            out[name] = np.ascontiguousarray(v, dtype=np.float32)
    report = {"unconsumed_layers": [n for n, c in consumed.items() if not c],
              "unassigned_layers": [n for n in layers if n not in layer_names],
              "skipped": skipped}
    if verbose:
        for key, title in (("unconsumed_layers", "The following layers were not consumed from .h5 file:"),
                           ("unassigned_layers", "The following layers were not assigned any weights:"),
                           ("skipped", "Skipped because of a mismatch:")):
            if report[key]:
                print(title)
                for n in report[key]:
                    print("- " + n)
    return out, report
