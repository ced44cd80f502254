"""Window / stride-mask fixtures from the REFERENCE's own H36mSequenceGenerator.  Run in the build container:

    python tests/golden/make_windows_golden.py          # needs /root/reference -> tests/golden/windows_expected.npz

The class (common/dataset/uplifiting_dataset.py:213-428) is pure numpy, but its module imports tensorflow at the top, so
the module cannot be imported here.  This script therefore takes the class definition out of the reference file's AST at
run time and executes THAT code (nothing of it is stored in this repository) with numpy / math in scope, on seeded
synthetic videos, for the modes tests/test_windows_gpu.py uses; it stores the first windows of an epoch and float64
checksums over the whole epoch.  tests/test_windows_gpu.py checks data.SequenceGenerator + uu3d_gather_windows against it."""
import ast
import math
import os

import numpy as np

SRC = "/root/reference/common/dataset/uplifiting_dataset.py"
tree = ast.parse(open(SRC).read())
node = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "H36mSequenceGenerator")
ns = {"np": np, "math": math}
exec(compile(ast.Module(body=[node], type_ignores=[]), SRC, "exec"), ns)
Gen = ns["H36mSequenceGenerator"]

HERE = os.path.dirname(os.path.abspath(__file__))
FLIP = [5, 4, 3, 2, 1, 0, 6, 7, 8, 9, 10, 16, 15, 14, 13, 12, 11]
LENS = (3, 40, 97, 26, 7)
RATES = [50, 100, 50, 100, 50]
rng = np.random.default_rng(5)
p2 = [rng.uniform(-1, 1, size=(n, 17, 2)).astype(np.float32) for n in LENS]
p3 = [rng.normal(0, 0.4, size=(n, 17, 3)).astype(np.float32) for n in LENS]
cams = [rng.normal(size=11).astype(np.float32) for _ in LENS]
subjects, actions = [1, 5, 6, 7, 8], [0, 3, 3, 14, 2]

MODES = {
    "eval41": dict(seq_len=41, stride=2, padding_type="copy", mask_stride=4, stride_mask_align_global=True, flip_augment=False, shuffle=False),
    "train71": dict(seq_len=71, stride=5, padding_type="copy", mask_stride=[5, 10, 20], rand_shift_stride_mask=True, flip_augment=True, shuffle=True, subsample=3),
    "inbatch9": dict(seq_len=9, stride=1, padding_type="zeros", mask_stride=None, flip_augment=True, in_batch_augment=True, shuffle=True),
    "zeros27": dict(seq_len=27, stride=3, padding_type="zeros", mask_stride=[3, 9], stride_mask_align_global=True, flip_augment=False, shuffle=False, subsample=2),
}
KEEP = 24
out = {"lens": np.array(LENS), "rates": np.array(RATES), "subjects": np.array(subjects), "actions": np.array(actions)}
for v, (a, b) in enumerate(zip(p2, p3)):
    out[f"video2d_{v}"] = a
    out[f"video3d_{v}"] = b
for tag, mode in MODES.items():
    g = Gen(p3, p2, cams, subjects, actions, RATES, "train", flip_lr_indices=FLIP, seed=3, verbose=False, **mode)
    s3, s2, mk, sb, ac, ii, sm = [], [], [], [], [], [], []
    chk = np.zeros(4, np.float64)
    n = 0
    for seq3, seq2, mask, cam, subject, action, i, stride_mask in g.next_epoch_iterator():
        if n < KEEP:
            s3.append(seq3.copy()); s2.append(seq2.copy()); mk.append(mask.copy()); sm.append(stride_mask.copy())
        sb.append(subject); ac.append(action); ii.append(i)
        w = 1.0 + (n % 7)
        chk += w * np.array([seq3.astype(np.float64).sum(), seq2.astype(np.float64).sum(), mask.sum(), stride_mask.sum()])
        n += 1
    assert n == len(g)
    out[f"{tag}/seq3d"] = np.stack(s3); out[f"{tag}/seq2d"] = np.stack(s2); out[f"{tag}/mask"] = np.stack(mk)
    out[f"{tag}/stride_mask"] = np.stack(sm)
    out[f"{tag}/subject"] = np.array(sb); out[f"{tag}/action"] = np.array(ac); out[f"{tag}/index"] = np.array(ii)
    out[f"{tag}/checksum"] = chk
    out[f"{tag}/count"] = np.int64(n)
    print(tag, n, chk)
np.savez_compressed(os.path.join(HERE, "windows_expected.npz"), **out)
