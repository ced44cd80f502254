"""AMASS window fixtures from the REFERENCE's own AMASSSequenceGenerator (uplifiting_dataset.py:431-661, numpy only) fed
with the reference's own AMASSDataset on tests/golden/amass_tiny.  The generator class is taken out of the reference file's
AST and executed at run time (its module imports tensorflow; nothing of it is stored here), see make_windows_golden.py.
    python tests/golden/make_amass_windows_golden.py      -> tests/golden/amass_windows_expected.npz"""
import ast
import math
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference")
from common.dataset.amass_dataset import AMASSDataset      # noqa: E402

SRC = "/root/reference/common/dataset/uplifiting_dataset.py"
tree = ast.parse(open(SRC).read())
node = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "AMASSSequenceGenerator")
ns = {"np": np, "math": math, "AMASSDataset": AMASSDataset}
exec(compile(ast.Module(body=[node], type_ignores=[]), SRC, "exec"), ns)
Gen = ns["AMASSSequenceGenerator"]

HERE = os.path.dirname(os.path.abspath(__file__))
FLIP = [5, 4, 3, 2, 1, 0, 6, 7, 8, 9, 10, 16, 15, 14, 13, 12, 11]
amass = AMASSDataset(os.path.join(HERE, "amass_tiny"), os.path.join(HERE, "h36m_tiny_3d.npz"), "train")
MODES = {
    "train9": dict(seq_len=9, stride=2, padding_type="copy", mask_stride=[2, 4, 8], rand_shift_stride_mask=True, flip_augment=True, shuffle=True),
    "eval27": dict(seq_len=27, stride=1, padding_type="zeros", mask_stride=5, stride_mask_align_global=True, flip_augment=False, shuffle=False),
    "inbatch5": dict(seq_len=5, stride=1, padding_type="copy", mask_stride=None, flip_augment=True, in_batch_augment=True, shuffle=True, subsample=2),
}
out = {}
for tag, mode in MODES.items():
    g = Gen(amass, flip_lr_indices=FLIP, seed=4, verbose=False, **mode)
    s3, cm, mk, sm, ii = [], [], [], [], []
    for seq3, cam, mask, subject, action, i, stride_mask in g.next_epoch_iterator():
        s3.append(seq3.copy()); cm.append(cam.copy()); mk.append(mask.copy()); sm.append(stride_mask.copy()); ii.append(i)
    assert len(s3) == len(g)
    out[f"{tag}/seq3d"] = np.stack(s3); out[f"{tag}/cams"] = np.stack(cm); out[f"{tag}/mask"] = np.stack(mk)
    out[f"{tag}/stride_mask"] = np.stack(sm); out[f"{tag}/index"] = np.array(ii)
    print(tag, len(s3))
np.savez_compressed(os.path.join(HERE, "amass_windows_expected.npz"), **out)
