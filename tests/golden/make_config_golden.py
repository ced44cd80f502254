"""Dumps what the REFERENCE's own config classes (common/utils/config.py, common/net/uplift_upsample_transformer_config.py --
no TensorFlow) make of every shipped config/*.json, of the class defaults, and of a text-mode file:
    python tests/golden/make_config_golden.py          # needs /root/reference -> tests/golden/config_expected.json
Checked by tests/test_config_cpu.py against this repository's UpliftUpsampleConfig."""
import glob
import json
import os
import sys
import tempfile

sys.path.insert(0, "/root/reference")
from common.net.uplift_upsample_transformer_config import UpliftUpsampleConfig      # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def public(c):
    out = {}
    for k in dir(c):
        if k.startswith("_") or callable(getattr(c, k)):
            continue
        out[k] = getattr(c, k)
    return out


res = {"__defaults__": public(UpliftUpsampleConfig())}
for path in sorted(glob.glob(os.path.join("/root/reference", "config", "*.json"))):
    res[os.path.basename(path)] = public(UpliftUpsampleConfig(config_file=path))
with tempfile.TemporaryDirectory() as d:
    t = os.path.join(d, "c.txt")
    open(t, "w").write("# comment\nSEQUENCE_LENGTH 11\nSTRIDES [3, 3]\nARCH 'x'\nMASK_STRIDE [2, 4]\n")
    try:
        res["__text_mode__"] = public(UpliftUpsampleConfig(config_file=t))
    except Exception as e:      # recorded as is: the test then expects the same failure class
        res["__text_mode__"] = {"__error__": type(e).__name__}
json.dump(res, open(os.path.join(HERE, "config_expected.json"), "w"), indent=1, sort_keys=True, default=str)
print({k: len(v) for k, v in res.items()})
