"""Per-launch HIP-event times of ONE forward under the throughput schedule (round 5), at a batch that fills the chip by itself: where the
CU-time goes once scheduling effects are out of the way.   python tools/kernel_share_exp.py [batch] [schedule]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import uplift_upsample_3dhpe_amd as pkg
from uplift_upsample_3dhpe_amd import synthetic as util
from tests import util as tutil

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sched = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = util.load_config("h36m_351"); arch = pkg.arch_from_config(cfg)
model = pkg.build_uplift_upsample_transformer(cfg, weights=pkg.init_weights(arch, seed=0))
x_np, m_np = util.synthetic_batch(cfg, B, seed=1000, mask_specs=[(5, 0)])
x = torch.from_numpy(x_np * m_np[:, :, None, None].astype(np.float32)).cuda(); m = torch.from_numpy(m_np).cuda()
for _ in range(3):
    tutil.direct_forward(model, x, m, sched)
model.set_profiling(True)
agg, order = {}, []
reps = 5
for _ in range(reps):
    tutil.direct_forward(model, x, m, sched)
    for e in model.read_profile():
        nm = e["name"]
        key = ("t." + nm.split(".", 1)[1]) if (nm[0] == "t" and "." in nm) else nm
        if key not in agg:
            agg[key] = dict(ms=0.0, n=0, kernel=e["kernel"], flops=0.0); order.append(key)
        agg[key]["ms"] += e["ms"]; agg[key]["n"] += 1; agg[key]["flops"] += e["flops"]
model.set_profiling(False)
tot = sum(a["ms"] for a in agg.values()) / reps
print(f"batch {B}, schedule {sched}: sum of launch times {tot:.4f} ms per forward = {tot / B * 128:.4f} ms per 128 sequences ({B / tot:.1f} k sequences/s if nothing overlapped)")
for k in sorted(agg, key=lambda k: -agg[k]["ms"]):
    a = agg[k]
    print(f"  {k:18s} {a['kernel']:34s} x{a['n'] // reps:2d}  {a['ms'] / reps:8.4f} ms  {100 * a['ms'] / reps / tot:5.1f} %   {a['flops'] / max(a['ms'], 1e-9) / 1e9:7.1f} TFLOP/s")
