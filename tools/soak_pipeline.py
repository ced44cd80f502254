"""Soak test of the four-queue pipeline: N batches cycling through 8 inputs, every result compared bitwise with what model(...) returned
for that input (a slot reading another slot's memory, a replay ahead of its inputs, a launch shape of the throughput schedule computing
something else would show up as a mismatch).   python tools/soak_pipeline.py [batches=20000] [config=h36m_351] [batch=128]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uplift_upsample_3dhpe_amd as pkg
from uplift_upsample_3dhpe_amd import synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
cfgname = sys.argv[2] if len(sys.argv) > 2 else "h36m_351"
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 128
cfg = synthetic.load_config(cfgname)
arch = pkg.arch_from_config(cfg)
model = pkg.build_uplift_upsample_transformer(cfg, weights=pkg.init_weights(arch, seed=0, perturb=0.1), device="cuda:0")
inputs, want = [], []
for k in range(8):
    x, m = synthetic.synthetic_batch(cfg, batch=batch, seed=40 + k)
    xt = torch.from_numpy(x * m[:, :, None, None].astype(np.float32)).cuda(); mt = torch.from_numpy(m).cuda()
    inputs.append((xt, mt))
    want.append(tuple(t.clone() for t in model([xt, mt], training=False)))
pipe = model.pipeline(batch)
bad = i = 0
t0 = time.perf_counter()
for full, cen in pipe.run(inputs[j % 8] for j in range(n)):
    fw, cw = want[i % 8]
    if not (torch.equal(full, fw) and torch.equal(cen, cw)):
        bad += 1
        if bad < 5:
            print(f"batch {i}: mismatch, max abs {float((full - fw).abs().max()):.3e}")
    i += 1
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{cfgname} batch {batch}: {n} batches through {pipe.depth} slots, {bad} mismatches ({batch * n / dt / 1e3:.1f} k sequences/s with the per-batch comparison)")
