"""Soak test: the same forward N times, every output compared bitwise with the first (a rare hazard glitch in a hand-scheduled
kernel shows up as a mismatch).   python tools/soak_determinism.py [iterations=1000] [config=h36m_351] [batch=128]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
pkg = importlib.import_module("uplift-upsample-3dhpe_amd".replace("-", "_")) if False else None
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import uplift_upsample_3dhpe_amd as pkg
from uplift_upsample_3dhpe_amd import synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
cfgname = sys.argv[2] if len(sys.argv) > 2 else "h36m_351"
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 128
cfg = synthetic.load_config(cfgname)
arch = pkg.arch_from_config(cfg)
model = pkg.build_uplift_upsample_transformer(cfg, weights=pkg.init_weights(arch, seed=0, perturb=0.1), device="cuda:0")
bad = 0
for mask_stride in (None, 10):
    x, m = synthetic.synthetic_batch(cfg, batch=batch, seed=11, mask_stride=mask_stride) if "mask_stride" in synthetic.synthetic_batch.__code__.co_varnames else synthetic.synthetic_batch(cfg, batch=batch, seed=11)
    xt = torch.from_numpy(x * m[:, :, None, None]).cuda(); mt = torch.from_numpy(m).cuda()
    full0, cen0 = model([xt, mt], training=False)
    full0, cen0 = full0.clone(), cen0.clone()
    for i in range(n):
        full, cen = model([xt, mt], training=False)
        if not (torch.equal(full, full0) and torch.equal(cen, cen0)):
            bad += 1
            if bad < 5: print(f"iteration {i}: mismatch, max abs {float((full - full0).abs().max()):.3e}")
    torch.cuda.synchronize()
    print(f"{cfgname} batch {batch} mask stride {mask_stride}: {n} forwards, {bad} mismatches")
sys.exit(1 if bad else 0)
