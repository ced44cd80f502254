"""Soak test of the training step's stream choreography: forward + backward N times on the same inputs and parameters, the
gradient buffer compared bitwise with the first run's (a missing cross-stream dependency shows up as a rare mismatch).
   python tools/soak_train_determinism.py [iterations=300] [batch=64]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uplift_upsample_3dhpe_amd as pkg
from uplift_upsample_3dhpe_amd import harness, synthetic, _capi
from uplift_upsample_3dhpe_amd.trainer import Trainer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cfg = synthetic.load_config("h36m_351_pt"); cfg.BATCH_SIZE = B
arch = pkg.arch_from_config(cfg)
model = pkg.build_uplift_upsample_transformer(cfg, weights=pkg.init_weights(arch, seed=0, perturb=0.1), device="cuda:0")
tr = Trainer(model, cfg, seed=100)
_capi.check(tr._lib, tr._lib.uu3d_train_set_grad_callback(model._h, _capi.GRAD_READY_FN(0), None), model._h)   # repeated backward passes without the optimizer
tr._buckets.wait = lambda: None
rng = np.random.default_rng(3000)
N, J = arch.num_frames, arch.num_keypoints
x = torch.from_numpy(rng.uniform(-1, 1, size=(B, N, J, 2)).astype(np.float32)).cuda()
gt = torch.from_numpy(rng.normal(0, 0.3, size=(B, N, J, 3)).astype(np.float32)).cuda()
m = torch.from_numpy(harness.stride_masks_train(N, cfg.SEQUENCE_STRIDE, cfg.MASK_STRIDE, B, rng, cfg.STRIDE_MASK_RAND_SHIFT)).cuda()
u = torch.rand(tr.drop_path_size(B), device="cuda")
loss0, _, _ = tr.forward_backward(x, gt, m, drop_path_uniform=u)
torch.cuda.synchronize()
g0 = tr.grads.clone(); l0 = loss0.clone()
bad = 0
for i in range(n):
    loss, _, _ = tr.forward_backward(x, gt, m, drop_path_uniform=u)
    torch.cuda.synchronize()
    if not (torch.equal(tr.grads, g0) and torch.equal(loss, l0)):
        bad += 1
        if bad < 5: print(f"iteration {i}: mismatch, max abs gradient difference {float((tr.grads - g0).abs().max()):.3e}")
print(f"batch {B}: {n} forward + backward passes, {bad} mismatches; |g| max {float(g0.abs().max()):.3e}")
sys.exit(1 if bad else 0)
