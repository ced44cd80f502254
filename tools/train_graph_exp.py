"""Experiment (round 1; re-run in round 5 on the three-stream step): replay uu3d_train_forward_backward from a hipGraph instead of ~400 eager launches.
   python tools/train_graph_exp.py [steps]"""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uplift_upsample_3dhpe_amd as pkg
from uplift_upsample_3dhpe_amd import harness
from uplift_upsample_3dhpe_amd.trainer import Trainer
from tests import util

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
cfg = util.load_config("h36m_351_pt"); B = 64; cfg.BATCH_SIZE = B
arch = pkg.arch_from_config(cfg)
model = pkg.build_uplift_upsample_transformer(cfg, weights=pkg.init_weights(arch, seed=0), device="cuda:0")
tr = Trainer(model, cfg, seed=100)
from uplift_upsample_3dhpe_amd import _capi
_capi.check(tr._lib, tr._lib.uu3d_train_set_grad_callback(model._h, _capi.GRAD_READY_FN(0), None), model._h)      # no bucket callback: it would only run at capture time
tr._buckets.wait = lambda: None
rng = np.random.default_rng(3000)
N, J = arch.num_frames, arch.num_keypoints
x = torch.from_numpy(rng.uniform(-1, 1, size=(B, N, J, 2)).astype(np.float32)).cuda()
gt = torch.from_numpy(rng.normal(0, 0.3, size=(B, N, J, 3)).astype(np.float32)).cuda()
m = torch.from_numpy(harness.stride_masks_train(N, cfg.SEQUENCE_STRIDE, cfg.MASK_STRIDE, B, rng, cfg.STRIDE_MASK_RAND_SHIFT)).cuda()
u = torch.rand(tr.drop_path_size(B), device="cuda")

def eager():
    tr.forward_backward(x, gt, m, drop_path_uniform=u); tr.apply_gradients()
for _ in range(10): eager()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps): eager()
torch.cuda.synchronize()
print(f"eager: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per step")
g_ref = tr.grads.clone()

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    tr.forward_backward(x, gt, m, drop_path_uniform=u)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    tr.forward_backward(x, gt, m, drop_path_uniform=u)
ustatic = u
def graphed():
    ustatic.uniform_()                       # fresh DropPath draws into the captured buffer (what a graphed Trainer would do)
    g.replay(); tr.apply_gradients()
for _ in range(10): graphed()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps): graphed()
torch.cuda.synchronize()
print(f"graph replay of forward_backward + eager optimizer: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per step")

g2 = tr.grads.clone()
print("gradients of a replay == eager (same draws):", end=" ")
ustatic.copy_(torch.rand(tr.drop_path_size(B), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)))
uu = ustatic.clone()
g.replay(); torch.cuda.synchronize(); a = tr.grads.clone()
tr.forward_backward(x, gt, m, drop_path_uniform=uu); torch.cuda.synchronize()
print(bool(torch.equal(a, tr.grads)))
t0 = time.perf_counter()
for _ in range(steps): g.replay()
torch.cuda.synchronize()
print(f"graph replay of forward_backward alone: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per step")
t0 = time.perf_counter()
for _ in range(steps): tr.forward_backward(x, gt, m, drop_path_uniform=u)
torch.cuda.synchronize()
print(f"eager forward_backward alone: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per step")
t0 = time.perf_counter()
for _ in range(steps): tr.apply_gradients()
torch.cuda.synchronize()
print(f"apply_gradients alone: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per step")
