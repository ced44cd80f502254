"""Round 5: the temporal chain (throughput schedule) against the round-4 launch chain (latency schedule) and the oracle, and what it does
to the pipelined step.   gpurun -- 'python tools/tchain_check.py'"""
import os, sys, time
os.environ["UU3D_TCHAIN"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uplift_upsample_3dhpe_amd as pkg
from uplift_upsample_3dhpe_amd.synthetic import load_config, synthetic_batch

def fwd(model, arch, x, m, schedule):
    xt, mt = torch.as_tensor(x).cuda(), torch.as_tensor(m).cuda()
    B = xt.shape[0]
    full = torch.empty((B, arch.num_frames, arch.num_keypoints, 3), dtype=torch.float32, device="cuda")
    cen = torch.empty((B, arch.num_keypoints, 3), dtype=torch.float32, device="cuda")
    model._forward(xt, model._mask_u8(mt), full, cen, 0, torch.cuda.current_stream(), schedule=schedule)
    torch.cuda.synchronize()
    return full.cpu().numpy(), cen.cpu().numpy()

for cfgname, batches in (("h36m_351", (128, 17, 40)),):
    cfg = load_config(cfgname)
    arch = pkg.arch_from_config(cfg)
    w = pkg.init_weights(arch, seed=9, perturb=0.1)
    model = pkg.build_uplift_upsample_transformer(cfg, weights=w)
    for mask_stride in (None, 5):
        for B in batches:
            x, m = synthetic_batch(cfg, B, seed=9) if mask_stride is None else synthetic_batch(cfg, B, seed=9, mask_specs=[(5, 0)])
            x = x * m[:, :, None, None]
            f0, c0 = fwd(model, arch, x, m, 0)
            f1, c1 = fwd(model, arch, x, m, 1)
            f2, c2 = fwd(model, arch, x, m, 1)
            print(f"{cfgname} batch {B} mask_stride {mask_stride}: |thr - lat| full {np.abs(f1 - f0).max():.3e} central {np.abs(c1 - c0).max():.3e}; "
                  f"repeat identical {np.array_equal(f1, f2) and np.array_equal(c1, c2)}; scale {np.abs(f0).max():.2f}; nan {np.isnan(f1).sum()}", flush=True)
    model.set_profiling(True)
    x, m = synthetic_batch(cfg, 128, seed=9); x = x * m[:, :, None, None]
    for sched in (1, 0):
        fwd(model, arch, x, m, sched); fwd(model, arch, x, m, sched)
        prof = model.read_profile()
        tot = sum(r["ms"] for r in prof)
        print(f"schedule {sched}: {len(prof)} launches, {tot * 1e3:.1f} us of kernels")
        if sched == 1:
            for r in prof: print(f"    {r['name']:24s} {r['kernel']:28s} {r['ms'] * 1e3:8.1f} us")
    model.set_profiling(False)
