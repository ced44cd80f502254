"""HBM roofline of the fused AdamW update: 28 B per parameter."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from uplift_upsample_3dhpe_amd import optim  # noqa: E402

for n in (10404902, 10404902 * 8, 10404902 * 32):
    p = torch.randn(n, device="cuda") * 0.05
    g = torch.randn(n, device="cuda") * 1e-3
    opt = optim.AdamW(p, weight_decay=2e-6, learning_rate=2e-5, epsilon=1e-8)
    for _ in range(3):
        opt.apply_gradients(g)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    iters = 20
    for _ in range(iters):
        opt.apply_gradients(g)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    gbs = 28.0 * n / ms / 1e6
    print(json.dumps({"kernel": "adamw_kernel", "params": n, "ms": round(ms, 4), "GB/s": round(gbs, 1),
                      "frac_of_8TBps": round(gbs / 8000, 3), "bytes_per_param": 28}))
