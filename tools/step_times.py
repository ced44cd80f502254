"""When does each of the K forwards of a short pipelined run complete?  (the driver's shape: bench.py --steps 20 --warmup 5)
Completion time of every step since the loop's start (ms), from timing events on the slots' streams -- shows what the fill of the empty
pipeline and its drain cost against the steady state.   python tools/step_times.py [steps] [depth]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import uplift_upsample_3dhpe_amd as pkg
from uplift_upsample_3dhpe_amd import synthetic as util

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
depth = int(sys.argv[2]) if len(sys.argv) > 2 else None
cfg = util.load_config("h36m_351"); arch = pkg.arch_from_config(cfg)
model = pkg.build_uplift_upsample_transformer(cfg, weights=pkg.init_weights(arch, seed=0))
B = 128
x_np, m_np = util.synthetic_batch(cfg, B, seed=1000, mask_specs=[(5, 0)])
x = torch.from_numpy(x_np * m_np[:, :, None, None].astype(np.float32)).cuda(); m = torch.from_numpy(m_np).cuda()
pipe = model.pipeline(B, depth=depth)
pipe.preload(x, m)
for rep in range(3):
    for _ in range(5):
        pipe.result(pipe.launch(wait_caller=False))
    torch.cuda.synchronize()
    start = torch.cuda.Event(enable_timing=True); start.record()
    torch.cuda.synchronize()
    evs, tickets = [], []
    import time
    t0 = time.perf_counter()
    for k in range(steps):
        if k >= pipe.depth:
            pipe.result(tickets[k - pipe.depth])
        tickets.append(pipe.launch(wait_caller=False))
        e = torch.cuda.Event(enable_timing=True); e.record(pipe._slots[tickets[-1] % pipe.depth].stream); evs.append(e)
    for t in tickets[max(0, steps - pipe.depth):]:
        pipe.result(t)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    ts = [start.elapsed_time(e) for e in evs]
    off = ts[0] - 0.0
    print(f"run {rep}: wall {wall:.3f} ms for {steps} steps ({wall / steps:.4f} ms per step), depth {pipe.depth}")
    ts = sorted(ts)
    d = np.diff([0.0] + ts)
    show = slice(0, None) if steps <= 40 else slice(steps - 32, None)
    print("   completion since start (ms): " + " ".join(f"{t:.2f}" for t in ts[show]))
    print("   gaps (ms):                   " + " ".join(f"{v:.2f}" for v in d[show]))
    if steps > 40:
        print(f"   steps {steps - 32}..{steps - 1}: {(ts[-1] - ts[steps - 33]) / 32:.4f} ms per step; largest gap {max(d[show]):.2f} ms")
