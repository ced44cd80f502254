"""Where the fixed ~1.6 ms of a short timed region goes (round 5): per-forward start / end times (HIP events on the slots' streams) and the host's
enqueue times for K steps out of an empty pipeline.   python tools/fill_drain_exp.py [depth] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import uplift_upsample_3dhpe_amd as pkg
from uplift_upsample_3dhpe_amd import synthetic as util

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
stagger_us = float(os.environ.get("STAGGER_US", "0"))
cfg = util.load_config("h36m_351"); arch = pkg.arch_from_config(cfg)
model = pkg.build_uplift_upsample_transformer(cfg, weights=pkg.init_weights(arch, seed=0))
x_np, m_np = util.synthetic_batch(cfg, 128, seed=1000, mask_specs=[(5, 0)])
x = torch.from_numpy(x_np * m_np[:, :, None, None].astype(np.float32)).cuda(); m = torch.from_numpy(m_np).cuda()
pipe = model.pipeline(128, depth=depth)
pipe.preload(x, m)
slots = pipe._slots
for rep in range(3):
    for _ in range(2 * depth):                                  # warm-up
        pipe.result(pipe.launch())
    torch.cuda.synchronize()
    st = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
    en = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
    e0 = torch.cuda.Event(enable_timing=True)
    host = []
    t0 = time.perf_counter()
    e0.record()
    for k in range(K):
        s = slots[k % depth]
        if stagger_us and k < depth and k > 0:
            t_next = t0 + k * stagger_us * 1e-6
            while time.perf_counter() < t_next:
                pass
        with torch.cuda.stream(s.stream):
            st[k].record(); s.graph.replay(); en[k].record()
        host.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    S = [e0.elapsed_time(e) for e in st]; E = [e0.elapsed_time(e) for e in en]
    print(f"depth {depth}, {K} steps: wall {wall:.3f} ms = {128 * K / wall:.1f} k/s; last end {max(E):.3f} ms; host enqueue of step k done at (ms): " + " ".join(f"{h:.2f}" for h in host))
    print("   start: " + " ".join(f"{v:.2f}" for v in S))
    print("   end:   " + " ".join(f"{v:.2f}" for v in E))
    print("   dur:   " + " ".join(f"{e - s:.2f}" for s, e in zip(S, E)))
    ends = sorted(E)
    print("   completion intervals: " + " ".join(f"{b - a:.2f}" for a, b in zip([0.0] + ends[:-1], ends)))

# ---- the same 20 steps through bench.py's loop, piece by piece
import bench
from uplift_upsample_3dhpe_amd.harness import per_joint_error
J = arch.num_keypoints
gt = torch.cat([torch.randn(128, J, 3, device="cuda") * 0.3, torch.ones(128, J, 1, device="cuda")], -1)
pipe.close()


class NoGather:
    mode = "end"
    def reset(self): pass
    def step(self, e): pass
    def finish(self): pass


def timed(label, pipe, gather, reps=5):
    out = []
    for _ in range(reps):
        bench.run_pipelined_steps(pipe, 5, pipe.depth, gather)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bench.run_pipelined_steps(pipe, K, pipe.depth, gather)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        out.append(((time.perf_counter() - t0) * 1e3, (t1 - t0) * 1e3))
    print(f"{label}: wall ms " + " ".join(f"{w:.2f}" for w, _ in out) + " | host loop ms " + " ".join(f"{h:.2f}" for _, h in out))


errs = {}
def post(full, central, i):
    if i not in errs:
        errs[i] = torch.empty((128, J), dtype=torch.float64, device="cuda")
    return per_joint_error(central, gt, cfg.ROOT_KEYTPOINT, out=errs[i])
p0 = model.pipeline(128, depth=depth, post=lambda f, c, i: None); p0.preload(x, m)
timed("bench loop, no error kernel, no gather copy", p0, NoGather())
p0.close()
p1 = model.pipeline(128, depth=depth, post=post); p1.preload(x, m)
timed("bench loop, error kernel in the graph, no gather copy", p1, NoGather())
g = bench.ErrorGather("end", K, 128, J, 1, "cuda", False)
timed("bench loop, error kernel + gather copy on the slot's stream", p1, g)
g.mode = "step"
timed("bench loop, error kernel + gather copy on the caller's stream (result)", p1, g)
