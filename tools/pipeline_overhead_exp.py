"""Where ForwardPipeline.submit / result lose time against bare graph replays (tools/streams_exp.py): per-step extras added one at a time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import uplift_upsample_3dhpe_amd as pkg
from uplift_upsample_3dhpe_amd import synthetic as util

cfg = util.load_config("h36m_351"); arch = pkg.arch_from_config(cfg); w = pkg.init_weights(arch, seed=0)
B = 128
x_np, m_np = util.synthetic_batch(cfg, B, seed=1000)
x = torch.from_numpy(x_np * m_np[:, :, None, None].astype(np.float32)).cuda(); m = torch.from_numpy(m_np).cuda()
model = pkg.build_uplift_upsample_transformer(cfg, weights=w)
for S in (1, 2, 4):
    pipe = model.pipeline(B, depth=S, graph=True)
    slots = pipe._slots
    main = torch.cuda.current_stream()
    def loop(n, mode):
        pend = []
        for k in range(n):
            s = slots[k % S]
            if mode >= 3: s.stream.wait_stream(main)
            with torch.cuda.stream(s.stream):
                if mode >= 4:
                    s.x.copy_(x, non_blocking=True); s.m.copy_(model._mask_u8(m), non_blocking=True)
                s.graph.replay()
                if mode >= 2: s.done.record(s.stream)
            if mode >= 2:
                pend.append(s)
                if len(pend) == S: main.wait_event(pend.pop(0).done)
    for mode, label in ((1, "replay only"), (2, "+ event record / main waits"), (3, "+ slot waits for main"), (4, "+ input copies"), (5, "pipe.submit / result")):
        def run(n):
            if mode < 5: return loop(n, mode)
            t = []
            for _ in range(n):
                t.append(pipe.submit(x, m))
                if len(t) == S: pipe.result(t.pop(0))
            for q in t: pipe.result(q)
        run(30); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(300); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"depth {S} {label:32s}: {B * 300 / dt:9.0f} sequences/s, {1e3 * dt / 300:.4f} ms per step", flush=True)
    pipe.close()
