"""Which HIP streams share a hardware queue, and which sharing pattern the forward pipeline likes.
HIP deals streams to GPU_MAX_HW_QUEUES (default 4) hardware queues; two streams on one queue run in order.  `same_queue` finds out by
blocking one stream with a spin kernel and watching whether a tiny kernel on the other gets through.
  python tools/queue_map_exp.py [--batch 128] [--steps 150]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="h36m_351")
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--pool", type=int, default=24)
    args = ap.parse_args()
    import numpy as np
    import torch
    import uplift_upsample_3dhpe_amd as pkg
    from uplift_upsample_3dhpe_amd import synthetic as util
    from uplift_upsample_3dhpe_amd.pipeline import ForwardPipeline
    cfg = util.load_config(args.config)
    arch = pkg.arch_from_config(cfg)
    model = pkg.build_uplift_upsample_transformer(cfg, weights=pkg.init_weights(arch, seed=0))
    dev = model.device
    tiny = torch.zeros(64, device=dev)
    # calibrate the spin
    torch.cuda.synchronize()
    t0 = time.perf_counter(); torch.cuda._sleep(10_000_000); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    cyc_per_ms = 10_000_000 / (dt * 1e3)
    print(f"spin: {cyc_per_ms:.0f} cycles per ms")

    def same_queue(a, b):
        torch.cuda.synchronize()
        with torch.cuda.stream(a):
            torch.cuda._sleep(int(4 * cyc_per_ms))
        eb = torch.cuda.Event()
        with torch.cuda.stream(b):
            tiny.add_(1.0)
            eb.record(b)
        time.sleep(0.0015)
        done = eb.query()
        torch.cuda.synchronize()
        return not done

    cur = torch.cuda.current_stream(dev)
    first = None
    if os.environ.get("QUEUE_MAP_FIRST"):       # what a fresh process gets (bench.py): the pipeline's probe before anything else used a stream
        from uplift_upsample_3dhpe_amd.pipeline import distinct_queue_streams as dqs
        first = dqs(dev, want=4, per_queue=2)
    pool = [torch.cuda.Stream(device=dev) for _ in range(args.pool)]
    reps, cls = [], []
    for s in pool:
        for ci, r in enumerate(reps):
            if same_queue(r, s):
                cls.append(ci); break
        else:
            reps.append(s); cls.append(len(reps) - 1)
    cur_cls = next((ci for ci, r in enumerate(reps) if same_queue(r, cur)), -1)
    print("queue class of pool streams:", cls, "| current stream:", cur_cls, "| classes:", len(reps))
    if first is not None:
        print("distinct_queue_streams(per_queue=2) called first in the process: classes by this tool's probe",
              [next((ci for ci, r in enumerate(reps) if same_queue(r, q)), -1) for q in first])

    x_np, m_np = util.synthetic_batch(cfg, args.batch, seed=1000, mask_specs=[(5, 0)])
    x = torch.from_numpy(x_np * m_np[:, :, None, None].astype(np.float32)).to(dev)
    m = torch.from_numpy(m_np).to(dev)

    resident = bool(os.environ.get("QUEUE_MAP_R5"))

    def measure(label, idx):
        pipe = ForwardPipeline(model, args.batch, depth=len(idx), streams=[pool[i] for i in idx])
        d = len(idx)
        if resident:
            pipe.preload(x, m)

        def run(n):
            t = []
            for _ in range(n):
                t.append(pipe.launch() if resident else pipe.submit(x, m))
                if len(t) == d:
                    pipe.result(t.pop(0))
            for q in t:
                pipe.result(q)
        run(30); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(args.steps); torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        pipe.close()
        print(f"{label:44s} streams {idx} classes {[cls[i] for i in idx]}: {ms:.4f} ms per step = {args.batch / ms:.1f} k sequences/s")

    by = {}
    for i, c in enumerate(cls):
        by.setdefault(c, []).append(i)
    K = sorted(by)
    nq = len(K)
    pick = lambda pattern: [by[K[c % nq]][k] for c, k in pattern]
    for n in range(1, nq + 1):
        measure(f"{n} slots, {n} queues", pick([(c, 0) for c in range(n)]))
    if os.environ.get("QUEUE_MAP_DISTINCT_ONLY"):
        return
    if resident:                                # round 5: eight slots (the temporal chain's launches), inputs resident in the slots
        from uplift_upsample_3dhpe_amd.pipeline import distinct_queue_streams
        for pq in (1, 2, 3):
            qs = distinct_queue_streams(dev, want=4, per_queue=pq)
            qc = [next((ci for ci, r in enumerate(reps) if same_queue(r, q)), -1) for q in qs]
            print(f"distinct_queue_streams(per_queue={pq}): {len(qs)} streams, classes by this tool's probe {qc}, distinct handles {len({q.cuda_stream for q in qs})}")
            n0 = len(pool); pool.extend(qs); cls.extend(qc)
            measure(f"{len(qs)} slots from distinct_queue_streams(per_queue={pq})", list(range(n0, n0 + len(qs))))
        for rep in range(2):
            measure("4 slots, 4 queues", pick([(c, 0) for c in range(4)]))
            measure("8 slots, class-major (ABCDABCD)", pick([(c, k) for k in range(2) for c in range(4)]))
            measure("8 slots, queue-major (AABBCCDD)", pick([(c, k) for c in range(4) for k in range(2)]))
            measure("8 slots, pool order", list(range(8)))
            measure("8 slots, pool order from 8", list(range(8, 16)))
            measure("12 slots, class-major", pick([(c, k) for k in range(3) for c in range(4)]))
            measure("12 slots, pool order", list(range(12)))
            measure("16 slots, pool order", list(range(16)))
        return
    # patterns: (class, k-th stream of that class)
    if nq >= 2:
        measure("2 slots, 2 queues", pick([(0, 0), (1, 0)]))
        measure("2 slots, 1 queue", pick([(0, 0), (0, 1)]))
        measure("4 slots, 2 queues (AABB)", pick([(0, 0), (0, 1), (1, 0), (1, 1)]))
        measure("4 slots, 2 queues (ABAB)", pick([(0, 0), (1, 0), (0, 1), (1, 1)]))
        measure("6 slots, 2 queues (ABABAB)", pick([(0, 0), (1, 0), (0, 1), (1, 1), (0, 2), (1, 2)]))
    if nq >= 3:
        measure("3 slots, 3 queues", pick([(0, 0), (1, 0), (2, 0)]))
        measure("6 slots, 3 queues (ABCABC)", pick([(0, 0), (1, 0), (2, 0), (0, 1), (1, 1), (2, 1)]))
        measure("4 slots, 3 queues (ABCA)", pick([(0, 0), (1, 0), (2, 0), (0, 1)]))
    if nq >= 4:
        measure("4 slots, 4 queues", pick([(0, 0), (1, 0), (2, 0), (3, 0)]))
        measure("8 slots, 4 queues (ABCDABCD)", pick([(0, 0), (1, 0), (2, 0), (3, 0), (0, 1), (1, 1), (2, 1), (3, 1)]))
        measure("6 slots, 4 queues (ABCDAB)", pick([(0, 0), (1, 0), (2, 0), (3, 0), (0, 1), (1, 1)]))
    # without the current stream's queue
    other = [c for c in K if c != cur_cls]
    if cur_cls >= 0 and len(other) >= 2:
        o = lambda pattern: [by[other[c % len(other)]][k] for c, k in pattern]
        measure("2 slots, 2 queues, not the caller's", o([(0, 0), (1, 0)]))
        if len(other) >= 3:
            measure("3 slots, 3 queues, not the caller's", o([(0, 0), (1, 0), (2, 0)]))
            measure("6 slots, 3 queues, not the caller's", o([(0, 0), (1, 0), (2, 0), (0, 1), (1, 1), (2, 1)]))
        measure("4 slots, 2 queues, not the caller's (ABAB)", o([(0, 0), (1, 0), (0, 1), (1, 1)]))


if __name__ == "__main__":
    main()
