"""Throughput of K forwards issued round robin on S HIP streams (one hipGraph + workspace + output buffers per stream): does the
latency-bound tail of one batch (strided blocks 2-3, heads: ~160 us of short launches) overlap the big kernels of the next?
  python tools/streams_exp.py [--config h36m_351] [--batch 128] [--steps 200]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="h36m_351")
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--streams", default="1,2,3")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--s-in", type=int, default=0, help="mask stride of every sequence (bench.py uses 5 = every frame real); 0 = the mixed evaluation masks")
    ap.add_argument("--prio", default="", help="comma list of stream priorities, cycled (0 / -1)")
    args = ap.parse_args()
    import numpy as np
    import torch
    import uplift_upsample_3dhpe_amd as pkg
    from uplift_upsample_3dhpe_amd import synthetic as util
    from uplift_upsample_3dhpe_amd.harness import per_joint_error
    cfg = util.load_config(args.config)
    arch = pkg.arch_from_config(cfg)
    w = pkg.init_weights(arch, seed=0)
    B, J = args.batch, arch.num_keypoints
    x_np, m_np = util.synthetic_batch(cfg, B, seed=1000, mask_specs=[(args.s_in, 0)] if args.s_in else None)
    x = torch.from_numpy(x_np * m_np[:, :, None, None].astype(np.float32)).cuda()
    m = torch.from_numpy(m_np).cuda()
    gt = torch.randn(B, J, 4, device="cuda")
    for S in [int(s) for s in args.streams.split(",")]:
        models = [pkg.build_uplift_upsample_transformer(cfg, weights=w) for _ in range(S)]
        pr = [int(p) for p in args.prio.split(",")] if args.prio else [0]
        streams = [torch.cuda.Stream(priority=pr[i % len(pr)]) for i in range(S)]
        errs = [torch.empty((B, J), dtype=torch.float64, device="cuda") for _ in range(S)]
        runs = []
        for i in range(S):
            def compute(i=i):
                full, central = models[i]([x, m], training=False)
                per_joint_error(central, gt, cfg.ROOT_KEYTPOINT, out=errs[i])
            with torch.cuda.stream(streams[i]):
                for _ in range(2):
                    compute()
            torch.cuda.synchronize()
            if args.no_graph:
                runs.append(compute)
            else:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=streams[i]):
                    compute()
                runs.append(g.replay)
        torch.cuda.synchronize()
        def loop(n):
            for k in range(n):
                with torch.cuda.stream(streams[k % S]):
                    runs[k % S]()
        loop(20)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop(args.steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"s_in {args.s_in} prio {args.prio or '-'} queues {os.environ.get('GPU_MAX_HW_QUEUES', '-')} streams {S} tail {'on' if os.environ.get('UU3D_TAIL') == '1' else 'off'} graph {not args.no_graph}: {B * args.steps / dt:9.0f} sequences/s, {1e3 * dt / args.steps:.4f} ms per step", flush=True)
        del models, runs


if __name__ == "__main__":
    main()
