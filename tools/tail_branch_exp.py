"""TIMING EXPERIMENT (round 5; results wrong by construction): would a forward's few-row tail (strided blocks 2.., head2, range check) cost less as a PARALLEL BRANCH of the slot's
graph -- beside the body of the slot's next batch -- than at the end of the forward's own chain of launches?  Per slot one hipGraph of
  (A) the whole forward                       (what pipeline.ForwardPipeline replays)
  (B) body || tail on two workspaces          (uu3d_forward_ex schedule bits 0x200 / 0x400: the two halves of the launch list, no data flow between them)
  (C) the body alone                          (the bound)
replayed round robin, `depth` slots on evenly dealt hardware queues.   python tools/tail_branch_exp.py [steps]"""
import os, sys, time
os.environ["UU3D_TIMING_PARTS"] = "1"        # (uu3d_forward_ex accepts the 0x200 / 0x400 schedule bits only with this)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import uplift_upsample_3dhpe_amd as pkg
from uplift_upsample_3dhpe_amd import synthetic as util
from uplift_upsample_3dhpe_amd.pipeline import distinct_queue_streams

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cfg = util.load_config("h36m_351"); arch = pkg.arch_from_config(cfg)
model = pkg.build_uplift_upsample_transformer(cfg, weights=pkg.init_weights(arch, seed=0))
B = 128
x_np, m_np = util.synthetic_batch(cfg, B, seed=1000, mask_specs=[(5, 0)])
x = torch.from_numpy(x_np * m_np[:, :, None, None].astype(np.float32)).cuda(); m = model._mask_u8(torch.from_numpy(m_np).cuda())
streams = distinct_queue_streams(model.device, want=4, per_queue=2)
depth = len(streams)
sides = [torch.cuda.Stream() for _ in range(depth)]


def build(mode):
    graphs, keep = [], []
    for i, s in enumerate(streams):
        full = torch.empty((B, arch.num_frames, arch.num_keypoints, 3), device="cuda"); cen = torch.empty((B, arch.num_keypoints, 3), device="cuda")
        full2 = torch.empty_like(full); cen2 = torch.empty_like(cen)
        keep += [full, cen, full2, cen2]

        def launch():
            if mode == "A":
                model._forward(x, m, full, cen, ("exp", mode, i), s, schedule=1)
            else:
                if mode == "B":
                    sides[i].wait_stream(s)
                    model._forward(x, m, full2, cen2, ("exp", mode, i, "tail"), sides[i], schedule=1 | 0x400)
                model._forward(x, m, full, cen, ("exp", mode, i), s, schedule=1 | 0x200)
                if mode == "B":
                    s.wait_stream(sides[i])
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            launch()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            launch()
        graphs.append(g)
    return graphs


def run(graphs, n):
    for k in range(n):
        with torch.cuda.stream(streams[k % depth]):
            graphs[k % depth].replay()


for mode, label in (("A", "whole forward per graph"), ("B", "body || tail branches"), ("C", "body only"), ("A", "whole forward per graph")):
    gs = build(mode)
    run(gs, 3 * depth); torch.cuda.synchronize()
    out = []
    for _ in range(3):
        t0 = time.perf_counter(); run(gs, steps); torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / steps * 1e3)
    print(f"{label:28s} ms per step " + " ".join(f"{v:.4f}" for v in out), flush=True)
    del gs
