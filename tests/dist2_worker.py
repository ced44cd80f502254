"""One rank of tests/test_dist2_gpu.py: two of these processes share cuda:0 and talk over gloo (device tensors: gloo stages them
through the host, and its all_reduce is issued from -- and ordered behind -- the stream that is current at the call, which is how
dist.BucketedAllReduce hands it the library's own HIP stream).  RCCL refuses two ranks on one device, so this is the closest a
one-GPU box gets to the N > 1 ordering of the collectives.

    python tests/dist2_worker.py <rank> <world> <port> <outdir>
"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import uplift_upsample_3dhpe_amd as pkg                                    # noqa: E402
from uplift_upsample_3dhpe_amd import dist as udist                        # noqa: E402
from uplift_upsample_3dhpe_amd import eval as ev                           # noqa: E402
from uplift_upsample_3dhpe_amd import synthetic as util                    # noqa: E402
from uplift_upsample_3dhpe_amd.trainer import Trainer                      # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def main():
    rank, world, port, outdir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    out = {"rank": rank}

    # ---- (a) run_eval sharded over the ranks (ragged split), central predictions all-gathered ----
    cfg = util.load_config("h36m_81")
    cfg.BATCH_SIZE = 16
    cfg.MASK_STRIDE = cfg.MASK_STRIDE[0] if isinstance(cfg.MASK_STRIDE, list) else cfg.MASK_STRIDE
    arch = pkg.arch_from_config(cfg)
    model = pkg.build_uplift_upsample_transformer(cfg, weights=pkg.init_weights(arch, seed=2, perturb=0.1))
    # a ragged split whatever the number of windows is: rank 0 takes three more than half (run_eval all-gathers shards of any size)
    even_split = udist.shard_bounds
    shard = {}

    def uneven(n, r, w):
        cut = min(n, n // 2 + 3)
        lo, hi = (0, cut) if r == 0 else (cut, n)
        shard["n"] = hi - lo
        return lo, hi
    udist.shard_bounds = uneven
    rep = ev.run_eval(cfg, "h36m", os.path.join(G, "h36m_tiny_3d.npz"), os.path.join(G, "h36m_tiny_2d.npz"), "S9", model=model,
                      action_wise=False, log=lambda *a: None)
    udist.shard_bounds = even_split
    out["shard"] = shard["n"]
    out["eval"] = {"all_frames": rep["all_frames"], "keyframes": rep["keyframes"], "num_forwarded": rep["num_forwarded"],
                   "num_windows": rep["num_windows"]}

    # ---- (b) a training step whose four gradient buckets are all-reduced from the library's stream while the backward pass runs ----
    cfgt = util.load_config("h36m_351_pt")
    Bl = 40                                                                   # per rank: a long backward pass (~50 ms eager) behind the first bucket
    cfgt.BATCH_SIZE = Bl * world
    cfgt.EMA_ENABLED = False
    archt = pkg.arch_from_config(cfgt)
    w = pkg.init_weights(archt, seed=5, perturb=0.1)
    xg, mg = util.synthetic_batch(cfgt, Bl * world, seed=6)
    gtg = np.random.default_rng(7).normal(0, 0.3, size=(Bl * world, archt.num_frames, 17, 3)).astype(np.float32)
    lo, hi = udist.shard_bounds(Bl * world, rank, world)
    T_ = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    x, m, gt = T_(xg[lo:hi]), T_(mg[lo:hi]), T_(gtg[lo:hi])

    mdl_a = pkg.build_uplift_upsample_transformer(cfgt, weights=w)
    tr_a = Trainer(mdl_a, cfgt)
    seen = []
    orig = tr_a._buckets.ready
    tr_a._buckets.ready = lambda first, count, stream=None: (seen.append((int(first), int(count), stream)), orig(first, count, stream))[1]
    tr_a.forward_backward(x, gt, m, drop_path_uniform=None)
    tr_a._buckets.wait()
    torch.cuda.synchronize()
    bucketed = tr_a.grads.clone()

    mdl_b = pkg.build_uplift_upsample_transformer(cfgt, weights=w)
    tr_b = Trainer(mdl_b, cfgt)
    tr_b._buckets._active = lambda: False                                    # no collective: this rank's shard gradient
    tr_b.forward_backward(x, gt, m, drop_path_uniform=None)
    tr_b._buckets.wait()
    torch.cuda.synchronize()
    local = tr_b.grads.clone()
    flat = local.clone()
    dist.all_reduce(flat)                                                     # one flat sum of the two shard gradients
    torch.cuda.synchronize()
    out["buckets"] = len(seen)
    out["buckets_from_library_stream"] = all(s is not None for _, _, s in seen)
    out["bucketed_equals_flat_bitwise"] = bool(torch.equal(bucketed, flat))
    out["sum_differs_from_local"] = bool(not torch.equal(flat, local))     # (the other rank did contribute)
    out["grad_l2"] = float(flat.double().norm().item())

    # ---- (c) a non-finite gradient on ONE rank: every rank skips the update, the replicas stay identical ----
    before = tr_a.params.clone()
    xin = x.clone()
    if rank == 1:
        xin[0, 0, 0, 0] = float("inf")
    tr_a.train_step(xin, gt, m, drop_path_uniform=None)
    torch.cuda.synchronize()
    out["skipped"] = bool(tr_a.nonfinite())
    out["params_unchanged_after_nonfinite_step"] = bool(torch.equal(tr_a.params, before))
    tr_a.train_step(x, gt, m, drop_path_uniform=None)                        # a finite step afterwards moves the weights again, identically
    torch.cuda.synchronize()
    out["finite_step_applied"] = bool((not tr_a.nonfinite()) and not torch.equal(tr_a.params, before))
    psum = tr_a.params.double().sum().reshape(1).cpu()
    parts = [torch.zeros_like(psum) for _ in range(world)]
    dist.all_gather(parts, psum)
    out["replicas_identical"] = bool(all(torch.equal(p, parts[0]) for p in parts))
    ph = torch.frombuffer(bytearray(tr_a.params.cpu().numpy().tobytes()), dtype=torch.uint8)
    out["params_crc"] = int(np.bitwise_xor.reduce(np.frombuffer(ph.numpy().tobytes(), dtype=np.uint32)))

    with open(os.path.join(outdir, f"rank{rank}.json"), "w") as f:
        json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
