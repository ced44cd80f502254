"""GPU: the RCCL code paths on ONE GPU (backend "nccl" at world size 1) -- the all-gather of the (B, J) float64 error block that
bench.py / eval issue per step, and a Trainer.train_step whose gradient buckets go through
dist.BucketedAllReduce's ExternalStream branch (the collective is issued from the library's own HIP stream).  With one rank a sum
over ranks is the identity, so results must equal the run without a process group bit for bit."""
import socket

import numpy as np
import pytest

import uplift_upsample_3dhpe_amd as pkg
from tests import util

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture()
def nccl_world1():
    import torch.distributed as dist
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        yield dist
    finally:
        dist.destroy_process_group()


def test_error_block_all_gather_and_bucketed_gradients_on_rccl(nccl_world1):
    dist = nccl_world1
    from uplift_upsample_3dhpe_amd import dist as udist
    from uplift_upsample_3dhpe_amd.harness import per_joint_error
    from uplift_upsample_3dhpe_amd.trainer import Trainer
    # ---- inference: forward -> per-joint error (float64) -> all_gather_into_tensor, as bench.py does per step ----
    cfg = util.load_config("h36m_351")
    arch = pkg.arch_from_config(cfg)
    model = pkg.build_uplift_upsample_transformer(cfg, weights=pkg.init_weights(arch, seed=1, perturb=0.1))
    B, J = 16, arch.num_keypoints
    x, m = util.synthetic_batch(cfg, B, seed=3)
    xm = torch.from_numpy(x * m[:, :, None, None].astype(np.float32)).cuda()
    gt = torch.cat([torch.randn(B, J, 3, device="cuda") * 0.3, torch.ones(B, J, 1, device="cuda")], -1)
    err = torch.empty((B, J), dtype=torch.float64, device="cuda")
    gathered = torch.empty((B, J), dtype=torch.float64, device="cuda")
    pipe = model.pipeline(B, depth=2, graph=True, post=lambda f, c, i: per_joint_error(c, gt, cfg.ROOT_KEYTPOINT))
    t = pipe.submit(xm, torch.from_numpy(m).cuda())
    e = pipe.result(t)[2]
    dist.all_gather_into_tensor(gathered, e)
    _, cen = util.direct_forward(model, xm, torch.from_numpy(m).cuda(), 1)       # a quiet call under the slots' (throughput) schedule
    per_joint_error(cen, gt, cfg.ROOT_KEYTPOINT, out=err)
    torch.cuda.synchronize()
    assert torch.equal(gathered, err)
    assert udist.mean_valid_mm(gathered) > 0
    pipe.close()

    # ---- training: bucketed all-reduce on the library's side stream (ExternalStream) vs no collective at all ----
    cfgt = util.load_config("h36m_81")
    cfgt.BATCH_SIZE = 4
    cfgt.EMA_ENABLED = False
    archt = pkg.arch_from_config(cfgt)
    w = pkg.init_weights(archt, seed=5, perturb=0.1)
    xt, mt = util.synthetic_batch(cfgt, 4, seed=6)
    gtt = np.random.default_rng(7).normal(0, 0.3, size=(4, archt.num_frames, 17, 3)).astype(np.float32)
    T_ = lambda a: torch.from_numpy(a).cuda()
    results = []
    for force in (False, True):
        mdl = pkg.build_uplift_upsample_transformer(cfgt, weights=w)
        tr = Trainer(mdl, cfgt)
        tr._buckets.force = force
        seen = []
        if force:
            orig = tr._buckets.ready
            tr._buckets.ready = lambda first, count, stream=None: (seen.append((first, count, stream)), orig(first, count, stream))[1]
        for _ in range(2):
            tr.train_step(T_(xt), T_(gtt), T_(mt), drop_path_uniform=None)
        torch.cuda.synchronize()
        if force:
            assert len(seen) >= 4 and all(s is not None for _, _, s in seen), "the buckets were not issued from the library's stream"
        results.append((tr.params.clone(), tr.grads.clone()))
    assert torch.equal(results[0][0], results[1][0]) and torch.equal(results[0][1], results[1][1])
