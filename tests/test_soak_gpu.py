"""Soak tests of the hand-scheduled pieces: the same work many times over, every result bitwise equal to the first.

* forward: the spatial stack's packed-f32 arithmetic by name relies on hand-placed hazard fences (MFMA and transcendental
  results read by inline asm, docs/HISTORY.md E.12) -- a missing one reads a stale register only when the timing lines up;
* training step: three parameter-gradient streams with event marks -- a missing cross-stream
  dependency shows up as a rare mismatch in the gradient buffer.
The long versions are tools/soak_determinism.py (50 000 forwards: 0 mismatches), tools/soak_train_determinism.py (2 000 passes: 0) and
tools/soak_pipeline.py (20 000 batches through the four-queue pipeline: 0; profiles/r03_soak.txt)."""
import numpy as np
import pytest

import uplift_upsample_3dhpe_amd as pkg
from tests import util

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfgname,batch", [("h36m_351", 128), ("h36m_81", 37)])
def test_forward_is_bitwise_repeatable_over_many_launches(cfgname, batch):
    cfg = util.load_config(cfgname)
    arch = pkg.arch_from_config(cfg)
    model = pkg.build_uplift_upsample_transformer(cfg, weights=pkg.init_weights(arch, seed=0, perturb=0.1), device="cuda:0")
    x, m = util.synthetic_batch(cfg, batch=batch, seed=11)
    xt = torch.from_numpy(x * m[:, :, None, None].astype(np.float32)).cuda()
    mt = torch.from_numpy(m).cuda()
    full0, cen0 = (t.clone() for t in model([xt, mt], training=False))
    assert torch.isfinite(full0).all()
    for i in range(400):
        full, cen = model([xt, mt], training=False)
        assert torch.equal(full, full0) and torch.equal(cen, cen0), f"launch {i} differs from the first"


def test_training_pass_is_bitwise_repeatable_over_many_steps():
    from uplift_upsample_3dhpe_amd import _capi, harness
    from uplift_upsample_3dhpe_amd.trainer import Trainer
    B = 24
    cfg = util.load_config("h36m_351_pt")
    cfg.BATCH_SIZE = B
    arch = pkg.arch_from_config(cfg)
    model = pkg.build_uplift_upsample_transformer(cfg, weights=pkg.init_weights(arch, seed=0, perturb=0.1), device="cuda:0")
    tr = Trainer(model, cfg, seed=100)
    # repeated backward passes without the optimizer in between: no bucket bookkeeping
    _capi.check(tr._lib, tr._lib.uu3d_train_set_grad_callback(model._h, _capi.GRAD_READY_FN(0), None), model._h)
    tr._buckets.wait = lambda: None
    rng = np.random.default_rng(3000)
    N, J = arch.num_frames, arch.num_keypoints
    x = torch.from_numpy(rng.uniform(-1, 1, size=(B, N, J, 2)).astype(np.float32)).cuda()
    gt = torch.from_numpy(rng.normal(0, 0.3, size=(B, N, J, 3)).astype(np.float32)).cuda()
    m = torch.from_numpy(harness.stride_masks_train(N, cfg.SEQUENCE_STRIDE, cfg.MASK_STRIDE, B, rng, cfg.STRIDE_MASK_RAND_SHIFT)).cuda()
    u = torch.rand(tr.drop_path_size(B), device="cuda")
    loss0, _, _ = tr.forward_backward(x, gt, m, drop_path_uniform=u)
    torch.cuda.synchronize()
    g0, l0 = tr.grads.clone(), loss0.clone()
    for i in range(60):
        loss, _, _ = tr.forward_backward(x, gt, m, drop_path_uniform=u)
        torch.cuda.synchronize()
        assert torch.equal(tr.grads, g0) and torch.equal(loss, l0), f"pass {i} differs from the first"


def test_pipelined_forwards_stay_bitwise_right_over_many_batches():
    """Four forwards in flight on four hardware queues (pipeline.ForwardPipeline, the bench's and run_eval's path), 1200 batches cycling through
    six different inputs: every result equals what a quiet throughput-schedule call returned for that input -- a slot reading another slot's workspace, a graph
    replayed before its inputs landed, or the throughput schedule's launch shapes computing something else would show up here."""
    cfg = util.load_config("h36m_351")
    arch = pkg.arch_from_config(cfg)
    model = pkg.build_uplift_upsample_transformer(cfg, weights=pkg.init_weights(arch, seed=0, perturb=0.1), device="cuda:0")
    B = 128
    inputs, want = [], []
    for k in range(6):
        x, m = util.synthetic_batch(cfg, batch=B, seed=20 + k)
        xt = torch.from_numpy(x * m[:, :, None, None].astype(np.float32)).cuda(); mt = torch.from_numpy(m).cuda()
        inputs.append((xt, mt))
        want.append(tuple(t.clone() for t in util.direct_forward(model, xt, mt, 1)))     # a quiet call under the slots' schedule
    pipe = model.pipeline(B)
    assert pipe.depth >= 2
    n = 0
    for full, cen in pipe.run(inputs[i % 6] for i in range(1200)):
        fw, cw = want[n % 6]
        assert torch.equal(full, fw) and torch.equal(cen, cw), f"batch {n} differs"
        n += 1
    assert n == 1200
    pipe.close()


def test_pipelined_temporal_chain_stays_bitwise_right_over_many_batches(monkeypatch):
    """The temporal chain (csrc/uu3d_tchain.h; forced on at this batch, UU3D_TCHAIN=1) through the four-queue pipeline: 600 batches cycling
    through six inputs, every result equal to what a quiet throughput-schedule forward returned for that input -- the chain's lane-private
    scratch slabs belong to a slot's workspace, its ring and counted waits are hand-scheduled: a slot reading another slot's scratch, or a
    wait that counts one operation too few, shows up here as a rare mismatch."""
    monkeypatch.setenv("UU3D_TCHAIN", "1")
    cfg = util.load_config("h36m_351")
    arch = pkg.arch_from_config(cfg)
    model = pkg.build_uplift_upsample_transformer(cfg, weights=pkg.init_weights(arch, seed=0, perturb=0.1), device="cuda:0")
    B = 128
    inputs, want = [], []
    for k in range(6):
        x, m = util.synthetic_batch(cfg, batch=B, seed=60 + k)
        xt = torch.from_numpy(x * m[:, :, None, None].astype(np.float32)).cuda(); mt = torch.from_numpy(m).cuda()
        inputs.append((xt, mt))
        full = torch.empty((B, arch.num_frames, arch.num_keypoints, 3), dtype=torch.float32, device="cuda")
        cen = torch.empty((B, arch.num_keypoints, 3), dtype=torch.float32, device="cuda")
        model._forward(xt, model._mask_u8(mt), full, cen, 0, torch.cuda.current_stream(), schedule=1)
        torch.cuda.synchronize()
        want.append((full, cen))
    model.set_profiling(True)
    model._forward(inputs[0][0], model._mask_u8(inputs[0][1]), full.clone(), cen.clone(), 0, torch.cuda.current_stream(), schedule=1)
    assert any(e["kernel"] == "tchain" for e in model.read_profile())
    model.set_profiling(False)
    pipe = model.pipeline(B)
    assert pipe.depth >= 2
    n = 0
    for full, cen in pipe.run(inputs[i % 6] for i in range(600)):
        fw, cw = want[n % 6]
        assert torch.equal(full, fw) and torch.equal(cen, cw), f"batch {n} differs"
        n += 1
    assert n == 600
    pipe.check_range()
    pipe.close()


def test_schedules_agree_over_random_batch_sizes_and_masks():
    """Throughput against latency schedule (the temporal chain with its ragged last tiles, the split-K shapes beside it, attn_h3_kernel on fragment-ordered
    q | k | v) over random batch sizes between 15 and 150 sequences and random stride masks, both shipped architectures: within 3e-5, and bitwise run to run."""
    rng = np.random.default_rng(2025)
    for cfgname in ("h36m_351", "h36m_81"):
        cfg = util.load_config(cfgname)
        arch = pkg.arch_from_config(cfg)
        model = pkg.build_uplift_upsample_transformer(cfg, weights=pkg.init_weights(arch, seed=8, perturb=0.1))
        worst = 0.0
        for trial in range(10):
            B = int(rng.integers(15, 151)) if cfgname == "h36m_351" else int(rng.integers(26, 200))
            specs = [(int(rng.choice([4, 5, 10, 20])), int(rng.integers(0, 4))) for _ in range(int(rng.integers(1, 4)))]
            x, m = util.synthetic_batch(cfg, B, seed=100 + trial, mask_specs=specs)
            xt, mt = torch.from_numpy(x * m[:, :, None, None].astype(np.float32)).cuda(), torch.from_numpy(m).cuda()
            f0, c0 = util.direct_forward(model, xt, mt, 0)
            f1, c1 = util.direct_forward(model, xt, mt, 1)
            f2, c2 = util.direct_forward(model, xt, mt, 1)
            assert torch.equal(f1, f2) and torch.equal(c1, c2), (cfgname, B, specs)
            d = max(float((f0 - f1).abs().max()), float((c0 - c1).abs().max()))
            worst = max(worst, d)
            assert d <= 3e-5, (cfgname, B, specs, d)
        assert model.check_range(raise_error=False) is False
        print(f"{cfgname}: throughput vs latency schedule over 10 random batches: max {worst:.2e}")
