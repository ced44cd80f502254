"""Training-step rows that need no back-propagation: oracle on CPU, kernels on GPU."""
import numpy as np
import pytest
import torch

from oracle import train_oracle as T
from tests import util


# ---------------------------------------------------------------- CPU: the oracle itself
def test_schedules_known_values():
    cfg = util.load_config("h36m_351_pt")
    sp = cfg.SCHEDULE_PARAMS
    assert cfg.SCHEDULE == "ExponentialDecay" and sp["staircase"] is True
    f = lambda s: T.exponential_decay(sp["initial_learning_rate"], sp["decay_steps"], sp["decay_rate"], s, True)
    assert f(0) == np.float32(2e-5) and f(5999) == np.float32(2e-5)
    assert f(6000) == pytest.approx(2e-5 * 0.99, rel=1e-6) and f(12000) == pytest.approx(2e-5 * 0.99 ** 2, rel=1e-6)
    g = lambda s: T.exponential_decay_with_steps(1e-3, 12000, 0.95, 60000, 0.5, s)
    assert g(0) == np.float32(1e-3) and g(12000) == pytest.approx(0.95e-3, rel=1e-6)
    assert g(60000) == pytest.approx(1e-3 * 0.95 ** 4 * 0.5, rel=1e-6)      # p = 5 - 1
    assert T.ema_decay_value(0.999, 0) == np.float32(0.1) and T.ema_decay_value(0.999, 10 ** 6) == np.float32(0.999)
    from uplift_upsample_3dhpe_amd import optim
    sched = optim.scheduler_by_name(cfg.SCHEDULE)(**sp)
    assert sched(6000) == float(f(6000)) and optim.ExponentialDecayWithSteps(1e-3, 12000, 0.95, 60000, 0.5)(60000) == float(g(60000))


def test_loss_gradient_matches_finite_differences():
    rng = np.random.default_rng(0)
    B, N, J = 2, 5, 17
    pf, pc = rng.normal(size=(B, N, J, 3)), rng.normal(size=(B, J, 3))
    gt = rng.normal(size=(B, N, J, 3))
    o = T.train_loss(pf, pc, gt, 6, 0.5, 0.5, 4)
    d = (gt - gt[:, :, 6:7])
    cen = np.linalg.norm(d[:, N // 2] - pc, axis=-1).sum() / (4 * J)
    seq = np.linalg.norm(d - pf, axis=-1).sum() / (4 * N * J)
    assert o["central"] == pytest.approx(cen, rel=1e-5) and o["seq"] == pytest.approx(seq, rel=1e-5)
    assert o["loss"] == pytest.approx(0.5 * cen + 0.5 * seq, rel=1e-5)
    # analytic gradient of tf.norm(gt - pred): (pred - gt) / ||pred - gt||, scaled by w / normaliser
    gf = 0.5 / (4 * N * J) * (pf - d) / np.linalg.norm(pf - d, axis=-1, keepdims=True)
    gc = 0.5 / (4 * J) * (pc - d[:, N // 2]) / np.linalg.norm(pc - d[:, N // 2], axis=-1, keepdims=True)
    assert np.allclose(o["grad_full"], gf, rtol=1e-4, atol=1e-8) and np.allclose(o["grad_central"], gc, rtol=1e-4, atol=1e-8)
    o2 = T.train_loss(None, pc, gt, 6, 0.5, 0.5, 4)
    assert o2["loss"] == pytest.approx(cen, rel=1e-5)        # fallback (w_c + w_s) * central, train.py:491-494


def test_adamw_oracle_first_step_and_decay():
    # first step: m = (1-b1) g, v = (1-b2) g^2, alpha-scaled update = lr * g / (|g| + eps*...) ~ lr * sign(g)
    w = np.array([1.0, -2.0, 0.5], np.float32); g = np.array([0.1, -0.2, 0.0], np.float32)
    var, m, v = T.adamw_update(w, np.zeros(3), np.zeros(3), g, lr=1e-3, wd=1e-2, beta1=0.9, beta2=0.999, eps=1e-8, step=1)
    assert np.allclose(m, 0.1 * g) and np.allclose(v, 0.001 * g * g, rtol=1e-5)
    assert np.allclose(var, w * (1 - 1e-2) - 1e-3 * np.sign(g), atol=1e-6)     # decay is NOT scaled by lr (tfa)
    assert np.allclose(T.ema_update(np.ones(3), np.zeros(3), 0.9), 0.9)


def test_adamw_oracle_amsgrad():
    """Keras Adam(amsgrad=True) (the config CLASS default OPTIMIZER_PARAMS, config.py:88): vhat never decreases and replaces v
    in the denominator; with a growing |g| it equals v, so the update is the plain one."""
    w = np.array([1.0, -2.0], np.float32)
    var, m, v, vh = T.adamw_update(w, np.zeros(2), np.zeros(2), np.array([0.1, -0.2], np.float32), 1e-3, 0.0, 0.9, 0.999, 1e-8, 1, vhat=np.zeros(2))
    plain = T.adamw_update(w, np.zeros(2), np.zeros(2), np.array([0.1, -0.2], np.float32), 1e-3, 0.0, 0.9, 0.999, 1e-8, 1)
    assert np.array_equal(vh, v) and np.array_equal(var, plain[0])
    var2, m2, v2, vh2 = T.adamw_update(var, m, v, np.zeros(2, np.float32), 1e-3, 0.0, 0.9, 0.999, 1e-8, 2, vhat=vh)
    assert np.all(v2 < v) and np.array_equal(vh2, vh)          # v decays, vhat holds the maximum
    assert not np.array_equal(var2, T.adamw_update(var, m, v, np.zeros(2, np.float32), 1e-3, 0.0, 0.9, 0.999, 1e-8, 2)[0])


# ---------------------------------------------------------------- GPU: kernels through the C ABI
@pytest.mark.gpu
def test_adamw_amsgrad_kernel_bit_exact_over_steps():
    from uplift_upsample_3dhpe_amd import optim
    rng = np.random.default_rng(4)
    n = 50001
    w0 = rng.normal(0, 0.05, n).astype(np.float32)
    params = torch.from_numpy(w0.copy()).cuda()
    opt = optim.AdamW(params, weight_decay=1e-6, learning_rate=1e-4, epsilon=1e-8, amsgrad=True)
    var, m, v, vh = w0.copy(), np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
    for it in range(5):
        g = (rng.normal(0, 1e-2, n) * (1.0 if it % 2 == 0 else 0.1)).astype(np.float32)     # v falls below vhat on odd steps
        var, m, v, vh = T.adamw_update(var, m, v, g, 1e-4, 1e-6, 0.9, 0.999, 1e-8, it + 1, vhat=vh)
        opt.apply_gradients(torch.from_numpy(g).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(opt.vhat.cpu().numpy(), vh) and np.array_equal(opt.v.cpu().numpy(), v)
    assert np.array_equal(params.cpu().numpy(), var)
    assert (vh > v).any()



@pytest.mark.gpu
def test_adamw_kernel_bit_exact_over_steps():
    from uplift_upsample_3dhpe_amd import optim
    rng = np.random.default_rng(1)
    n = 100003                                             # not a multiple of 4: exercises the tail
    w0 = rng.normal(0, 0.05, n).astype(np.float32)
    cfg = util.load_config("h36m_351_pt")
    lr_s = optim.ExponentialDecay(**cfg.SCHEDULE_PARAMS)
    wd_s = optim.ExponentialDecay(**dict(cfg.SCHEDULE_PARAMS, initial_learning_rate=cfg.WEIGHT_DECAY))
    params = torch.from_numpy(w0.copy()).cuda()
    opt = optim.AdamW(params, weight_decay=wd_s, learning_rate=lr_s, epsilon=1e-8)
    opt.iterations = 5998                                  # crosses the staircase boundary at 6000
    var, m, v = w0.copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
    for it in range(4):
        g = rng.normal(0, 1e-2, n).astype(np.float32)
        step = opt.iterations
        lr = T.exponential_decay(2e-5, 6000, 0.99, step, True); wd = T.exponential_decay(2e-6, 6000, 0.99, step, True)
        var, m, v = T.adamw_update(var, m, v, g, lr, wd, 0.9, 0.999, 1e-8, step + 1)
        opt.apply_gradients(torch.from_numpy(g).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(opt.m.cpu().numpy(), m) and np.array_equal(opt.v.cpu().numpy(), v)
    assert np.array_equal(params.cpu().numpy(), var)       # separately rounded float32 ops: bit-exact


@pytest.mark.gpu
def test_adamw_full_model_size_and_ema():
    import uplift_upsample_3dhpe_amd as pkg
    from uplift_upsample_3dhpe_amd import optim
    arch = pkg.arch_from_config(util.load_config("h36m_351_pt"))
    w = pkg.init_weights(arch, seed=0)
    flat = np.concatenate([w[k].ravel() for k in w]).astype(np.float32)
    assert flat.size == 10404902
    g = np.random.default_rng(2).normal(0, 1e-3, flat.size).astype(np.float32)
    params = torch.from_numpy(flat.copy()).cuda()
    opt = optim.AdamW(params, weight_decay=2e-6, learning_rate=2e-5, epsilon=1e-8)
    opt.apply_gradients(torch.from_numpy(g).cuda())
    ref, _, _ = T.adamw_update(flat, np.zeros_like(flat), np.zeros_like(flat), g, 2e-5, 2e-6, 0.9, 0.999, 1e-8, 1)
    assert np.array_equal(params.cpu().numpy(), ref)
    ema = torch.from_numpy(flat.copy()).cuda()
    d = optim.ema_decay_value(0.999, 3)
    optim.ema_update(ema, params, d)
    assert np.array_equal(ema.cpu().numpy(), T.ema_update(flat, ref, np.float32(d)))


@pytest.mark.gpu
@pytest.mark.parametrize("with_full", [True, False])
def test_loss_kernel_matches_oracle(with_full):
    from uplift_upsample_3dhpe_amd import optim
    cfg = util.load_config("h36m_351_pt")
    rng = np.random.default_rng(3)
    B, N, J = 6, 71, 17
    pf = rng.normal(0, 0.3, (B, N, J, 3)).astype(np.float32); pc = rng.normal(0, 0.3, (B, J, 3)).astype(np.float32)
    gt = rng.normal(0, 0.3, (B, N, J, 3)).astype(np.float32)
    o = T.train_loss(pf if with_full else None, pc, gt, cfg.ROOT_KEYTPOINT, cfg.LOSS_WEIGHT_CENTER, cfg.LOSS_WEIGHT_SEQUENCE, cfg.BATCH_SIZE)
    loss, gf, gc = optim.train_loss(torch.from_numpy(pf).cuda() if with_full else None, torch.from_numpy(pc).cuda(),
                                    torch.from_numpy(gt).cuda(), cfg)
    loss = loss.cpu().numpy()
    assert loss[0] == pytest.approx(o["loss"], rel=2e-6) and loss[1] == pytest.approx(o["central"], rel=2e-6)
    assert np.abs(gc.cpu().numpy() - o["grad_central"]).max() <= 1e-9 + 2e-6 * np.abs(o["grad_central"]).max()
    if with_full:
        assert loss[2] == pytest.approx(o["seq"], rel=2e-6)
        assert np.abs(gf.cpu().numpy() - o["grad_full"]).max() <= 1e-9 + 2e-6 * np.abs(o["grad_full"]).max()
    else:
        assert gf is None
    l2, _, _ = optim.train_loss(torch.from_numpy(pf).cuda() if with_full else None, torch.from_numpy(pc).cuda(),
                                torch.from_numpy(gt).cuda(), cfg)
    assert np.array_equal(l2.cpu().numpy(), loss)          # deterministic reduction
