"""Backward building blocks (include/uu3d_ops.h) against numpy / torch-autograd restatements of the
TensorFlow gradients they replace (train.py:477,498).  Tolerances are relative to each tensor's scale."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from uplift_upsample_3dhpe_amd import _capi
    return _capi.load_library()


def _d(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _scratch(lib):
    n = int(lib.uu3d_op_scratch_floats())
    return torch.empty(n, dtype=torch.float32, device="cuda"), n


def _close(a, b, rel):
    scale = max(np.abs(b).max(), 1e-30)
    assert np.abs(a - b).max() <= rel * scale, (np.abs(a - b).max(), scale)


@pytest.mark.parametrize("R,P,Q", [(9088, 384, 1152), (2944, 2304, 384), (128, 384, 52), (154496, 32, 96), (777, 64, 64)])
def test_gemm_tn(lib, R, P, Q):
    rng = np.random.default_rng(0)
    a = rng.normal(size=(R, P)).astype(np.float32); b = rng.normal(size=(R, Q)).astype(np.float32)
    ad, bd = _d(a), _d(b)
    c = torch.empty(P, Q, device="cuda"); sc, n = _scratch(lib)
    assert lib.uu3d_op_gemm_tn(_p(ad), P, _p(bd), Q, R, P, Q, _p(c), Q, _p(sc), n, None) == 0
    ref = a.astype(np.float64).T @ b.astype(np.float64)
    _close(c.cpu().numpy(), ref, 2e-6)
    c2 = torch.empty_like(c)
    assert lib.uu3d_op_gemm_tn(_p(ad), P, _p(bd), Q, R, P, Q, _p(c2), Q, _p(sc), n, None) == 0
    assert torch.equal(c, c2)                                 # deterministic split-K


@pytest.mark.parametrize("R,P,Q", [(9088, 384, 1152), (4544, 384, 384), (1472, 2304, 384), (4544, 544, 384), (777, 132, 200), (33, 128, 128)])
def test_gemm_tn_h3(lib, R, P, Q):
    """Weight-gradient product on the f16 matrix cores (f16x3, fragments through ds_read_b64_tr_b16): ragged tiles, rows
    that are no multiple of the k-step, operands with gradient-sized entries (f16 denormal range)."""
    rng = np.random.default_rng(2)
    a = rng.normal(size=(R, P)).astype(np.float32)
    b = (rng.normal(size=(R, Q)) * np.exp(rng.uniform(-14, 0, size=(R, 1)))).astype(np.float32)      # rows from 1e-6 to 1
    ad, bd = _d(a), _d(b)
    c = torch.full((P, Q), float("nan"), device="cuda"); sc, n = _scratch(lib)
    assert lib.uu3d_op_gemm_tn_h3(_p(ad), P, _p(bd), Q, R, P, Q, _p(c), Q, _p(sc), n, None) == 0
    ref = a.astype(np.float64).T @ b.astype(np.float64)
    _close(c.cpu().numpy(), ref, 2e-6)
    c2 = torch.empty_like(c)
    assert lib.uu3d_op_gemm_tn_h3(_p(ad), P, _p(bd), Q, R, P, Q, _p(c2), Q, _p(sc), n, None) == 0
    assert torch.equal(c, c2)


@pytest.mark.parametrize("M,N,K", [(9088, 384, 1152), (384, 768, 384), (128, 64, 384)])
def test_gemm_nt(lib, M, N, K):
    rng = np.random.default_rng(1)
    a = rng.normal(size=(M, K)).astype(np.float32); w = rng.normal(size=(N, K)).astype(np.float32)
    c = torch.empty(M, N, device="cuda"); sc, n = _scratch(lib)
    assert lib.uu3d_op_gemm_nt(_p(_d(a)), K, _p(_d(w)), K, M, N, K, _p(c), N, _p(sc), n, None) == 0
    _close(c.cpu().numpy(), a.astype(np.float64) @ w.astype(np.float64).T, 2e-6)


def test_colsum_plain_period_mask(lib):
    rng = np.random.default_rng(2)
    R, Cc, N = 128 * 71, 384, 71
    x = rng.normal(size=(R, Cc)).astype(np.float32); m = (rng.random(R) < 0.4)
    sc, n = _scratch(lib)
    out = torch.empty(Cc, device="cuda")
    assert lib.uu3d_op_colsum(_p(_d(x)), Cc, R, Cc, 0, None, 0, _p(out), 0, _p(sc), n, None) == 0
    _close(out.cpu().numpy(), x.astype(np.float64).sum(0), 2e-6)
    outp = torch.empty(N, Cc, device="cuda")                                     # positional-encoding gradient
    assert lib.uu3d_op_colsum(_p(_d(x)), Cc, R, Cc, N, None, 0, _p(outp), 0, _p(sc), n, None) == 0
    _close(outp.cpu().numpy(), x.astype(np.float64).reshape(128, N, Cc).sum(0), 2e-6)
    outm = torch.ones(Cc, device="cuda")                                         # token gradient: masked rows, accumulate
    assert lib.uu3d_op_colsum(_p(_d(x)), Cc, R, Cc, 0, _p(_d(m.astype(np.uint8))), 0, _p(outm), 1, _p(sc), n, None) == 0
    _close(outm.cpu().numpy(), 1.0 + x[~m].astype(np.float64).sum(0), 2e-6)


@pytest.mark.parametrize("nb,P,Cc", [(4544, 17, 32), (64, 24, 384), (3, 1, 384), (5, 7, 30)])
def test_colsum_periodic_shapes(lib, nb, P, Cc):
    """Positional-encoding gradients on colsum_period4_kernel (C % 4 == 0: the spatial stack's 17 x 32 over 4544 frames, a strided
    block's 24 x 384, a one-token sequence) and on the generic kernel (C % 4 != 0)."""
    rng = np.random.default_rng(12)
    x = rng.normal(size=(nb * P, Cc)).astype(np.float32)
    sc, n = _scratch(lib)
    out = torch.full((P, Cc), 7.0, device="cuda")
    assert lib.uu3d_op_colsum(_p(_d(x)), Cc, nb * P, Cc, P, None, 0, _p(out), 0, _p(sc), n, None) == 0
    _close(out.cpu().numpy(), x.astype(np.float64).reshape(nb, P, Cc).sum(0), 2e-6)


@pytest.mark.parametrize("M,D", [(9088, 384), (5000, 32), (3, 384)])
def test_layernorm_backward(lib, M, D):
    rng = np.random.default_rng(3)
    x = (rng.normal(size=(M, D)) * 2 + 0.5).astype(np.float32); dy = rng.normal(size=(M, D)).astype(np.float32)
    g = (1 + 0.1 * rng.normal(size=D)).astype(np.float32); b = (0.1 * rng.normal(size=D)).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True); gt = torch.tensor(g, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.layer_norm(xt, (D,), gt, bt, 1e-5)
    y.backward(torch.tensor(dy, dtype=torch.float64))
    xd, dyd = _d(x), _d(dy)
    stats = torch.empty(M, 2, device="cuda"); sc, n = _scratch(lib)
    assert lib.uu3d_op_row_stats(_p(xd), D, D, M, 1e-5, _p(stats), None) == 0
    dx = torch.full((M, D), 0.25, device="cuda"); dg = torch.empty(D, device="cuda"); db = torch.empty(D, device="cuda")
    assert lib.uu3d_op_ln_bwd(_p(xd), _p(dyd), _p(stats), _p(_d(g)), D, D, M, _p(dx), 1, _p(dg), _p(db), _p(sc), n, None) == 0
    _close(dx.cpu().numpy() - 0.25, xt.grad.numpy(), 5e-6)        # accumulate=1 adds to the existing gradient
    _close(dg.cpu().numpy(), gt.grad.numpy(), 5e-6)
    _close(db.cpu().numpy(), bt.grad.numpy(), 5e-6)


@pytest.mark.parametrize("B,L,H,dh,masked", [(6, 71, 8, 48, True), (5, 23, 8, 48, False), (4, 3, 8, 48, False),
                                             (40, 17, 8, 4, False), (3, 96, 8, 48, True)])
def test_attention_forward_backward(lib, B, L, H, dh, masked):
    rng = np.random.default_rng(4)
    D = H * dh
    qkv = rng.normal(size=(B * L, 3 * D)).astype(np.float32); dout = rng.normal(size=(B * L, D)).astype(np.float32)
    m = (rng.random((B, L)) < 0.5); m[0] = False                   # one all-masked sequence
    t = torch.tensor(qkv, dtype=torch.float32, requires_grad=True)
    q, k, v = [t[:, i * D:(i + 1) * D].reshape(B, L, H, dh).permute(0, 2, 1, 3) for i in range(3)]
    logits = torch.matmul(q, k.transpose(-1, -2)) / torch.sqrt(torch.tensor(float(dh)))
    if masked:
        logits = logits + (1.0 - torch.tensor(m.astype(np.float32)))[:, None, None, :] * -1e9
    o = torch.matmul(torch.softmax(logits, -1), v).permute(0, 2, 1, 3).reshape(B * L, D)
    o.backward(torch.tensor(dout))
    md = _d(m.astype(np.uint8)) if masked else None
    out = torch.empty(B * L, D, device="cuda"); dqkv = torch.empty(B * L, 3 * D, device="cuda")
    qd = _d(qkv)
    assert lib.uu3d_op_attn_fwd(_p(qd), 3 * D, D, B, L, H, dh, _p(md), _p(out), D, None) == 0
    assert lib.uu3d_op_attn_bwd(_p(qd), _p(_d(dout)), 3 * D, D, B, L, H, dh, _p(md), _p(dqkv), D, None) == 0
    assert lib.uu3d_op_attn_bwd(_p(qd), _p(_d(dout)), 3 * D, D, B, 128, H, dh, _p(md), _p(dqkv), D, None) == 2   # unsupported
    _close(out.cpu().numpy(), o.detach().numpy(), 5e-6)
    _close(dqkv.cpu().numpy(), t.grad.numpy(), 2e-5)
