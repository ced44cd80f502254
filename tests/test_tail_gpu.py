"""GPU: the XCD-cooperative tail kernel (csrc/uu3d_tail.h: the last StridedTransformerBlock + strided_temporal_fc of
common/net/uplift_upsample_transformer.py:93-160,414-416 as one launch) against the CPU oracle, against the launch chain it
replaces (the default), and its own protocol diagnostics: no bounded spin may give up and no workgroup may ever observe
data stamped by a foreign XCC id."""
import numpy as np
import pytest

import uplift_upsample_3dhpe_amd as pkg
from tests import util

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _model(cfg, w, tail=True, **kw):
    """The tail kernel is opt-in (UU3D_TAIL=1 when the model is created; DESIGN.md: it ties with the launch chain)."""
    import os
    old = os.environ.get("UU3D_TAIL")
    os.environ["UU3D_TAIL"] = "1" if tail else "0"
    try:
        return pkg.build_uplift_upsample_transformer(cfg, weights=w, precision="f16x3", **kw)
    finally:
        if old is None:
            del os.environ["UU3D_TAIL"]
        else:
            os.environ["UU3D_TAIL"] = old


def _forward(model, x, m):
    xm = x * m[:, :, None, None].astype(np.float32)
    full, central = model([torch.from_numpy(xm).cuda(), torch.from_numpy(m).cuda()], training=False)
    torch.cuda.synchronize()
    return full.cpu().numpy(), central.cpu().numpy(), xm


# ragged groups (batch not a multiple of 8), a single sequence, fewer sequences than groups, the bench batches
@pytest.mark.parametrize("cfgname,batch", [("h36m_351", 1), ("h36m_351", 3), ("h36m_351", 8), ("h36m_351", 21), ("h36m_351", 128),
                                           ("h36m_81", 5), ("h36m_81", 19), ("h36m_81", 256)])
def test_tail_matches_oracle_and_chain(cfgname, batch, monkeypatch):
    from oracle import uplift_oracle as O
    cfg = util.load_config(cfgname)
    arch = pkg.arch_from_config(cfg)
    w = pkg.init_weights(arch, seed=5, perturb=0.1)
    x, m = util.synthetic_batch(cfg, batch=batch, seed=batch)
    model = _model(cfg, w)
    full, central, xm = _forward(model, x, m)
    st = model.tail_status(batch)
    print(cfgname, batch, st)
    assert st["err"] == 0, f"tail kernel protocol error word {st['err']:#x} (1: spin timeout, 2: foreign XCC id)"
    groups = min(8, batch)                       # ceil(B / ceil(B / 8)) non-empty groups, at most 8
    per = -(-batch // 8)
    groups = -(-batch // per)
    assert all(o != 0 for o in st["owner"][:groups]), "a non-empty sequence group was never claimed: the tail kernel did not run"
    assert all(o == 0 for o in st["owner"][groups:])
    assert sum(st["census"]) > 0
    # every owner word names an XCD that really had workgroups
    for o in st["owner"][:groups]:
        assert st["census"][o - 1] > 0
    chain = _model(cfg, w, tail=False)
    full_c, central_c, _ = _forward(chain, x, m)
    for mdl, want_tail in ((model, True), (chain, False)):     # which launches ran: per-launch profile of one more forward
        mdl.set_profiling(True)
        _forward(mdl, x, m)
        names = [e["kernel"] for e in mdl.read_profile()]
        mdl.set_profiling(False)
        assert ("strided_tail" in names) == want_tail, names
    n_oracle = min(batch, 12)                    # the oracle on the first sequences (sequences are independent)
    f32, c32 = O.forward(util.hp_from_arch(arch), w, xm[:n_oracle], m[:n_oracle], torch.float32)
    err = np.abs(central[:n_oracle] - c32).max()
    err_c = np.abs(central_c[:n_oracle] - c32).max()
    dev = np.abs(central - central_c).max()
    print(f"{cfgname} batch {batch}: tail vs oracle {err:.3e}, chain vs oracle {err_c:.3e}, tail vs chain {dev:.3e}")
    assert np.isfinite(central).all()
    assert err <= util.TOL_MAX_ABS and err_c <= util.TOL_MAX_ABS
    bad = np.nonzero(np.abs(central - central_c).reshape(batch, -1).max(axis=1) > 5e-5)[0]
    assert dev <= 5e-5, f"sequences that differ from the launch chain: {bad.tolist()}"
    assert np.array_equal(full, full_c)          # head1 / the temporal stack are untouched by the switch
    # run-to-run bitwise: ticket order and XCD placement must not matter (every sum has a fixed order)
    for _ in range(5):
        _, central2, _ = _forward(model, x, m)
        assert np.array_equal(central, central2)
        assert model.tail_status(batch)["err"] == 0


def test_tail_under_concurrent_streams():
    """Two forwards on two streams at once (concurrent_halves): both tail kernels are in flight together, each workgroup
    still only joins groups of its own XCD; results equal the single-chain run bit for bit."""
    cfg = util.load_config("h36m_351")
    arch = pkg.arch_from_config(cfg)
    w = pkg.init_weights(arch, seed=7, perturb=0.1)
    x, m = util.synthetic_batch(cfg, batch=128, seed=11)
    one = _model(cfg, w)
    two = _model(cfg, w, concurrent_halves=True)
    _, c1, _ = _forward(one, x, m)
    for _ in range(10):
        _, c2, _ = _forward(two, x, m)
        assert two.tail_status(64, 0)["err"] == 0 and two.tail_status(64, 1)["err"] == 0
        # the halves see groups of 8 sequences instead of 16: same arithmetic per sequence, same bits
        assert np.abs(c1 - c2).max() <= 2e-5


def test_tail_soak_bitwise():
    """500 forwards of the bench batch: every central pose bit-identical to the first run, error word 0 every time (the hand-offs
    between the phases are races by nature: tickets, done counters, sc1 loads -- a rare stale read shows up here)."""
    cfg = util.load_config("h36m_351")
    arch = pkg.arch_from_config(cfg)
    w = pkg.init_weights(arch, seed=2, perturb=0.1)
    x, m = util.synthetic_batch(cfg, batch=128, seed=4)
    model = _model(cfg, w)
    xm = torch.from_numpy(x * m[:, :, None, None].astype(np.float32)).cuda()
    mt = torch.from_numpy(m).cuda()
    _, ref = model([xm, mt], training=False)
    ref = ref.clone()
    bad = 0
    for i in range(500):
        _, c = model([xm, mt], training=False)
        bad += int(not torch.equal(c, ref))
        if i % 50 == 49:
            assert model.tail_status(128)["err"] == 0
    assert bad == 0, f"{bad} of 500 forwards differ from the first one"
