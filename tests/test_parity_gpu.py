"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on seeded inputs."""
import numpy as np
import pytest

import uplift_upsample_3dhpe_amd as pkg
from tests import util

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _run_hip(cfg, weights, x, m, precision):
    model = pkg.build_uplift_upsample_transformer(cfg, weights=weights, precision=precision)
    xm = x * m[:, :, None, None].astype(np.float32)          # caller zeroes masked frames (eval.py:67)
    xt = torch.from_numpy(xm).cuda()
    mt = torch.from_numpy(m).cuda()
    full, central = model([xt, mt], training=False)
    torch.cuda.synchronize()
    return full.cpu().numpy(), central.cpu().numpy(), xm


@pytest.mark.parametrize("cfgname", ["h36m_351", "h36m_81"])
@pytest.mark.parametrize("seed", [0, 1, 2])
@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_forward_matches_oracle(cfgname, seed, precision):
    from oracle import uplift_oracle as O
    cfg = util.load_config(cfgname)
    arch = pkg.arch_from_config(cfg)
    w = pkg.init_weights(arch, seed=seed, perturb=0.1)
    x, m = util.synthetic_batch(cfg, batch=6, seed=seed)
    full, central, xm = _run_hip(cfg, w, x, m, precision)
    hp = util.hp_from_arch(arch)
    f32, c32 = O.forward(hp, w, xm, m, torch.float32)
    f64, c64 = O.forward(hp, w, xm, m, torch.float64)
    assert np.isfinite(full).all() and np.isfinite(central).all()
    err32 = max(np.abs(full - f32).max(), np.abs(central - c32).max())
    # The float64 twin only bounds rounding where it is meaningful: for an ALL-masked row the
    # reference's fp32 `logits + mask * -1e9` rounds every logit of temporal block 1 to exactly
    # -1e9 (uniform attention), which float64 does not reproduce.  fp32 is the reference.
    rows = m.any(axis=1)
    err64 = max(np.abs(full - f64)[rows].max(), np.abs(central - c64)[rows].max())
    print(f"{cfgname} seed {seed} {precision}: max-abs vs oracle f32 {err32:.3e}, vs f64 {err64:.3e}")
    assert err32 <= util.TOL_MAX_ABS
    assert err64 <= util.TOL_MAX_ABS
