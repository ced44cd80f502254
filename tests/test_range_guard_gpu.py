"""GPU: the range guard of precision f16x3 (include/uu3d.h, RANGE CONTRACT; round-4 verdict, weak point 9).  The reference computes in
float32 end to end (SURVEY section 8 header); the f16x3 products split operands into f16 planes, so an activation of magnitude >= 65504
turns into Inf / NaN.  With weights scaled so that the hidden activations of a temporal block exceed 7e4 the model must either match
the oracle (the exact-f32 fallback of model(...)) or fail loudly -- never return NaN with status 0."""
import warnings

import numpy as np
import pytest

import uplift_upsample_3dhpe_amd as pkg
from uplift_upsample_3dhpe_amd import _capi
from tests import util

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _overflowing_weights(arch, scale=3.0e4, block="temporal_block_2"):
    """fc1 of one block times `scale`, its fc2 divided by it: the same function in exact arithmetic, hidden activations `scale` times larger."""
    w = dict(pkg.init_weights(arch, seed=2, perturb=0.1))
    w[block + "/mlp/fc1/kernel"] = w[block + "/mlp/fc1/kernel"] * np.float32(scale)
    w[block + "/mlp/fc1/bias"] = w[block + "/mlp/fc1/bias"] * np.float32(scale)
    w[block + "/mlp/fc2/kernel"] = w[block + "/mlp/fc2/kernel"] / np.float32(scale)
    return w


@pytest.mark.parametrize("cfgname,batch", [("h36m_351", 6), ("h36m_351", 20), ("h36m_81", 30)])
def test_overflowing_activations_fall_back_to_exact_f32(cfgname, batch):
    from oracle import uplift_oracle as O
    cfg = util.load_config(cfgname)
    arch = pkg.arch_from_config(cfg)
    w = _overflowing_weights(arch)
    x, m = util.synthetic_batch(cfg, batch=batch, seed=3)
    xm = x * m[:, :, None, None].astype(np.float32)
    xt, mt = torch.from_numpy(xm).cuda(), torch.from_numpy(m).cuda()
    n = min(batch, 6)
    f32, c32 = O.forward(util.hp_from_arch(arch), w, xm[:n], m[:n], torch.float32)
    # the premise: the hidden activations of that block really leave the f16 range (else the test tests nothing)
    raw = pkg.build_uplift_upsample_transformer(cfg, weights=w, range_guard=False)
    fr, cr = raw([xt, mt], training=False)
    torch.cuda.synchronize()
    assert not bool(torch.isfinite(cr).all()), "the scaled weights did not overflow the f16x3 path: raise the scale"
    with pytest.raises(_capi.Uu3dRangeError):
        raw.check_range()                                          # ... and the unguarded model says so when asked
    assert raw.check_range(raise_error=False) is False             # the word is sticky until read, then clear
    # the guarded model (the default): same call, finite outputs that match the oracle, one warning
    model = pkg.build_uplift_upsample_transformer(cfg, weights=w)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        full, cen = model([xt, mt], training=False)
        torch.cuda.synchronize()
    assert any("f16 range" in str(r.message) for r in rec)
    full, cen = full.cpu().numpy(), cen.cpu().numpy()
    assert np.isfinite(full).all() and np.isfinite(cen).all()
    scale = max(np.abs(f32).max(), np.abs(c32).max())
    err = max(np.abs(full[:n] - f32).max(), np.abs(cen[:n] - c32).max())
    print(f"{cfgname} batch {batch}: exact-f32 fallback vs oracle {err:.3e} (outputs up to {scale:.1f})")
    assert err <= 1e-4 * max(1.0, scale)
    assert model.check_range(raise_error=False) is False           # the fallback leaves no stale flag behind
    # a pipeline does not check per batch; its check_range() does, once
    pipe = model.pipeline(batch, depth=2)
    for _ in range(3):
        pipe.result(pipe.submit(xt, mt))
    with pytest.raises(_capi.Uu3dRangeError):
        pipe.check_range()
    pipe.close()


def test_ordinary_weights_never_trip_the_guard_and_weights_are_checked_at_commit():
    cfg = util.load_config("h36m_351")
    arch = pkg.arch_from_config(cfg)
    w = pkg.init_weights(arch, seed=0, perturb=0.1)
    model = pkg.build_uplift_upsample_transformer(cfg, weights=w)
    x, m = util.synthetic_batch(cfg, batch=12, seed=0)
    xt, mt = torch.from_numpy(x * m[:, :, None, None].astype(np.float32)).cuda(), torch.from_numpy(m).cuda()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        full, cen = model([xt, mt], training=False)
    assert model.check_range(raise_error=False) is False
    # non-finite INPUTS are reported the same way (the fallback cannot help: it raises)
    xbad = xt.clone(); xbad[3, int(np.nonzero(m[3])[0][0]), 2, 0] = float("nan")           # (a frame the stride mask keeps: masked frames are never read)
    with pytest.raises(_capi.Uu3dRangeError), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model([xbad, mt], training=False)
    # a kernel value the f16 planes cannot hold is refused when the weights are committed
    wbad = dict(w)
    k = wbad["temporal_block_1/mlp/fc1/kernel"].copy(); k[0, 0] = 7.0e4
    wbad["temporal_block_1/mlp/fc1/kernel"] = k
    with pytest.raises(_capi.Uu3dError) as ei:
        pkg.build_uplift_upsample_transformer(cfg, weights=wbad)
    assert ei.value.status == _capi.UU3D_ERR_RANGE
    pkg.build_uplift_upsample_transformer(cfg, weights=wbad, precision="f32")      # float32 holds it
