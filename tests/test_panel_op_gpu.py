"""The forward's row-panel path for a LayerNorm-fed Dense layer (csrc/uu3d_gemm_panel.h: ln_split_frag_kernel +
gemm_h3_panel_kernel) on its own through the C ABI (include/uu3d_ops.h), against a float64 restatement of
kl.LayerNormalization + kl.Dense (vision_transformer.py:46-68,135-137,183,188) -- at the ragged sizes the model never
produces by itself (fewer rows than a 32-row panel, one row past a panel / a 128-row tile, a single row) as well as
the model's own.  Tolerance 2e-5 max-abs on outputs of unit scale (the whole model is held to 1e-4)."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

K = 384


@pytest.fixture(scope="module")
def lib():
    from uplift_upsample_3dhpe_amd import _capi
    return _capi.load_library()


def _p(t):
    return C.c_void_p(t.data_ptr())


def _reference(x, g, b, eps, w, bias):
    x = x.astype(np.float64)
    mean = x.mean(-1, keepdims=True)
    var = ((x - mean) ** 2).mean(-1, keepdims=True)
    y = (x - mean) / np.sqrt(var + eps) * g.astype(np.float64) + b.astype(np.float64)
    return y @ w.astype(np.float64) + bias.astype(np.float64)


@pytest.mark.parametrize("relu", [0, 1])
@pytest.mark.parametrize("M,N", [(1, 384), (31, 768), (32, 1152), (33, 768), (127, 1152), (128, 768), (129, 1152),
                                 (300, 32), (1207, 1152), (4544, 768), (9088, 1152)])
def test_ln_dense_panel(lib, M, N, relu):
    rng = np.random.default_rng(M * 7 + N)
    x = (rng.normal(size=(M, 1)) + (0.5 + np.abs(rng.normal(size=(M, 1)))) * rng.normal(size=(M, K))).astype(np.float32)
    g = (1.0 + 0.1 * rng.normal(size=K)).astype(np.float32)
    b = (0.1 * rng.normal(size=K)).astype(np.float32)
    w = (0.05 * rng.normal(size=(K, N))).astype(np.float32)          # Keras (in, out)
    bias = (0.1 * rng.normal(size=N)).astype(np.float32)
    if M > 3:
        x[3, :] *= 1e-6                                               # a row of tiny values: f16 denormal range of the split
    ref = _reference(x, g, b, 1e-5, w, bias)

    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    xd, gd, bd, biasd = dev(x), dev(g), dev(b), dev(bias)
    operand = torch.empty(int(lib.uu3d_op_panel_operand_bytes(N)), dtype=torch.uint8, device="cuda")
    w_host = np.ascontiguousarray(w)
    assert lib.uu3d_op_panel_pack(w_host.ctypes.data_as(C.c_void_p), N, _p(operand), None) == 0
    scratch = torch.empty(int(lib.uu3d_op_panel_a_bytes(M)), dtype=torch.uint8, device="cuda")
    if relu:
        out = torch.full((2, M, N), float("nan"), dtype=torch.float16, device="cuda")
    else:
        out = torch.full((M + 1, N), float("nan"), dtype=torch.float32, device="cuda")      # one guard row: nothing past M may be written
    st = lib.uu3d_op_ln_dense_panel(_p(xd), K, M, _p(gd), _p(bd), 1e-5, _p(operand), _p(biasd), N, relu, _p(scratch), _p(out), N, None)
    assert st == 0
    torch.cuda.synchronize()
    if relu:
        o = out.cpu().numpy().astype(np.float64)
        got = o[0] + o[1] / 2048.0
        want = np.maximum(ref, 0.0)
    else:
        o = out.cpu().numpy()
        assert np.isnan(o[M]).all(), "rows past M were written"
        got, want = o[:M].astype(np.float64), ref
    assert np.isfinite(got).all()
    err = np.abs(got - want).max()
    print(f"M={M} N={N} relu={relu}: max-abs {err:.3e} (scale {np.abs(want).max():.2f})")
    assert err <= 2e-5
    # run-to-run bitwise
    out2 = torch.empty_like(out)
    assert lib.uu3d_op_ln_dense_panel(_p(xd), K, M, _p(gd), _p(bd), 1e-5, _p(operand), _p(biasd), N, relu, _p(scratch), _p(out2), N, None) == 0
    torch.cuda.synchronize()
    a, c = out.cpu().numpy(), out2.cpu().numpy()
    if relu:
        assert np.array_equal(a, c)
    else:
        assert np.array_equal(a[:M], c[:M])
