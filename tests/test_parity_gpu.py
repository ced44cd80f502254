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


@pytest.mark.parametrize("cfgname,batch", [("h36m_351", 17), ("h36m_81", 27)])
def test_row_panel_gemm_path_matches_oracle(cfgname, batch, monkeypatch):
    """B * N >= 1024 token rows: the LayerNorm-fed Dense layers run on the row-panel GEMM
    (csrc/uu3d_gemm_panel.h: ln_split_frag_kernel + gemm_h3_panel_kernel).  17 * 71 = 1207 and
    27 * 41 = 1107 rows are not multiples of 32 (ragged last panel) nor of 128 (idle waves in the
    last workgroup).  Checked against the oracle and against the tiled kernels (UU3D_NO_PANEL=1)."""
    from oracle import uplift_oracle as O
    cfg = util.load_config(cfgname)
    arch = pkg.arch_from_config(cfg)
    w = pkg.init_weights(arch, seed=3, perturb=0.1)
    x, m = util.synthetic_batch(cfg, batch=batch, seed=3)
    full, central, xm = _run_hip(cfg, w, x, m, "f16x3")          # the product path: ln_split_frag + panel GEMM for the LayerNorm-fed Dense layers
    monkeypatch.setenv("UU3D_NO_PANEL", "1")
    full_t, central_t, _ = _run_hip(cfg, w, x, m, "f16x3")
    monkeypatch.delenv("UU3D_NO_PANEL")
    hp = util.hp_from_arch(arch)
    f32, c32 = O.forward(hp, w, xm, m, torch.float32)
    err = max(np.abs(full - f32).max(), np.abs(central - c32).max())
    err_t = max(np.abs(full_t - f32).max(), np.abs(central_t - c32).max())
    dev = max(np.abs(full - full_t).max(), np.abs(central - central_t).max())
    print(f"{cfgname} batch {batch}: panel vs oracle {err:.3e}, tiled vs oracle {err_t:.3e}, panel vs tiled {dev:.3e}")
    assert err_t <= util.TOL_MAX_ABS
    assert np.isfinite(full).all() and np.isfinite(central).all()
    assert err <= util.TOL_MAX_ABS
    assert dev > 0.0, "UU3D_NO_PANEL=1 did not change the path: the panel kernels were not exercised"
    full2, central2, _ = _run_hip(cfg, w, x, m, "f16x3")
    assert np.array_equal(full, full2) and np.array_equal(central, central2)      # run-to-run bitwise


@pytest.mark.parametrize("n_frames,strides,batch", [(75, [3, 5, 5], 6), (63, [3, 3, 7], 8), (80, [4, 4, 5], 5), (49, [3, 4, 4], 8), (63, [3, 3, 7], 6), (75, [3, 5, 5], 8)])
def test_wave_per_head_attention_on_other_lengths(n_frames, strides, batch, monkeypatch):
    """attn_head_wave_kernel (uu3d_attn.h) serves every sequence of 4-5 query tiles (49..80 tokens); the shipped configs only
    have 71.  Other lengths (ragged and full last tiles, 4 and 5 tiles, grids that are and are not a multiple of 8 workgroups)
    against the oracle and against the workgroup-per-item kernel (UU3D_ATTN_WG=1), masked block included.  (This test found a
    GPU memory fault of the 4-tile instantiation: an SGPR hazard behind inline asm, see the kernel.)"""
    from oracle import uplift_oracle as O
    cfg = util.load_config("h36m_351")
    cfg.SEQUENCE_LENGTH = n_frames
    cfg.STRIDES = list(strides)
    cfg.PADDINGS = [[0, 0]] * 3
    arch = pkg.arch_from_config(cfg)
    w = pkg.init_weights(arch, seed=5, perturb=0.1)
    x, m = util.synthetic_batch(cfg, batch=batch, seed=5)
    full, central, xm = _run_hip(cfg, w, x, m, "f16x3")
    f32, c32 = O.forward(util.hp_from_arch(arch), w, xm, m, torch.float32)
    err = max(np.abs(full - f32).max(), np.abs(central - c32).max())
    monkeypatch.setenv("UU3D_ATTN_WG", "1")
    full_wg, central_wg, _ = _run_hip(cfg, w, x, m, "f16x3")
    dev = max(np.abs(full - full_wg).max(), np.abs(central - central_wg).max())
    print(f"N={n_frames} strides {strides} batch {batch}: max-abs vs oracle f32 {err:.3e}, vs the workgroup-per-item kernel {dev:.3e}")
    assert np.isfinite(full).all() and err <= util.TOL_MAX_ABS
    assert dev <= 2e-5


@pytest.mark.parametrize("n_frames,strides,batch", [(351, [3, 9, 13], 2), (117, [3, 3, 13], 3), (200, [5, 8, 5], 2), (416, [4, 8, 13], 1)])
def test_long_sequences_match_oracle(n_frames, strides, batch, monkeypatch):
    """Sequences beyond the 128 tokens the exact-f32 attention kernels hold: SURVEY 8(d)'s "synthetic dense-351"
    (351 -> 117 -> 13 -> 1, not a shipped config), other lengths, and the largest supported one (416 tokens = 13 key tiles),
    on attn_h3_kernel (f16x3 products, online softmax over 32-key tiles, uu3d_attn_h3.h).  Stride masks with masked and
    all-masked rows; the masked first temporal block included."""
    from oracle import uplift_oracle as O
    cfg = util.load_config("dense_351")
    cfg.SEQUENCE_LENGTH = n_frames
    cfg.STRIDES = list(strides)
    arch = pkg.arch_from_config(cfg)
    w = pkg.init_weights(arch, seed=6, perturb=0.1)
    x, m = util.synthetic_batch(cfg, batch=batch, seed=6)
    full, central, xm = _run_hip(cfg, w, x, m, "f16x3")
    f32, c32 = O.forward(util.hp_from_arch(arch), w, xm, m, torch.float32)
    err = max(np.abs(full - f32).max(), np.abs(central - c32).max())
    print(f"N={n_frames} strides {strides} batch {batch}: max-abs vs oracle f32 {err:.3e}")
    assert np.isfinite(full).all() and np.isfinite(central).all()
    assert err <= util.TOL_MAX_ABS
    full2, central2, _ = _run_hip(cfg, w, x, m, "f16x3")
    assert np.array_equal(full, full2) and np.array_equal(central, central2)


def test_sequence_length_limits():
    """417 tokens do not fit the attention kernel's LDS image; precision f32 keeps the 128-token limit of the exact-f32 kernels."""
    from uplift_upsample_3dhpe_amd import _capi
    cfg = util.load_config("dense_351")
    cfg.SEQUENCE_LENGTH, cfg.STRIDES = 417, [4, 8, 13]
    with pytest.raises(_capi.Uu3dError, match="416"):
        pkg.build_uplift_upsample_transformer(cfg)
    cfg = util.load_config("dense_351")
    with pytest.raises(_capi.Uu3dError, match="128"):
        pkg.build_uplift_upsample_transformer(cfg, precision="f32")


def test_f16x3_attention_agrees_with_the_f32_kernels(monkeypatch):
    """h36m_351 (71 tokens, masked block included): attn_h3_kernel vs the exact-f32 wave-per-head kernel (UU3D_ATTN_F32=1)."""
    cfg = util.load_config("h36m_351")
    arch = pkg.arch_from_config(cfg)
    w = pkg.init_weights(arch, seed=8, perturb=0.1)
    x, m = util.synthetic_batch(cfg, batch=12, seed=8)
    full, central, _ = _run_hip(cfg, w, x, m, "f16x3")
    monkeypatch.setenv("UU3D_ATTN_F32", "1")
    full_f, central_f, _ = _run_hip(cfg, w, x, m, "f16x3")
    dev = max(np.abs(full - full_f).max(), np.abs(central - central_f).max())
    print(f"f16x3 attention vs exact-f32 attention: {dev:.3e}")
    assert 0.0 < dev <= 3e-5


@pytest.mark.parametrize("cfgname,batch", [("h36m_81", 5), ("h36m_351", 17)])
def test_output_bn_heads_match_oracle(cfgname, batch, tmp_path):
    """OUTPUT_BN = true (BatchNormalization in front of temporal_fc / strided_temporal_fc, u_u_t.py:275-285) at inference: the
    library folds the per-channel affine of the moving statistics into the head operands at commit time.  Against the oracle's
    explicit BatchNorm; a checkpoint with the 8 extra tensors round-trips through the Keras .h5 walk."""
    from oracle import uplift_oracle as O
    cfg = util.load_config(cfgname)
    cfg.OUTPUT_BN = True
    arch = pkg.arch_from_config(cfg)
    w = pkg.init_weights(arch, seed=4, perturb=0.2)
    x, m = util.synthetic_batch(cfg, batch=batch, seed=4)
    full, central, xm = _run_hip(cfg, w, x, m, "f16x3")
    n = min(batch, 6)
    f32, c32 = O.forward(util.hp_from_arch(arch), w, xm[:n], m[:n], torch.float32)
    err = max(np.abs(full[:n] - f32).max(), np.abs(central[:n] - c32).max())
    print(f"{cfgname} OUTPUT_BN: max-abs vs oracle {err:.3e}")
    assert err <= util.TOL_MAX_ABS
    model = pkg.build_uplift_upsample_transformer(cfg, weights=w)
    assert len(model.weights) == len(model.trainable_variables) + 4
    path = str(tmp_path / "bn.h5")
    model.save_weights(path)
    other = pkg.build_uplift_upsample_transformer(cfg, weights=pkg.init_weights(arch, seed=9))
    other.load_weights(path)
    xt, mt = torch.from_numpy(xm).cuda(), torch.from_numpy(m).cuda()
    a, b = model([xt, mt], training=False), other([xt, mt], training=False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    # (the TRAINING form -- batch statistics, moving-average update -- exists since late round 3: tests/test_train_step_gpu.py)
    ft, ct = model([xt, mt], training=True)
    assert torch.isfinite(ft).all() and torch.isfinite(ct).all() and (ft - a[0]).abs().max() > 1e-4


@pytest.mark.parametrize("variant", ["no_temporal_blocks", "no_strided_blocks", "neither", "no_temporal_blocks_no_mask"])
def test_structural_variants_match_oracle(variant):
    """TEMPORAL_TRANSFORMER_BLOCKS = 0 and / or STRIDES = [] (u_u_t.py:356,372-380,411-413; no shipped config uses them): without
    temporal blocks the FIRST strided block takes the key mask and there is no full-sequence head; without strided blocks the central
    token x[:, N // 2] feeds strided_temporal_fc.  Against the oracle (fp32 and, where no row is all-masked, float64)."""
    from oracle import uplift_oracle as O
    cfg = util.load_config("h36m_81")
    if variant in ("no_temporal_blocks", "neither", "no_temporal_blocks_no_mask"):
        cfg.TEMPORAL_TRANSFORMER_BLOCKS = 0
    if variant in ("no_strided_blocks", "neither"):
        cfg.STRIDES, cfg.PADDINGS = [], []
    if variant == "no_temporal_blocks_no_mask":
        cfg.MASK_STRIDE = None
    arch = pkg.arch_from_config(cfg)
    assert arch.temporal_depth == (0 if "temporal" in variant or variant == "neither" else 4)
    w = pkg.init_weights(arch, seed=6, perturb=0.1)
    for batch in (3, 40):                       # 40 x 41 = 1640 token rows: the row-panel / fused-MLP path where temporal blocks exist
        if arch.has_strided_input:
            x, m = util.synthetic_batch(cfg, batch=batch, seed=6)
        else:
            x, m = np.random.default_rng(6).uniform(-1, 1, size=(batch, arch.num_frames, 17, 2)).astype(np.float32), None
        model = pkg.build_uplift_upsample_transformer(cfg, weights=w)
        n = min(batch, 6)
        if arch.has_strided_input:
            xin = x * m[:, :, None, None].astype(np.float32)
            full, central = model([torch.from_numpy(xin).cuda(), torch.from_numpy(m).cuda()], training=False)
            f32, c32 = O.forward(util.hp_from_arch(arch), w, xin[:n], m[:n], torch.float32)
        else:
            xin = x
            full, central = model(torch.from_numpy(xin).cuda(), training=False)
            f32, c32 = O.forward(util.hp_from_arch(arch), w, xin[:n], None, torch.float32)
        torch.cuda.synchronize()
        assert (full is None) == (f32 is None) == (arch.temporal_depth == 0)
        err = np.abs(central.cpu().numpy()[:n] - c32).max()
        if full is not None:
            err = max(err, np.abs(full.cpu().numpy()[:n] - f32).max())
        print(f"{variant} batch {batch}: max-abs vs oracle {err:.3e}")
        assert np.isfinite(central.cpu().numpy()).all() and err <= util.TOL_MAX_ABS


@pytest.mark.parametrize("cfgname,batch", [("h36m_81", 3), ("h36m_351", 17)])
def test_return_attention_matches_oracle(cfgname, batch):
    """return_attention=True (u_u_t.py:176,365,418-419): (full, central, att_list) with the softmax weights of every temporal block,
    the masked block's keys included (probability exactly 0 for masked keys of a row with at least one valid key)."""
    from oracle import uplift_oracle as O
    cfg = util.load_config(cfgname)
    arch = pkg.arch_from_config(cfg)
    w = pkg.init_weights(arch, seed=2, perturb=0.1)
    x, m = util.synthetic_batch(cfg, batch=batch, seed=2)
    xm = x * m[:, :, None, None].astype(np.float32)
    model = pkg.build_uplift_upsample_transformer(cfg, weights=w, return_attention=True)
    full, central, att = model([torch.from_numpy(xm).cuda(), torch.from_numpy(m).cuda()], training=False)
    torch.cuda.synchronize()
    plain = pkg.build_uplift_upsample_transformer(cfg, weights=w)
    f0, c0 = plain([torch.from_numpy(xm).cuda(), torch.from_numpy(m).cuda()], training=False)
    assert torch.equal(full, f0) and torch.equal(central, c0)            # the maps are a side output
    n = min(batch, 4)
    f32, c32, a32 = O.forward(util.hp_from_arch(arch), w, xm[:n], m[:n], torch.float32, return_attention=True)
    assert len(att) == len(a32) == arch.temporal_depth
    for i, (got, want) in enumerate(zip(att, a32)):
        g = got.cpu().numpy()
        assert g.shape == (batch, arch.num_heads, arch.num_frames, arch.num_frames)
        assert np.abs(g.sum(-1) - 1.0).max() <= 1e-5
        err = np.abs(g[:n] - want).max()
        print(f"{cfgname} temporal block {i + 1}: attention maps max-abs vs oracle {err:.2e}")
        assert err <= 2e-5
    rows = m.any(axis=1)
    masked_keys = att[0].cpu().numpy()[rows][:, :, :, :] * (~m[rows])[:, None, None, :]
    assert masked_keys.max() == 0.0
