"""GPU: the temporal chain (csrc/uu3d_tchain.h; the throughput schedule's path from 1024 token rows on, UU3D_TCHAIN=0 switches it off) -- every row-local stage of a
vit.TransformerBlock (common/net/vision_transformer.py:176-195: projection + residual, LayerNorm 2, fc1, ReLU, fc2 + residual, the next
block's LayerNorm 1 + QKV) as ONE launch per block -- against the CPU oracle, against the launch chain it replaces, run to run, with
whole and ragged row tiles, with and without key masks, with return_attention, and for the structural variants of the constructor
(no strided blocks, one temporal block)."""
import os

import numpy as np
import pytest

import uplift_upsample_3dhpe_amd as pkg
from tests import util

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _model(cfg, w, chain=True, **kw):
    old = os.environ.get("UU3D_TCHAIN")
    os.environ["UU3D_TCHAIN"] = "1" if chain else "0"
    try:
        return pkg.build_uplift_upsample_transformer(cfg, weights=w, precision="f16x3", **kw)
    finally:
        if old is None:
            del os.environ["UU3D_TCHAIN"]
        else:
            os.environ["UU3D_TCHAIN"] = old


def _forward(model, arch, xm, m, schedule, attn=False):
    xt, mt = torch.from_numpy(xm).cuda(), torch.from_numpy(m).cuda()
    B = xt.shape[0]
    full = torch.empty((B, arch.num_frames, arch.num_keypoints, 3), dtype=torch.float32, device="cuda") if model.full_output else None
    cen = torch.empty((B, arch.num_keypoints, 3), dtype=torch.float32, device="cuda")
    maps = None
    if attn:
        maps = [torch.empty((B, arch.num_heads, arch.num_frames, arch.num_frames), dtype=torch.float32, device="cuda") for _ in range(arch.temporal_depth)]
    model._forward(xt, model._mask_u8(mt) if model.has_strided_input else None, full, cen, 0, torch.cuda.current_stream(), attn=maps, schedule=schedule)
    torch.cuda.synchronize()
    return (full.cpu().numpy() if full is not None else None), cen.cpu().numpy(), ([a.cpu().numpy() for a in maps] if attn else None)


def _kernels(model, arch, xm, m, schedule):
    model.set_profiling(True)
    _forward(model, arch, xm, m, schedule)
    names = [e["kernel"] for e in model.read_profile()]
    model.set_profiling(False)
    return names


# 15 * 71 = 1065 rows: nine row tiles, the last with 41 live rows; 128 * 71 = 9088: 71 whole tiles (the benchmark batch); 33 * 71: 2343 rows
@pytest.mark.parametrize("batch,mask_specs", [(15, None), (33, [(5, 0)]), (128, None), (40, [(20, 0), (10, 5)])])
def test_chain_matches_oracle_and_the_launch_chain(batch, mask_specs):
    from oracle import uplift_oracle as O
    cfg = util.load_config("h36m_351")
    arch = pkg.arch_from_config(cfg)
    w = pkg.init_weights(arch, seed=3, perturb=0.1)
    x, m = util.synthetic_batch(cfg, batch=batch, seed=batch, mask_specs=mask_specs)
    xm = x * m[:, :, None, None].astype(np.float32)
    model = _model(cfg, w)
    full, cen, _ = _forward(model, arch, xm, m, 1)
    names = _kernels(model, arch, xm, m, 1)
    assert names.count("tchain") == arch.temporal_depth + 2, names                       # LN1 + QKV | one per temporal block | head of strided block 1
    assert not any(k.startswith("mlp_fused") or k.startswith("gemm_panel8<BiasSplitQ>") for k in names), names
    assert "tchain" not in _kernels(model, arch, xm, m, 0)                               # the latency schedule keeps its launches
    full_l, cen_l, _ = _forward(model, arch, xm, m, 0)
    plain = _model(cfg, w, chain=False)
    assert "tchain" not in _kernels(plain, arch, xm, m, 1)                               # UU3D_TCHAIN=0: the round-4 launches
    n = min(batch, 8)
    f32, c32 = O.forward(util.hp_from_arch(arch), w, xm[:n], m[:n], torch.float32)
    e = max(np.abs(full[:n] - f32).max(), np.abs(cen[:n] - c32).max())
    d = max(np.abs(full - full_l).max(), np.abs(cen - cen_l).max())
    print(f"batch {batch}: chain vs oracle {e:.3e}, chain vs launch chain {d:.3e}")
    assert np.isfinite(full).all() and np.isfinite(cen).all()
    assert e <= util.TOL_MAX_ABS
    assert d <= 3e-5                                                                     # (other summation orders, LayerNorm affine folded into the weights)
    for _ in range(3):                                                                   # fixed summation orders: bitwise run to run
        f2, c2, _ = _forward(model, arch, xm, m, 1)
        assert np.array_equal(f2, full) and np.array_equal(c2, cen)


def test_chain_sequence_independence_and_attention_maps():
    """A sequence's result does not depend on its neighbours in the row tile (lane-private arithmetic; dead lanes of a ragged tile read row
    M - 1); and return_attention=True keeps the round-4 launches (the maps are recomputed from row-major q | k planes, the chain writes
    q | k | v in fragment order): the maps of a model with the switch on equal the latency schedule's."""
    cfg = util.load_config("h36m_351")
    arch = pkg.arch_from_config(cfg)
    w = pkg.init_weights(arch, seed=4, perturb=0.1)
    x, m = util.synthetic_batch(cfg, batch=20, seed=11)
    xm = x * m[:, :, None, None].astype(np.float32)
    model = _model(cfg, w)
    full, cen, _ = _forward(model, arch, xm, m, 1)
    assert "tchain" in _kernels(model, arch, xm, m, 1)
    perm = np.random.default_rng(0).permutation(20)
    fp, cp, _ = _forward(model, arch, xm[perm], m[perm], 1)
    assert np.array_equal(fp, full[perm]) and np.array_equal(cp, cen[perm])
    f17, c17, _ = _forward(model, arch, xm[:17], m[:17], 1)                              # other tiles, another ragged tail
    assert np.array_equal(f17, full[:17])                                                # (everything the chain computes feeds `full`)
    assert np.abs(c17 - cen[:17]).max() <= 1e-5                                          # (the strided blocks' split-K depth depends on the row count)
    amodel = _model(cfg, w, return_attention=True)
    fa, ca, maps = _forward(amodel, arch, xm, m, 1, attn=True)
    fl, cl, maps_l = _forward(amodel, arch, xm, m, 0, attn=True)
    assert np.array_equal(fa, fl) and np.array_equal(ca, cl)                              # the same launches in both schedules
    for a, b in zip(maps, maps_l):
        assert np.array_equal(a, b) and np.abs(a.sum(-1) - 1.0).max() <= 1e-5


def test_chain_at_41_tokens():
    """config/h36m_81.json (41 tokens: below the 49 from which attn_h3_kernel is the attention kernel of the other paths): the chain's fragment-ordered q | k | v
    cost no split epilogue, so the throughput schedule takes the chain + attn_h3_kernel there too (UU3D_TCHAIN_SHORT=0: not)."""
    from oracle import uplift_oracle as O
    cfg = util.load_config("h36m_81")
    arch = pkg.arch_from_config(cfg)
    w = pkg.init_weights(arch, seed=3, perturb=0.1)
    x, m = util.synthetic_batch(cfg, batch=40, seed=40, mask_specs=[(20, 0), (10, 5)])
    xm = x * m[:, :, None, None].astype(np.float32)
    model = _model(cfg, w)
    full, cen, _ = _forward(model, arch, xm, m, 1)
    names = _kernels(model, arch, xm, m, 1)
    assert names.count("tchain") == arch.temporal_depth + 2 and names.count("attn_h3") == arch.temporal_depth + 1, names
    assert "tchain" not in _kernels(model, arch, xm, m, 0)
    old = os.environ.get("UU3D_TCHAIN_SHORT")
    os.environ["UU3D_TCHAIN_SHORT"] = "0"
    try:
        plain = _model(cfg, w)
    finally:
        if old is None:
            del os.environ["UU3D_TCHAIN_SHORT"]
        else:
            os.environ["UU3D_TCHAIN_SHORT"] = old
    assert "tchain" not in _kernels(plain, arch, xm, m, 1)
    full_l, cen_l, _ = _forward(model, arch, xm, m, 0)
    f32, c32 = O.forward(util.hp_from_arch(arch), w, xm[:8], m[:8], torch.float32)
    e = max(np.abs(full[:8] - f32).max(), np.abs(cen[:8] - c32).max())
    d = max(np.abs(full - full_l).max(), np.abs(cen - cen_l).max())
    print(f"h36m_81 batch 40: chain vs oracle {e:.3e}, vs launch chain {d:.3e}")
    assert e <= util.TOL_MAX_ABS and d <= 3e-5
    f2, c2, _ = _forward(model, arch, xm, m, 1)
    assert np.array_equal(f2, full) and np.array_equal(c2, cen)


@pytest.mark.parametrize("variant", ["no_strided", "one_temporal", "no_mask"])
def test_chain_structural_variants(variant):
    """The stage sets the constructor can ask for: no strided blocks (the last temporal block ends at the residual stream),
    a single temporal block, a model without strided input (no key mask, no token blend)."""
    from oracle import uplift_oracle as O
    cfg = util.load_config("h36m_351")
    if variant == "no_strided":
        cfg.STRIDES, cfg.PADDINGS = [], []
    elif variant == "one_temporal":
        cfg.TEMPORAL_TRANSFORMER_BLOCKS = 1
    else:
        cfg.MASK_STRIDE = None
    arch = pkg.arch_from_config(cfg)
    w = pkg.init_weights(arch, seed=6, perturb=0.1)
    rng = np.random.default_rng(5)
    B = 16
    x = rng.uniform(-1, 1, size=(B, arch.num_frames, arch.num_keypoints, 2)).astype(np.float32)
    m = np.ones((B, arch.num_frames), dtype=bool)
    if arch.has_strided_input:
        m[:, 1::2] = False
        m[3] = False
    xm = x * m[:, :, None, None].astype(np.float32)
    model = _model(cfg, w)
    full, cen, _ = _forward(model, arch, xm, m, 1)
    names = _kernels(model, arch, xm, m, 1)
    assert names.count("tchain") == arch.temporal_depth + 1 + (1 if len(arch.strides) > 0 else 0), names
    f32, c32 = O.forward(util.hp_from_arch(arch), w, xm[:6], m[:6] if arch.has_strided_input else None, torch.float32)
    e = np.abs(cen[:6] - c32).max()
    if full is not None:
        e = max(e, np.abs(full[:6] - f32).max())
    print(f"{variant}: chain vs oracle {e:.3e}")
    assert e <= util.TOL_MAX_ABS


def test_chain_is_the_default_of_the_throughput_schedule():
    """Default (no UU3D_TCHAIN in the environment): the chain runs under the throughput schedule from 1024 token rows on (8 row tiles: below
    that the tile's 128 rows are mostly dead lanes), at the benchmark's batch of 128 (71 tiles) and at the reference's own eval BATCH_SIZE
    of 512 windows alike; the latency schedule never takes it, UU3D_TCHAIN_MIN_TILES moves the threshold."""
    from oracle import uplift_oracle as O
    cfg = util.load_config("h36m_351")
    arch = pkg.arch_from_config(cfg)
    w = pkg.init_weights(arch, seed=3, perturb=0.1)
    old = os.environ.pop("UU3D_TCHAIN", None)
    try:
        model = pkg.build_uplift_upsample_transformer(cfg, weights=w)
        os.environ["UU3D_TCHAIN_MIN_TILES"] = "256"
        try:
            late = pkg.build_uplift_upsample_transformer(cfg, weights=w)
        finally:
            del os.environ["UU3D_TCHAIN_MIN_TILES"]
    finally:
        if old is not None:
            os.environ["UU3D_TCHAIN"] = old
    x, m = util.synthetic_batch(cfg, batch=464, seed=1)
    xm = x * m[:, :, None, None].astype(np.float32)
    assert _kernels(model, arch, xm, m, 1).count("tchain") == arch.temporal_depth + 2
    assert "tchain" not in _kernels(model, arch, xm, m, 0)
    assert _kernels(model, arch, xm[:128], m[:128], 1).count("tchain") == arch.temporal_depth + 2
    assert _kernels(model, arch, xm[:15], m[:15], 1).count("tchain") == arch.temporal_depth + 2      # 1065 rows
    assert "tchain" not in _kernels(model, arch, xm[:14], m[:14], 1)                                  # 994 rows
    assert "tchain" not in _kernels(late, arch, xm[:128], m[:128], 1)                                 # 71 tiles < 256
    assert _kernels(late, arch, xm, m, 1).count("tchain") == arch.temporal_depth + 2                  # 258 tiles
    full, cen, _ = _forward(model, arch, xm, m, 1)
    full_l, cen_l, _ = _forward(model, arch, xm, m, 0)
    idx = [0, 1, 127, 128, 300, 463]                                 # sequences from the first, a middle and the last (ragged) row tiles
    f32, c32 = O.forward(util.hp_from_arch(arch), w, xm[idx], m[idx], torch.float32)
    e = max(np.abs(full[idx] - f32).max(), np.abs(cen[idx] - c32).max())
    d = max(np.abs(full - full_l).max(), np.abs(cen - cen_l).max())
    print(f"batch 464: chain (default) vs oracle {e:.3e}, vs launch chain {d:.3e}")
    assert e <= util.TOL_MAX_ABS and d <= 3e-5


@pytest.mark.parametrize("cfgname,batch", [("h36m_81", 256), ("h36m_351", 512), ("h36m_351", 2048), ("h36m_81", 4096)])
def test_timed_workloads_match_oracle_at_their_batch(cfgname, batch):
    """The batches bench.py TIMES under the throughput schedule -- BASELINE configs[1] (config/h36m_81.json, batch 256: `secondary.h36m_81_batch256`)
    and the reference's own eval BATCH_SIZE of 512 windows (`secondary.eval_batch_512`) -- straight against the oracle on six sequences taken from
    the first, the middle and the last row tiles (metric: common/dataset/metrics.py:13-37 on what these forwards return).  The two large batches
    (2272 and 2624 row tiles of the chain, 145 k and 168 k token rows) are the size edge: every index of the path in 64-bit range, one forward = 9 / 10 waves
    of workgroups per chain launch."""
    from oracle import uplift_oracle as O
    cfg = util.load_config(cfgname)
    arch = pkg.arch_from_config(cfg)
    w = pkg.init_weights(arch, seed=3, perturb=0.1)
    x, m = util.synthetic_batch(cfg, batch=batch, seed=batch, mask_specs=None)                  # (the default cycle: keyframe-aligned, centre-masked and all-masked rows)
    xm = x * m[:, :, None, None].astype(np.float32)
    model = _model(cfg, w)
    full, cen, _ = _forward(model, arch, xm, m, 1)
    assert _kernels(model, arch, xm, m, 1).count("tchain") == arch.temporal_depth + 2
    idx = [0, 1, batch // 2 - 1, batch // 2, batch - 2, batch - 1]
    f32, c32 = O.forward(util.hp_from_arch(arch), w, xm[idx], m[idx], torch.float32)
    e = max(np.abs(full[idx] - f32).max(), np.abs(cen[idx] - c32).max())
    print(f"{cfgname} batch {batch}: throughput schedule vs oracle {e:.3e}")
    assert np.isfinite(full).all() and np.isfinite(cen).all()
    assert e <= util.TOL_MAX_ABS
