"""Shared helpers for the tests: the product's synthetic-workload module plus the oracle-side config mapping."""
from uplift_upsample_3dhpe_amd.synthetic import (ROOT, CONFIGS, TOL_MAX_ABS, TOL_MPJPE_MM, load_config,   # noqa: F401
                                                 eval_stride_mask, synthetic_batch)
from oracle.uplift_oracle import hp_from_arch                                                                # noqa: F401


def direct_forward(model, xt, mt, schedule):
    """One quiet uu3d_forward_ex call on the current stream under the given schedule (0 latency = what model(...) runs, 1 throughput = what a
    pipeline with more than one slot runs: from 1024 token rows on that is the temporal chain, whose sums have another order) -> (full, central)."""
    import torch
    a = model.arch
    B = xt.shape[0]
    full = torch.empty((B, a.num_frames, a.num_keypoints, 3), dtype=torch.float32, device=xt.device) if model._returns_full else None
    cen = torch.empty((B, a.num_keypoints, 3), dtype=torch.float32, device=xt.device)
    model._forward(xt.contiguous(), model._mask_u8(mt) if mt is not None else None, full, cen, 0, torch.cuda.current_stream(xt.device), schedule=schedule)
    torch.cuda.synchronize()
    return full, cen
