"""Batch-sharded multi-GPU evaluation: one process per GPU, replicated weights, no collective
on the data path; the only exchange is an all-gather of the per-sequence, per-joint error
block ``(B_local, J)`` float64 (RCCL over xGMI with backend "nccl"; "gloo" in the CPU tests).

The reference is single-GPU only (eval.py:368); this mirrors what its eval loop accumulates
(eval.py:185-196 -> action_wise_eval.py:25-26,43): per-example per-joint MPJPE, then the mean
over entries >= 0 in float64.
"""
import numpy as np


def shard_bounds(global_batch, rank, world):
    """Contiguous split of ``global_batch`` sequences; the first ``global_batch % world`` ranks get one more."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(int(global_batch), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allgather_errors(err_local, group=None):
    """All-gather a ``(B_local, J)`` float64 error block; ranks may hold different B_local.

    Returns the ``(B_global, J)`` tensor in rank order on every rank.
    """
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return err_local
    world = dist.get_world_size(group)
    n_local = torch.tensor([err_local.shape[0]], dtype=torch.int64, device=err_local.device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    nmax = max(counts)
    padded = torch.full((nmax,) + tuple(err_local.shape[1:]), -1.0, dtype=err_local.dtype, device=err_local.device)
    padded[: err_local.shape[0]] = err_local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)


def mean_valid_mm(err_global):
    """mean over entries >= 0, in millimetres (action_wise_eval.py:22,25-26,43)."""
    e = np.asarray(err_global.detach().cpu().numpy() if hasattr(err_global, "detach") else err_global, np.float64)
    e = e * 1000.0
    return float(np.mean(e[e >= 0]))


def allreduce_gradients(flat_grads, group=None):
    """Data-parallel training: sum the flat float32 gradient over ranks, in place.

    The loss is normalised by the GLOBAL ``config.BATCH_SIZE`` (train.py:482,488-489), so per-rank
    gradients simply add up -- no rescale.  One all-reduce of 10.4 M floats (41.6 MB) per step."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group)
    return flat_grads
