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
    if err_local.is_cuda and dist.get_backend(group) == "gloo":
        # gloo gathers host tensors only (its device-tensor support is broadcast / all_reduce): stage the few KB through the host
        return allgather_errors(err_local.cpu(), group).to(err_local.device)
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


class BucketedAllReduce(object):
    """Gradient all-reduce in buckets that start WHILE the backward pass is still running.

    ``uu3d_train_forward_backward`` reports every finished contiguous range of the flat gradient buffer through
    ``uu3d_train_set_grad_callback``; ``ready(first, count, stream)`` starts an asynchronous sum over ranks of exactly
    that range (the communication stream first waits for ``stream``, the HIP stream the range was written on),
    ``wait()`` makes the current stream wait for all of them.  The loss is normalised by the GLOBAL batch size, so rank
    gradients add with no rescale (train.py:482,488-489).  Elementwise sums do not depend on how the buffer is cut, so
    the result is bit-identical to one flat all-reduce; with world size 1 (or no process group) nothing happens.
    For the shipped h36m_351 model the ranges are: strided blocks + heads (21 MB), temporal blocks 4-3 and 2-1
    (9.4 MB each), everything in front of the temporal blocks (1.3 MB) -- four collectives instead of one 41.6 MB one
    behind the backward pass; the ring is per-link bound (xGMI), so the first 21 MB overlap ~2 ms of backward work.
    """

    def __init__(self, flat_grads, group=None):
        self.flat = flat_grads.view(-1)
        self.group = group
        self.works = []
        self.ranges = []
        self.error = None
        self.force = False          # issue the collectives at world size 1 as well (tests: the RCCL / ExternalStream path on one GPU)

    def begin(self):
        """Start of a backward pass: wait for collectives a previous pass left outstanding (gradient inspection, a step that
        failed halfway) and forget their ranges, so that this pass's ranges tile the buffer on their own."""
        works, self.works, self.ranges, self.error = self.works, [], [], None
        for w in works:
            w.wait()

    def raise_pending(self):
        """Re-raise what ``ready`` caught inside the C callback (ctypes would only print it and go on)."""
        if self.error is not None:
            e, self.error = self.error, None
            raise e

    def _active(self):
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size(self.group) > 1 or self.force)

    def ready(self, first, count, stream=None):
        import torch
        import torch.distributed as dist
        try:
            if self._active():
                view = self.flat[first:first + count]
                if stream is not None and self.flat.is_cuda:
                    # the range was written on a raw HIP stream of the library: issue the collective from it, so that the
                    # process group's communication stream waits for exactly that work
                    with torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=self.flat.device)):
                        self.works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                else:
                    self.works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.ranges.append((int(first), int(count)))       # only a range whose collective was issued counts as covered
        except BaseException as e:                              # raised again by raise_pending() / wait()
            if self.error is None:
                self.error = e

    def wait(self):
        """Current stream (GPU) / host (CPU tensors) waits for every started bucket; checks the ranges tiled the buffer."""
        works, ranges = self.works, self.ranges
        self.works, self.ranges = [], []                        # cleared whatever happens below
        self.raise_pending()
        for w in works:
            w.wait()
        covered = sorted(ranges)
        pos = 0
        for f, c in covered:
            if f != pos:
                raise RuntimeError(f"gradient ranges do not tile the buffer: gap or overlap at {pos} (next range starts at {f})")
            pos = f + c
        if covered and pos != self.flat.numel():
            raise RuntimeError(f"gradient ranges end at {pos}, the buffer has {self.flat.numel()} elements")
