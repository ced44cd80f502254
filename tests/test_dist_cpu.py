"""N > 1 path on CPU: world_size-2 gloo run of the shard + all-gather protocol used by bench.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from uplift_upsample_3dhpe_amd import dist as ud


def test_shard_bounds_cover_batch():
    for g, w in [(1024, 8), (10, 3), (2, 4), (7, 1)]:
        spans = [ud.shard_bounds(g, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == g
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
    with pytest.raises(ValueError):
        ud.shard_bounds(4, 2, 2)


def _worker(rank, world, port, global_batch, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(0)                        # same table on every rank
    table = rng.uniform(0, 0.2, size=(global_batch, 17))
    table[3, 5] = -1.0                                    # an invalid joint
    lo, hi = ud.shard_bounds(global_batch, rank, world)
    local = torch.from_numpy(table[lo:hi].copy())
    out = ud.allgather_errors(local)
    q.put((rank, out.numpy(), ud.mean_valid_mm(out)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("global_batch", [8, 7])
def test_allgather_errors_gloo_world2(global_batch):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, global_batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    table = np.random.default_rng(0).uniform(0, 0.2, size=(global_batch, 17)); table[3, 5] = -1.0
    expect = float(np.mean((table * 1000)[table >= 0]))
    for rank, out, mean_mm in res:
        assert np.array_equal(out, table)                 # rank order, ragged shards, bit-exact
        assert mean_mm == pytest.approx(expect, rel=1e-12)


def _grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.full((1000,), float(rank + 1))
    ud.allreduce_gradients(g)
    q.put((rank, g.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_gradients_gloo_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, g in res:
        assert np.all(g == 3.0)                           # 1 + 2 on every rank, no rescale
    assert ud.allreduce_gradients(torch.ones(3)).sum() == 3    # no process group: identity


def _bucket_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(100 + rank)
    n = 10007
    g = torch.from_numpy(rng.normal(0, 1e-3, n).astype(np.float32))
    flat = g.clone()
    ud.allreduce_gradients(flat)                                   # the one-collective form
    b = g.clone()
    br = ud.BucketedAllReduce(b)
    # ranges in the order the backward pass finishes them: tail of the buffer first, the front last (uu3d_train_step.inc)
    for first, count in [(7000, 3007), (4000, 3000), (1500, 2500), (0, 1500)]:
        br.ready(first, count)
    br.wait()
    bad = ud.BucketedAllReduce(g.clone())
    bad.ready(10, n - 10)
    try:
        bad.wait(); tiled = True
    except RuntimeError:
        tiled = False
    q.put((rank, flat.numpy().copy(), b.numpy().copy(), tiled))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_equals_flat_bitwise_gloo_world2():
    """gradient buckets started in backward order sum to exactly what one flat all-reduce gives, and a set of
    ranges that does not tile the buffer is an error (a gradient tensor would silently stay rank-local)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = sum(np.random.default_rng(100 + r).normal(0, 1e-3, 10007).astype(np.float32) for r in range(2))
    for _, flat, bucketed, tiled in res:
        assert np.array_equal(flat, bucketed)
        assert np.array_equal(flat, expect.astype(np.float32))
        assert not tiled
    # no process group: ready / wait are bookkeeping only
    t = torch.ones(8)
    b = ud.BucketedAllReduce(t)
    b.ready(4, 4); b.ready(0, 4); b.wait()
    assert t.sum() == 8


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher (what a driver may call): the parent spawns two ranks as child
    processes before touching a GPU and returns their exit code (--spawn-check: ranks print and exit, no GPU needed)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--spawn-check"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "rank 0/2" in r.stdout and "rank 1/2" in r.stdout


class _StubPipe:
    """The surface of pipeline.ForwardPipeline that bench.run_pipelined_steps drives, with a forward that costs nothing: launch() -> ticket,
    result(ticket) -> (full, central, err), after(ticket, fn), join(); the error block of step k on rank r is a known table."""

    def __init__(self, rank, B, J, depth):
        self.rank, self.B, self.J, self.depth, self.n = rank, B, J, depth, 0
        self.live = {}

    @staticmethod
    def block(rank, k, B, J):
        return torch.arange(B * J, dtype=torch.float64).reshape(B, J) + 1000.0 * k + 1.0e6 * rank

    def launch(self, wait_caller=True):
        t = self.n
        self.n += 1
        assert len(self.live) < self.depth, "more batches in flight than slots"
        self.live[t] = self.block(self.rank, t, self.B, self.J)
        return t

    def result(self, t):
        return None, None, self.live.pop(t)

    def after(self, t, fn):                                          # (the real one runs fn with the slot's stream current)
        return fn(None, None, self.live.pop(t))

    def join(self):
        self.joined = self.n


def _bench_loop_worker(rank, world, port, mode, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    B, J, steps, depth = 5, 17, 7, 3
    g = bench.ErrorGather(mode, steps, B, J, world, "cpu", True)
    pipe = _StubPipe(rank, B, J, depth)
    bench.run_pipelined_steps(pipe, 2, depth, g)                     # "warm-up": the loop is re-entrant (reset)
    pipe.n = 0
    bench.run_pipelined_steps(pipe, steps, depth, g)
    out = {"local": g.local.numpy().copy(), "k": g.k, "gathered": None if g.gathered is None or mode != "end" else g.gathered.view(world, steps, B, J).numpy().copy(),
           "step_out": None if g.step_out is None else g.step_out.numpy().copy()}
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["end", "step"])
def test_bench_step_loop_gathers_over_two_ranks_gloo(mode):
    """bench.py's REAL step loop (run_pipelined_steps + ErrorGather: what `python bench.py --gpus N` times) at world size 2 over gloo
    with a stub forward: every step's (B, J) block of every rank arrives, in rank and step order -- mode "end": one collective behind
    the last result; mode "step": one per step (the last one is checked)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_loop_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    B, J, steps = 5, 17, 7
    for rank in (0, 1):
        o = res[rank]
        assert o["k"] == steps
        for k in range(steps):
            assert np.array_equal(o["local"][k], _StubPipe.block(rank, k, B, J).numpy())
        if mode == "end":
            for r in (0, 1):
                for k in range(steps):
                    assert np.array_equal(o["gathered"][r, k], _StubPipe.block(r, k, B, J).numpy())
        else:
            for r in (0, 1):
                assert np.array_equal(o["step_out"][r * B:(r + 1) * B], _StubPipe.block(r, steps - 1, B, J).numpy())
