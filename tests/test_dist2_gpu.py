"""GPU: the N > 1 paths with DEVICE tensors and two real ranks on one GPU (round-3 verdict, "next" item 4).  RCCL will not put two
ranks on one device, so the two child processes (tests/dist2_worker.py) use gloo, whose device-tensor all_reduce is issued from the
current stream and staged through the host: the ordering contract dist.BucketedAllReduce relies on (the collective waits for the
library's HIP stream that wrote the bucket) is the same one RCCL is given.  Checked:

  (a) eval.run_eval sharded over two ranks with a ragged split = the one-rank report (same windows, other batch compositions);
  (b) Trainer.forward_backward with the four buckets all-reduced from the library's stream while the backward pass is still running
      == ONE flat all-reduce of the two shard gradients, bit for bit, on a batch large enough that a bucket reduced before its
      gradients were written would show;
  (c) an Inf in ONE rank's input: both ranks skip the optimizer step, the replicas stay identical."""
import json
import os
import socket
import subprocess
import sys

import pytest

import uplift_upsample_3dhpe_amd as pkg
from tests import util

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
G = os.path.join(util.ROOT, "tests", "golden")


def test_two_ranks_on_one_gpu(tmp_path):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=util.ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(util.ROOT, "tests", "dist2_worker.py"), str(r), "2", str(port), str(tmp_path)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-3000:] for l in logs)
    res = [json.load(open(os.path.join(tmp_path, f"rank{r}.json"))) for r in range(2)]

    # (a) both ranks hold the same report; it equals the one-rank report (batches are composed differently: 1e-3 mm as in test_eval_gpu)
    from uplift_upsample_3dhpe_amd import eval as ev
    cfg = util.load_config("h36m_81")
    cfg.BATCH_SIZE = 16
    cfg.MASK_STRIDE = cfg.MASK_STRIDE[0] if isinstance(cfg.MASK_STRIDE, list) else cfg.MASK_STRIDE
    arch = pkg.arch_from_config(cfg)
    model = pkg.build_uplift_upsample_transformer(cfg, weights=pkg.init_weights(arch, seed=2, perturb=0.1))
    one = ev.run_eval(cfg, "h36m", os.path.join(G, "h36m_tiny_3d.npz"), os.path.join(G, "h36m_tiny_2d.npz"), "S9", model=model,
                      action_wise=False, log=lambda *a: None)
    assert res[0]["eval"] == res[1]["eval"]
    assert res[0]["eval"]["num_forwarded"] == one["num_forwarded"] == res[0]["shard"] + res[1]["shard"]
    assert res[0]["shard"] != res[1]["shard"] and min(res[0]["shard"], res[1]["shard"]) > 0, "the split is meant to be ragged"
    for part in ("all_frames", "keyframes"):
        if one[part] is None:
            assert res[0]["eval"][part] is None
            continue
        for k, v in one[part].items():
            assert abs(res[0]["eval"][part][k] - v) <= 1e-3, (part, k)

    # (b) bucketed (overlapped with the backward pass, issued from the library's stream) == flat, bit for bit, on both ranks
    for r in res:
        assert r["buckets"] >= 4 and r["buckets_from_library_stream"]
        assert r["bucketed_equals_flat_bitwise"], r
        assert r["sum_differs_from_local"] and r["grad_l2"] > 0
    assert res[0]["grad_l2"] == res[1]["grad_l2"]

    # (c) one rank's Inf: all ranks skip, replicas identical before and after the next (finite) step
    for r in res:
        assert r["skipped"] and r["params_unchanged_after_nonfinite_step"], r
        assert r["finite_step_applied"] and r["replicas_identical"], r
    assert res[0]["params_crc"] == res[1]["params_crc"]
