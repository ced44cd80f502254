"""Builds the in-tree HIP shared library ``csrc/libuu3d.so`` for gfx950 with hipcc.

hipcc cross-compiles without a GPU; the built .so is git-ignored but travels to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libuu3d.so")
SOURCES = ["uu3d_api.hip", "uu3d_ops.hip"]
HEADERS = ["uu3d_gemm.h", "uu3d_attn.h", "uu3d_spatial.h", "uu3d_misc.h", "uu3d_train.h", "uu3d_bwd.h", "uu3d_launch.h", "uu3d_train_kernels.h", "uu3d_train_step.inc", os.path.join("..", "..", "include", "uu3d_ops.h"), os.path.join("..", "..", "include", "uu3d.h")]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False, extra_flags=()):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC",
           "-Wall", "-Wno-unused-function", *extra_flags,
           "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
