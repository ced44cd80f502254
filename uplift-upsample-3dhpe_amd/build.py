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
HEADERS = ["uu3d_tchain.h", "uu3d_gemm.h", "uu3d_gemm_h3.h", "uu3d_gemm_panel.h", "uu3d_gemm_panel8.h", "uu3d_gemm_wt.h", "uu3d_mlp_fused.h", "uu3d_tail.h", "uu3d_attn.h", "uu3d_attn_h3.h", "uu3d_spatial.h", "uu3d_spatial_h3.h", "uu3d_pk.h", "uu3d_misc.h", "uu3d_train.h", "uu3d_bwd.h", "uu3d_launch.h", "uu3d_train_kernels.h", "uu3d_dropout.h", "uu3d_train_step.inc", os.path.join("..", "..", "include", "uu3d_ops.h"), os.path.join("..", "..", "include", "uu3d.h")]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


# Packed fp32 VALU ops (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32) are turned OFF for the device code.
# Measured on MI355X with ROCm 7.2: with them on, the LayerNorm prologue of the f16x3
# GEMM came out as v_pk_mul_f32 / v_pk_fma_f32 with op_sel broadcasts from a VGPR pair, and the low half of
# the result was intermittently 0 in lanes 48-63 -- whole output rows wrong by O(1), different on every run.
# Without the feature the same source is bit-reproducible (tools/gemm_bench DET=1: 0 mismatches in 32 of 32
# configurations against 19 of 32 failing) and the forward pass costs 1-3 %.
DEVICE_FLAGS = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
_HOST_NOISE = "is not a recognized feature for this target"      # the host half of the compile ignores the feature


def build(force=False, verbose=False, extra_flags=()):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC",
           "-Wall", "-Wno-unused-function", *DEVICE_FLAGS, *extra_flags,
           "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, cwd=CSRC, stderr=subprocess.PIPE, text=True)
    err = "".join(l for l in r.stderr.splitlines(True) if _HOST_NOISE not in l)
    if err.strip():
        sys.stderr.write(err)
    if r.returncode != 0:
        raise subprocess.CalledProcessError(r.returncode, cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
