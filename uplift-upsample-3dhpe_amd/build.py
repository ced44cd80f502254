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
HEADERS = ["uu3d_tchain16.h", "uu3d_gemm.h", "uu3d_gemm_h3.h", "uu3d_gemm_panel.h", "uu3d_gemm_panel8.h", "uu3d_gemm_wt.h", "uu3d_mlp_fused.h", "uu3d_attn.h", "uu3d_attn_h3.h", "uu3d_spatial.h", "uu3d_spatial_h3.h", "uu3d_pk.h", "uu3d_misc.h", "uu3d_train.h", "uu3d_bwd.h", "uu3d_launch.h", "uu3d_train_kernels.h", "uu3d_dropout.h", "uu3d_train_step.inc", os.path.join("..", "..", "include", "uu3d_ops.h"), os.path.join("..", "..", "include", "uu3d.h")]


def _fingerprint(extra_flags=()):
    """sha256 over every source / header of the library, the compiler flags and hipcc's version: what the binary was built FROM."""
    import hashlib
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        h.update(f.encode()); h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(repr((DEVICE_FLAGS, tuple(extra_flags))).encode())
    try:
        h.update(subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--version"], capture_output=True, text=True).stdout.encode())
    except OSError:
        pass
    return h.hexdigest()


def _stale(lib, extra_flags=()):
    """The library is rebuilt unless the fingerprint written next to it (``<lib>.sha256``) matches the sources -- not by file times: a shipped
    binary whose sources were edited and touched back, or checked out again, would otherwise pass for current."""
    stamp = lib + ".sha256"
    if not os.path.exists(lib) or not os.path.exists(stamp):
        return True
    return open(stamp).read().strip() != _fingerprint(extra_flags)


# Packed fp32 VALU ops (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32) are turned OFF for the device code.
# Measured on MI355X with ROCm 7.2 (docs/HISTORY.md E.12): with them on, the LayerNorm prologue of the f16x3
# GEMM came out as v_pk_mul_f32 / v_pk_fma_f32 with op_sel broadcasts from a VGPR pair, and the low half of
# the result was intermittently 0 in lanes 48-63 -- whole output rows wrong by O(1), different on every run.
# Without the feature the same source is bit-reproducible (tools/gemm_bench DET=1: 0 mismatches in 32 of 32
# configurations against 19 of 32 failing) and the forward pass costs 1-3 %.
DEVICE_FLAGS = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
_HOST_NOISE = "is not a recognized feature for this target"      # the host half of the compile ignores the feature


TIMING_LIB = os.path.join(CSRC, "libuu3d_timing.so")      # build(timing=True): -DUU3D_TIMING_BUILD (UU3D_SKIP / UU3D_TIMING_PARTS honoured; tools/ only, load it through UU3D_LIB)


def build(force=False, verbose=False, extra_flags=(), timing=False):
    lib = TIMING_LIB if timing else LIB
    extra_flags = tuple(extra_flags) + (("-DUU3D_TIMING_BUILD",) if timing else ())
    if not force and not _stale(lib, extra_flags):
        return lib
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC",
           "-Wall", "-Wno-unused-function", *DEVICE_FLAGS, *extra_flags,
           "-o", lib] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, cwd=CSRC, stderr=subprocess.PIPE, text=True)
    err = "".join(l for l in r.stderr.splitlines(True) if _HOST_NOISE not in l)
    if err.strip():
        sys.stderr.write(err)
    if r.returncode != 0:
        raise subprocess.CalledProcessError(r.returncode, cmd)
    with open(lib + ".sha256", "w") as f:
        f.write(_fingerprint(extra_flags) + "\n")
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, timing="--timing" in sys.argv))
