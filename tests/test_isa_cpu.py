"""Compile-time guards on the generated gfx950 code (hipcc cross-compiles without a GPU).

* The LDS-DMA GEMM (uu3d_gemm_h3.h, gemm_h3g_kernel) waits with a COUNTED s_waitcnt vmcnt(N): it is only correct
  when every dma() call is exactly N global_load_lds instructions.  A divergent loader once made hipcc emit one
  DMA per control-flow path (rare wrong rows in the strided conv); this test pins the instruction count.
* Packed-fp32 VALU ops are switched off for the device code (build.py DEVICE_FLAGS): none may
  appear.
"""
import os
import re
import subprocess
import tempfile

import pytest

from tests import util

CSRC = os.path.join(util.ROOT, "uplift-upsample-3dhpe_amd", "csrc")
SRC = r'''
#include "uu3d_gemm_h3.h"
#include "uu3d_gemm_panel.h"
using namespace uu3d;
template __global__ void uu3d::gemm_h3_panel_kernel<24, PanelEpBias>(const _Float16*, const _Float16*, const float*, int, int, int, int, const PanelEpBias, int, float);
template __global__ void uu3d::gemm_h3_panel_kernel<24, PanelEpBias, true>(const _Float16*, const _Float16*, const float*, int, int, int, int, const PanelEpBias, int, float);
template __global__ void uu3d::gemm_h3_panel_kernel<24, PanelEpBiasReluSplit>(const _Float16*, const _Float16*, const float*, int, int, int, int, const PanelEpBiasReluSplit, int, float);
template __global__ void uu3d::ln_split_frag_kernel<24, 8>(const float*, int, int, float, const float*, const float*, _Float16*);
template __global__ void uu3d::gemm_h3g_kernel<1, 1, GLoadConv3, EpSlab, 3>(const GLoadConv3, const _Float16*, const _Float16*, int, int, int, int, int, int, const EpSlab);
template __global__ void uu3d::gemm_h3g_kernel<1, 2, GLoadPlain, EpBiasResidual, 3>(const GLoadPlain, const _Float16*, const _Float16*, int, int, int, int, int, int, const EpBiasResidual);
template __global__ void uu3d::gemm_h3_kernel<1, 2, ALoadLayerNorm, EpBias>(const ALoadLayerNorm, const _Float16*, const _Float16*, int, int, int, int, int, int, const EpBias);
'''


@pytest.fixture(scope="module")
def asm():
    import importlib.util
    spec = importlib.util.spec_from_file_location("uu3d_build", os.path.join(util.ROOT, "uplift-upsample-3dhpe_amd", "build.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    with tempfile.TemporaryDirectory() as d:
        src, out = os.path.join(d, "k.hip"), os.path.join(d, "k.s")
        open(src, "w").write(SRC)
        subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", *b.DEVICE_FLAGS, "-I", CSRC,
                        "-I", os.path.join(util.ROOT, "include"), "-S", "--cuda-device-only", "-o", out, src],
                       check=True, stderr=subprocess.DEVNULL)
        return open(out).read()


def _kernels(asm):
    out = {}
    for m in re.finditer(r"^(_ZN4uu3d\w+):.*?s_endpgm", asm, re.S | re.M):
        out[m.group(1)] = m.group(0)
    return out


def test_lds_dma_instruction_counts(asm):
    ks = _kernels(asm)
    conv = next(v for k, v in ks.items() if "gemm_h3g_kernel" in k and "GLoadConv3" in k)
    plain = next(v for k, v in ks.items() if "gemm_h3g_kernel" in k and "GLoadPlain" in k)
    # 64x64 tile: NP = 2*(1+1) = 4 DMAs per k-tile; 64x128: NP = 2*(1+2) = 6; two prologue tiles + one in the loop
    assert conv.count("global_load_lds_dwordx4") == 3 * 4
    assert plain.count("global_load_lds_dwordx4") == 3 * 6
    for body, n in ((conv, 4), (plain, 6)):
        assert f"s_waitcnt vmcnt({n}) lgkmcnt(0)" in body       # counted wait + LDS reads retired before the barrier
        assert body.count("s_barrier") == 2
        assert "scratch_" not in body


def test_row_panel_gemm_code_shape(asm):
    """uu3d_gemm_panel.h: the counted waits assume exactly PANEL_PIECES = 12 LDS-DMAs per wave and k-step (one k-step
    left in flight: vmcnt(12)); the kernel only works as designed without scratch (a spill reload is a vmcnt(0) that
    drains the ring) and with its register budget of one wave per SIMD."""
    ks = _kernels(asm)
    for ep in ("PanelEpBiasE", "PanelEpBiasReluSplit"):
        body = next(v for k, v in ks.items() if "gemm_h3_panel_kernel" in k and ep in k)
        assert "scratch_" not in body
        assert "s_waitcnt vmcnt(12) lgkmcnt(0)" in body
        n_dma = body.count("global_load_lds_dwordx4")
        # 2 prologue k-steps + one refill k-step per copy of the chunk body (ping-pong x whole / ragged panel, plus whatever
        # hipcc peels): every copy must hold exactly 12 DMAs next to its 72 MFMAs
        assert n_dma % 12 == 0 and n_dma >= 2 * 12 + 4 * 12, n_dma
        assert body.count("v_mfma_f32_32x32x16_f16") == (n_dma // 12 - 2) * 72
        assert re.search(r"ds_read_b128 v\[\d+:\d+\], v\d+ offset:\d+\n\tds_read_b128", body)     # the asm fragment reads survived


def test_no_packed_fp32_valu_ops(asm):
    assert re.search(r"v_pk_(mul|fma|add)_f32", asm) is None
    assert "v_mfma_f32_32x32x16_f16" in asm
