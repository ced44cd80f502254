"""Compile-time guards on the generated gfx950 code (hipcc cross-compiles without a GPU).

* The LDS-DMA GEMM (uu3d_gemm_h3.h, gemm_h3g_kernel) waits with a COUNTED s_waitcnt vmcnt(N): it is only correct
  when every dma() call is exactly N global_load_lds instructions.  A divergent loader once made hipcc emit one
  DMA per control-flow path (rare wrong rows in the strided conv); this test pins the instruction count.
* Packed-fp32 VALU ops are switched off for the device code (build.py DEVICE_FLAGS, docs/HISTORY.md E.12): none may
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
#include "uu3d_gemm_panel8.h"
#include "uu3d_tchain16.h"
#include "uu3d_attn.h"
#include "uu3d_attn_h3.h"
#include "uu3d_bwd.h"
#include "uu3d_spatial_h3.h"
using namespace uu3d;
template __global__ void uu3d::attn_head_wave_kernel<5, 48, true>(const float*, int, int, int, int, const uint8_t*, float*, int, size_t, int);
template __global__ void uu3d::attn_f32_kernel<3, 48, true>(const float*, int, int, int, int, const uint8_t*, float*, int, size_t);
template __global__ void uu3d::attn_h3_kernel<48, 3, 3, false>(const _Float16*, const _Float16*, int, int, int, int, const uint8_t*, _Float16*, size_t, int, int, int);
template __global__ void uu3d::attn_h3_kernel<48, 12, 3, true>(const _Float16*, const _Float16*, int, int, int, int, const uint8_t*, _Float16*, size_t, int, int, int);
template __global__ void uu3d::gemm_tn_h3_kernel<TnLoadLayerNorm, EpSlab>(const TnLoadLayerNorm, const float*, int, int, int, int, int, int, int, const EpSlab);
template __global__ void uu3d::gemm_tn_kernel<TnLoadLayerNorm, EpSlab>(const TnLoadLayerNorm, const float*, int, int, int, int, int, int, int, const EpSlab);
template __global__ void uu3d::spatial_stack_h3_kernel<17, 3, 1, false>(const float*, const SpatialParams, const _Float16*, float*, _Float16*, _Float16*, const SpatialTrainIO);
template __global__ void uu3d::spatial_stack_h3_kernel<17, 3, 1, true>(const float*, const SpatialParams, const _Float16*, float*, _Float16*, _Float16*, const SpatialTrainIO);
template __global__ void uu3d::gemm_h3_panel_kernel<24, PanelEpBias>(const _Float16*, const _Float16*, const float*, int, int, int, int, const PanelEpBias);
template __global__ void uu3d::gemm_h3_panel_kernel<24, PanelEpBiasReluSplit>(const _Float16*, const _Float16*, const float*, int, int, int, int, const PanelEpBiasReluSplit);
template __global__ void uu3d::gemm_h3_panel_kernel<24, PanelEpBiasResidual, 4>(const _Float16*, const _Float16*, const float*, int, int, int, int, const PanelEpBiasResidual);
template __global__ void uu3d::gemm_h3_panel8_kernel<PanelEpBiasSplitQ, 12, 3>(const _Float16*, const _Float16*, const float*, int, int, int, const PanelEpBiasSplitQ);
template __global__ void uu3d::gemm_h3_panel8_kernel<PanelEpBiasResidual, 4, 3>(const _Float16*, const _Float16*, const float*, int, int, int, const PanelEpBiasResidual);
template __global__ void uu3d::gemm_h3_panel8_kernel<PanelEpBiasResidualLn, 12, 3>(const _Float16*, const _Float16*, const float*, int, int, int, const PanelEpBiasResidualLn);
template __global__ void uu3d::tchain16_kernel<TC_PROJ | TC_MLP | TC_QKV>(const TChainArgs);
template __global__ void uu3d::tchain16_kernel<TC_PROJ | TC_MLP | TC_QKV | TC_PE>(const TChainArgs);
template __global__ void uu3d::tchain16_kernel<TC_PROJ | TC_MLP>(const TChainArgs);
template __global__ void uu3d::tchain16_kernel<TC_QKV>(const TChainArgs);
template __global__ void uu3d::tchain16_kernel<TC_PROJ | TC_FC1_PLANES>(const TChainArgs);
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
    # the in-place residual form (attention projection, chunk loop unrolled: CPW = 4): every residual value is requested straight into
    # an accumulation register by name, and nothing but the prologue / the ends of the two panel variants drains the ring -- a load
    # hipcc can see brought a vmcnt(0) per chunk (parked in an AGPR at once) or per loop iteration (in flight across the back edge)
    body = next(v for k, v in ks.items() if "gemm_h3_panel_kernel" in k and "PanelEpBiasResidual" in k)
    assert "scratch_" not in body
    n_dma = body.count("global_load_lds_dwordx4")
    assert n_dma % 12 == 0 and body.count("v_mfma_f32_32x32x16_f16") == (n_dma // 12 - 2) * 72
    res = re.findall(r"global_load_dword (\w+), v\d+, s\[\d+:\d+\]", body)
    assert len(res) == 2 * 4 * 16 and all(r.startswith("a") for r in res), res[:4]
    assert len(re.findall(r"s_waitcnt vmcnt\(0\)", body)) <= 5
    assert len(re.findall(r"s_waitcnt vmcnt\(12\) lgkmcnt\(0\)", body)) == 2 * 4


def test_eight_wave_row_panel_gemm_code_shape(asm):
    """uu3d_gemm_panel8.h (round 4): two code paths (whole row tile / ragged last tile), each a straight line of CPW chunks.  What the
    kernel relies on: no scratch (a register that was requested by name must never be spilled or copied before its wait -- the
    branch between the paths sits in front of the first request for that reason); per chunk and wave 36 MFMAs, 6 LDS-DMA pieces
    (3 per half-interval) and two barriers; the A fragments requested by name (24 per path, never by a load hipcc sees); every
    residual / bias request by name into an architectural register; 256 registers = two waves per SIMD."""
    ks = _kernels(asm)
    for ep, cpw, requests in (("PanelEpBiasSplitQ", 12, 1), ("PanelEpBiasResidualE", 4, 9), ("PanelEpBiasResidualLn", 12, 9)):
        body = next(v for k, v in ks.items() if "gemm_h3_panel8_kernel" in k and ep in k)
        assert "scratch_" not in body, ep
        assert body.count("v_mfma_f32_32x32x16_f16") == 2 * cpw * 36, ep
        # prologue: half-chunks 0 .. 4 = 15 pieces per path (hipcc may hoist the first 6, which do not depend on the path, above the branch:
        # 24 instead of 30 in the text); every chunk issues 6 more, 3 per half-interval (the tail's are clamped re-reads)
        assert body.count("global_load_lds_dwordx4") - 12 * cpw in (24, 30), ep
        per_interval = [seg.count("global_load_lds_dwordx4") for seg in body.split("s_barrier")]
        assert sorted(set(per_interval[1:])) in ([0, 3], [0, 3, 9]), (ep, per_interval)
        assert body.count("s_barrier") == 2 * (2 * cpw + 2 + (1 if "Ln" in ep else 0)), ep
        a_frag = re.findall(r"global_load_dwordx4 v\[\d+:\d+\], v\d+, s\[\d+:\d+\]", body)
        assert len(a_frag) == 2 * 24 + (0 if "Ln" not in ep else 0), (ep, len(a_frag))
        by_name = re.findall(r"global_load_dword (\w+), v\d+, s\[\d+:\d+\]", body)
        assert len(by_name) == 2 * cpw * requests and all(r.startswith("v") for r in by_name), (ep, len(by_name))
        # the waits are counted: nothing but the two ends of a path (and the LayerNorm tail) drains vector memory
        assert len(re.findall(r"s_waitcnt vmcnt\(0\)", body)) <= (2 * 5 if "Ln" in ep else 2), ep
    # the kernel descriptors: at most 256 registers (two waves per SIMD), all of the LDS
    for name in (k for k in ks if "gemm_h3_panel8_kernel" in k):
        d = asm[asm.index(".amdhsa_kernel " + name):]
        d = d[:d.index(".end_amdhsa_kernel")]
        assert int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", d).group(1)) <= 256, name


def _loops(body):
    """{header block: text of every basic block of that INNERMOST loop} from hipcc's block comments ("=>This Inner Loop Header", "in Loop: Header=BBn_m"):
    a loop's blocks are not contiguous in the text (the compiler rotates and sinks them), so label-to-branch ranges miss pieces."""
    blocks, cur, head = [], None, None
    for line in body.split("\n"):
        m = re.match(r"^\.(LBB\d+_\d+):\s*;?(.*)$", line)
        if m:
            cur = [m.group(1), m.group(2), []]
            blocks.append(cur)
        elif cur is not None:
            cur[2].append(line)
    inner = {name for name, comment, _ in blocks if "Inner Loop Header" in comment}          # innermost loops only
    loops = {}
    for name, comment, lines in blocks:
        m = re.search(r"in Loop: Header=(BB\d+_\d+)", comment)
        hdr = "L" + m.group(1) if m else name
        if hdr in inner:
            loops.setdefault(hdr, []).extend(lines)
    return {k: "\n".join(v) for k, v in loops.items()}


def test_temporal_chain_16_token_panels_code_shape(asm):
    """uu3d_tchain16.h (the temporal chain; round 5's form added its residuals with global_atomic_add_f32): 64 rows per workgroup on eight waves (16-token panels, v_mfma_f32_16x16x32_f16), everything of a
    temporal block's row-local stages on chip.  Pinned: no float atomic, no scratch, 256 registers = two waves per SIMD, per chunk body 36 MFMAs, 6 LDS-DMA
    pieces and two barriers, no vector-memory load inside a rolled loop, every wait inside it one of the hand-written counted ones."""
    ks = _kernels(asm)
    chains = {k: v for k, v in ks.items() if "tchain16_kernel" in k}
    assert len(chains) == 5
    for name, body in chains.items():
        assert "global_atomic" not in body and "flat_atomic" not in body and "buffer_atomic" not in body, name
        assert "scratch_" not in body, name
        assert "v_pk_mul_f32" not in body and "v_pk_fma_f32" not in body and "v_pk_add_f32" not in body, name
        assert "v_mfma_f32_32x32x16_f16" not in body, name
        n_mfma, n_dma = body.count("v_mfma_f32_16x16x32_f16"), body.count("global_load_lds_dwordx4")
        assert n_mfma % 36 == 0 and n_dma == 6 * (n_mfma // 36) + 15, (name, n_mfma, n_dma)
        loops = {h: t for h, t in _loops(body).items() if t.count("v_mfma_f32_16x16x32_f16") == 4 * 36}
        assert loops or "Li3E" in name, name
        for h, t in loops.items():
            back = re.search(r"s_cbranch_\w+ \." + re.escape(h) + r"\b", t)
            assert back, (name, h)
            t = t[:back.start()]
            assert "v_readlane" not in t and "v_writelane" not in t, name
            assert not re.search(r"\b(global|buffer|flat)_load_(dword|ubyte|ushort|short)", t), name
            assert t.count("s_barrier") == 4 * 2, name
            in_asm = sum(blk.count("s_waitcnt") for blk in re.findall(r";;#ASMSTART(.*?);;#ASMEND", t, re.S))
            assert in_asm >= 4 * 14 and t.count("s_waitcnt") == in_asm, (name, in_asm, t.count("s_waitcnt"))
            assert not re.search(r"s_waitcnt vmcnt\(0\)", t), name
            assert t.count("global_load_lds_dwordx4") == 4 * 6, (name, t.count("global_load_lds_dwordx4"))
        d = asm[asm.index(".amdhsa_kernel " + name):]
        d = d[:d.index(".end_amdhsa_kernel")]
        assert int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", d).group(1)) <= 256, name


def test_no_packed_fp32_valu_ops(asm):
    """docs/HISTORY.md E.12: a packed-f32 op whose op_sel reads the OTHER half of a register pair can lose that operand next to
    a busy matrix pipe.  hipcc never emits packed f32 (target feature off); the only ones in the library are written by name in
    the spatial stack (uu3d_pk.h, namespace pk), and none of them carries op_sel / op_sel_hi."""
    ks = _kernels(asm)
    for name, body in ks.items():
        pk = re.findall(r"v_pk_(?:mul|fma|add)_f32[^\n]*", body)
        if "spatial_stack_h3_kernel" in name:
            assert len(pk) > 500, (name, len(pk))
            assert not any("op_sel" in l for l in pk), name
            assert "scratch_" not in body, name
        else:
            assert not pk, name
    assert "v_mfma_f32_32x32x16_f16" in asm


def _exec_guarded_loads(body):
    """Loads that hipcc put inside a divergent branch together with the wait for their data (source: `if (ok) x = *p;`):
    each one is a serial memory round trip.  Counts s_cbranch_execz -> global_load -> s_waitcnt vmcnt within one block."""
    lines, n = body.split("\n"), 0
    for i, l in enumerate(lines):
        if "s_cbranch_execz" not in l:
            continue
        seg = lines[i + 1:i + 25]
        gl = [j for j, x in enumerate(seg) if "global_load" in x]
        if gl and not any(".LBB" in x for x in seg[:gl[0]]) and any("s_waitcnt vmcnt" in x for x in seg[gl[0]:gl[0] + 12]):
            n += 1
    return n


def test_attention_kernels_code_shape(asm):
    """uu3d_attn.h: no scratch, no branch-guarded loads (mask bytes, K / V / Q rows are clamped and selected instead), the
    V operand reads of the wave-per-item kernel stay the by-name ds_read_b32 batches issued ahead of their MFMAs, and the
    f16 planes leave as 16-byte stores."""
    ks = _kernels(asm)
    hw = next(v for k, v in ks.items() if "attn_head_wave_kernel" in k)
    wg = next(v for k, v in ks.items() if "attn_f32_kernel" in k)
    for body in (hw, wg):
        assert "scratch_" not in body
        assert _exec_guarded_loads(body) == 0
        assert "global_store_dwordx4" in body and "global_store_short" not in body
    assert hw.count("s_barrier") == 0
    # inline-asm loads with a scalar base: the compiler cannot pad the VALU-writes-SGPR -> VMEM-reads-SGPR hazard (5 wait
    # states) for them; an s_nop 4 must stand between the last v_readfirstlane and the first such load
    code0 = [l.strip() for l in hw.split("\n") if l.strip() and not l.strip().startswith(";")]
    first_s = next(i for i, l in enumerate(code0) if re.match(r"global_load_dwordx4 v\[\d+:\d+\], v\d+, s\[", l))
    last_rfl = max(i for i, l in enumerate(code0[:first_s]) if l.startswith("v_readfirstlane_b32"))
    assert any(l.startswith("s_nop 4") for l in code0[last_rfl + 1:first_s])
    code = [l.strip() for l in hw.split("\n") if l.strip() and not l.strip().startswith(";")]
    run = best = 0
    for l in code:
        run = run + 1 if l.startswith("ds_read_b32") else 0
        best = max(best, run)
    assert best >= 24, best                                  # two key tiles' V values (2 x 12) issued back to back, ahead of the MFMAs
    assert "s_waitcnt lgkmcnt(15)" in hw and "s_waitcnt lgkmcnt(12)" in hw
    # no scalar load between the first by-name V read and the last counted wait (scalar loads return out of order: a counted
    # lgkmcnt would then not mean "the oldest reads are back")
    first = next(i for i, l in enumerate(code) if l.startswith("ds_read_b32"))
    last = max(i for i, l in enumerate(code) if l.startswith("s_waitcnt lgkmcnt(12)") or l.startswith("s_waitcnt lgkmcnt(15)"))
    assert not any(l.startswith(("s_load", "s_buffer_load")) for l in code[first:last + 1])
    back = code[max(0, first - 40):first]
    z = max(i for i, l in enumerate(back) if l.startswith("s_waitcnt lgkmcnt(0)"))             # the reads start from an empty counter ...
    assert not any(l.startswith(("ds_", "s_load", "s_buffer_load")) for l in back[z + 1:])      # ... and nothing else is issued on it in between


def test_weight_gradient_gemm_code_shape(asm):
    """uu3d_bwd.h: the f16x3 weight-gradient GEMM reads its fragments with ds_read_b64_tr_b16 (32 per k-step: 8 fragments x 2
    planes x 2 reads) for 24 MFMAs, fits two workgroups per CU (<= 256 registers), and neither it nor the f32 kernel waits for a
    load inside a divergent branch (TnLoad*::fetch / finish)."""
    ks = _kernels(asm)
    h3 = next(v for k, v in ks.items() if "gemm_tn_h3_kernel" in k)
    f32 = next(v for k, v in ks.items() if "gemm_tn_kernel" in k)
    assert "scratch_" not in h3 and "scratch_" not in f32
    assert h3.count("ds_read_b64_tr_b16") == 32
    assert h3.count("v_mfma_f32_32x32x16_f16") == 24
    assert _exec_guarded_loads(h3) == 0 and _exec_guarded_loads(f32) == 0
    m = re.search(r"gemm_tn_h3_kernel\w*\n(?:.*\n)*?\s*\.vgpr_count:\s+(\d+)", asm)
    assert m and int(m.group(1)) <= 256, m and m.group(1)


def test_spatial_stack_weight_fragments_are_prefetched(asm):
    """uu3d_spatial_h3.h: the weight fragments of a product are loaded by name ahead of it (12 loads in one batch at the top of
    a block for q / k / v) with counted waits; hipcc had sunk every pair of loads to its MFMAs."""
    ks = _kernels(asm)
    sp = next(v for k, v in ks.items() if "spatial_stack_h3_kernel" in k)
    assert "scratch_" not in sp
    assert re.search(r"(global_load_dwordx4 v\[\d+:\d+\], v\[\d+:\d+\], off\n(?:\t[sv]_\w+.*\n){0,6}?\t?){12}", sp) or sp.count("global_load_dwordx4") >= 32
    for n in (8, 4, 0):
        assert f"s_waitcnt vmcnt({n})" in sp
    # the per-head attention reads its key pairs (then its value pairs), two 16-byte reads each, in by-name batches of 10 and 8
    code = [l.strip() for l in sp.split("\n") if l.strip() and not l.strip().startswith(";")]
    run = best = 0
    for l in code:
        run = run + 1 if l.startswith("ds_read_b128") else 0
        best = max(best, run)
    assert best >= 10, best                                  # first batch: 5 key pairs x 2 reads
    m = re.search(r"spatial_stack_h3_kernel\w*\n(?:.*\n)*?\s*\.vgpr_count:\s+(\d+)", asm)
    assert m and int(m.group(1)) <= 168, m and m.group(1)    # three waves per SIMD
