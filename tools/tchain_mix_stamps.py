"""(RECORD: needs the round-5 chain kernel in the library and its -DUU3D_TC_STAMP hooks, both removed after commit ed9e71b.)
How long the temporal chain's workgroups take INSIDE the pipelined mix (library built with -DUU3D_TC_STAMP):
   python uplift-upsample-3dhpe_amd/build.py ... ; gpurun -- 'UU3D_TCHAIN=1 python tools/tchain_mix_stamps.py [slots]'"""
import os, sys, ctypes as C, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["UU3D_TCHAIN"] = "1"
import uplift_upsample_3dhpe_amd as pkg
from uplift_upsample_3dhpe_amd import _capi
from uplift_upsample_3dhpe_amd.synthetic import load_config, synthetic_batch
slots = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = load_config("h36m_351"); arch = pkg.arch_from_config(cfg)
model = pkg.build_uplift_upsample_transformer(cfg, weights=pkg.init_weights(arch, seed=0))
x, m = synthetic_batch(cfg, 128, seed=1000, mask_specs=[(5, 0)])
xt, mt = torch.from_numpy(x).cuda(), torch.from_numpy(m).cuda()
lib = _capi.load_library()
lib.uu3d_debug_tchain_stamps.argtypes = [C.c_void_p, C.c_int]
lib.uu3d_debug_tchain_acc.argtypes = [C.c_void_p, C.c_int]
def acc(tag):
    buf = np.zeros(32 * 4, np.uint64)
    assert lib.uu3d_debug_tchain_acc(buf.ctypes.data, 1) == 0
    a = buf.reshape(32, 4)
    for fl, name in ((4, "first: LN1 + QKV (36 chunks)"), (7, "mid (96 chunks)"), (23, "mid + pe (96 chunks)"), (9, "strided 1 head (36 chunks)")):
        if a[fl, 2]:
            print(f"  {tag}: {name}: {a[fl, 1] / a[fl, 2] / 100.0:7.1f} us per workgroup, {a[fl, 0] / a[fl, 2] / 1e3:7.1f} k cycles, clock {a[fl, 0] / a[fl, 1] / 10.0:.2f} GHz, {int(a[fl, 2])} workgroups")
def stamps(tag):
    buf = np.zeros(256 * 32, np.uint64)
    assert lib.uu3d_debug_tchain_stamps(buf.ctypes.data, buf.size) == 0
    st = buf.reshape(256, 32)[:71]
    cyc = (st[:, 18] - st[:, 0]).astype(np.float64); us = (st[:, 19] - st[:, 1]).astype(np.float64) / 100.0
    ok = (st[:, 18] > st[:, 0])
    span = (st[ok, 19].max() - st[ok, 1].min()) / 100.0
    print(f"{tag}: last chain launch: per workgroup {np.median(us[ok]):.1f} us median ({us[ok].min():.1f} .. {us[ok].max():.1f}), clock {np.median(cyc[ok] / us[ok]) / 1e3:.2f} GHz, "
          f"first start to last end {span:.1f} us, start spread {(st[ok, 1].max() - st[ok, 1].min()) / 100.0:.1f} us")
for depth in (1, slots):
    pipe = model.pipeline(128, depth=depth, graph=True)
    if depth == 1:
        pipe.close(); 
        full = torch.empty((128, 71, 17, 3), device="cuda"); cen = torch.empty((128, 17, 3), device="cuda")
        for _ in range(20): model._forward(xt, model._mask_u8(mt), full, cen, 0, torch.cuda.current_stream(), schedule=1)
        torch.cuda.synchronize(); acc("alone"); continue
    pipe.preload(xt, mt)
    torch.cuda.synchronize(); acc("(capture / warm-up, discarded)")
    tickets = []
    t0 = time.perf_counter()
    for i in range(300):
        tickets.append(pipe.launch())
        if len(tickets) == depth: pipe.result(tickets.pop(0))
    for t in tickets: pipe.result(t)
    torch.cuda.synchronize()
    print(f"{depth} slots: {(time.perf_counter() - t0) / 300 * 1e3:.4f} ms per step")
    acc(f"{depth} slots in flight")
    pipe.close()
