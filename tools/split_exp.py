"""Experiment: one batch as k independent sub-batches on k streams inside one hipGraph (tails of one chain overlap the
next chain's ramp-up).  Prints ms per whole-batch step for k = 1, 2, 4."""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import uplift_upsample_3dhpe_amd as pkg
from tests import util

cfg = util.load_config("h36m_351")
B = int(os.environ.get("B", 128))
model = pkg.build_uplift_upsample_transformer(cfg, precision=os.environ.get("PREC", "f16x3"))
a = model.arch
x, m = util.synthetic_batch(cfg, B, seed=0)
x = torch.as_tensor(x).cuda(); m = torch.as_tensor(m).cuda().to(torch.uint8)
x = (x * m[:, :, None, None]).contiguous()
lib, h = model._lib, model._h
lib.uu3d_workspace_bytes.restype = C.c_size_t

def make(k):
    sizes = [B // k + (1 if i < B % k else 0) for i in range(k)]
    offs = [sum(sizes[:i]) for i in range(k)]
    full = torch.empty(B, a.num_frames, 17, 3, device="cuda"); cen = torch.empty(B, 17, 3, device="cuda")
    wss = [torch.empty(lib.uu3d_workspace_bytes(h, s) + 256, dtype=torch.uint8, device="cuda") for s in sizes]
    streams = [torch.cuda.Stream() for _ in range(k - 1)]
    def run():
        main = torch.cuda.current_stream()
        for i in range(k):
            st = main if i == 0 else streams[i - 1]
            if i: st.wait_stream(main)
        for i in range(k):
            st = main if i == 0 else streams[i - 1]
            o, s = offs[i], sizes[i]
            ws = wss[i]; wp = (ws.data_ptr() + 255) // 256 * 256
            rc = lib.uu3d_forward(h, C.c_void_p(x[o:o + s].data_ptr()), C.c_void_p(m[o:o + s].data_ptr()), s,
                                  C.c_void_p(full[o:o + s].data_ptr()), C.c_void_p(cen[o:o + s].data_ptr()),
                                  C.c_void_p(wp), C.c_size_t(ws.numel() - 256), C.c_void_p(st.cuda_stream))
            assert rc == 0, rc
        for i in range(1, k):
            main.wait_stream(streams[i - 1])
    return run, full, cen

ref = None
for k in (1, 2, 3, 4):
    run, full, cen = make(k)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            run()
        for _ in range(10): g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100): g.replay()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 10
    if ref is None: ref = (full.clone(), cen.clone())
    print(f"k={k}: {ms:.4f} ms/step  {B / ms * 1e3:.0f} seq/s   max|diff vs k=1| {float((full - ref[0]).abs().max()):.2e} {float((cen - ref[1]).abs().max()):.2e}", flush=True)
