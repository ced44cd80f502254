"""What the strided stack costs the pipelined step (round 5): the bench's loop on config/h36m_351.json as shipped, without its strided blocks (STRIDES = []: the central
output comes from the temporal stack's middle token), and with one temporal block less -- same batch, same slots.   python tools/tail_cost_exp.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import uplift_upsample_3dhpe_amd as pkg
from uplift_upsample_3dhpe_amd import synthetic as util

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200


def run(label, mutate):
    cfg = util.load_config("h36m_351")
    mutate(cfg)
    arch = pkg.arch_from_config(cfg)
    model = pkg.build_uplift_upsample_transformer(cfg, weights=pkg.init_weights(arch, seed=0))
    x_np, m_np = util.synthetic_batch(cfg, 128, seed=1000, mask_specs=[(5, 0)])
    x = torch.from_numpy(x_np * m_np[:, :, None, None].astype(np.float32)).cuda(); m = torch.from_numpy(m_np).cuda()
    pipe = model.pipeline(128)
    pipe.preload(x, m if model.has_strided_input else None)

    def loop(n):
        t = []
        for _ in range(n):
            t.append(pipe.launch(wait_caller=False))
            if len(t) == pipe.depth:
                pipe.after(t.pop(0), lambda *a: None)
        for k in t:
            pipe.after(k, lambda *a: None)
        pipe.join()
    loop(24); torch.cuda.synchronize()
    out = []
    for _ in range(3):
        t0 = time.perf_counter(); loop(steps); torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / steps * 1e3)
    print(f"{label:52s} ms per step " + " ".join(f"{v:.4f}" for v in out), flush=True)
    pipe.close()


def no_strided(c): c.STRIDES, c.PADDINGS = [], []
def three_temporal(c): c.TEMPORAL_TRANSFORMER_BLOCKS = 3
def one_strided(c): c.STRIDES, c.PADDINGS = [3], [[1, 1]] if False else None
run("config/h36m_351.json as shipped", lambda c: None)
run("without the strided blocks (STRIDES = [])", no_strided)
run("three temporal blocks instead of four", three_temporal)
run("config/h36m_351.json as shipped", lambda c: None)
