"""What an evaluation loop reaches end to end (eval.predict_windows: window descriptors -> device gather -> flip-batched forward ->
un-flip / average), on a synthetic Human3.6M-sized table: windows/s at several pipeline depths, against bench.py's forward-only figure.
   python tools/eval_throughput_exp.py [--windows 40000] [--batch 64]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="h36m_351")
    ap.add_argument("--videos", type=int, default=40)
    ap.add_argument("--frames", type=int, default=2500)
    ap.add_argument("--batch", type=int, default=64, help="windows per batch (x 2 sequences with EVAL_FLIP)")
    args = ap.parse_args()
    import numpy as np, torch
    import uplift_upsample_3dhpe_amd as pkg
    from uplift_upsample_3dhpe_amd import synthetic as util
    from uplift_upsample_3dhpe_amd import eval as ev
    from uplift_upsample_3dhpe_amd.data import PoseTable, SequenceGenerator
    cfg = util.load_config(args.config)
    cfg.MASK_STRIDE = cfg.MASK_STRIDE[0] if isinstance(cfg.MASK_STRIDE, list) else cfg.MASK_STRIDE
    arch = pkg.arch_from_config(cfg)
    model = pkg.build_uplift_upsample_transformer(cfg, weights=pkg.init_weights(arch, seed=0))
    rng = np.random.default_rng(0)
    p2 = [rng.uniform(-1, 1, size=(args.frames, 17, 2)).astype(np.float32) for _ in range(args.videos)]
    p3 = [rng.normal(0, 0.3, size=(args.frames, 17, 3)).astype(np.float32) for _ in range(args.videos)]
    table = PoseTable(p2, p3, subjects=["S9"] * args.videos, actions=["Walking"] * args.videos, frame_rates=[50] * args.videos, device=model.device)
    gen = SequenceGenerator(table, seq_len=cfg.SEQUENCE_LENGTH, target_frame_rate=50, subsample=1, stride=cfg.SEQUENCE_STRIDE,
                            padding_type=cfg.PADDING_TYPE, flip_augment=False, flip_lr_indices=cfg.AUGM_FLIP_KEYPOINT_ORDER,
                            mask_stride=cfg.MASK_STRIDE, stride_mask_align_global=True, rand_shift_stride_mask=False, shuffle=False)
    desc = gen.descriptors()
    need = ev.needed_windows(desc[:, 1].copy(), cfg)
    run = desc[need]
    print(f"{len(desc)} windows, {len(run)} forwarded (keyframe centres), batch {args.batch} windows = {2 * args.batch} sequences with flip")
    for depth, graph in ((1, False), (1, True), (2, True), (3, True), (4, True), (8, True), (None, True)):       # None: run_eval's default (two slots per hardware queue up to 256 sequences per batch, one above)
        ev.predict_windows(model, gen, run[:args.batch * 8], cfg, args.batch, flip=True, depth=depth, graph=graph)     # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = ev.predict_windows(model, gen, run, cfg, args.batch, flip=True, depth=depth, graph=graph)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"depth {depth} graph {graph}: {len(run) / dt:9.0f} windows/s = {2 * len(run) / dt:9.0f} sequences/s ({1e3 * dt / (len(run) / args.batch):.3f} ms per batch)", flush=True)


if __name__ == "__main__":
    main()
