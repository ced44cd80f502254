"""Do the parallel branches of a captured hipGraph run concurrently?  (round 5: the premise of deferring a forward's few-row tail into a branch of the slot's next launch)
   python tools/graph_branch_exp.py"""
import time, torch
dev = torch.device("cuda:0")
torch.cuda.synchronize()
cycles = 2_000_000
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(cycles); e1.record(); e1.synchronize()
one = e0.elapsed_time(e1)
print(f"one spin: {one:.3f} ms")
main, side = torch.cuda.Stream(), torch.cuda.Stream()
for nb in (1, 2, 3):
    sides = [torch.cuda.Stream() for _ in range(nb - 1)]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=main):
        for s in sides:
            s.wait_stream(main)
        torch.cuda._sleep(cycles)
        for s in sides:
            with torch.cuda.stream(s):
                torch.cuda._sleep(cycles)
        for s in sides:
            main.wait_stream(s)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    print(f"graph with {nb} parallel spin branch(es): {(time.perf_counter() - t0) * 100:.3f} ms per replay ({one:.3f} = concurrent, {nb * one:.3f} = serial)")
# two graphs on two streams, each with 2 branches: do four branches overlap?
