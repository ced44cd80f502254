"""Several independent batches in flight: the inference loop of eval.py:147-152 (one ``model(...)`` call per batch, each
waiting for the previous one) as a software pipeline over HIP streams.

A forward of this model is a chain of ~50 dependent launches whose last third (strided blocks 2-3, the heads: ~160 us of short,
latency-bound kernels on a few CUs) leaves the chip almost idle; batches are independent, so batch k + 1's spatial stack and
temporal blocks can run beside batch k's tail.  ``ForwardPipeline`` keeps ``depth`` batches in flight, each on its own stream with
its own workspace, static input / output buffers and (optionally) its own hipGraph of the whole forward:

    pipe = ForwardPipeline(model, batch=128, depth=2)          # model.pipeline(128)
    for full, central in pipe.run(batches):                    # batches: iterable of (x, stride_mask) or x
        ...                                                    # outputs are valid until `depth` more batches were submitted

Measured on MI355X, h36m_351, batch 128, hipGraph replay (one box each; box to box +-10 %): `bench.py` (s_in = 5: every frame real)
1 / 2 / 3 / 4 / 6 / 8 batches in flight = 126 / 162 / 166-169 / 152 / 167-169 / 155 k sequences/s; `tools/streams_exp.py` (same workload,
another process layout) 2 / 3 / 4 / 6 = 166 / 151 / 170 / 168 k.  The gain is +28-34 %; the depths that lose ~10 % differ between the two
programs -- HIP deals streams to its 4 hardware queues in creation order, and a depth whose slots collide on a queue serialises them --
hence ``tune_depth`` below (bench.py's default): try 2, 3, 4 and 6 slots, keep the fastest pipeline object.  h36m_81 batch 256: 305 /
322-324 / 326 k at 2 / 3 / 6.
Results are bit-identical to ``model(...)``: the same launches on the same data, only on another
stream.  Latency of ONE batch does not improve (0.9 ms); use ``model(...)`` for that.
"""
import ctypes as C


class _Slot(object):
    __slots__ = ("stream", "x", "m", "full", "central", "graph", "done", "busy", "n", "extra")


class ForwardPipeline(object):

    def __init__(self, model, batch, depth=2, graph=True, post=None):
        """``post(full, central, slot_index)``: optional device work appended to every forward ON THE SLOT'S STREAM (and into its
        graph), e.g. the per-joint error kernel; what it returns is handed out by ``result`` as a third element."""
        import torch
        if depth < 1:
            raise ValueError("depth >= 1")
        self._torch = torch
        self.model, self.batch, self.depth, self.post = model, int(batch), int(depth), post
        a = model.arch
        dev = model.device
        self._slots = []
        self._submitted = 0
        model._sync_from_trainer()
        # workspaces of its own (keys in the model's workspace table): a model(...) call on the caller's stream and a pipeline slot
        # must never share scratch memory -- they run on different streams
        self._keys = [("pipeline", id(self), i) for i in range(depth)]
        cur = torch.cuda.current_stream(dev)
        for i in range(depth):
            s = _Slot()
            s.stream = torch.cuda.Stream(device=dev)
            s.stream.wait_stream(cur)
            s.x = torch.zeros((batch, a.num_frames, a.num_keypoints, 2), dtype=torch.float32, device=dev)
            s.m = torch.ones((batch, a.num_frames), dtype=torch.uint8, device=dev) if model.has_strided_input else None
            s.full = torch.empty((batch, a.num_frames, a.num_keypoints, 3), dtype=torch.float32, device=dev) if model._returns_full else None
            s.central = torch.empty((batch, a.num_keypoints, 3), dtype=torch.float32, device=dev)
            s.graph, s.done, s.busy, s.n, s.extra = None, torch.cuda.Event(), False, 0, None
            self._slots.append(s)
        if graph:
            for i, s in enumerate(self._slots):
                with torch.cuda.stream(s.stream):
                    self._launch(i, batch)                         # warm-up outside the capture (lazy attribute calls, allocations)
                s.stream.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s.stream):
                    self._launch(i, batch)
                s.graph = g
            torch.cuda.synchronize(dev)

    def _launch(self, i, n):
        s = self._slots[i]
        lib, h = self.model._lib, self.model._h
        # launches shaped for CU-microseconds, not latency: the forwards in flight share the chip (include/uu3d.h, uu3d_set_schedule)
        if self.depth > 1:
            lib.uu3d_set_schedule(h, 1)
        try:
            self.model._forward(s.x[:n], s.m[:n] if s.m is not None else None, s.full[:n] if s.full is not None else None,
                                s.central[:n], self._keys[i], s.stream)
        finally:
            if self.depth > 1:
                lib.uu3d_set_schedule(h, 0)
        if self.post is not None:
            s.extra = self.post(s.full[:n] if s.full is not None else None, s.central[:n], i)

    def submit(self, x, stride_mask=None):
        """Enqueue one batch (n <= batch sequences).  Returns a ticket for ``result``.  The inputs are read on the slot's stream
        after everything the caller's current stream has enqueued so far."""
        torch = self._torch
        model = self.model
        i = self._submitted % self.depth
        s = self._slots[i]
        if s.busy:
            raise RuntimeError("slot still holds an unread result: call result() for the oldest ticket first")
        n = int(x.shape[0])
        if n > self.batch:
            raise ValueError(f"batch of {n} > pipeline batch {self.batch}")
        if model.has_strided_input != (stride_mask is not None):
            raise ValueError("stride_mask must be given iff the model has strided input")
        if model._weights_dirty or getattr(model, "_pending_assigns", False):
            self.drain()
            model._sync_from_trainer()                              # (weights changed: the packs are rewritten on the caller's stream)
        cur = torch.cuda.current_stream(model.device)
        s.stream.wait_stream(cur)
        with torch.cuda.stream(s.stream):
            s.x[:n].copy_(x, non_blocking=True)
            if s.m is not None:
                s.m[:n].copy_(model._mask_u8(stride_mask), non_blocking=True)
            if s.graph is not None and n == self.batch:
                s.graph.replay()
            else:
                self._launch(i, n)
            s.done.record(s.stream)
        s.busy, s.n = True, n
        t = self._submitted
        self._submitted += 1
        return t

    def result(self, ticket):
        """(full, central[, post's value]) of a submitted batch; the caller's current stream waits for it.  The tensors are the
        slot's static buffers: valid until ``depth`` more batches have been submitted."""
        torch = self._torch
        if not (self._submitted - self.depth <= ticket < self._submitted):
            raise ValueError("ticket is not in flight")
        s = self._slots[ticket % self.depth]
        if not s.busy:
            raise RuntimeError("result already taken")
        torch.cuda.current_stream(self.model.device).wait_event(s.done)
        s.busy = False
        out = (s.full[:s.n] if s.full is not None else None, s.central[:s.n])
        return out + ((s.extra,) if self.post is not None else ())

    def close(self):
        """Wait for everything in flight and give the slots' workspaces back."""
        if getattr(self, "_slots", None):
            for s in self._slots:
                s.stream.synchronize()
                s.graph = None
            for k in self._keys:
                self.model._ws.pop(k, None)
            self._slots = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def drain(self):
        for s in self._slots:
            if s.busy:
                self._torch.cuda.current_stream(self.model.device).wait_event(s.done)
                s.busy = False

    def run(self, batches):
        """Generator over the results of ``batches`` (items: ``(x, stride_mask)`` or ``x``), in order, ``depth`` in flight."""
        pending = []
        for item in batches:
            x, m = item if isinstance(item, (tuple, list)) else (item, None)
            pending.append(self.submit(x, m))
            if len(pending) == self.depth:
                yield self.result(pending.pop(0))
        while pending:
            yield self.result(pending.pop(0))


def tune_depth(model, batch, candidates=(2, 3, 4, 6), steps=40, graph=True, post=None):
    """Pick the number of slots by trying them: HIP deals streams to its hardware queues in creation order, and which depths collide on
    a queue (and lose ~10 %) depends on what else the process has created -- measured, not predictable.  Builds
    one pipeline per candidate, runs ``steps`` forwards of its (zero) static inputs, keeps the fastest and closes the others: the
    pipeline that is returned is the very object that was measured (same streams, same queues).
    Returns ``(pipeline, {depth: seconds per step})``."""
    import time
    import torch
    timings, best = {}, None
    for d in candidates:
        pipe = ForwardPipeline(model, batch, depth=d, graph=graph, post=post)
        x, m = pipe._slots[0].x, pipe._slots[0].m

        def run(n):
            t = []
            for _ in range(n):
                t.append(pipe.submit(x, m))
                if len(t) == d:
                    pipe.result(t.pop(0))
            for q in t:
                pipe.result(q)
        run(max(4, steps // 4))
        torch.cuda.synchronize(model.device)
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize(model.device)
        timings[d] = (time.perf_counter() - t0) / steps
        if best is None or timings[d] < timings[best.depth]:
            if best is not None:
                best.close()
            best = pipe
        else:
            pipe.close()
    return best, timings
