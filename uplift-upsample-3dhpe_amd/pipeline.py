"""Several independent batches in flight: the inference loop of eval.py:147-152 (one ``model(...)`` call per batch, each
waiting for the previous one) as a software pipeline over HIP streams.

A forward of this model is a chain of ~50 dependent launches whose last third (strided blocks 2-3, the heads: ~160 us of short,
latency-bound kernels on a few CUs) leaves the chip almost idle; batches are independent, so batch k + 1's spatial stack and
temporal blocks can run beside batch k's tail.  ``ForwardPipeline`` keeps ``depth`` batches in flight, each on its own stream with
its own workspace, static input / output buffers and (optionally) its own hipGraph of the whole forward:

    pipe = ForwardPipeline(model, batch=128)                   # model.pipeline(128): two slots per hardware queue (8)
    for full, central in pipe.run(batches):                    # batches: iterable of (x, stride_mask) or x
        ...                                                    # outputs are valid until `depth` more batches were submitted

How many batches in flight, and on which streams: HIP deals a process's streams to 4 hardware queues (GPU_MAX_HW_QUEUES) at their FIRST USE, in an
order that is not the creation order (24 pool streams on one box: 0 1 2 2 1 0 3 2 1 0 3 2 ...), two streams on one queue run in order, and what the
pipeline reaches depends on how evenly its slots are dealt over the queues (`tools/queue_map_exp.py`, h36m_351, batch 128, one box, round 5 with the
temporal chain): 1 / 2 / 3 / 4 slots on as many queues = 128-130 / 146-153 / 180 / 182-187 k sequences/s; eight or twelve slots dealt evenly
(ABCDABCD) 186-192 k; the same eight in pool order (queue classes 0 1 2 2 1 0 3 2) 160 k.  Raising GPU_MAX_HW_QUEUES does not help (6 / 8 / 12 queues:
118-145 k, round 4).  So the default (``depth=None``) asks ``distinct_queue_streams`` for two streams of every hardware queue: it finds out which
streams share a queue by blocking one with a spin kernel and timing a tiny kernel on the other with HIP events (~100 ms once per process; the
probe's kernel and every stream are used once BEFORE they are timed, and a positive has to repeat -- a first use looks like queueing, and one
misfiled stream is an uneven deal).  What runs is four forwards at a time in lock step, one per queue (`tools/fill_drain_exp.py`: 20 steps = five
rounds of ~2.7 ms), the second slot of a queue keeping it fed; the consumer's work belongs on the slot's stream (``after``), not behind a wait on
the caller's (``result``), whose queue a quarter of the slots share.
Results are bit-identical to ONE quiet call under the same schedule (depth 1: the latency schedule = ``model(...)``; more slots: the throughput
schedule, ``model.call_scheduled(inputs, "throughput")`` -- from 1024 token rows on that is the temporal chain, within 3e-5 of ``model(...)``): the
same launches on the same data, only on another stream.  Latency of ONE batch does not improve (0.9 ms); use ``model(...)`` for that.
"""
import os
import ctypes as C


_QUEUE_STREAMS = {}


def distinct_queue_streams(device, want=4, pool=16, spin_ms=2.0, per_queue=1):
    """Up to ``want`` torch streams that sit on pairwise DIFFERENT HIP hardware queues (found once per process and device).

    Two streams share a queue iff work on one waits for work on the other: a spin kernel of ``spin_ms`` goes to stream a, a tiny kernel to
    stream b, and HIP events tell when b's kernel ran -- behind the spin (same queue) or at once.  Streams come from torch's pool; the
    first of every new queue class is kept.  ``per_queue`` > 1 (round 5): that many streams of EVERY class, returned class-major
    (q0 q1 q2 q3 q0 q1 q2 q3): slots dealt over them share the hardware queues evenly -- an uneven deal loses 10 % (module docstring)."""
    import torch
    key = (str(device), want, per_queue)
    if key in _QUEUE_STREAMS:
        return list(_QUEUE_STREAMS[key])
    torch.cuda.synchronize(device)
    with torch.cuda.device(device):
        tiny = torch.zeros(64, device=device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cycles = 4_000_000                                      # calibrate the spin (its clock is not the shader clock)
        e0.record(); torch.cuda._sleep(cycles); e1.record(); e1.synchronize()
        per_ms = cycles / max(e0.elapsed_time(e1), 1e-3)

        def same_queue(a, b):
            t0, ta, tb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            b.wait_stream(a)                                    # (both idle and ordered before the probe)
            with torch.cuda.stream(a):
                t0.record(a)
                torch.cuda._sleep(int(spin_ms * per_ms))
                ta.record(a)
            with torch.cuda.stream(b):
                tiny.add_(1.0)
                tb.record(b)
            ta.synchronize(); tb.synchronize()
            return t0.elapsed_time(tb) > 0.5 * t0.elapsed_time(ta)

        tiny.add_(1.0)                                          # (first launch of the probe's kernel: module load, not queueing)
        torch.cuda.synchronize(device)
        reps, members = [], []
        for _ in range(pool * per_queue):
            s = torch.cuda.Stream(device=device)
            with torch.cuda.stream(s):
                tiny.add_(1.0)                                  # (first use of the stream: not queueing either)
            s.synchronize()
            # a late tiny kernel for any other reason reads as "same queue", never the other way round: a positive has to repeat
            cls = next((i for i, r in enumerate(reps) if same_queue(r, s) and same_queue(r, s)), None)
            if cls is None:
                if len(reps) < want:
                    reps.append(s); members.append([s])
            elif len(members[cls]) < per_queue:
                members[cls].append(s)
            if len(reps) == want and all(len(mm) == per_queue for mm in members):
                break
        torch.cuda.synchronize(device)
    n = min(len(mm) for mm in members) if members else 0                  # (a class that never got its share: fall back to what every class has)
    out = [mm[k] for k in range(max(n, 1)) for mm in members if k < len(mm)]
    _QUEUE_STREAMS[key] = list(out)
    return list(out)


class _Slot(object):
    __slots__ = ("stream", "x", "m", "full", "central", "graph", "done", "busy", "n", "extra")


class ForwardPipeline(object):

    def __init__(self, model, batch, depth=None, graph=True, post=None, streams=None):
        """``depth``: batches in flight; None = TWO per HIP hardware queue for batches up to 128 sequences, one above (``distinct_queue_streams(per_queue=2)``: 8 / 4 -- round 5: with the temporal
        chain's launches of one workgroup per 128 rows, eight forwards in flight keep the chip full where four do not: 187 k against 181 k
        sequences/s at batch 128; the round-4 launches measure the same with four and eight), an int up to that number takes that many of
        those streams (class-major: the first four sit on four different queues), more falls back to fresh pool streams.
        ``post(full, central, slot_index)``: optional device work appended to every forward ON THE SLOT'S STREAM (and into its
        graph), e.g. the per-joint error kernel; what it returns is handed out by ``result`` as a third element.
        ``streams``: the slots' streams (``depth`` of them), whatever queues they are on."""
        import torch
        if depth is not None and depth < 1:
            raise ValueError("depth >= 1")
        if streams is None and (depth is None or depth > 1):
            q = distinct_queue_streams(model.device, want=int(os.environ.get("UU3D_PIPE_QUEUES", "4")), per_queue=2)
            if depth is None:
                # two slots per queue for batches up to 128 sequences, one above (forward only, 2 / 4 / 8 slots -- 256 per batch: 186 / 198 / 191 k sequences/s,
                # 512: 197 / 204 / 198 k, 1024: 203 / 208 / 203 k; at 128 four and eight measure the same unless the consumer waits on the caller's stream)
                depth = len(q) if int(batch) <= 128 else max(1, len(q) // 2)
            if depth <= len(q):
                streams = q[:depth]
        if streams is not None and depth is None:
            depth = len(streams)
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
            s.stream = streams[i] if streams is not None else torch.cuda.Stream(device=dev)
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
        # launches shaped for CU-microseconds, not latency, when the forwards in flight share the chip: an ARGUMENT of the call
        # (include/uu3d.h, uu3d_forward_ex), so a model(...) call on another thread keeps its own schedule
        self.model._forward(s.x[:n], s.m[:n] if s.m is not None else None, s.full[:n] if s.full is not None else None,
                            s.central[:n], self._keys[i], s.stream, schedule=1 if self.depth > 1 else 0)
        if self.post is not None:
            s.extra = self.post(s.full[:n] if s.full is not None else None, s.central[:n], i)

    def _next_slot(self, n):
        model = self.model
        i = self._submitted % self.depth
        s = self._slots[i]
        if s.busy:
            raise RuntimeError("slot still holds an unread result: call result() for the oldest ticket first")
        if n > self.batch:
            raise ValueError(f"batch of {n} > pipeline batch {self.batch}")
        if model._weights_dirty or getattr(model, "_pending_assigns", False):
            self.drain()
            model._sync_from_trainer()                              # (weights changed: the packs are rewritten on the caller's stream)
            cur = self._torch.cuda.current_stream(model.device)
            for sl in self._slots:                                  # (... which every slot waits for, whatever wait_caller says)
                sl.stream.wait_stream(cur)
        return i, s

    def acquire(self, n=None, wait_caller=True):
        """The next slot's STATIC input buffers, for a producer that writes the batch in place (no copy, no temporaries):

            x_buf, m_buf, stream = pipe.acquire(n)      # (n, N, J, 2) float32, (n, N) uint8 or None, the slot's torch stream
            generator.gather(desc, out=(x_buf, m_buf), stream=stream)
            ticket = pipe.launch(n)

        Everything that fills the buffers must be enqueued on ``stream`` (or on a stream ``stream`` has been made to wait for) before
        ``launch``.  ``stream`` already waits for the caller's current stream at this point (``wait_caller=False``: it does not -- for a
        producer whose inputs do not depend on anything the caller's stream has enqueued; see ``launch``).  The slot's previous forward has
        finished reading the buffers as far as ``stream`` is concerned (same stream: in order)."""
        torch = self._torch
        n = self.batch if n is None else int(n)
        i, s = self._next_slot(n)
        if wait_caller:
            s.stream.wait_stream(torch.cuda.current_stream(self.model.device))
        return s.x[:n], (s.m[:n] if s.m is not None else None), s.stream

    def preload(self, x, stride_mask=None):
        """Copy one batch into EVERY slot's static input buffers (e.g. a benchmark that forwards the same resident batch again and
        again: ``preload`` once, then ``launch()`` per step -- no per-step input copy)."""
        torch = self._torch
        n = int(x.shape[0])
        cur = torch.cuda.current_stream(self.model.device)
        for s in self._slots:
            s.stream.wait_stream(cur)
            with torch.cuda.stream(s.stream):
                s.x[:n].copy_(x, non_blocking=True)
                if s.m is not None:
                    s.m[:n].copy_(self.model._mask_u8(stride_mask), non_blocking=True)
            cur.wait_stream(s.stream)                               # (the sources may be dropped / rewritten by the caller afterwards)

    def launch(self, n=None, wait_caller=True):
        """Enqueue the forward of the next slot on the buffers it holds (``acquire`` + a producer, or ``preload``): ``n`` sequences.
        The slot's stream first waits for the caller's current stream (which may still read the slot's previous outputs).  Returns a ticket.
        ``wait_caller=False`` (round 5) skips that wait -- for loops whose consumer runs on the slot's stream (``after``) and whose inputs are
        resident or written on the slot's stream: the wait is an event on the caller's stream, i.e. a marker packet in the hardware queue that
        stream shares with a quarter of the slots, BEHIND their running forwards -- every launch then waits for that queue's current forward."""
        torch = self._torch
        n = self.batch if n is None else int(n)
        i, s = self._next_slot(n)
        if wait_caller:
            s.stream.wait_stream(torch.cuda.current_stream(self.model.device))
        with torch.cuda.stream(s.stream):
            if s.graph is not None and n == self.batch:
                s.graph.replay()
            else:
                self._launch(i, n)
            s.done.record(s.stream)
        s.busy, s.n = True, n
        t = self._submitted
        self._submitted += 1
        return t

    def submit(self, x, stride_mask=None):
        """Enqueue one batch (n <= batch sequences) given as tensors: they are copied into the slot's static buffers on the slot's
        stream, after everything the caller's current stream has enqueued so far.  Returns a ticket for ``result``.  The copies
        run on ANOTHER stream than the one the tensors were allocated on, so they are registered with the caching allocator
        (``record_stream``): a temporary that the caller drops right after ``submit`` is not handed out again before the copy ran."""
        model = self.model
        n = int(x.shape[0])
        if model.has_strided_input != (stride_mask is not None):
            raise ValueError("stride_mask must be given iff the model has strided input")
        xb, mb, stream = self.acquire(n)
        torch = self._torch
        with torch.cuda.stream(stream):
            if x.is_cuda:
                x.record_stream(stream)
            xb.copy_(x, non_blocking=True)
            if mb is not None:
                mu = model._mask_u8(stride_mask)
                if mu.is_cuda:
                    mu.record_stream(stream)
                    if stride_mask.is_cuda and mu.data_ptr() != stride_mask.data_ptr():
                        stride_mask.record_stream(stream)
                mb.copy_(mu, non_blocking=True)
        return self.launch(n)

    def check_range(self):
        """Range guard of precision f16x3 (include/uu3d.h, RANGE CONTRACT) for everything submitted so far: waits for every slot, then raises
        ``Uu3dRangeError`` if a forward produced non-finite outputs (activations beyond the f16 range, or non-finite inputs).  Pipelines do
        not check per batch (that would be a host synchronisation per batch); callers check once -- ``eval.predict_windows`` at its end."""
        for s in self._slots:
            s.stream.synchronize()
        return self.model.check_range()

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

    def after(self, ticket, fn):
        """Follow-up device work for a submitted batch ON ITS SLOT'S STREAM: ``fn(full, central[, post's value])`` runs with the slot's stream
        current, i.e. what it enqueues runs behind the slot's forward, in order, with no wait between streams -- returns fn's value and frees
        the slot for its next submit (same stream: the next forward runs behind fn's work).  The alternative to ``result`` for consumers that
        only move the outputs on (eval.predict_windows' copy into its result table, bench.py's copy of the error block): ``result`` makes the
        CALLER'S stream wait for the slot, and a wait on the caller's stream is a barrier packet in the hardware queue that stream shares with
        a quarter of the slots -- every forward queued behind it on that queue stalls (round 5, h36m_351 batch 128, eight slots: 164 k
        sequences/s with the consumer's copy on the caller's stream, 186 k with it on the slot's).  ``join()`` before the caller's stream
        (or the host) reads what fn wrote."""
        torch = self._torch
        if not (self._submitted - self.depth <= ticket < self._submitted):
            raise ValueError("ticket is not in flight")
        s = self._slots[ticket % self.depth]
        if not s.busy:
            raise RuntimeError("result already taken")
        s.busy = False
        out = (s.full[:s.n] if s.full is not None else None, s.central[:s.n])
        with torch.cuda.stream(s.stream):
            return fn(*(out + ((s.extra,) if self.post is not None else ())))

    def join(self):
        """The caller's current stream waits for everything enqueued on the slots' streams so far (forwards and ``after`` work)."""
        cur = self._torch.cuda.current_stream(self.model.device)
        for s in self._slots:
            cur.wait_stream(s.stream)

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
        """The caller's stream waits for everything on the slots' streams (forwards and ``after`` work); unread results are dropped."""
        cur = self._torch.cuda.current_stream(self.model.device)
        for s in self._slots:
            cur.wait_stream(s.stream)
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
