"""Host-side mirror of the reference's Keras model object
(``common/net/uplift_upsample_transformer.py:163-421``) over the C-ABI HIP library.

    full, central = model([x, stride_mask], training=False)      # has_strided_input
    full, central = model(x, training=False)                     # otherwise

``x``: (B, N, J, 2) float32 torch tensor on the model's ROCm device (the caller zeroes
masked frames, eval.py:67); ``stride_mask``: (B, N) bool/uint8, 1 = real input present.
Returns ``full`` (B, N, J, 3) (``None`` when the reference would return None, :399-404)
and ``central`` (B, J, 3).  ``training=True`` (train.py:478) runs the training-mode forward:
DropPath with draws from the model's (or its Trainer's) generator, on the LIVE weights of
the Trainer when one is attached.  ``model.weights`` / ``model.trainable_variables``
(train.py:496,503) are name -> array views in the reference's creation order.
PyTorch is used for device memory and streams only; every FLOP runs in
``csrc/libuu3d.so``.  There is no CPU fallback.
"""
import ctypes as C

import numpy as np

from .. import _capi
from ..arch import UpliftArch
from ..weights import init_weights, weight_spec


class WeightView(object):
    """One entry of ``model.weights`` / ``model.trainable_variables``: the role a tf.Variable plays at train.py:496,503
    (``name``, ``shape``, ``numpy()``, ``assign(value)``) over the model's current weights."""

    def __init__(self, model, name, shape):
        self._model, self.name, self.shape = model, name, tuple(shape)

    def numpy(self):
        return self._model._get_one(self.name, self.shape)

    def assign(self, value):
        """tf.Variable.assign: ONE uu3d_set_weight; the operand packs (and an attached Trainer's master buffer) are refreshed once,
        by the next call that reads the weights -- a loop over model.weights (train.py:503) stays linear."""
        self._model._assign_one(self.name, self.shape, value)

    def assign_sub(self, delta):
        """tf.Variable.assign_sub, the call the reference's EMA update makes (train.py:503: ema_w.assign_sub((1 - d) * (ema_w - w)))."""
        self._model._assign_one(self.name, self.shape, self.numpy() - np.asarray(delta, np.float32).reshape(self.shape))

    def __repr__(self):
        return f"<WeightView {self.name} {self.shape}>"


class UpliftUpsampleTransformer(object):

    # with concurrent_halves=True a batch of at least this many sequences runs as two independent half batches on two
    # HIP streams.  That was +1..7 % while every kernel left CUs idle; since the LayerNorm-fed Dense layers run on the
    # row-panel GEMM (one workgroup per CU, a launch fills the chip) a single chain is 2.5-6 % FASTER (h36m_351 batch 128:
    # 111.8 k vs 108.6 k sequences/s, h36m_81 batch 256: 230 k vs 221 k), so it is off by default.
    SPLIT_MIN_BATCH = 64

    def __init__(self, arch: UpliftArch, device=None, seed=0, weights=None, return_attention=False, precision="f16x3",
                 concurrent_halves=False, range_guard=True):
        import torch
        # return_attention=True (u_u_t.py:176,418-419; never used by the reference's scripts): model(...) returns (full, central, att_list),
        # att_list = the (B, heads, N, N) softmax weights of every temporal block, recomputed by a separate kernel (uu3d_forward_attention)
        self.return_attention = bool(return_attention)
        if not torch.cuda.is_available():
            raise _capi.Uu3dLibraryError("no ROCm device visible: the uplift path has no CPU fallback")
        self._torch = torch
        self.arch = arch
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._lib = _capi.load_library()
        # attributes the reference's callers read (eval.py:66,171)
        self.full_output = arch.full_output
        self.has_strided_input = arch.has_strided_input
        self.num_frames = arch.num_frames
        self._returns_full = arch.full_output and arch.temporal_depth > 0

        cfg = _capi.Uu3dConfig()
        cfg.num_frames, cfg.num_keypoints = arch.num_frames, arch.num_keypoints
        cfg.d_spatial, cfg.d_temporal = arch.d_spatial, arch.d_temporal
        cfg.h_spatial, cfg.h_temporal = arch.h_spatial, arch.h_temporal
        cfg.spatial_depth, cfg.temporal_depth = arch.spatial_depth, arch.temporal_depth
        if len(arch.strides) > _capi.UU3D_MAX_STRIDED:
            raise ValueError("too many strided blocks")
        cfg.num_strided = len(arch.strides)
        for i, (s, p) in enumerate(zip(arch.strides, arch.paddings)):
            cfg.strides[i], cfg.pad_left[i], cfg.pad_right[i] = s, p[0], p[1]
        cfg.num_heads = arch.num_heads
        cfg.qkv_bias = int(arch.qkv_bias)
        cfg.has_strided_input = int(arch.has_strided_input)
        cfg.first_strided_token_attention_layer = arch.first_strided_token_attention_layer
        cfg.full_output = int(arch.full_output)
        cfg.output_bn = int(bool(getattr(arch, "output_bn", False)))
        cfg.learnable_masked_token = int(bool(getattr(arch, "learnable_masked_token", False)))
        # "f16x3": forward GEMMs as three f16 MFMA passes on hi/lo-split operands (f32-grade error);
        # "f32": exact f32-input MFMA everywhere
        if precision not in ("f32", "f16x3"):
            raise ValueError("precision must be 'f32' or 'f16x3'")
        self.precision = precision
        # (The sticky range word is ONE per model: a guarded model(...) call must not run concurrently with a pipeline or another thread's call on
        # the same handle -- it would take, and clear, their flag.  uu3d_range_status reads and clears it in one atomic exchange.)
        # range_guard (precision f16x3): model(...) checks its outputs after the call (one stream synchronisation) and repeats a batch whose
        # activations left the f16 range on the exact-f32 kernels -- include/uu3d.h, RANGE CONTRACT; False: the caller checks (check_range())
        self.range_guard = bool(range_guard)
        self._range_warned = False
        cfg.precision = _capi.UU3D_PREC_F16X3 if precision == "f16x3" else _capi.UU3D_PREC_F32
        handle = C.c_void_p()
        st = self._lib.uu3d_create(C.byref(cfg), self.device.index or 0, C.byref(handle))
        _capi.check(self._lib, st, None)
        self._h = handle
        self._ws = {}
        self._trainer = None              # a trainer.Trainer owns the live weights once attached
        self._weights_dirty = False       # live (trainer) weights newer than the host / inference copies
        self._holds_ema = False           # Trainer.export_to_model(use_ema=True): the model holds the EMA weights until the next step
        self._train_params = None         # master buffer of a training-mode forward without a Trainer
        self._train_ws = None
        self._seed = seed
        self._rng = None
        self._profiling = False
        self._halves = bool(concurrent_halves)
        self._side_stream = None
        self._spec = self._query_spec()
        expected = [(n, tuple(s)) for n, s in weight_spec(arch)]
        if self._spec != expected:
            raise RuntimeError("weight inventory of libuu3d.so disagrees with weights.weight_spec")
        self.set_weights_dict(weights if weights is not None else init_weights(arch, seed=seed))

    # ---- weights (model.weights / get_weights / set_weights analogues) ----------------------
    def _query_spec(self):
        n = self._lib.uu3d_num_weights(self._h)
        out = []
        for i in range(n):
            name = C.c_char_p()
            ndim = C.c_int32()
            dims = (C.c_int64 * 4)()
            _capi.check(self._lib, self._lib.uu3d_weight_info(self._h, i, C.byref(name), C.byref(ndim), C.byref(dims)), self._h)
            out.append((name.value.decode(), tuple(int(dims[k]) for k in range(ndim.value))))
        return out

    @property
    def weight_names(self):
        return [n for n, _ in self._spec]

    # ---- live weights of an attached Trainer -------------------------------------------------
    def _attach_trainer(self, trainer):
        self._trainer, self._train_params = trainer, None

    def _sync_from_trainer(self):
        """save_weights / get_weights / inference calls see the trained weights (train.py:393,706,719 use the live model)."""
        if self._trainer is not None and self._weights_dirty:
            self._trainer.export_to_model()
        self._flush_assigns()

    # ---- single-variable access (WeightView) ----------------------------------------------------
    def _get_one(self, name, shape):
        if self._trainer is not None and self._weights_dirty:   # (a model that holds the EMA weights reads them: that is what was asked for)
            self._trainer.export_to_model()
        a = np.empty(shape, np.float32)
        _capi.check(self._lib, self._lib.uu3d_get_weight(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), a.size), self._h)
        return a

    def _assign_one(self, name, shape, value):
        # the other variables keep their TRAINED values: a stale model is refreshed first, and so is a model that holds the EMA
        # weights after Trainer.export_to_model(use_ema=True) -- otherwise the flush below would reload the master buffer with
        # "EMA weights + the one assigned tensor" and the trained weights would be gone
        if self._trainer is not None and (self._weights_dirty or self._holds_ema):
            self._trainer.export_to_model()
        a = np.ascontiguousarray(np.asarray(value, dtype=np.float32).reshape(shape))
        _capi.check(self._lib, self._lib.uu3d_set_weight(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), a.size), self._h)
        self._pending_assigns = True

    def _flush_assigns(self):
        """Deferred half of WeightView.assign: one commit (+ one reload of an attached Trainer's master buffer) for any number of
        assigned variables."""
        if getattr(self, "_pending_assigns", False):
            self._pending_assigns = False
            self._commit()
            self._holds_ema = False
            if self._trainer is not None:
                self._trainer.reload_from_model()
            self._train_params = None

    @property
    def weights(self):
        return [WeightView(self, n, s) for n, s in self._spec]

    @property
    def trainable_variables(self):
        """model.trainable_variables: everything but the BatchNorm moving statistics of OUTPUT_BN."""
        return [v for v in self.weights if not v.name.endswith(("/moving_mean", "/moving_variance"))]

    def set_weights_dict(self, weights):
        for name, shape in self._spec:
            if name not in weights:
                raise ValueError(f"missing weight {name}")
            a = np.ascontiguousarray(np.asarray(weights[name], dtype=np.float32))
            if tuple(a.shape) != shape:
                raise ValueError(f"weight {name}: shape {a.shape} != {shape}")
            st = self._lib.uu3d_set_weight(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), a.size)
            _capi.check(self._lib, st, self._h)
        self._commit()
        self._weights_dirty = False
        self._pending_assigns = False
        self._holds_ema = False
        if self._trainer is not None:       # the trainer's master buffer follows (Adam moments are kept, like tf.Variable.assign)
            self._trainer.reload_from_model()
        self._train_params = None

    def set_weights(self, weight_list):
        if len(weight_list) != len(self._spec):
            raise ValueError(f"expected {len(self._spec)} arrays, got {len(weight_list)}")
        self.set_weights_dict({n: w for (n, _), w in zip(self._spec, weight_list)})

    def get_weights_dict(self):
        self._sync_from_trainer()
        out = {}
        for name, shape in self._spec:
            a = np.empty(shape, np.float32)
            st = self._lib.uu3d_get_weight(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), a.size)
            _capi.check(self._lib, st, self._h)
            out[name] = a
        return out

    def get_weights(self):
        d = self.get_weights_dict()
        return [d[n] for n, _ in self._spec]

    def save_weights(self, filepath):
        """``model.save_weights("x.h5")`` (train.py:706,719): Keras HDF5 weight file, top-level layer names of the reference."""
        from ..utils import weight_io
        weight_io.save_keras_h5(filepath, self.get_weights_dict(), self._spec)

    def load_weights(self, filepath, skip_mismatch=False, callbacks=(), verbose=False):
        """By-name / by-position load of a Keras ``.h5`` (reference ``weight_io.load_weights_with_callback``)."""
        from ..utils import weight_io
        return weight_io.load_weights_with_callback(self, filepath, skip_mismatch=skip_mismatch, callbacks=callbacks, verbose=verbose)

    def _commit(self):
        torch = self._torch
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream().cuda_stream
            _capi.check(self._lib, self._lib.uu3d_commit_weights(self._h, C.c_void_p(stream)), self._h)

    # ---- forward -----------------------------------------------------------------------------
    def _workspace(self, batch, slot=0):
        """One workspace per concurrently running forward (slot 0 / 1), grown on demand."""
        ws = self._ws.get(slot)
        nbytes = int(self._lib.uu3d_workspace_bytes(self._h, batch))
        if ws is None or ws.numel() < nbytes:
            ws = self._torch.empty(nbytes, dtype=self._torch.uint8, device=self.device)
            self._ws[slot] = ws
        return ws

    def _forward(self, x, stride_mask, full, central, slot, stream, attn=None, schedule=0, exact_f32=False):
        """One uu3d_forward_ex call.  ``schedule`` (include/uu3d.h: 0 = latency, 1 = throughput) is an argument of THIS call;
        ``exact_f32``: this call on the exact-f32 kernels whatever the model's precision (UU3D_SCHEDULE_EXACT_F32)."""
        if exact_f32:
            schedule = int(schedule) | _capi.UU3D_SCHEDULE_EXACT_F32
        B = x.shape[0]
        ws = self._workspace(B, slot)
        ptrs = None
        if attn is not None:
            ptrs = (C.c_void_p * max(len(attn), 1))(*[t.data_ptr() for t in attn])
        st = self._lib.uu3d_forward_ex(self._h, C.c_void_p(x.data_ptr()),
                                       C.c_void_p(stride_mask.data_ptr()) if stride_mask is not None else None, B,
                                       C.c_void_p(full.data_ptr()) if full is not None else None,
                                       C.c_void_p(central.data_ptr()), ptrs, C.c_void_p(ws.data_ptr()),
                                       C.c_size_t(ws.numel()), int(schedule), C.c_void_p(stream.cuda_stream))
        _capi.check(self._lib, st, self._h)

    # ---- range guard of precision f16x3 (include/uu3d.h, RANGE CONTRACT) -------------------------
    def check_range(self, stream=None, raise_error=True):
        """Synchronises ``stream`` (default: the current one) and reports whether any f16x3 forward since the last check produced non-finite
        outputs (activations beyond the f16 range 65504, or non-finite inputs): raises ``Uu3dRangeError`` / returns True.  Clears the flag."""
        torch = self._torch
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        flag = C.c_int32(0)
        st = self._lib.uu3d_range_status(self._h, C.c_void_p(s.cuda_stream), C.byref(flag))
        if st not in (_capi.UU3D_OK, _capi.UU3D_ERR_RANGE):
            _capi.check(self._lib, st, self._h)
        if flag.value and raise_error:
            raise _capi.Uu3dRangeError(_capi.UU3D_ERR_RANGE, self._lib.uu3d_last_error(self._h).decode())
        return bool(flag.value)

    def _exact_f32_available(self):
        return bool(self.arch.compiled_dims) and self.arch.num_frames <= 128

    def _mask_u8(self, stride_mask):
        """(B, N) bool / uint8 mask as uint8 bytes on the model's device.  A bool tensor is reinterpreted (torch bools are
        one byte, 0 / 1): no conversion kernel inside the forward."""
        torch = self._torch
        m = stride_mask
        if m.device != self.device:
            m = m.to(self.device)
        if m.dtype == torch.bool:
            return m.contiguous().view(torch.uint8)
        if m.dtype == torch.uint8:
            return m.contiguous()
        return (m != 0).contiguous().view(torch.uint8)

    def _set_dropout(self, rng, seed=None):
        """The Dropout layers of this training-mode call (uu3d_train_set_dropout): the config's rates and a fresh seed of the mask
        stream drawn from ``rng`` (or the given ``seed``); with both rates 0 nothing is drawn."""
        a = self.arch
        if a.drop_rate > 0.0 or a.attention_drop_rate > 0.0:
            if seed is None:
                seed = int(self._torch.randint(0, 2 ** 62, (1,), generator=rng, device=self.device, dtype=self._torch.int64).item())
        else:
            seed = 0
        _capi.check(self._lib, self._lib.uu3d_train_set_dropout(self._h, float(a.drop_rate), float(a.attention_drop_rate), int(seed)), self._h)
        self.last_dropout_seed = seed                           # (what a test needs to evaluate the same masks on the CPU)
        return seed

    def _training_forward(self, x, stride_mask, full, central):
        """model(inputs, training=True) (train.py:478): DropPath active, live weights.  Without a Trainer the model keeps
        its own master buffer (uu3d_train_init) and generator."""
        from ..arch import training_unsupported
        torch = self._torch
        bad = training_unsupported(self.arch)
        if bad:
            raise NotImplementedError("training=True with " + "; ".join(bad) + " is not implemented")
        a, B = self.arch, x.shape[0]
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        if self._trainer is not None:
            params, rng, rates = self._trainer.params, self._trainer._rng, self._trainer.drop_path_rates
        else:
            if self._train_params is None:
                self._train_params = torch.empty(int(self._lib.uu3d_num_params(self._h)), dtype=torch.float32, device=self.device)
                _capi.check(self._lib, self._lib.uu3d_train_init(self._h, C.c_void_p(self._train_params.data_ptr()), stream), self._h)
            if self._rng is None:
                self._rng = torch.Generator(device=self.device)
                self._rng.manual_seed(int(self._seed))
            params, rng, rates = self._train_params, self._rng, np.asarray(a.drop_path_rate, np.float32)
        n_draws = a.spatial_depth * 2 * B * a.num_frames + a.temporal_depth * 2 * B
        if float(rates[2]) > 0.0:
            n_draws += len(a.strides) * 2 * B
        u = torch.rand(n_draws, generator=rng, device=self.device, dtype=torch.float32)
        nbytes = int(self._lib.uu3d_train_workspace_bytes(self._h, B))
        if self._train_ws is None or self._train_ws.numel() < nbytes:
            self._train_ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        r3 = (C.c_float * 3)(*[float(r) for r in rates])
        tm = torch.rand((B, a.num_frames), generator=rng, device=self.device, dtype=torch.float32) if a.token_mask_rate > 0.0 else None   # u_u_t.py:299
        self._set_dropout(rng)
        st = self._lib.uu3d_train_forward_backward_masked(
            self._h, C.c_void_p(params.data_ptr()), C.c_void_p(x.data_ptr()),
            C.c_void_p(stride_mask.data_ptr()) if stride_mask is not None else None, None, B, int(a.batch_size), 0.0, 0.0, 0,
            r3, C.c_void_p(u.data_ptr()), C.c_void_p(tm.data_ptr()) if tm is not None else None, float(a.token_mask_rate),
            None, C.c_void_p(full.data_ptr()) if full is not None else None,
            C.c_void_p(central.data_ptr()), None, C.c_void_p(self._train_ws.data_ptr()), self._train_ws.numel(), stream)
        _capi.check(self._lib, st, self._h)

    def __call__(self, inputs, training=None, mask=None):
        torch = self._torch
        if self.has_strided_input:
            x, stride_mask = inputs[0], inputs[1]
        else:
            x, stride_mask = inputs, None
        a = self.arch
        if x.dim() != 4 or tuple(x.shape[1:]) != (a.num_frames, a.num_keypoints, 2):
            raise ValueError(f"x must be (B, {a.num_frames}, {a.num_keypoints}, 2), got {tuple(x.shape)}")
        if x.device != self.device:
            raise ValueError(f"x is on {x.device}, model is on {self.device}")
        B = x.shape[0]
        x = x.to(torch.float32).contiguous()
        if stride_mask is not None:
            if tuple(stride_mask.shape) != (B, a.num_frames):
                raise ValueError(f"stride_mask must be (B, {a.num_frames})")
            stride_mask = self._mask_u8(stride_mask)
        full = torch.empty((B, a.num_frames, a.num_keypoints, 3), dtype=torch.float32, device=self.device) \
            if self._returns_full else None
        central = torch.empty((B, a.num_keypoints, 3), dtype=torch.float32, device=self.device)
        if training:
            if self.return_attention:
                raise NotImplementedError("return_attention=True with training=True")
            self._flush_assigns()
            self._training_forward(x, stride_mask, full, central)
            return full, central
        self._sync_from_trainer()
        main = torch.cuda.current_stream(self.device)
        # Range guard (include/uu3d.h): the f16x3 products cannot represent activations of magnitude >= 65504 (the reference is float32 end to
        # end).  A direct call checks its own result -- one stream synchronisation, skipped inside a stream capture and with range_guard=False --
        # and repeats an overflowed batch on the exact-f32 kernels; where those do not exist it raises.  Pipelines check once (check_range()).
        guard = self.range_guard and self.precision == "f16x3" and not torch.cuda.is_current_stream_capturing()

        def guarded(run):
            run(False)
            if guard and self.check_range(main, raise_error=False):
                if not self._exact_f32_available():
                    raise _capi.Uu3dRangeError(_capi.UU3D_ERR_RANGE, "activations beyond the f16 range of the f16x3 products (or non-finite inputs), and "
                                               "this model has no exact-f32 forward to fall back to (generic dims or > 128 tokens)")
                if not self._range_warned:
                    self._range_warned = True
                    import warnings
                    warnings.warn("uu3d: a batch left the f16 range of the f16x3 products (|activation| >= 65504) and was repeated on the exact-f32 "
                                  "kernels; build the model with precision='f32' if this is the rule for these weights", RuntimeWarning)
                run(True)
                torch.cuda.current_stream(self.device).synchronize()
                if not bool(torch.isfinite(central).all()) or (full is not None and not bool(torch.isfinite(full).all())):
                    raise _capi.Uu3dRangeError(_capi.UU3D_ERR_RANGE, "non-finite outputs even in exact f32: the inputs or weights are not finite")
        if self.return_attention:
            att = [torch.empty((B, a.num_heads, a.num_frames, a.num_frames), dtype=torch.float32, device=self.device)
                   for _ in range(a.temporal_depth)]
            guarded(lambda f32: self._forward(x, stride_mask, full, central, 0, main, attn=att, exact_f32=f32))
            return full, central, att
        if not (self._halves and B >= self.SPLIT_MIN_BATCH and not self._profiling):
            guarded(lambda f32: self._forward(x, stride_mask, full, central, 0, main, exact_f32=f32))
            return full, central
        # Sequences are independent, so the batch runs as two chains of kernels that the GPU interleaves: the
        # ramp-up and tail of one chain's (short, 10-50 us) kernels overlap the other chain's work.  Measured
        # +7 % sequences/s at B = 128 (tools/split_exp.py); three or four chains are slower again.  The fork and
        # join are stream waits, so the call is still capturable into a hipGraph.
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=self.device)
        side = self._side_stream
        h = (B + 1) // 2

        def run_halves(f32):
            side.wait_stream(main)
            self._forward(x[:h], stride_mask[:h] if stride_mask is not None else None,
                          full[:h] if full is not None else None, central[:h], 0, main, exact_f32=f32)
            self._forward(x[h:], stride_mask[h:] if stride_mask is not None else None,
                          full[h:] if full is not None else None, central[h:], 1, side, exact_f32=f32)
            main.wait_stream(side)
        guarded(run_halves)                                # (the range guard covers both halves: the sticky word is the model's)
        return full, central

    def call_scheduled(self, inputs, schedule):
        """One inference call on the current stream under a NAMED launch schedule of include/uu3d.h: "latency" (what ``model(...)`` runs) or
        "throughput" (what a pipeline's slots run: launches shaped for CU-microseconds, the temporal chain from 1024 token rows on) ->
        (full, central).  No range guard (``check_range()`` afterwards).  For profiling and for tests that compare a pipeline with a quiet call."""
        torch = self._torch
        sched = {"latency": 0, "throughput": 1}[schedule]
        x, stride_mask = (inputs[0], inputs[1]) if self.has_strided_input else (inputs, None)
        a = self.arch
        B = x.shape[0]
        x = x.to(torch.float32).contiguous()
        if stride_mask is not None:
            stride_mask = self._mask_u8(stride_mask)
        self._sync_from_trainer()
        full = torch.empty((B, a.num_frames, a.num_keypoints, 3), dtype=torch.float32, device=self.device) if self._returns_full else None
        central = torch.empty((B, a.num_keypoints, 3), dtype=torch.float32, device=self.device)
        self._forward(x, stride_mask, full, central, 0, torch.cuda.current_stream(self.device), schedule=sched)
        return full, central

    # ---- graph replay / several batches in flight ------------------------------------------------
    def pipeline(self, batch, depth=None, graph=True, post=None):
        """``depth`` independent batches in flight on ``depth`` HIP streams, each replaying its own hipGraph of the forward
        (pipeline.ForwardPipeline): the throughput path of an evaluation loop (eval.py:147-152)."""
        from ..pipeline import ForwardPipeline
        return ForwardPipeline(self, batch, depth=depth, graph=graph, post=post)

    def capture(self, batch):
        """One forward at a fixed batch size as a hipGraph: ``f = model.capture(128); full, central = f([x, mask])`` replays
        it (one launch instead of ~50; outputs are static buffers, overwritten by the next call)."""
        pipe = self.pipeline(batch, depth=1, graph=True)

        def replay(inputs):
            x, m = (inputs[0], inputs[1]) if self.has_strided_input else (inputs, None)
            return pipe.result(pipe.submit(x, m))
        replay.pipeline = pipe
        return replay

    def set_profiling(self, enabled):
        self._profiling = bool(enabled)          # per-launch events time one chain, so the batch is not split
        _capi.check(self._lib, self._lib.uu3d_set_profiling(self._h, int(bool(enabled))), self._h)

    def read_profile(self):
        n = C.c_int32()
        _capi.check(self._lib, self._lib.uu3d_profile_read(self._h, None, 0, C.byref(n)), self._h)
        arr = (_capi.Uu3dProfileEntry * max(n.value, 1))()
        _capi.check(self._lib, self._lib.uu3d_profile_read(self._h, arr, n.value, C.byref(n)), self._h)
        return [dict(name=e.name.decode(), kernel=e.kernel.decode(), ms=float(e.ms), flops=float(e.flops),
                     bytes=float(e.bytes)) for e in arr[:n.value]]

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.uu3d_destroy(self._h)
                self._h = None
        except Exception:
            pass
