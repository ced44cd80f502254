"""Host-side mirror of the reference's train_step (train.py:464-506) over the C-ABI training entry points.

    trainer = Trainer(model, config)                 # uploads the model's weights into a flat master buffer
    loss = trainer.train_step(kp2d, kp3d, stride_masks, drop_path_uniform=None)   # fwd + bwd + AdamW (+ EMA)

All arithmetic runs in csrc/libuu3d.so (uu3d_train_forward_backward, uu3d_adamw_update, uu3d_ema_update);
PyTorch holds the device buffers and, with more than one rank, all-reduces the flat gradient over RCCL.
"""
import ctypes as C

import numpy as np

from . import _capi, optim
from .arch import training_unsupported
from .dist import BucketedAllReduce


class Trainer(object):

    def __init__(self, model, config, seed=0):
        import torch
        self._torch = torch
        self.model, self.config = model, config
        self._lib = model._lib
        bad = training_unsupported(model.arch)
        if bad:
            raise NotImplementedError("training with " + "; ".join(bad) + " is not implemented (the reference would train a "
                                      "different model than this step computes)")
        lib = self._lib
        self.n_params = int(lib.uu3d_num_params(model._h))
        dev = model.device
        self.params = torch.empty(self.n_params, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
        self.loss = torch.zeros(3, dtype=torch.float32, device=dev)
        self._ws = None
        self._ws_batch = 0
        self._stream = lambda: C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _capi.check(lib, lib.uu3d_train_init(model._h, C.c_void_p(self.params.data_ptr()), self._stream()), model._h)
        sched = optim.scheduler_by_name(config.SCHEDULE)
        lr = sched(**config.SCHEDULE_PARAMS)
        if config.OPTIMIZER == "AdamW":
            wd = sched(**dict(config.SCHEDULE_PARAMS, initial_learning_rate=config.WEIGHT_DECAY))   # train.py:408-411
        elif config.OPTIMIZER == "Adam":
            wd = 0.0
        else:
            raise ValueError(config.OPTIMIZER)
        extra = dict(config.OPTIMIZER_PARAMS)                 # amsgrad / beta_1 / beta_2 (/ epsilon for Adam), train.py:412-417
        unknown = set(extra) - {"amsgrad", "beta_1", "beta_2", "epsilon"}
        if unknown:
            raise NotImplementedError(f"OPTIMIZER_PARAMS {sorted(unknown)} are not implemented")
        if config.OPTIMIZER == "AdamW":
            if "epsilon" in extra:                            # tfa.optimizers.AdamW(..., epsilon=1e-8, **OPTIMIZER_PARAMS): duplicate keyword
                raise TypeError("AdamW() got multiple values for keyword argument 'epsilon'")
            extra["epsilon"] = 1e-8
        # the optimizer sees the TRAINABLE prefix of the flat buffer: BatchNorm's moving statistics (OUTPUT_BN) are the last tensors of the
        # inventory, updated by the training-mode forward itself (Keras: non-trainable weights), never decayed or stepped
        from .weights import weight_spec
        self.n_trainable = self.n_params - sum(int(np.prod(shape)) for name, shape in weight_spec(model.arch) if "/moving_" in name)
        tail = [name for name, _ in model._spec if "/moving_" in name]
        if [name for name, _ in model._spec][len(model._spec) - len(tail):] != tail:
            raise RuntimeError("the non-trainable weights are expected at the end of the inventory")
        self.optimizer = optim.AdamW(self.params[:self.n_trainable], weight_decay=wd, learning_rate=lr, **extra)
        self.ema = self.params.clone() if config.EMA_ENABLED else None
        self.global_step = 0
        self._rng = torch.Generator(device=dev)
        self._rng.manual_seed(seed)
        self.drop_path_rates = np.asarray(config.DROP_PATH_RATE if isinstance(config.DROP_PATH_RATE, list)
                                          else [config.DROP_PATH_RATE] * 3, np.float32)
        # bucketed gradient all-reduce: the library reports finished ranges of self.grads during the backward pass
        self._buckets = BucketedAllReduce(self.grads)
        # world size > 1: the skip decision of a non-finite step is taken on the REDUCED gradients (identical on every rank)
        self._skip_word = torch.zeros(1, dtype=torch.int32, device=dev)
        self._ready_cb = _capi.GRAD_READY_FN(lambda user, first, count, stream: self._buckets.ready(first, count, stream))
        _capi.check(lib, lib.uu3d_train_set_grad_callback(model._h, self._ready_cb, None), model._h)
        model._attach_trainer(self)

    def _workspace(self, batch):
        if self._ws is None or batch > self._ws_batch:
            nbytes = int(self._lib.uu3d_train_workspace_bytes(self.model._h, batch))
            self._ws = self._torch.empty(nbytes, dtype=self._torch.uint8, device=self.model.device)
            self._ws_bytes, self._ws_batch = nbytes, batch
        return self._ws

    def drop_path_size(self, batch):
        a = self.model.arch
        n = a.spatial_depth * 2 * batch * a.num_frames + a.temporal_depth * 2 * batch
        if float(self.drop_path_rates[2]) > 0.0:                              # strided blocks: two gates per sequence and block (u_u_t.py:132-137)
            n += len(a.strides) * 2 * batch
        return n

    def forward_backward(self, keypoints2d, keypoints3d, stride_masks, drop_path_uniform="draw", token_mask_uniform="draw", dropout_seed=None):
        """Training-mode forward + loss + backward.  Returns (loss[3] tensor, full, central); gradients in self.grads.

        drop_path_uniform: "draw" = fresh U[0,1) draws, None = DropPath disabled, or a flat tensor of draws.
        token_mask_uniform (TOKEN_MASK_RATE > 0, u_u_t.py:287-311): "draw", None = no token masking, or a (B, N) tensor of draws.
        dropout_seed (DROP_RATE / ATTENTION_DROP_RATE > 0): the seed of this step's Dropout masks; None = drawn from the trainer's
        generator.  ``self.last_dropout_seed`` keeps what was used."""
        torch = self._torch
        a, cfg = self.model.arch, self.config
        self.model._flush_assigns()                                          # WeightView.assign() since the last step: into the master buffer first
        B = keypoints2d.shape[0]
        dev = self.model.device
        x = keypoints2d.to(device=dev, dtype=torch.float32)
        m_ptr = None
        if self.model.has_strided_input:
            sm8 = self.model._mask_u8(stride_masks)
            x = x * sm8[:, :, None, None]                                   # train.py:474 (mask is 0 / 1)
            m_ptr = C.c_void_p(sm8.data_ptr())
        x = x.contiguous()
        gt = keypoints3d.to(device=dev, dtype=torch.float32).contiguous()
        if isinstance(drop_path_uniform, str):
            u = torch.rand(self.drop_path_size(B), generator=self._rng, device=dev, dtype=torch.float32)
        else:
            u = drop_path_uniform
        tm = None
        if a.token_mask_rate > 0.0 and token_mask_uniform is not None:
            tm = (torch.rand((B, a.num_frames), generator=self._rng, device=dev, dtype=torch.float32) if isinstance(token_mask_uniform, str)
                  else token_mask_uniform.to(device=dev, dtype=torch.float32).contiguous())
        full = torch.empty((B, a.num_frames, a.num_keypoints, 3), dtype=torch.float32, device=dev)
        central = torch.empty((B, a.num_keypoints, 3), dtype=torch.float32, device=dev)
        ws = self._workspace(B)
        rates = (C.c_float * 3)(*[float(r) for r in self.drop_path_rates])
        self.last_dropout_seed = self.model._set_dropout(self._rng, dropout_seed)
        self._buckets.begin()                                               # a previous pass without apply_gradients leaves nothing behind
        st = self._lib.uu3d_train_forward_backward_masked(
            self.model._h, C.c_void_p(self.params.data_ptr()), C.c_void_p(x.data_ptr()), m_ptr, C.c_void_p(gt.data_ptr()), B,
            int(cfg.BATCH_SIZE), float(cfg.LOSS_WEIGHT_CENTER), float(cfg.LOSS_WEIGHT_SEQUENCE), int(cfg.ROOT_KEYTPOINT),
            rates, None if u is None else C.c_void_p(u.data_ptr()),
            None if tm is None else C.c_void_p(tm.data_ptr()), float(a.token_mask_rate), C.c_void_p(self.loss.data_ptr()),
            C.c_void_p(full.data_ptr()), C.c_void_p(central.data_ptr()), C.c_void_p(self.grads.data_ptr()),
            C.c_void_p(ws.data_ptr()), self._ws_bytes, self._stream())
        _capi.check(self._lib, st, self.model._h)
        self._buckets.raise_pending()                                       # an exception inside the gradient-ready callback (ctypes only prints it)
        if a.output_bn:
            self.model._weights_dirty = True                                # the forward moved BatchNorm's running statistics inside the master buffer
        return self.loss, full, central

    def apply_gradients(self):
        """optimizer.apply_gradients (+ EMA), then refresh the operand packs.  With more than one rank the gradient buckets
        were started by forward_backward while the backward pass ran; here the stream only waits for them."""
        self._buckets.wait()                                                 # loss normaliser is the GLOBAL batch size: sums, no rescale
        # a backward pass that produced non-finite gradients (loss-scaled f16x3 overflow) leaves weights and moments alone:
        # the flag is read on the device, the host never waits (nonfinite() reads it back for logging).  With several ranks the
        # library's flag only knows THIS rank's gradients, raised before the all-reduce: a rank whose shard was finite would
        # apply the Inf / NaN sum it received and diverge from the one that skipped.  The decision is therefore
        # taken on the reduced buffer, which is bit-identical on every rank: all ranks skip or none does.
        world = self._world()
        if world > 1:
            torch = self._torch
            self._skip_word.copy_((~torch.isfinite(self.grads).all()).to(torch.int32).reshape(1))
            flag_ptr = self._skip_word.data_ptr()
        else:
            flag_ptr = self._lib.uu3d_train_nonfinite_flag(self.model._h)
        self.optimizer.apply_gradients(self.grads[:self.n_trainable], skip_flag_ptr=flag_ptr)
        if world > 1 and self.n_trainable < self.n_params:
            # OUTPUT_BN: every rank moved the running statistics with ITS shard's batch statistics; the mean over ranks keeps the
            # replicas (and with them inference weights, EMA and the sharded evaluation) identical.  Batch statistics stay per replica.
            import torch.distributed as dist
            tail = self.params[self.n_trainable:]
            dist.all_reduce(tail, op=dist.ReduceOp.SUM, group=self._buckets.group)
            tail.div_(float(world))
        if self.ema is not None:
            optim.ema_update(self.ema, self.params, optim.ema_decay_value(self.config.EMA_DECAY, self.global_step))
        _capi.check(self._lib, self._lib.uu3d_train_repack(self.model._h, C.c_void_p(self.params.data_ptr()), self._stream()), self.model._h)
        self.global_step += 1
        self.model._weights_dirty = True                                      # the model's host / inference weights are now stale
        self.model._holds_ema = False

    def _world(self):
        import torch.distributed as dist
        return dist.get_world_size(self._buckets.group) if (dist.is_available() and dist.is_initialized()) else 1

    def nonfinite(self):
        """True when the last step was skipped for non-finite gradients (world size 1: this rank's backward pass flagged them;
        more ranks: the reduced gradients were not finite -- the same answer on every rank).  Synchronises."""
        if self._world() > 1:
            return bool(int(self._skip_word.item()))
        out = C.c_int32()
        _capi.check(self._lib, self._lib.uu3d_train_nonfinite(self.model._h, C.byref(out)), self.model._h)
        return out.value != 0

    def train_step(self, keypoints2d, keypoints3d, stride_masks, drop_path_uniform="draw", token_mask_uniform="draw", dropout_seed=None):
        loss, _, _ = self.forward_backward(keypoints2d, keypoints3d, stride_masks, drop_path_uniform, token_mask_uniform, dropout_seed)
        self.apply_gradients()
        return loss

    def reload_from_model(self):
        """The model's weights were replaced (set_weights / load_weights): take them as the new master weights."""
        _capi.check(self._lib, self._lib.uu3d_train_init(self.model._h, C.c_void_p(self.params.data_ptr()), self._stream()), self.model._h)
        _capi.check(self._lib, self._lib.uu3d_train_set_grad_callback(self.model._h, self._ready_cb, None), self.model._h)

    def export_to_model(self, use_ema=False):
        """Copy the trained (or EMA) weights back into the inference model (val_model, train.py:400-401)."""
        src = self.ema if (use_ema and self.ema is not None) else self.params
        _capi.check(self._lib, self._lib.uu3d_train_export(self.model._h, C.c_void_p(src.data_ptr()), self._stream()), self.model._h)
        # The model now holds exactly what was asked for.  After an EMA export it must KEEP the EMA weights for the validation
        # forward (val_model = ema_model, train.py:398-401) -- the next apply_gradients / load_state_dict marks it stale again;
        # a "dirty" flag here made model(...) re-export the live weights over them.
        self.model._weights_dirty = False
        self.model._holds_ema = bool(use_ema and self.ema is not None)

    # ---- checkpoint / resume (the role of tf.train.Checkpoint(model, optimizer, ema_model) in train.py:420-436) ----
    def state_dict(self):
        """Everything a resumed run needs to continue bit-identically: master weights, Adam moments, iteration counters,
        EMA weights and the DropPath generator state (host numpy arrays)."""
        sd = {"params": self.params.cpu().numpy(), "adam_m": self.optimizer.m.cpu().numpy(), "adam_v": self.optimizer.v.cpu().numpy(),
              "iterations": np.int64(self.optimizer.iterations), "global_step": np.int64(self.global_step),
              "rng_state": self._rng.get_state().cpu().numpy()}
        if self.ema is not None:
            sd["ema"] = self.ema.cpu().numpy()
        if self.optimizer.amsgrad:
            sd["adam_vhat"] = self.optimizer.vhat.cpu().numpy()
        return sd

    def load_state_dict(self, sd):
        torch = self._torch
        if sd["params"].shape != (self.n_params,):
            raise ValueError(f"checkpoint holds {sd['params'].shape[0]} parameters, the model has {self.n_params}")
        self.params.copy_(torch.from_numpy(np.asarray(sd["params"], np.float32)))
        self.optimizer.m.copy_(torch.from_numpy(np.asarray(sd["adam_m"], np.float32)))
        self.optimizer.v.copy_(torch.from_numpy(np.asarray(sd["adam_v"], np.float32)))
        if self.optimizer.amsgrad:
            if "adam_vhat" not in sd:
                raise ValueError("amsgrad is on but the checkpoint has no vhat slot")
            self.optimizer.vhat.copy_(torch.from_numpy(np.asarray(sd["adam_vhat"], np.float32)))
        self.optimizer.iterations = int(sd["iterations"])
        self.global_step = int(sd["global_step"])
        self._rng.set_state(torch.from_numpy(np.asarray(sd["rng_state"], np.uint8)))
        if self.ema is not None:
            if "ema" not in sd:
                raise ValueError("EMA is enabled but the checkpoint has no EMA weights")
            self.ema.copy_(torch.from_numpy(np.asarray(sd["ema"], np.float32)))
        _capi.check(self._lib, self._lib.uu3d_train_repack(self.model._h, C.c_void_p(self.params.data_ptr()), self._stream()), self.model._h)
        self.model._weights_dirty = True
        self.model._holds_ema = False

    def save_checkpoint(self, path):
        np.savez(path, **self.state_dict())

    def load_checkpoint(self, path):
        with np.load(path) as z:
            self.load_state_dict({k: z[k] for k in z.files})

    def grads_dict(self):
        out, o = {}, 0
        flat = self.grads.cpu().numpy()
        for name, shape in self.model._spec:
            n = int(np.prod(shape))
            out[name] = flat[o:o + n].reshape(shape)
            o += n
        return out

    def params_dict(self):
        """The live master weights by name (Keras layouts), moving statistics of OUTPUT_BN included."""
        out, o = {}, 0
        flat = self.params.cpu().numpy()
        for name, shape in self.model._spec:
            n = int(np.prod(shape))
            out[name] = flat[o:o + n].reshape(shape)
            o += n
        return out
