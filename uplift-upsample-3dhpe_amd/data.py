"""Window / stride-mask generator over pose tables that stay resident in HBM.

The reference's ``H36mSequenceGenerator`` (common/dataset/uplifiting_dataset.py:213-428) slices, pads, masks and flips
one window at a time in numpy and ships (N, J, C) copies through ``tf.data``.  Here the videos are uploaded once
(``PoseTable``), a batch is described by one small descriptor per window (``SequenceGenerator``: the same sample list,
shuffling and random draws as the reference, in the same order) and ``uu3d_gather_windows`` builds the batch on the
device: 2D windows already multiplied by their stride mask (eval.py:67, train.py:474), the masks, and the 3D targets.
"""
import ctypes as C

import numpy as np

from . import _capi


class PoseTable(object):
    """All videos back to back on the device.  poses_2d / poses_3d: lists of (F_v, J, 2) / (F_v, J, 3) arrays."""

    def __init__(self, poses_2d, poses_3d=None, subjects=None, actions=None, frame_rates=None, device=None):
        import torch
        self.torch = torch
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        main = poses_2d if poses_2d is not None else poses_3d          # AMASS: 3D sequences only, 2D comes from the camera projection
        self.lens = np.array([len(v) for v in main], np.int32)
        self.starts = np.concatenate([[0], np.cumsum(self.lens)[:-1]]).astype(np.int64)
        self.J = main[0].shape[1]
        self.kp2d = None if poses_2d is None else torch.from_numpy(np.concatenate(poses_2d, 0).astype(np.float32)).to(self.device)
        self.kp3d = None
        if poses_3d is not None:
            assert all(len(a) == len(b) for a, b in zip(main, poses_3d))
            self.kp3d = torch.from_numpy(np.concatenate(poses_3d, 0).astype(np.float32)).to(self.device)
        self.d_starts = torch.from_numpy(self.starts).to(self.device)
        self.d_lens = torch.from_numpy(self.lens).to(self.device)
        n = len(main)
        self.subjects = np.asarray(subjects if subjects is not None else np.zeros(n, np.int64))
        self.actions = np.asarray(actions if actions is not None else np.zeros(n, np.int64))
        self.frame_rates = np.asarray(frame_rates if frame_rates is not None else np.full(n, 50), np.int64)


class SequenceGenerator(object):
    """Same constructor vocabulary as the reference class (uplifiting_dataset.py:215-219)."""

    def __init__(self, table, seq_len, target_frame_rate=50, subsample=1, stride=1, padding_type="zeros",
                 flip_augment=True, in_batch_augment=False, flip_lr_indices=None, mask_stride=None,
                 stride_mask_align_global=False, rand_shift_stride_mask=False, shuffle=True, seed=0):
        if padding_type not in ("zeros", "copy"):
            raise ValueError(f"Padding type not supported: {padding_type}")
        self.table, self.seq_len, self.stride, self.subsample = table, seq_len, stride, subsample
        self.target_frame_rate = target_frame_rate
        self.pad_edge = padding_type == "copy"
        self.flip_augment, self.in_batch_augment = flip_augment, in_batch_augment
        if flip_augment and flip_lr_indices is None:
            raise ValueError("flip_augment needs flip_lr_indices")
        self.flip_lr_indices = None if flip_lr_indices is None else np.asarray(flip_lr_indices, np.int32)
        self.abs_mask_stride = mask_stride
        if mask_stride is not None:
            self.abs_mask_stride = list(mask_stride) if isinstance(mask_stride, (list, tuple)) else [mask_stride]
            for ams in self.abs_mask_stride:
                if ams < stride or ams % stride != 0:
                    raise ValueError("every mask stride must be a multiple of the sequence stride")
        self.align_global, self.rand_shift = stride_mask_align_global, rand_shift_stride_mask
        if rand_shift_stride_mask and stride_mask_align_global:
            raise ValueError("rand_shift_stride_mask excludes stride_mask_align_global")
        self.shuffle, self.seed = shuffle, seed
        self.rng = np.random.default_rng(seed)
        self.stride_shift_rng = np.random.default_rng(seed)
        self.mask_stride_rng = np.random.default_rng(seed)
        rows = []
        for s_i, n in enumerate(table.lens):
            pos = np.arange(0, n, subsample)
            z = np.zeros_like(pos)
            blk = np.stack([np.full_like(pos, s_i), pos, z, np.full_like(pos, table.frame_rates[s_i])], -1)
            if not in_batch_augment and flip_augment:
                blk = np.concatenate([blk, np.stack([blk[:, 0], pos, 1 - z, blk[:, 3]], -1)], 0)
            rows.append(blk)
        self.sequence_locations = np.concatenate(rows, 0).astype(np.int64)
        self._torch = table.torch
        self._d_flip = None if self.flip_lr_indices is None else self._torch.from_numpy(self.flip_lr_indices).to(table.device)

    def __len__(self):
        n = len(self.sequence_locations)
        return 2 * n if (self.in_batch_augment and self.flip_augment) else n

    def descriptors(self):
        """One epoch of window descriptors (W, 6) int32 = (video, centre, stride, abs mask stride, mask shift, flip), in
        the order and with the random draws of next_epoch_iterator (:303-428)."""
        locs = self.sequence_locations
        n_cams = getattr(self, "_n_cameras", None)         # AmassSequenceGenerator: one camera drawn per sample from self.rng
        if self.shuffle:
            locs = locs.copy()
            self.rng.shuffle(locs)
        else:
            if n_cams is not None:
                self.rng = np.random.default_rng(self.seed)             # uplifiting_dataset.py:549-551: deterministic cameras in eval
            self.stride_shift_rng = np.random.default_rng(self.seed)
            self.mask_stride_rng = np.random.default_rng(self.seed)
        out = []
        self.camera_indices = []
        for s_i, i, do_flip, frame_rate in locs:
            stride, mult = self.stride, 1
            if frame_rate % self.target_frame_rate != 0:
                raise ValueError("frame rate must be a multiple of the target frame rate")
            if frame_rate != self.target_frame_rate:
                mult = int(frame_rate // self.target_frame_rate)
                stride *= mult
            if self.abs_mask_stride is None:
                ams = stride
            else:
                if len(self.abs_mask_stride) == 1:
                    ams = self.abs_mask_stride[0]
                else:
                    ams = self.abs_mask_stride[self.mask_stride_rng.integers(low=0, high=len(self.abs_mask_stride), endpoint=False)]
                ams *= mult
            shift = 0
            if self.align_global:
                shift = int(i)
            elif self.rand_shift:
                r = ams // stride
                max_shift = int(np.ceil((r - 1) / 2))
                shift = int(self.stride_shift_rng.integers(low=-max_shift, high=max_shift, endpoint=(r % 2 != 0))) * stride
            out.append((s_i, i, stride, ams, shift, int(do_flip)))
            if n_cams is not None:
                self.camera_indices.append(int(self.rng.integers(low=0, high=n_cams, size=1)[0]))
            if self.in_batch_augment and self.flip_augment:
                out.append((s_i, i, stride, ams, shift, 1 - int(do_flip)))
                if n_cams is not None:
                    self.camera_indices.append(self.camera_indices[-1])      # the flipped copy keeps the sample's camera
        return np.array(out, dtype=np.int32).reshape(-1, 6)

    def gather(self, desc, zero_masked=True, with_3d=True, out=None, stream=None):
        """Build one batch on the device from (B, 6) descriptors -> dict of device tensors (+ host metadata).

        ``out = (kp2d_buf, stride_mask_buf)``: write the 2D windows / stride masks into these tensors ((B, N, J, 2) float32 and (B, N)
        uint8 or None) instead of fresh ones -- e.g. a ``pipeline.ForwardPipeline`` slot's static input buffers (``acquire``).
        ``stream``: the torch stream the gather runs on (default: the current one); the descriptor upload is registered with it."""
        torch, t = self._torch, self.table
        lib = _capi.load_library()
        desc = np.ascontiguousarray(desc, np.int32)
        B, N, J = len(desc), self.seq_len, t.J
        tstream = stream if stream is not None else torch.cuda.current_stream(t.device)
        with torch.cuda.stream(tstream):
            # pinned staging + asynchronous copy (round 5): a pageable upload blocks the HOST until the stream reaches it -- behind the running
            # forward of whichever slot shares the stream's hardware queue (eight slots on four queues: 141 k instead of 171 k sequences/s end to end)
            d_desc = torch.from_numpy(desc).pin_memory().to(t.device, non_blocking=True)
            stream = tstream.cuda_stream
            kp2d = out[0] if out is not None else torch.empty((B, N, J, 2), dtype=torch.float32, device=t.device)
            smask = out[1] if (out is not None and out[1] is not None) else torch.empty((B, N), dtype=torch.uint8, device=t.device)
            pmask = torch.empty((B, N), dtype=torch.uint8, device=t.device)
        if tuple(kp2d.shape) != (B, N, J, 2) or kp2d.dtype != torch.float32 or not kp2d.is_contiguous() or \
                tuple(smask.shape) != (B, N) or smask.dtype != torch.uint8 or not smask.is_contiguous():
            raise ValueError("out buffers must be contiguous (B, N, J, 2) float32 and (B, N) uint8")
        fl = C.c_void_p(self._d_flip.data_ptr()) if self._d_flip is not None else None
        st = lib.uu3d_gather_windows(C.c_void_p(t.kp2d.data_ptr()), C.c_void_p(t.d_starts.data_ptr()), C.c_void_p(t.d_lens.data_ptr()),
                                     C.c_void_p(d_desc.data_ptr()), fl, B, N, J, 2, int(self.pad_edge), int(zero_masked),
                                     C.c_void_p(kp2d.data_ptr()), C.c_void_p(smask.data_ptr()), C.c_void_p(pmask.data_ptr()),
                                     C.c_void_p(stream))
        _capi.check(lib, st, None)
        out = {"kp2d": kp2d, "stride_mask": smask, "mask": pmask, "subjects": t.subjects[desc[:, 0]],
               "actions": t.actions[desc[:, 0]], "index": desc[:, 1].copy()}
        if with_3d and t.kp3d is not None:
            with torch.cuda.stream(tstream):                            # (allocated under the stream that writes them: ADVICE round 4)
                kp3d = torch.empty((B, N, J, 3), dtype=torch.float32, device=t.device)
                dummy = torch.empty((B, N), dtype=torch.uint8, device=t.device)
            st = lib.uu3d_gather_windows(C.c_void_p(t.kp3d.data_ptr()), C.c_void_p(t.d_starts.data_ptr()), C.c_void_p(t.d_lens.data_ptr()),
                                         C.c_void_p(d_desc.data_ptr()), fl, B, N, J, 3, int(self.pad_edge), 0,
                                         C.c_void_p(kp3d.data_ptr()), C.c_void_p(dummy.data_ptr()), None, C.c_void_p(stream))
            _capi.check(lib, st, None)
            out["kp3d"] = kp3d
        return out

    def batches(self, batch_size, drop_remainder=False, **kw):
        desc = self.descriptors()
        for b in range(0, len(desc), batch_size):
            blk = desc[b:b + batch_size]
            if drop_remainder and len(blk) < batch_size:
                break
            yield self.gather(blk, **kw)


class AmassSequenceGenerator(SequenceGenerator):
    """``AMASSSequenceGenerator`` (uplifiting_dataset.py:431-661): 3D windows of world-frame sequences, each with one of the
    Human3.6M cameras drawn at random (``amass.camera_table``); the 2D input and the camera-frame target come from
    ``world_to_cam_and_2d`` afterwards.  A flip mirrors the pose sequence only, never the camera (:641-650)."""

    def __init__(self, table, cameras, seq_len, **kw):
        super().__init__(table, seq_len, **kw)
        self.cameras = np.ascontiguousarray(cameras, np.float32)
        self._n_cameras = len(self.cameras)

    def descriptors(self):
        """As the base class, plus ``self.camera_indices``.  Reference quirk kept for parity: without in-batch augmentation
        the reference tests ``if do_flip is True`` on a numpy bool (uplifiting_dataset.py:583,641), which is never true, so
        the samples listed as flipped come out UNFLIPPED (duplicates); only the in-batch copy (:654-659) is mirrored."""
        desc = super().descriptors()
        if not self.in_batch_augment:
            desc[:, 5] = 0
        return desc

    def gather(self, desc, camera_indices=None):
        """-> {"kp3d" (B, N, J, 3) world frame, "cams" (B, 18), "stride_mask", "mask", "index"} on the device."""
        torch, t = self._torch, self.table
        lib = _capi.load_library()
        desc = np.ascontiguousarray(desc, np.int32)
        B, N, J = len(desc), self.seq_len, t.J
        d_desc = torch.from_numpy(desc).to(t.device)
        stream = torch.cuda.current_stream(t.device).cuda_stream
        kp3d = torch.empty((B, N, J, 3), dtype=torch.float32, device=t.device)
        smask = torch.empty((B, N), dtype=torch.uint8, device=t.device)
        pmask = torch.empty((B, N), dtype=torch.uint8, device=t.device)
        fl = C.c_void_p(self._d_flip.data_ptr()) if self._d_flip is not None else None
        st = lib.uu3d_gather_windows(C.c_void_p(t.kp3d.data_ptr()), C.c_void_p(t.d_starts.data_ptr()), C.c_void_p(t.d_lens.data_ptr()),
                                     C.c_void_p(d_desc.data_ptr()), fl, B, N, J, 3, int(self.pad_edge), 0,
                                     C.c_void_p(kp3d.data_ptr()), C.c_void_p(smask.data_ptr()), C.c_void_p(pmask.data_ptr()),
                                     C.c_void_p(stream))
        _capi.check(lib, st, None)
        out = {"kp3d": kp3d, "stride_mask": smask, "mask": pmask, "index": desc[:, 1].copy()}
        if camera_indices is not None:
            out["cams"] = torch.from_numpy(self.cameras[np.asarray(camera_indices, np.int64)]).to(t.device)
        return out

    def batches(self, batch_size, drop_remainder=False):
        desc = self.descriptors()
        cams = np.asarray(self.camera_indices, np.int64)
        for b in range(0, len(desc), batch_size):
            if drop_remainder and len(desc) - b < batch_size:
                break
            yield self.gather(desc[b:b + batch_size], cams[b:b + batch_size])


def world_to_cam_and_2d(sequences_3d, cams):
    """tf_world_to_cam_and_2d (uplifiting_dataset.py:669-761) for a batch on the device: sequences_3d (B, N, J, 3) world
    coordinates, cams (B, 18) = orientation quaternion | translation | 11 intrinsics (``amass.camera_table``; the reference
    documents "4+3+11" and slices ``cam[7:19]``, which on an 18-vector is those 11) -> (camera-space 3D (B, N, J, 3),
    projected 2D (B, N, J, 2))."""
    import torch
    lib = _capi.load_library()
    x = sequences_3d.to(torch.float32).contiguous()
    c = cams.to(device=x.device, dtype=torch.float32).contiguous()
    B, N, J = x.shape[0], x.shape[1], x.shape[2]
    if tuple(c.shape) != (B, 18) or x.shape[3] != 3:
        raise ValueError("sequences_3d must be (B, N, J, 3) and cams (B, 18)")
    cam3d = torch.empty_like(x)
    kp2d = torch.empty((B, N, J, 2), dtype=torch.float32, device=x.device)
    st = lib.uu3d_world_to_cam_2d(C.c_void_p(x.data_ptr()), C.c_void_p(c.data_ptr()), B, N, J, C.c_void_p(cam3d.data_ptr()),
                                  C.c_void_p(kp2d.data_ptr()), C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    _capi.check(lib, st, None)
    return cam3d, kp2d
