"""ctypes binding of ``include/uu3d.h`` (the C-ABI shared library ``csrc/libuu3d.so``).

There is NO fallback: if the HIP library is missing or fails to load, importing a symbol
raises ``Uu3dLibraryError`` -- the product path never computes on the CPU.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (UU3D_LIB: another build of the same library, e.g. csrc/libuu3d_timing.so of `build.py --timing` for the marginal-cost experiments of tools/)
LIB_PATH = os.environ.get("UU3D_LIB") or os.path.join(_HERE, "csrc", "libuu3d.so")

UU3D_MAX_STRIDED = 8
UU3D_PREC_F32 = 0
UU3D_PREC_F16X3 = 1

(UU3D_OK, UU3D_ERR_INVALID_ARGUMENT, UU3D_ERR_UNSUPPORTED, UU3D_ERR_SHAPE, UU3D_ERR_NOT_READY,
 UU3D_ERR_WORKSPACE, UU3D_ERR_HIP, UU3D_ERR_NO_DEVICE, UU3D_ERR_RANGE) = range(9)
UU3D_SCHEDULE_LATENCY, UU3D_SCHEDULE_THROUGHPUT, UU3D_SCHEDULE_EXACT_F32 = 0, 1, 0x100

# every symbol include/uu3d.h declares
EXPORTED_SYMBOLS = (
    "uu3d_version", "uu3d_status_string", "uu3d_last_error", "uu3d_create", "uu3d_destroy",
    "uu3d_num_weights", "uu3d_weight_info", "uu3d_set_weight", "uu3d_get_weight",
    "uu3d_commit_weights", "uu3d_workspace_bytes", "uu3d_forward", "uu3d_forward_attention", "uu3d_forward_ex", "uu3d_range_status", "uu3d_mpjpe",
    "uu3d_set_schedule", "uu3d_set_profiling", "uu3d_profile_read", "uu3d_gather_windows", "uu3d_world_to_cam_2d",
    "uu3d_mpjpe_loss", "uu3d_adamw_update", "uu3d_adamw_update_guarded", "uu3d_train_nonfinite_flag", "uu3d_train_nonfinite", "uu3d_ema_update",
    "uu3d_num_params", "uu3d_train_init", "uu3d_train_repack", "uu3d_train_export",
    "uu3d_train_workspace_bytes", "uu3d_train_forward_backward", "uu3d_train_forward_backward_masked", "uu3d_train_set_grad_callback", "uu3d_train_set_dropout",
)
# include/uu3d_ops.h
OPS_SYMBOLS = (
    "uu3d_op_gemm_tn", "uu3d_op_gemm_tn_h3", "uu3d_op_gemm_nt", "uu3d_op_colsum", "uu3d_op_row_stats", "uu3d_op_ln_bwd",
    "uu3d_op_attn_fwd", "uu3d_op_attn_bwd", "uu3d_op_scratch_floats",
    "uu3d_op_panel_operand_bytes", "uu3d_op_panel_a_bytes", "uu3d_op_panel_pack", "uu3d_op_ln_dense_panel",
)


class Uu3dLibraryError(RuntimeError):
    pass


class Uu3dError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"uu3d status {status}: {message}")
        self.status = status


class Uu3dRangeError(Uu3dError):
    """A forward produced non-finite outputs: activations left the f16 range of the f16x3 products (include/uu3d.h, RANGE CONTRACT)."""


class Uu3dConfig(C.Structure):
    _fields_ = [
        ("num_frames", C.c_int32), ("num_keypoints", C.c_int32),
        ("d_spatial", C.c_int32), ("d_temporal", C.c_int32),
        ("h_spatial", C.c_int32), ("h_temporal", C.c_int32),
        ("spatial_depth", C.c_int32), ("temporal_depth", C.c_int32),
        ("num_strided", C.c_int32),
        ("strides", C.c_int32 * UU3D_MAX_STRIDED),
        ("pad_left", C.c_int32 * UU3D_MAX_STRIDED),
        ("pad_right", C.c_int32 * UU3D_MAX_STRIDED),
        ("num_heads", C.c_int32), ("qkv_bias", C.c_int32), ("has_strided_input", C.c_int32),
        ("first_strided_token_attention_layer", C.c_int32), ("full_output", C.c_int32),
        ("precision", C.c_int32), ("output_bn", C.c_int32), ("learnable_masked_token", C.c_int32),
    ]


class Uu3dProfileEntry(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("kernel", C.c_char * 32), ("ms", C.c_float),
                ("flops", C.c_double), ("bytes", C.c_double)]


# void (*uu3d_grad_ready_fn)(void* user, int64_t first, int64_t count, void* stream)
GRAD_READY_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p)

_lib = None


def load_library(path=None):
    """dlopen the HIP library and declare every prototype.  Raises if it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise Uu3dLibraryError(
            f"{p} not found: build it with `python __graft_entry__.py build` "
            f"(hipcc --offload-arch=gfx950); there is no CPU fallback")
    try:
        lib = C.CDLL(p)
    except OSError as e:  # pragma: no cover
        raise Uu3dLibraryError(f"cannot load {p}: {e}") from e
    vp, i32, i64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t
    lib.uu3d_version.restype = C.c_char_p
    lib.uu3d_version.argtypes = []
    lib.uu3d_status_string.restype = C.c_char_p
    lib.uu3d_status_string.argtypes = [C.c_int]
    lib.uu3d_last_error.restype = C.c_char_p
    lib.uu3d_last_error.argtypes = [vp]
    lib.uu3d_create.restype = C.c_int
    lib.uu3d_create.argtypes = [C.POINTER(Uu3dConfig), C.c_int, C.POINTER(vp)]
    lib.uu3d_destroy.restype = None
    lib.uu3d_destroy.argtypes = [vp]
    lib.uu3d_num_weights.restype = C.c_int
    lib.uu3d_num_weights.argtypes = [vp]
    lib.uu3d_weight_info.restype = C.c_int
    lib.uu3d_weight_info.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(i32), C.POINTER(i64 * 4)]
    lib.uu3d_set_weight.restype = C.c_int
    lib.uu3d_set_weight.argtypes = [vp, C.c_char_p, vp, i64]
    lib.uu3d_get_weight.restype = C.c_int
    lib.uu3d_get_weight.argtypes = [vp, C.c_char_p, vp, i64]
    lib.uu3d_commit_weights.restype = C.c_int
    lib.uu3d_commit_weights.argtypes = [vp, vp]
    lib.uu3d_workspace_bytes.restype = sz
    lib.uu3d_workspace_bytes.argtypes = [vp, i32]
    lib.uu3d_forward.restype = C.c_int
    lib.uu3d_forward.argtypes = [vp, vp, vp, i32, vp, vp, vp, sz, vp]
    lib.uu3d_forward_attention.restype = C.c_int
    lib.uu3d_forward_attention.argtypes = [vp, vp, vp, i32, vp, vp, C.POINTER(vp), vp, sz, vp]
    lib.uu3d_forward_ex.restype = C.c_int
    lib.uu3d_forward_ex.argtypes = [vp, vp, vp, i32, vp, vp, C.POINTER(vp), vp, sz, i32, vp]
    lib.uu3d_range_status.restype = C.c_int
    lib.uu3d_range_status.argtypes = [vp, vp, C.POINTER(i32)]
    lib.uu3d_mpjpe.restype = C.c_int
    lib.uu3d_mpjpe.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    lib.uu3d_gather_windows.restype = C.c_int
    lib.uu3d_gather_windows.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp]
    lib.uu3d_world_to_cam_2d.restype = C.c_int
    lib.uu3d_world_to_cam_2d.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp]
    lib.uu3d_set_schedule.restype = C.c_int
    lib.uu3d_set_schedule.argtypes = [vp, i32]
    lib.uu3d_set_profiling.restype = C.c_int
    lib.uu3d_set_profiling.argtypes = [vp, i32]
    lib.uu3d_profile_read.restype = C.c_int
    lib.uu3d_profile_read.argtypes = [vp, C.POINTER(Uu3dProfileEntry), i32, C.POINTER(i32)]
    lib.uu3d_mpjpe_loss.restype = C.c_int
    lib.uu3d_mpjpe_loss.argtypes = [vp, vp, vp, i32, i32, i32, i32, C.c_float, C.c_float, i32, vp, vp, vp, vp, vp]
    lib.uu3d_adamw_update.restype = C.c_int
    lib.uu3d_adamw_update.argtypes = [vp, vp, vp, vp, vp, i64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, i64, vp]
    lib.uu3d_adamw_update_guarded.restype = C.c_int
    lib.uu3d_adamw_update_guarded.argtypes = [vp, vp, vp, vp, vp, i64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, i64, vp, vp]
    lib.uu3d_train_nonfinite_flag.restype = vp
    lib.uu3d_train_nonfinite_flag.argtypes = [vp]
    lib.uu3d_train_nonfinite.restype = C.c_int
    lib.uu3d_train_nonfinite.argtypes = [vp, C.POINTER(C.c_int32)]
    lib.uu3d_ema_update.restype = C.c_int
    lib.uu3d_ema_update.argtypes = [vp, vp, i64, C.c_float, vp]
    lib.uu3d_num_params.restype = i64
    lib.uu3d_num_params.argtypes = [vp]
    lib.uu3d_train_init.restype = C.c_int
    lib.uu3d_train_init.argtypes = [vp, vp, vp]
    lib.uu3d_train_repack.restype = C.c_int
    lib.uu3d_train_repack.argtypes = [vp, vp, vp]
    lib.uu3d_train_export.restype = C.c_int
    lib.uu3d_train_export.argtypes = [vp, vp, vp]
    lib.uu3d_train_workspace_bytes.restype = sz
    lib.uu3d_train_workspace_bytes.argtypes = [vp, i32]
    lib.uu3d_train_forward_backward.restype = C.c_int
    lib.uu3d_train_forward_backward.argtypes = [vp, vp, vp, vp, vp, i32, i32, C.c_float, C.c_float, i32,
                                                C.POINTER(C.c_float), vp, vp, vp, vp, vp, vp, sz, vp]
    lib.uu3d_train_forward_backward_masked.restype = C.c_int
    lib.uu3d_train_forward_backward_masked.argtypes = [vp, vp, vp, vp, vp, i32, i32, C.c_float, C.c_float, i32,
                                                       C.POINTER(C.c_float), vp, vp, C.c_float, vp, vp, vp, vp, vp, sz, vp]
    lib.uu3d_train_set_grad_callback.restype = C.c_int
    lib.uu3d_train_set_grad_callback.argtypes = [vp, GRAD_READY_FN, vp]
    lib.uu3d_train_set_dropout.restype = C.c_int
    lib.uu3d_train_set_dropout.argtypes = [vp, C.c_float, C.c_float, C.c_uint64]
    lib.uu3d_op_scratch_floats.restype = sz
    lib.uu3d_op_scratch_floats.argtypes = []
    lib.uu3d_op_gemm_tn.restype = C.c_int
    lib.uu3d_op_gemm_tn.argtypes = [vp, i32, vp, i32, i32, i32, i32, vp, i32, vp, sz, vp]
    lib.uu3d_op_gemm_tn_h3.restype = C.c_int
    lib.uu3d_op_gemm_tn_h3.argtypes = [vp, i32, vp, i32, i32, i32, i32, vp, i32, vp, sz, vp]
    lib.uu3d_op_gemm_nt.restype = C.c_int
    lib.uu3d_op_gemm_nt.argtypes = [vp, i32, vp, i32, i32, i32, i32, vp, i32, vp, sz, vp]
    lib.uu3d_op_colsum.restype = C.c_int
    lib.uu3d_op_colsum.argtypes = [vp, i32, i32, i32, i32, vp, i32, vp, i32, vp, sz, vp]
    lib.uu3d_op_row_stats.restype = C.c_int
    lib.uu3d_op_row_stats.argtypes = [vp, i32, i32, i32, C.c_float, vp, vp]
    lib.uu3d_op_ln_bwd.restype = C.c_int
    lib.uu3d_op_ln_bwd.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, i32, vp, vp, vp, sz, vp]
    lib.uu3d_op_attn_fwd.restype = C.c_int
    lib.uu3d_op_attn_fwd.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp]
    lib.uu3d_op_attn_bwd.restype = C.c_int
    lib.uu3d_op_attn_bwd.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp]
    lib.uu3d_op_panel_operand_bytes.restype = sz
    lib.uu3d_op_panel_operand_bytes.argtypes = [i32]
    lib.uu3d_op_panel_a_bytes.restype = sz
    lib.uu3d_op_panel_a_bytes.argtypes = [i32]
    lib.uu3d_op_panel_pack.restype = C.c_int
    lib.uu3d_op_panel_pack.argtypes = [vp, i32, vp, vp]
    lib.uu3d_op_ln_dense_panel.restype = C.c_int
    lib.uu3d_op_ln_dense_panel.argtypes = [vp, i32, i32, vp, vp, C.c_float, vp, vp, i32, i32, vp, vp, i32, vp]
    if path is None:
        _lib = lib
    return lib


def check(lib, status, handle=None):
    if status != UU3D_OK:
        detail = lib.uu3d_last_error(handle).decode() if True else ""
        base = lib.uu3d_status_string(status).decode()
        raise Uu3dError(status, f"{base}: {detail}" if detail else base)
