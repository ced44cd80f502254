"""MI355X-native uplift/upsample 3D-HPE transformer forward path (gfx950, hand-written HIP).

Importable as ``uplift_upsample_3dhpe_amd`` (see the alias package of that name at the
repo root).  Public surface mirrors the reference's boundary
(``common/net/uplift_upsample_transformer_constructor.py:14``):

    from uplift_upsample_3dhpe_amd import UpliftUpsampleConfig, build_uplift_upsample_transformer
    model = build_uplift_upsample_transformer(UpliftUpsampleConfig("config/h36m_351.json"))
    full, central = model([x, stride_mask], training=False)
"""
from .net.uplift_upsample_transformer_config import Config, UpliftUpsampleConfig  # noqa: F401
from .arch import UpliftArch, arch_from_config, flops_per_sequence  # noqa: F401
from .weights import weight_spec, init_weights, count_params  # noqa: F401


def build_uplift_upsample_transformer(config, **kwargs):
    from .net.uplift_upsample_transformer_constructor import build_uplift_upsample_transformer as _b
    return _b(config, **kwargs)


def library_version():
    """``uu3d_version()`` of the loaded HIP library (a timing build, csrc/libuu3d_timing.so through UU3D_LIB, says so: bench.py refuses it)."""
    from . import _capi
    return _capi.load_library().uu3d_version().decode()
