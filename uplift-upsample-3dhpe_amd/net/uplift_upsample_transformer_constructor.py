"""Model factory -- same name and meaning as the reference's
``common/net/uplift_upsample_transformer_constructor.py:14-50``:

    model = build_uplift_upsample_transformer(config)

decides ``has_strided_input`` from ``MASK_STRIDE`` (:16-21), maps the config onto the model's
hyper-parameters (:23-43) and "builds" it for ``BATCH_SIZE`` (:44-49) -- here: creates the
device model, uploads seeded Keras-default weights and reserves the workspace.
"""
from ..arch import arch_from_config
from .uplift_upsample_transformer import UpliftUpsampleTransformer


def build_uplift_upsample_transformer(config, **kwargs):
    arch = arch_from_config(config)
    return UpliftUpsampleTransformer(arch, **kwargs)
