"""Shared helpers for the tests: the product's synthetic-workload module plus the oracle-side config mapping."""
from uplift_upsample_3dhpe_amd.synthetic import (ROOT, CONFIGS, TOL_MAX_ABS, TOL_MPJPE_MM, load_config,   # noqa: F401
                                                 eval_stride_mask, synthetic_batch)
from oracle.uplift_oracle import hp_from_arch                                                                # noqa: F401
