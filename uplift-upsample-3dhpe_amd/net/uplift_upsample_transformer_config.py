"""Config schema for the uplift/upsample transformer path.

Mirrors the behaviour of the reference's two config classes so that the unmodified
``config/h36m_*.json`` files load:

* ``Config`` base (reference ``common/utils/config.py:21-111``): class attributes are the
  defaults; every key of the JSON root object is assigned over them without validation
  (unknown keys are tolerated, missing keys fall back to the defaults); a ``.txt`` mode
  of ``KEY <json-value>`` lines exists too; ``copy()``, ``dump()``, ``display()``.
* ``UpliftUpsampleConfig`` defaults (reference
  ``common/net/uplift_upsample_transformer_config.py:13-106``), including the schema's
  misspelt key ``ROOT_KEYTPOINT``.
"""
import copy as _copy
import json
import warnings
import os


def _public_items(obj):
    out = []
    for key in dir(obj):
        if key.startswith("__") and key.endswith("__"):
            continue
        value = getattr(obj, key)
        if callable(value):
            continue
        out.append((key, value))
    return out


class Config(object):
    """Attribute bag: class attributes are defaults, a config file overrides them."""

    def __init__(self, config_file=None, file_mode=None):
        if config_file is not None:
            self.load(config_file, file_mode)

    # -- loading -----------------------------------------------------------------------
    def load(self, config_file, file_mode=None):
        if not os.path.exists(config_file):
            raise AssertionError(f"config file not found: {config_file}")
        if file_mode is None:
            ext = os.path.splitext(config_file)[1]
            if ext not in (".txt", ".json"):
                raise AssertionError(f"unsupported config extension: {ext}")
            file_mode = "txt" if ext == ".txt" else "json"
        if file_mode == "txt":
            with open(config_file, "r") as f:
                for raw in f:
                    line = raw.strip("\r\n ")
                    if not line or line.startswith("#"):
                        continue
                    parts = line.split(" ", 1)
                    if len(parts) < 2 or not parts[1].strip():
                        continue
                    text = parts[1].strip()
                    if "'" in text:            # same warning as the reference (common/utils/config.py:78-81), then treated as double quotes
                        warnings.warn("Avoid single quotes literals in config files. Use double quotes instead")
                        text = text.replace("'", '"')
                    setattr(self, parts[0], json.loads(text))
        else:
            with open(config_file, "r") as f:
                root = json.load(f)
            for key, value in root.items():
                setattr(self, key, value)

    # -- utilities ---------------------------------------------------------------------
    def to_dict(self):
        return {k: v for k, v in _public_items(self)}

    def copy(self):
        new = self.__class__()
        for key, value in _public_items(self):
            setattr(new, key, _copy.deepcopy(value))
        return new

    def dump(self, config_file):
        root = {}
        for key, value in _public_items(self):
            if hasattr(value, "tolist"):
                value = value.tolist()
            root[key] = value
        with open(config_file, "w") as f:
            json.dump(root, f, indent=4, sort_keys=True)

    def display(self):
        print("\nConfigurations:")
        for key, value in _public_items(self):
            print("{:30} {}".format(key, value))
        print("\n")


class UpliftUpsampleConfig(Config):
    # Execution
    GPU_ID = 0
    BATCH_SIZE = 256
    ARCH = "UpliftUpsampleTransformer"
    SHUFFLE_SEED = 0

    # Architecture
    SPATIAL_EMBED_DIM = 32
    TEMPORAL_EMBED_DIM = 348
    MLP_RATIO = 2
    NUM_HEADS = 8
    SPATIAL_TRANSFORMER_BLOCKS = 4
    TEMPORAL_TRANSFORMER_BLOCKS = 4
    STRIDES = [3, 3, 3]
    PADDINGS = None  # None means [[1, 1]] per strided block
    QKV_BIAS = True
    DROP_PATH_RATE = [0.1, 0.1, 0.0]
    DROP_RATE = 0.0
    ATTENTION_DROP_RATE = 0.0
    OUTPUT_BN = False

    # Refine module (not on the hot path)
    USE_REFINE = False
    REFINE_FC_SIZE = 1024
    REFINE_DROP_RATE = 0.5

    # Token masking
    TOKEN_MASK_RATE = 0.0
    LEARNABLE_MASKED_TOKEN = False

    # Objective
    NUM_KEYPOINTS = 17
    SEQUENCE_LENGTH = 27
    PADDING_TYPE = "copy"
    SEQUENCE_STRIDE = 1
    TEST_STRIDED_EVAL = True
    MASK_STRIDE = None
    STRIDE_MASK_RAND_SHIFT = False
    FIRST_STRIDED_TOKEN_ATTENTION_LAYER = 0
    LOSS_WEIGHT_SEQUENCE = 1.0
    LOSS_WEIGHT_CENTER = 1.0

    # Data handling and augmentation
    ROOT_KEYTPOINT = 6
    AUGM_FLIP_KEYPOINT_ORDER = [5, 4, 3, 2, 1, 0, 6, 7, 8, 9, 10, 16, 15, 14, 13, 12, 11]
    AUGM_FLIP_PROB = 0.5
    IN_BATCH_AUGMENT = False

    # Training
    EPOCHS = 120
    STEPS_PER_EPOCH = 6000
    DATASET_TRAIN_3D_SUBSAMPLE_STEP = 1
    DATASET_VAL_3D_SUBSAMPLE_STEP = 4
    DATASET_TEST_3D_SUBSAMPLE_STEP = 1

    # Validation
    VALIDATION_INTERVAL = 1
    VALIDATION_EXAMPLES = -1
    EVAL_FLIP = True
    EVAL_DISABLE_LEARNED_UPSAMPLING = False

    # Optimizer and schedule
    OPTIMIZER = "Adam"
    OPTIMIZER_PARAMS = {"amsgrad": True, "epsilon": 1e-08}
    SCHEDULE = "ExponentialDecayWithSteps"
    SCHEDULE_PARAMS = {
        "initial_learning_rate": 1e-3,
        "decay_steps": 12000,
        "decay_rate": 0.95,
        "large_decay_steps": 60000,
        "large_decay_rate": 0.5,
    }
    WEIGHT_DECAY = None
    EMA_ENABLED = False
    EMA_DECAY = None

    # Checkpoints
    CHECKPOINT_INTERVAL = 10
    BEST_CHECKPOINT_METRIC = "AW-MPJPE"
