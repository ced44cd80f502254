"""Architecture description derived from an ``UpliftUpsampleConfig``.

This is the host-side restatement of what the reference's constructor and model
``__init__`` decide (``common/net/uplift_upsample_transformer_constructor.py:14-50``,
``common/net/uplift_upsample_transformer.py:165-285``): which sub-layers exist, their
sizes, the strided-block sequence lengths, and whether the model takes the
``[x, stride_mask]`` input pair.
"""
import math
from dataclasses import dataclass, field
from typing import List, Tuple


@dataclass(frozen=True)
class UpliftArch:
    num_frames: int            # N  = SEQUENCE_LENGTH (token count, not receptive field)
    num_keypoints: int         # J
    d_spatial: int
    d_temporal: int
    spatial_depth: int
    temporal_depth: int
    strides: Tuple[int, ...]
    paddings: Tuple[Tuple[int, int], ...]
    num_heads: int
    mlp_ratio: float
    qkv_bias: bool
    has_strided_input: bool
    first_strided_token_attention_layer: int
    full_output: bool
    drop_path_rate: Tuple[float, float, float]
    batch_size: int
    # training-only regularisers (Keras Dropout / random token masking are identities with training=False)
    drop_rate: float = 0.0
    attention_drop_rate: float = 0.0
    token_mask_rate: float = 0.0
    learnable_masked_token: bool = False   # TOKEN_MASK_RATE > 0 and LEARNABLE_MASKED_TOKEN: one more weight (u_u_t.py:219-220)
    output_bn: bool = False      # BatchNormalization in front of both heads (u_u_t.py:275-285); inference form only
    # derived
    strided_lengths: Tuple[int, ...] = field(default=())   # L_0 .. L_len(strides)

    @property
    def h_spatial(self) -> int:
        return int(self.d_spatial * self.mlp_ratio)

    @property
    def h_temporal(self) -> int:
        return int(self.d_temporal * self.mlp_ratio)

    @property
    def out_dim(self) -> int:
        return 3 * self.num_keypoints

    @property
    def compiled_dims(self) -> bool:
        """The dims the specialised HIP kernels are compiled for (every shipped config/*.json).  Anything else the reference's
        constructor accepts (u_u_t_constructor.py:26-32) runs on the library's generic kernels -- forward and training step, correct
        and untuned (csrc/uu3d_api.hip: uu3d_create)."""
        return (self.num_keypoints == 17 and self.d_spatial == 32 and self.h_spatial == 64 and self.num_heads == 8
                and self.d_temporal == 384)


def _pe_length_recurrence(n: int, strides, paddings) -> List[int]:
    """PE lengths as the reference computes them (u_u_t.py:210-216): ceil((L+p0+p1-2)/s)."""
    out = [n]
    for s, p in zip(strides, paddings):
        out.append(math.ceil((out[-1] + p[0] + p[1] - 2) / s))
    return out


def _conv_length_recurrence(n: int, strides, paddings) -> List[int]:
    """Actual output length of ZeroPadding1D(p) + Conv1D(k=3, stride s, 'valid')."""
    out = [n]
    for s, p in zip(strides, paddings):
        out.append((out[-1] + p[0] + p[1] - 3) // s + 1)
    return out


def training_unsupported(a: "UpliftArch"):
    """Config options that change the TRAINING-mode forward and that this build does not implement; the caller raises.
    Empty since round 4: the Dropout layers (DROP_RATE: u_u_t.py:201,324, vit.py:57-58,63-67,153-154, u_u_t.py:78-79,84-89;
    ATTENTION_DROP_RATE: vit.py:87-88,127-128) are implemented with counter-based masks (csrc/uu3d_dropout.h), like
    random_token_masking (u_u_t.py:287-311), DropPath in all three stacks and training-mode BatchNormalization (u_u_t.py:276-283)
    before them.  Rates outside [0, 1) are rejected, as Keras does."""
    out = []
    if not (0.0 <= a.drop_rate < 1.0):
        out.append(f"DROP_RATE = {a.drop_rate} (must be in [0, 1))")
    if not (0.0 <= a.attention_drop_rate < 1.0):
        out.append(f"ATTENTION_DROP_RATE = {a.attention_drop_rate} (must be in [0, 1))")
    return out


def arch_from_config(config) -> UpliftArch:
    # Options that would build a DIFFERENT model than the one the HIP path computes are rejected, not ignored.
    has_strided_input = config.MASK_STRIDE is not None
    if has_strided_input:
        ms = config.MASK_STRIDE
        if type(ms) is int and ms == 1:
            has_strided_input = False
        if type(ms) is list and ms[0] == 1:
            has_strided_input = False

    strides = tuple(int(s) for s in config.STRIDES)
    if config.PADDINGS is None:
        paddings = tuple((1, 1) for _ in strides)
    else:
        paddings = tuple((int(p[0]), int(p[1])) for p in config.PADDINGS)
    if len(paddings) != len(strides):
        raise ValueError("PADDINGS must have one [left, right] pair per entry of STRIDES")

    n = int(config.SEQUENCE_LENGTH)
    pe_len = _pe_length_recurrence(n, strides, paddings)
    conv_len = _conv_length_recurrence(n, strides, paddings)
    # The reference asserts x.shape[1] == pos_encoding.shape[1] in every strided block
    # (u_u_t.py:127); a config where the two recurrences disagree fails there.
    if pe_len[:-1] != conv_len[:-1]:
        raise AssertionError(
            f"strided PE lengths {pe_len[:-1]} do not match the conv output lengths {conv_len[:-1]}")
    if len(strides) > 0 and conv_len[-1] != 1:
        # einops "b n (p c) -> (b n) p c" with n=1 fails in the reference (u_u_t.py:416).
        raise ValueError(f"STRIDES/PADDINGS must reduce the sequence to one token, got {conv_len}")

    d_t = int(config.TEMPORAL_EMBED_DIM)
    d_s = int(config.SPATIAL_EMBED_DIM)
    heads = int(config.NUM_HEADS)
    if d_t % heads != 0 or (int(config.SPATIAL_TRANSFORMER_BLOCKS) > 0 and d_s % heads != 0):
        raise AssertionError("embedding dims must be divisible by NUM_HEADS")  # vit.py:79

    dpr = config.DROP_PATH_RATE
    if type(dpr) is list:
        dpr3 = (float(dpr[0]), float(dpr[1]), float(dpr[2]))
    else:
        dpr3 = (float(dpr),) * 3

    return UpliftArch(
        num_frames=n,
        num_keypoints=int(config.NUM_KEYPOINTS),
        d_spatial=d_s,
        d_temporal=d_t,
        spatial_depth=int(config.SPATIAL_TRANSFORMER_BLOCKS),
        temporal_depth=int(config.TEMPORAL_TRANSFORMER_BLOCKS),
        strides=strides,
        paddings=paddings,
        num_heads=heads,
        mlp_ratio=float(config.MLP_RATIO),
        qkv_bias=bool(config.QKV_BIAS),
        has_strided_input=has_strided_input,
        first_strided_token_attention_layer=int(config.FIRST_STRIDED_TOKEN_ATTENTION_LAYER),
        full_output=not bool(config.USE_REFINE),
        drop_path_rate=dpr3,
        batch_size=int(config.BATCH_SIZE),
        drop_rate=float(getattr(config, "DROP_RATE", 0.0)),
        attention_drop_rate=float(getattr(config, "ATTENTION_DROP_RATE", 0.0)),
        token_mask_rate=float(getattr(config, "TOKEN_MASK_RATE", 0.0)),
        # the layer (and its weight) exists only with both settings (u_u_t.py:219-220)
        learnable_masked_token=float(getattr(config, "TOKEN_MASK_RATE", 0.0)) > 0.0 and bool(getattr(config, "LEARNABLE_MASKED_TOKEN", False)),
        output_bn=bool(getattr(config, "OUTPUT_BN", False)),
        strided_lengths=tuple(conv_len),
    )


def flops_per_sequence(a: UpliftArch) -> dict:
    """Algorithmic FLOPs (2*MAC, GEMM-shaped work only) per sequence, SURVEY.md section 8(d)."""
    N, J, ds, dt, H = a.num_frames, a.num_keypoints, a.d_spatial, a.d_temporal, a.num_heads
    hs, ht = a.h_spatial, a.h_temporal
    Ls, Lt = a.spatial_depth, a.temporal_depth
    f = {}
    f["kp_embed"] = 2 * N * J * 2 * ds if Ls > 0 else 0
    f["sp_qkvo"] = 4 * Ls * 2 * N * J * ds * ds
    f["sp_attn"] = Ls * N * H * 4 * J * J * (ds // H)
    f["sp_mlp"] = Ls * 2 * 2 * N * J * ds * hs
    f["s2t"] = 2 * N * (J * ds if Ls > 0 else J * 2) * dt
    f["t_qkvo"] = 4 * Lt * 2 * N * dt * dt
    f["t_attn"] = Lt * H * 4 * N * N * (dt // H)
    f["t_mlp"] = Lt * 2 * 2 * N * dt * ht
    f["head1"] = 2 * N * dt * 3 * J if (a.full_output and Lt > 0) else 0
    st_qkvo = st_attn = st_mlp = 0
    L = a.strided_lengths
    for i in range(len(a.strides)):
        st_qkvo += 4 * 2 * L[i] * dt * dt
        st_attn += H * 4 * L[i] * L[i] * (dt // H)
        st_mlp += 2 * L[i] * dt * ht + 2 * L[i + 1] * 3 * ht * dt
    f["st_qkvo"], f["st_attn"], f["st_mlp"] = st_qkvo, st_attn, st_mlp
    f["head2"] = 2 * dt * 3 * J
    f["total"] = sum(f.values())
    return f
