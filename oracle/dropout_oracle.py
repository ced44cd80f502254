"""CPU ORACLE (test infrastructure, NOT the product path): the Dropout masks of csrc/uu3d_dropout.h in numpy.

Keras' ``Dropout`` (training mode) keeps an element iff its uniform draw is >= rate and scales what it keeps by 1 / (1 - rate)
(``common/net/vision_transformer.py:57-58,63-67,87-90,127-128,153-154``; ``common/net/uplift_upsample_transformer.py:78-79,84-89,201,324``).
The reference's draws are TensorFlow's private random stream; the HIP library instead derives the draw of element ``index`` of
Dropout layer ``site`` from a 32-bit integer hash of (seed, site, index), and this file evaluates the same integer function, so the
oracle and the library drop exactly the same elements and their outputs and gradients can be compared to rounding.

PARITY UNPINNED against TensorFlow (as the rest of the transformer oracle): what is restated is the published semantics of
``tf.nn.dropout`` -- mask = uniform >= rate, output = x * mask / (1 - rate).
"""
import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def drop_hash(seed, site, index):
    """uint32 hash of csrc/uu3d_dropout.h::drop_hash for an array of element indices (uint64)."""
    index = np.asarray(index, np.uint64)
    seed = int(seed)
    lo, hi = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    h = ((index & _M32) * np.uint64(0x9E3779B1) + lo) & _M32
    h ^= (((index >> np.uint64(32)) * np.uint64(0x85EBCA77)) + np.uint64(site) * np.uint64(0xC2B2AE3D) + hi) & _M32
    h ^= h >> np.uint64(16); h = (h * np.uint64(0x7FEB352D)) & _M32
    h ^= h >> np.uint64(15); h = (h * np.uint64(0x846CA68B)) & _M32
    h ^= h >> np.uint64(16)
    return h.astype(np.uint32)


def drop_factor(shape, rate, seed, site):
    """The factor tensor of one Dropout layer: 1 / (1 - rate) where kept, 0 where dropped (float32, row-major element indices)."""
    n = int(np.prod(shape))
    u = (drop_hash(seed, site, np.arange(n, dtype=np.uint64)) >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    keep = u >= np.float32(rate)
    return np.where(keep, np.float32(1.0) / (np.float32(1.0) - np.float32(rate)), np.float32(0.0)).astype(np.float32).reshape(shape)


SITE_TOKEN = 1


def site_spatial(i, which):      # which: 0 attention weights, 1 projection output, 3 fc2 output
    return 10 + 4 * i + which


def site_temporal(i, which):     # which: 0 attention weights, 1 projection output, 2 hidden behind the ReLU, 3 fc2 output
    return 100 + 4 * i + which


def site_strided(j, which):      # which: 0 attention weights, 1 projection output, 2 hidden, 3 strided-convolution output
    return 200 + 4 * j + which
