"""Counter-based uniform generator shared (bit for bit) with the HIP dropout kernel.

The reference's Dropout is Caffe2's (lib/models/lfb_helper.py:259,314,334;
lib/models/resnet_video.py:323) whose RNG stream is unspecified and non-reproducible, so the
mask generator is ours: u(seed, i) = two rounds of the murmur3 32-bit finaliser over the
REFERENCE-layout linear index i, top 24 bits -> [0,1).  keep = u >= ratio.
(csrc/vlfb_head.hip: mix32 / dropout_uniform)
"""
import numpy as np

_M = np.uint64(0xFFFFFFFF)


def _mix32(x):
    x = x & _M
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & _M
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & _M
    x ^= x >> np.uint64(16)
    return x


def uniform(seed, n):
    """u(seed, i) for i in [0, n) as float32"""
    i = np.arange(n, dtype=np.uint64)
    lo, hi = i & _M, i >> np.uint64(32)
    s0, s1 = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    h = _mix32(lo ^ s0)
    h = _mix32((h + np.uint64(0x9E3779B9) + (hi ^ s1)) & _M)
    return ((h >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)).astype(np.float32)


def dropout_keep_mask(seed, shape, ratio):
    """boolean keep-mask over a tensor of `shape` in the reference's (row-major) layout"""
    n = int(np.prod(shape))
    return (uniform(seed, n) >= np.float32(ratio)).reshape(shape)
