"""Counter-hash synthetic data for the SSD300 path.

Everything here is exact integer / IEEE arithmetic (no libm calls), so the same
(seed, stream) produces bit-identical arrays in this container and on the GPU
box.  Golden fixtures therefore store only outputs; inputs are regenerated.

Streams:  images  uniform[0,1) fp32 (N,3,S,S)              -- train.py:111 contract
          targets (N,G,4+C) rows [cx,cy,w,h,onehot], zero-padded -- utils.py:8-16 / dataset.py:92-116
          weights kaiming(fan_out) pseudo-normal, BN gamma=1 beta=0 -- ssd.py:135-158
"""
import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def _mix32(x: np.ndarray) -> np.ndarray:
    """lowbias32-style avalanche on uint32 lanes (computed in uint64 to avoid overflow warnings)."""
    x = x.astype(np.uint64) & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & _M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & _M32
    x ^= x >> np.uint64(16)
    return x


def hash_u32(seed: int, stream: int, n: int, offset: int = 0) -> np.ndarray:
    """n uint32 words for counters offset..offset+n-1 of (seed, stream)."""
    ctr = np.arange(offset, offset + n, dtype=np.uint64)
    key = np.uint64((seed * 0x9E3779B1 + stream * 0x85EBCA77 + 0x165667B1) & 0xFFFFFFFF)
    return _mix32(_mix32(ctr ^ key) + np.uint64(0x9E3779B9) * (ctr >> np.uint64(32)) + key)


def uniform(seed: int, stream: int, n: int, offset: int = 0) -> np.ndarray:
    """float64 uniform [0,1) with 24 random bits (exactly representable in fp32)."""
    return (hash_u32(seed, stream, n, offset) >> np.uint64(8)).astype(np.float64) * (1.0 / 16777216.0)


def pseudo_normal(seed: int, stream: int, n: int) -> np.ndarray:
    """Zero-mean unit-variance Irwin-Hall(4) variate; exact arithmetic only."""
    u = uniform(seed, stream, 4 * n).reshape(n, 4)
    return ((u[:, 0] + u[:, 1]) + (u[:, 2] + u[:, 3]) - 2.0) * 1.7320508075688772


def images(seed: int, n: int, size: int = 300) -> np.ndarray:
    return uniform(seed, 1, n * 3 * size * size).astype(np.float32).reshape(n, 3, size, size)


def targets(seed: int, n: int, g_max: int = 8, num_classes: int = 21,
            wh_lo: float = 0.05, wh_hi: float = 0.65, g_fixed: int = 0) -> np.ndarray:
    """(N, G, 4+C) fp32.  Per image G~U{1..g_max} (or g_fixed), w,h~U(lo,hi), box inside the image,
    label~U{1..C-1} one-hot at column 4+label (column 4 = void), padded rows all-zero."""
    c = num_classes
    t = np.zeros((n, g_max, 4 + c), dtype=np.float32)
    u = uniform(seed, 2, n * (1 + 5 * g_max)).reshape(n, 1 + 5 * g_max)
    for i in range(n):
        g = g_fixed if g_fixed else 1 + int(u[i, 0] * g_max)
        for j in range(g):
            r = u[i, 1 + 5 * j: 6 + 5 * j]
            w = wh_lo + (wh_hi - wh_lo) * r[0]
            h = wh_lo + (wh_hi - wh_lo) * r[1]
            cx = w / 2 + (1.0 - w) * r[2]
            cy = h / 2 + (1.0 - h) * r[3]
            lab = 1 + int(r[4] * (c - 1))
            t[i, j, :4] = (cx, cy, w, h)
            t[i, j, 4 + lab] = 1.0
    return t


def logits(seed: int, n: int, p: int = 8732, c: int = 21, void_bias: float = 0.0, loc_scale: float = 0.5) -> np.ndarray:
    """(N,P,4+C) fp32 synthetic head outputs: loc ~ loc_scale*N(0,1), conf ~ N(0,1) with column 4 += void_bias."""
    x = pseudo_normal(seed, 3, n * p * (4 + c)).reshape(n, p, 4 + c)
    x[:, :, :4] *= loc_scale
    x[:, :, 4] += void_bias
    return x.astype(np.float32)
