"""Synthetic input: 5-octave value noise from a stateless 64-bit hash (numpy form).

The same generator exists as a HIP kernel (sg_synth_vnoise5) for on-device input; both produce
identical bytes: lattice periods 4, 8, 16, 32, 64 px, amplitudes 20, 28, 36, 40, 40 around 128,
rounded half-to-even and clamped to 8 bits, b = g = r.
"""
import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _sm64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M
    return x ^ (x >> np.uint64(31))


def _lat(ix, iy, k, seed):
    c = (ix.astype(np.int64).astype(np.uint64) & np.uint64(0xFFFFFFFF)) | \
        ((iy.astype(np.int64).astype(np.uint64) & np.uint64(0xFFFFFFFF)) << np.uint64(32))
    h = _sm64(np.uint64(seed) ^ _sm64(c) ^ _sm64(np.uint64(0x1000 + k)))
    return ((h >> np.uint64(11)).astype(np.float64) * 2.0 ** -53) * 2 - 1


def vnoise5(width, height, seed):
    """Returns an (height, width, 3) uint8 image."""
    L = (4.0, 8.0, 16.0, 32.0, 64.0)
    A = (20.0, 28.0, 36.0, 40.0, 40.0)
    with np.errstate(over="ignore"):
        xs = np.arange(width, dtype=np.float64)[None, :]
        ys = np.arange(height, dtype=np.float64)[:, None]
        v = np.full((height, width), 128.0)
        for k in range(5):
            x = np.broadcast_to(xs / L[k], (height, width))
            y = np.broadcast_to(ys / L[k], (height, width))
            fx0, fy0 = np.floor(x), np.floor(y)
            ix, iy = fx0.astype(np.int64), fy0.astype(np.int64)
            fx, fy = x - fx0, y - fy0
            fx = fx * fx * (3 - 2 * fx)
            fy = fy * fy * (3 - 2 * fy)
            a, b = _lat(ix, iy, k, seed), _lat(ix + 1, iy, k, seed)
            c, d = _lat(ix, iy + 1, k, seed), _lat(ix + 1, iy + 1, k, seed)
            v = v + A[k] * ((a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy)
    q = np.clip(np.rint(v), 0, 255).astype(np.uint8)
    return np.repeat(q[:, :, None], 3, axis=2)
