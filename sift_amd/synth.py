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


# ---- inputs that stress what smooth value noise barely touches -------------------------------------------------
# Flat regions give exact ties in the non-strict extremum test (reference sift/sift.cpp:479-523) and Hessians with
# |det| < 1e-10 (sift.cpp:738-739); straight edges give Det <= 0 and large Tr^2/Det in the edge test
# (sift.cpp:616-626); white noise gives far more extrema per pixel than any natural image.

def _gray3(q):
    return np.repeat(np.ascontiguousarray(q, dtype=np.uint8)[:, :, None], 3, axis=2)


def white_noise(width, height, seed):
    """Every pixel an independent hash of (x, y, seed); b = g = r."""
    with np.errstate(over="ignore"):
        xs = np.arange(width, dtype=np.uint64)[None, :]
        ys = np.arange(height, dtype=np.uint64)[:, None]
        h = _sm64(np.uint64(seed) ^ _sm64(xs | (ys << np.uint64(32))))
    return _gray3((h & np.uint64(255)).astype(np.uint8))


def checkerboard(width, height, cell=16, lo=0, hi=255):
    """Saturated squares with step edges and X-corners."""
    xs = np.arange(width)[None, :] // cell
    ys = np.arange(height)[:, None] // cell
    return _gray3(np.where((xs + ys) % 2 == 0, lo, hi))


def step_edges(width, height, seed, n=40):
    """Piecewise-constant image: n random axis-aligned rectangles of random gray level painted over mid-gray."""
    rng = np.random.RandomState(seed)
    img = np.full((height, width), 128, dtype=np.uint8)
    for _ in range(n):
        x0, y0 = rng.randint(0, width), rng.randint(0, height)
        w, h = rng.randint(4, max(5, width // 3)), rng.randint(4, max(5, height // 3))
        img[y0:y0 + h, x0:x0 + w] = rng.randint(0, 256)
    return _gray3(img)


def blobs(width, height, seed, n=60):
    """Bright and dark Gaussian blobs of random size on mid-gray (scale-space extrema by construction), rounded."""
    rng = np.random.RandomState(seed)
    ys, xs = np.mgrid[0:height, 0:width].astype(np.float64)
    v = np.full((height, width), 128.0)
    for _ in range(n):
        cx, cy = rng.uniform(0, width), rng.uniform(0, height)
        s = rng.uniform(1.5, 12.0)
        a = rng.choice([-1.0, 1.0]) * rng.uniform(30, 120)
        v += a * np.exp(-((xs - cx) ** 2 + (ys - cy) ** 2) / (2 * s * s))
    return _gray3(np.clip(np.rint(v), 0, 255))


def saturated_blocks(width, height, seed, n=12):
    """Value noise with rectangular blocks forced to 0 and to 255 (clipped highlights and shadows)."""
    rng = np.random.RandomState(seed)
    img = vnoise5(width, height, seed)[:, :, 0].copy()
    for i in range(n):
        x0, y0 = rng.randint(0, width), rng.randint(0, height)
        w, h = rng.randint(6, max(7, width // 4)), rng.randint(6, max(7, height // 4))
        img[y0:y0 + h, x0:x0 + w] = 255 if i % 2 else 0
    return _gray3(img)
