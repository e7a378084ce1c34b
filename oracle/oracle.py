"""ctypes binding of the CPU oracle (oracle/libsift_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may import this
module.  The product package (sift_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsift_oracle.so")


class Keypoint(C.Structure):
    """Mirror of the reference's struct Keypoint (sift/sift.h:75-102), 1120 bytes."""
    _fields_ = [
        ("octave", C.c_int), ("interval", C.c_int), ("offset_interval", C.c_double),
        ("x", C.c_int), ("y", C.c_int), ("scale", C.c_double), ("dx", C.c_double), ("dy", C.c_double),
        ("offset_x", C.c_double), ("offset_y", C.c_double), ("octave_scale", C.c_double),
        ("ori", C.c_double), ("descr_length", C.c_int), ("descriptor", C.c_double * 128),
        ("val", C.c_double),
    ]


KEYPOINT_DTYPE = np.dtype({
    "names": ["octave", "interval", "offset_interval", "x", "y", "scale", "dx", "dy", "offset_x",
              "offset_y", "octave_scale", "ori", "descr_length", "descriptor", "val"],
    "formats": ["<i4", "<i4", "<f8", "<i4", "<i4", "<f8", "<f8", "<f8", "<f8", "<f8", "<f8", "<f8",
                "<i4", ("<f8", (128,)), "<f8"],
    "offsets": [0, 4, 8, 16, 20, 24, 32, 40, 48, 56, 64, 72, 80, 88, 1112],
    "itemsize": 1120,
})


class Pyramid(C.Structure):
    _fields_ = [("octaves", C.c_int), ("intervals", C.c_int), ("rows", C.POINTER(C.c_int)),
                ("cols", C.POINTER(C.c_int)), ("gauss", C.POINTER(C.POINTER(C.c_double))),
                ("dog", C.POINTER(C.POINTER(C.c_double)))]


class Counters(C.Structure):
    _fields_ = [("cc00", C.c_long), ("cc0", C.c_long), ("cc1", C.c_long), ("cc2", C.c_long), ("cc3", C.c_long)]

    def as_tuple(self):
        return (self.cc00, self.cc0, self.cc1, self.cc2, self.cc3)


class Timings(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("init", "gauss", "dog", "detect", "orient", "descr", "sort", "total")]


def build(force=False):
    """Compile the oracle with gcc (seconds)."""
    src = os.path.join(_HERE, "sift_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libsift_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    dp = C.POINTER(C.c_double)
    u8p = C.POINTER(C.c_uint8)
    L.so_sizeof_keypoint.restype = C.c_size_t
    L.so_convert_to_gray.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_size_t, dp]
    L.so_upsample.argtypes = [dp, C.c_int, C.c_int, dp]
    L.so_downsample.argtypes = [dp, C.c_int, C.c_int, dp]
    L.so_gaussian_ksize.argtypes = [C.c_double]
    L.so_gaussian_ksize.restype = C.c_int
    L.so_gaussian_kernel.argtypes = [C.c_double, C.c_int, dp]
    L.so_gaussian_blur.argtypes = [dp, dp, C.c_int, C.c_int, C.c_double]
    L.so_create_init_smooth_gray.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_double, dp]
    L.so_num_octaves.argtypes = [C.c_int, C.c_int]
    L.so_num_octaves.restype = C.c_int
    L.so_pyramid_sigmas.argtypes = [C.c_double, C.c_int, dp]
    L.so_inverse3d.argtypes = [dp, dp]
    L.so_inverse3d.restype = C.c_int
    L.so_orientation_histogram.argtypes = [dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, dp]
    L.so_sift.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_double, C.c_int, C.c_int,
                          C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(Counters), C.POINTER(Timings)]
    L.so_sift.restype = C.POINTER(Keypoint)
    L.so_sift_debug.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_double, C.c_int, C.c_int,
                                C.POINTER(C.c_size_t), C.POINTER(C.POINTER(Keypoint)), C.POINTER(C.c_size_t),
                                C.POINTER(C.POINTER(Pyramid)), C.POINTER(Counters)]
    L.so_sift_debug.restype = C.POINTER(Keypoint)
    L.so_descriptor_one.argtypes = [C.POINTER(Keypoint), C.POINTER(Pyramid), dp]
    L.so_sort_features.argtypes = [C.POINTER(Keypoint), C.c_size_t]
    L.so_free.argtypes = [C.c_void_p]
    L.so_free_pyramid.argtypes = [C.POINTER(Pyramid)]
    L.so_sift_batch.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]
    L.so_sift_batch.restype = C.c_double
    L.so_match.argtypes = [C.POINTER(Keypoint), C.c_size_t, C.POINTER(Keypoint), C.c_size_t, C.c_double, C.c_int,
                           C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)]
    L.so_match.restype = C.c_size_t
    L.so_write_features.argtypes = [C.POINTER(Keypoint), C.c_size_t, C.c_char_p]
    L.so_write_features.restype = C.c_int
    L.so_vnoise5.argtypes = [u8p, C.c_int, C.c_int, C.c_uint64]
    assert L.so_sizeof_keypoint() == 1120 == C.sizeof(Keypoint) == KEYPOINT_DTYPE.itemsize
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _kp_array(ptr, n):
    """Copy n Keypoint records at ptr into a numpy structured array."""
    if n == 0:
        return np.zeros(0, dtype=KEYPOINT_DTYPE)
    buf = C.string_at(ptr, n * 1120)
    return np.frombuffer(buf, dtype=KEYPOINT_DTYPE).copy()


def vnoise5(width, height, seed):
    img = np.empty((height, width, 3), dtype=np.uint8)
    lib().so_vnoise5(_u8(img), height, width, seed)
    return img


def convert_to_gray(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, ch = img.shape
    out = np.empty((h, w), dtype=np.float64)
    lib().so_convert_to_gray(_u8(img), h, w, ch, w * ch, _dp(out))
    return out


def upsample(src):
    src = np.ascontiguousarray(src, dtype=np.float64)
    h, w = src.shape
    out = np.zeros((2 * h, 2 * w), dtype=np.float64)
    lib().so_upsample(_dp(src), h, w, _dp(out))
    return out


def downsample(src):
    src = np.ascontiguousarray(src, dtype=np.float64)
    h, w = src.shape
    out = np.empty((h // 2, w // 2), dtype=np.float64)
    lib().so_downsample(_dp(src), h, w, _dp(out))
    return out


def gaussian_kernel(sigma):
    n = lib().so_gaussian_ksize(sigma)
    k = np.empty(n, dtype=np.float64)
    lib().so_gaussian_kernel(sigma, n, _dp(k))
    return k


def gaussian_blur(src, sigma):
    src = np.ascontiguousarray(src, dtype=np.float64)
    h, w = src.shape
    out = np.empty_like(src)
    lib().so_gaussian_blur(_dp(src), _dp(out), h, w, sigma)
    return out


def create_init_smooth_gray(img, sigma=1.6):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, ch = img.shape
    out = np.empty((2 * h, 2 * w), dtype=np.float64)
    lib().so_create_init_smooth_gray(_u8(img), h, w, ch, w * ch, sigma, _dp(out))
    return out


def num_octaves(base_rows, base_cols):
    return lib().so_num_octaves(base_rows, base_cols)


def pyramid_sigmas(sigma=1.6, intervals=3):
    s = np.empty(intervals + 3, dtype=np.float64)
    lib().so_pyramid_sigmas(sigma, intervals, _dp(s))
    return s


def inverse3d(H):
    H = np.ascontiguousarray(H, dtype=np.float64).reshape(9)
    Hi = np.zeros(9, dtype=np.float64)
    ok = lib().so_inverse3d(_dp(H), _dp(Hi))
    return bool(ok), Hi.reshape(3, 3)


def sift(img, sigma=1.6, intervals=3, nointerval=False):
    """Full pipeline. Returns (features[KEYPOINT_DTYPE] sorted, n_keypoints, counters, timings)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, ch = img.shape
    n = C.c_size_t(0)
    nk = C.c_size_t(0)
    ctr = Counters()
    tm = Timings()
    p = lib().so_sift(_u8(img), h, w, ch, w * ch, sigma, intervals, int(nointerval),
                      C.byref(n), C.byref(nk), C.byref(ctr), C.byref(tm))
    feats = _kp_array(p, n.value)
    lib().so_free(p)
    return feats, nk.value, ctr.as_tuple(), {k: getattr(tm, k) for k, _ in Timings._fields_}


class Debug:
    """Everything so_sift_debug hands back: unsorted features, keypoints, pyramid levels."""

    def __init__(self, img, sigma=1.6, intervals=3, nointerval=False, prequant=False):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w, ch = img.shape
        L = lib()
        n = C.c_size_t(0)
        nk = C.c_size_t(0)
        kps = C.POINTER(Keypoint)()
        pyr = C.POINTER(Pyramid)()
        ctr = Counters()
        p = L.so_sift_debug(_u8(img), h, w, ch, w * ch, sigma, intervals, int(nointerval), C.byref(n),
                            C.byref(kps), C.byref(nk), C.byref(pyr), C.byref(ctr))
        self.features = _kp_array(p, n.value)           # emission order, descriptors filled
        self.keypoints = _kp_array(kps, nk.value)       # after CalculateScale + HalfFeatures
        self.counters = ctr.as_tuple()
        P = pyr.contents
        self.octaves = P.octaves
        self.intervals = P.intervals
        self.rows = [P.rows[o] for o in range(P.octaves)]
        self.cols = [P.cols[o] for o in range(P.octaves)]
        S = P.intervals
        self.gauss = []
        self.dog = []
        for o in range(P.octaves):
            shape = (self.rows[o], self.cols[o])
            for i in range(S + 3):
                self.gauss.append(np.ctypeslib.as_array(P.gauss[o * (S + 3) + i], shape=shape).copy())
            for i in range(S + 2):
                self.dog.append(np.ctypeslib.as_array(P.dog[o * (S + 2) + i], shape=shape).copy())
        self.prequant = None
        if prequant and n.value:
            pq = np.empty((n.value, 128), dtype=np.float64)
            tmp = (Keypoint * 1)()
            for i in range(n.value):
                C.memmove(tmp, C.byref(p[i]), 1120)
                L.so_descriptor_one(tmp, pyr, _dp(pq[i]))
            self.prequant = pq
        sorted_ = (Keypoint * max(1, n.value))()
        if n.value:
            C.memmove(sorted_, p, n.value * 1120)
            L.so_sort_features(sorted_, n.value)
        self.sorted_features = _kp_array(sorted_, n.value)
        L.so_free(p)
        L.so_free(kps)
        L.so_free_pyramid(pyr)


def sift_batch(imgs, threads):
    """imgs: (N,H,W,3) uint8. Returns (seconds, counts)."""
    imgs = np.ascontiguousarray(imgs, dtype=np.uint8)
    n, h, w, _ = imgs.shape
    counts = (C.c_size_t * n)()
    sec = lib().so_sift_batch(_u8(imgs), n, h, w, threads, counts)
    return sec, list(counts)


def write_features(feats, path):
    feats = np.ascontiguousarray(feats)
    assert feats.dtype == KEYPOINT_DTYPE
    return lib().so_write_features(feats.ctypes.data_as(C.POINTER(Keypoint)), len(feats), path.encode())


def match(f1, f2, ratio=0.5, accumulate_bug=False):
    """Brute-force matching with ratio test (demo main, 源.cpp:88-156). Returns (pairs[n,2] int32, best2[n1,2]
    squared distances smallest / second smallest)."""
    f1 = np.ascontiguousarray(f1)
    f2 = np.ascontiguousarray(f2)
    assert f1.dtype == KEYPOINT_DTYPE and f2.dtype == KEYPOINT_DTYPE
    i1 = np.zeros(max(len(f1), 1), dtype=np.int32)
    i2 = np.zeros(max(len(f1), 1), dtype=np.int32)
    best2 = np.zeros((max(len(f1), 1), 2), dtype=np.float64)
    n = lib().so_match(f1.ctypes.data_as(C.POINTER(Keypoint)), len(f1), f2.ctypes.data_as(C.POINTER(Keypoint)), len(f2),
                       ratio, int(accumulate_bug), i1.ctypes.data_as(C.POINTER(C.c_int)),
                       i2.ctypes.data_as(C.POINTER(C.c_int)), best2.ctypes.data_as(C.POINTER(C.c_double)))
    return np.stack([i1[:n], i2[:n]], axis=1), best2[:len(f1)]
