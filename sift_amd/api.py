"""ctypes binding of libsiftgpu (include/siftgpu.h).

The library is the product; this module only loads it and marshals numpy arrays.  There is no CPU
fallback: if the shared library is missing, or no HIP device is usable, construction fails loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SIFTGPU_LIB") or os.path.join(_HERE, "libsiftgpu.so")

SG_OK, SG_E_INVALID, SG_E_HIP, SG_E_CAPACITY, SG_E_UNSUPPORTED, SG_E_NOMEM = 0, -1, -2, -3, -4, -5
SG_F64, SG_F32, SG_F64_FMA = 0, 1, 2
STAGES = (("prep",) + tuple("blur_row%d" % i for i in range(6)) + tuple("blur_col%d" % i for i in range(6)) +
          ("downsample", "detect", "compact", "refine", "orient", "descr", "sort", "blur_pair", "pyr_tail"))

# Mirror of the reference's struct Keypoint (sift/sift.h:75-102), 1120 bytes.
KEYPOINT_DTYPE = np.dtype({
    "names": ["octave", "interval", "offset_interval", "x", "y", "scale", "dx", "dy", "offset_x",
              "offset_y", "octave_scale", "ori", "descr_length", "descriptor", "val"],
    "formats": ["<i4", "<i4", "<f8", "<i4", "<i4", "<f8", "<f8", "<f8", "<f8", "<f8", "<f8", "<f8",
                "<i4", ("<f8", (128,)), "<f8"],
    "offsets": [0, 4, 8, 16, 20, 24, 32, 40, 48, 56, 64, 72, 80, 88, 1112],
    "itemsize": 1120,
})

# sg_wire_feature: a Keypoint record with its descriptor as bytes, 224 bytes.
WIRE_DTYPE = np.dtype([("octave", "<i4"), ("interval", "<i4"), ("x", "<i4"), ("y", "<i4"), ("offset_interval", "<f8"),
                       ("scale", "<f8"), ("dx", "<f8"), ("dy", "<f8"), ("offset_x", "<f8"), ("offset_y", "<f8"),
                       ("octave_scale", "<f8"), ("ori", "<f8"), ("val", "<f8"), ("reserved", "<f8"),
                       ("desc", "u1", (128,))])
assert WIRE_DTYPE.itemsize == 224

PACKED_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("scale", "<f4"), ("ori", "<f4"), ("desc", "u1", (128,))])
assert PACKED_DTYPE.itemsize == 144

# Every symbol include/siftgpu.h declares.
EXPORTS = (
    "sg_default_params", "sg_strerror", "sg_last_error", "sg_device_count", "sg_create", "sg_destroy", "sg_sync", "sg_required_features",
    "sg_extract", "sg_extract_begin", "sg_extract_fetch", "sg_extract_batch_begin", "sg_extract_batch_fetch_wire",
    "sg_extract_batch_fetch_wire_begin", "sg_extract_batch_fetch_wire_wait",
    "sg_expand_wire", "sg_host_alloc", "sg_host_free", "sg_extract_batch_device", "sg_extract_batch", "sg_convert_to_gray", "sg_upsample",
    "sg_downsample", "sg_gaussian_smooth", "sg_create_init_smooth_gray", "sg_gaussian_pyramid", "sg_sub",
    "sg_dog_pyramid", "sg_detect_local_extrema", "sg_calculate_scale", "sg_half_features",
    "sg_orientation_assignment", "sg_descriptor_representation", "sg_sort_features", "sg_last_octaves",
    "sg_level_dims", "sg_last_pair_octaves", "sg_match_scratch_bytes", "sg_last_tail_octave", "sg_read_level", "sg_read_counters", "sg_read_work_counters", "sg_read_keypoints", "sg_read_features_unsorted",
    "sg_set_profiling", "sg_get_stage_times", "sg_reset_stage_times", "sg_malloc", "sg_free", "sg_memcpy_h2d",
    "sg_memcpy_d2h", "sg_synth_vnoise5", "sg_match", "sg_match_packed_device",
    "sg_match_packed_device_noalloc", "sg_comm_unique_id", "sg_comm_create", "sg_comm_destroy",
    "sg_allgather_features", "sg_compact_features",
)


class Params(C.Structure):
    _fields_ = [("sigma", C.c_double), ("intervals", C.c_int), ("precision", C.c_int), ("nointerval", C.c_int),
                ("max_width", C.c_int), ("max_height", C.c_int), ("max_batch", C.c_int), ("max_features", C.c_int),
                ("device", C.c_int), ("verbose", C.c_int), ("keep_prequant", C.c_int)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_ulonglong) for n in ("cc00", "cc0", "cc1", "cc2", "cc3")]

    def as_tuple(self):
        return (self.cc00, self.cc0, self.cc1, self.cc2, self.cc3)


class SiftGpuError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        super().__init__("libsiftgpu status %d%s" % (status, (": " + detail) if detail else ""))


_lib = None


def load_library():
    """Load libsiftgpu.so and declare argument types. Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            "%s not found: build it with `make -C sift_amd/csrc` (hipcc, gfx950); there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, dp, u8p = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint8)
    dpp = C.POINTER(dp)
    szp = C.POINTER(C.c_size_t)
    L.sg_default_params.argtypes = [C.POINTER(Params)]
    L.sg_default_params.restype = None
    L.sg_strerror.argtypes = [C.c_int]
    L.sg_strerror.restype = C.c_char_p
    L.sg_last_error.argtypes = [vp]
    L.sg_last_error.restype = C.c_char_p
    L.sg_device_count.restype = C.c_int
    L.sg_create.argtypes = [C.POINTER(Params), C.POINTER(vp)]
    L.sg_destroy.argtypes = [vp]
    L.sg_sync.argtypes = [vp]
    L.sg_required_features.argtypes = [vp, szp]
    L.sg_extract.argtypes = [vp, u8p, C.c_int, C.c_int, C.c_int, C.c_size_t, vp, C.c_size_t, szp]
    L.sg_extract_begin.argtypes = [vp, u8p, C.c_int, C.c_int, C.c_int, C.c_size_t, szp]
    L.sg_extract_fetch.argtypes = [vp, vp, C.c_size_t, szp]
    L.sg_extract_batch_begin.argtypes = [vp, C.POINTER(u8p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_int,
                                         C.POINTER(C.c_uint32)]
    L.sg_extract_batch_fetch_wire.argtypes = [vp, vp, C.c_size_t]
    L.sg_extract_batch_fetch_wire_begin.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.POINTER(C.c_int)]
    L.sg_extract_batch_fetch_wire_wait.argtypes = [vp, C.c_int]
    L.sg_expand_wire.argtypes = [vp, C.c_size_t, vp]
    L.sg_expand_wire.restype = None
    L.sg_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.sg_host_free.argtypes = [vp]
    L.sg_extract_batch_device.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t, vp,
                                          C.c_size_t, vp]
    L.sg_extract_batch.argtypes = [vp, C.POINTER(u8p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, vp,
                                   C.c_size_t, C.POINTER(C.c_uint32)]
    L.sg_convert_to_gray.argtypes = [vp, u8p, C.c_int, C.c_int, C.c_int, C.c_size_t, dp]
    L.sg_upsample.argtypes = [vp, dp, C.c_int, C.c_int, dp]
    L.sg_downsample.argtypes = [vp, dp, C.c_int, C.c_int, dp]
    L.sg_gaussian_smooth.argtypes = [vp, dp, dp, C.c_int, C.c_int, C.c_double]
    L.sg_create_init_smooth_gray.argtypes = [vp, u8p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_double, dp]
    L.sg_gaussian_pyramid.argtypes = [vp, dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, dpp]
    L.sg_sub.argtypes = [vp, dp, dp, dp, C.c_int, C.c_int]
    L.sg_dog_pyramid.argtypes = [vp, dpp, C.c_int, C.c_int, C.c_int, C.c_int, dpp]
    L.sg_detect_local_extrema.argtypes = [vp, dpp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_size_t, szp,
                                          C.POINTER(Counters)]
    L.sg_calculate_scale.argtypes = [vp, C.c_size_t, C.c_double, C.c_int]
    L.sg_half_features.argtypes = [vp, C.c_size_t]
    L.sg_orientation_assignment.argtypes = [vp, vp, C.c_size_t, dpp, C.c_int, C.c_int, C.c_int, vp, C.c_size_t, szp]
    L.sg_descriptor_representation.argtypes = [vp, vp, C.c_size_t, dpp, C.c_int, C.c_int, C.c_int]
    L.sg_sort_features.argtypes = [vp, vp, C.c_size_t]
    L.sg_match.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.c_double, vp, C.c_size_t, szp]
    L.sg_match_packed_device.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32, C.c_double, vp, vp]
    L.sg_match_packed_device_noalloc.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32, C.c_double, vp, vp]
    L.sg_comm_unique_id.argtypes = [vp]
    L.sg_comm_create.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(vp)]
    L.sg_comm_destroy.argtypes = [vp]
    L.sg_allgather_features.argtypes = [vp, vp, vp, vp, C.c_int, C.c_size_t, vp, C.c_size_t, vp, szp]
    L.sg_compact_features.argtypes = [vp, vp, vp, C.c_int, C.c_size_t, vp]
    L.sg_last_octaves.argtypes = [vp]
    L.sg_last_pair_octaves.argtypes = [vp]
    L.sg_last_pair_octaves.restype = C.c_uint
    L.sg_match_scratch_bytes.argtypes = [vp]
    L.sg_match_scratch_bytes.restype = C.c_size_t
    L.sg_last_tail_octave.argtypes = [vp]
    L.sg_level_dims.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.sg_read_level.argtypes = [vp, C.c_int, C.c_int, C.c_int, dp]
    L.sg_read_counters.argtypes = [vp, C.c_int, C.POINTER(Counters)]
    L.sg_read_work_counters.argtypes = [vp, C.c_int, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    L.sg_read_keypoints.argtypes = [vp, C.c_int, vp, C.c_size_t, szp]
    L.sg_read_features_unsorted.argtypes = [vp, C.c_int, vp, C.c_size_t, szp, dp]
    L.sg_set_profiling.argtypes = [vp, C.c_int]
    L.sg_get_stage_times.argtypes = [vp, dp, C.POINTER(C.c_ulonglong)]
    L.sg_reset_stage_times.argtypes = [vp]
    L.sg_malloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.sg_free.argtypes = [vp, vp]
    L.sg_memcpy_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    L.sg_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
    L.sg_synth_vnoise5.argtypes = [vp, vp, C.c_int, C.c_int, C.c_uint64]
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _level_ptrs(levels):
    arr = (C.POINTER(C.c_double) * len(levels))()
    for i, lv in enumerate(levels):
        arr[i] = _dp(lv)
    return arr


def comm_unique_id():
    """128 bytes identifying a new RCCL communicator: made on one rank, handed to all (sg_comm_unique_id)."""
    buf = C.create_string_buffer(128)
    rc = load_library().sg_comm_unique_id(buf)
    if rc != SG_OK:
        raise SiftGpuError(rc, "RCCL is not available")
    return buf.raw


class SiftGPU:
    """One libsiftgpu context (one HIP device, one stream)."""

    def __init__(self, max_width=1920, max_height=1080, max_batch=1, precision=SG_F64, sigma=1.6, intervals=3,
                 nointerval=False, max_features=0, device=0, keep_prequant=False):
        self.lib = load_library()
        p = Params()
        self.lib.sg_default_params(C.byref(p))
        p.sigma, p.intervals, p.precision, p.nointerval = sigma, intervals, precision, int(nointerval)
        p.max_width, p.max_height, p.max_batch = max_width, max_height, max_batch
        p.max_features, p.device, p.keep_prequant = max_features, device, int(keep_prequant)
        self.params = p
        self.ctx = C.c_void_p()
        rc = self.lib.sg_create(C.byref(p), C.byref(self.ctx))
        if rc != SG_OK:
            self.ctx = None
            raise SiftGpuError(rc, self.lib.sg_strerror(rc).decode())

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.sg_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != SG_OK:
            raise SiftGpuError(rc, (self.lib.sg_last_error(self.ctx) or b"").decode() or
                               self.lib.sg_strerror(rc).decode())

    # ---- whole path -------------------------------------------------------------------------
    def extract(self, img, cap=None):
        """Sift()/Sift_nointerval() on one host image (H,W,C>=3 uint8). Returns KEYPOINT_DTYPE records."""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w, ch = img.shape
        if cap is None:
            cap = max(4096, w * h // 40 + 4096)
        out = np.zeros(cap, dtype=KEYPOINT_DTYPE)
        n = C.c_size_t(0)
        rc = self.lib.sg_extract(self.ctx, _u8(img), w, h, ch, w * ch, out.ctypes.data, cap, C.byref(n))
        self._check(rc)
        return out[: n.value].copy()

    def extract_batch(self, imgs, cap_per_image):
        """Host batch form: imgs (N,H,W,C) uint8 -> list of PACKED_DTYPE arrays."""
        imgs = np.ascontiguousarray(imgs, dtype=np.uint8)
        n, h, w, ch = imgs.shape
        ptrs = (C.POINTER(C.c_uint8) * n)()
        for i in range(n):
            ptrs[i] = _u8(imgs[i])
        out = np.zeros((n, cap_per_image), dtype=PACKED_DTYPE)
        counts = (C.c_uint32 * n)()
        rc = self.lib.sg_extract_batch(self.ctx, ptrs, n, w, h, ch, w * ch, out.ctypes.data, cap_per_image, counts)
        self._check(rc)
        return [out[i, : counts[i]].copy() for i in range(n)]

    def extract_batch_wire(self, imgs, pinned=False, piece_records=0):
        """Several host images of one size in one batch -> list of KEYPOINT_DTYPE arrays (sg_extract_batch_begin,
        sg_extract_batch_fetch_wire, sg_expand_wire).  pinned: stage images and records in page-locked memory.
        piece_records > 0: the records come down in pieces of that many (sg_extract_batch_fetch_wire_begin / _wait),
        every piece expanded as soon as it has landed."""
        imgs = np.ascontiguousarray(imgs, dtype=np.uint8)
        n, h, w, ch = imgs.shape
        ptrs = (C.POINTER(C.c_uint8) * n)()
        host_in = C.c_void_p()
        if pinned:
            rc = self.lib.sg_host_alloc(imgs.nbytes, C.byref(host_in))
            if rc != SG_OK:
                raise SiftGpuError(rc, "sg_host_alloc")
            C.memmove(host_in, imgs.ctypes.data, imgs.nbytes)
        try:
            for i in range(n):
                ptrs[i] = (C.cast(host_in.value + i * imgs[0].nbytes, C.POINTER(C.c_uint8)) if pinned else _u8(imgs[i]))
            counts = (C.c_uint32 * n)()
            self._check(self.lib.sg_extract_batch_begin(self.ctx, ptrs, n, w, h, ch, w * ch, int(pinned), counts))
        finally:
            if pinned:
                self.lib.sg_host_free(host_in)
        total = sum(counts)
        wire = np.zeros(max(total, 1), dtype=WIRE_DTYPE)
        recs = np.zeros(max(total, 1), dtype=KEYPOINT_DTYPE)
        if piece_records > 0:
            n_pieces = C.c_int(0)
            self._check(self.lib.sg_extract_batch_fetch_wire_begin(self.ctx, wire.ctypes.data, total, piece_records,
                                                                   C.byref(n_pieces)))
            k = 0
            for i in range(n):  # pieces follow the images: every image is cut into pieces of its own
                first = sum(counts[:i])
                for off in range(0, counts[i], piece_records):
                    m = min(piece_records, counts[i] - off)
                    self._check(self.lib.sg_extract_batch_fetch_wire_wait(self.ctx, k))
                    self.lib.sg_expand_wire(wire.ctypes.data + (first + off) * WIRE_DTYPE.itemsize, m,
                                            recs.ctypes.data + (first + off) * KEYPOINT_DTYPE.itemsize)
                    k += 1
            assert k == n_pieces.value, (k, n_pieces.value)
        else:
            self._check(self.lib.sg_extract_batch_fetch_wire(self.ctx, wire.ctypes.data, total))
            self.lib.sg_expand_wire(wire.ctypes.data, total, recs.ctypes.data)
        out, at = [], 0
        for i in range(n):
            out.append(recs[at: at + counts[i]].copy())
            at += counts[i]
        return out

    def extract_batch_device(self, d_imgs, n_img, w, h, ch, d_out, cap_per_image, d_counts):
        """Device-pointer form (ints); asynchronous. Call sync() before reading the outputs."""
        rc = self.lib.sg_extract_batch_device(self.ctx, d_imgs, n_img, w, h, ch, w * ch, w * h * ch, d_out,
                                              cap_per_image, d_counts)
        self._check(rc)

    def sync(self):
        self._check(self.lib.sg_sync(self.ctx))

    def required_features(self):
        """max_features a new context needs after an internal-capacity overflow was reported."""
        n = C.c_size_t(0)
        self._check(self.lib.sg_required_features(self.ctx, C.byref(n)))
        return n.value

    # ---- device memory helpers ---------------------------------------------------------------
    def malloc(self, nbytes):
        p = C.c_void_p()
        self._check(self.lib.sg_malloc(self.ctx, nbytes, C.byref(p)))
        return p.value

    def free(self, ptr):
        self._check(self.lib.sg_free(self.ctx, ptr))

    def h2d(self, d_ptr, arr):
        arr = np.ascontiguousarray(arr)
        self._check(self.lib.sg_memcpy_h2d(self.ctx, d_ptr, arr.ctypes.data, arr.nbytes))

    def d2h(self, arr, d_ptr):
        assert arr.flags["C_CONTIGUOUS"]
        self._check(self.lib.sg_memcpy_d2h(self.ctx, arr.ctypes.data, d_ptr, arr.nbytes))

    def synth_vnoise5(self, d_ptr, w, h, seed):
        self._check(self.lib.sg_synth_vnoise5(self.ctx, d_ptr, w, h, seed))

    # ---- stage entry points -----------------------------------------------------------------------
    def convert_to_gray(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w, ch = img.shape
        out = np.empty((h, w), dtype=np.float64)
        self._check(self.lib.sg_convert_to_gray(self.ctx, _u8(img), w, h, ch, w * ch, _dp(out)))
        return out

    def upsample(self, src):
        src = np.ascontiguousarray(src, dtype=np.float64)
        h, w = src.shape
        out = np.empty((2 * h, 2 * w), dtype=np.float64)
        self._check(self.lib.sg_upsample(self.ctx, _dp(src), w, h, _dp(out)))
        return out

    def downsample(self, src):
        src = np.ascontiguousarray(src, dtype=np.float64)
        h, w = src.shape
        out = np.empty((h // 2, w // 2), dtype=np.float64)
        self._check(self.lib.sg_downsample(self.ctx, _dp(src), w, h, _dp(out)))
        return out

    def gaussian_smooth(self, src, sigma):
        src = np.ascontiguousarray(src, dtype=np.float64)
        h, w = src.shape
        out = np.empty_like(src)
        self._check(self.lib.sg_gaussian_smooth(self.ctx, _dp(src), _dp(out), w, h, sigma))
        return out

    def create_init_smooth_gray(self, img, sigma=1.6):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w, ch = img.shape
        out = np.empty((2 * h, 2 * w), dtype=np.float64)
        self._check(self.lib.sg_create_init_smooth_gray(self.ctx, _u8(img), w, h, ch, w * ch, sigma, _dp(out)))
        return out

    @staticmethod
    def _octave_shapes(h, w, octaves):
        shapes = []
        for o in range(octaves):
            if o > 0:
                ph, pw = shapes[-1]
                h, w = (ph, pw) if (ph <= 1 or pw <= 1) else (ph // 2, pw // 2)
            shapes.append((h, w))
        return shapes

    def gaussian_pyramid(self, base, octaves, intervals=3, sigma=1.6):
        base = np.ascontiguousarray(base, dtype=np.float64)
        h, w = base.shape
        shapes = self._octave_shapes(h, w, octaves)
        levels = [np.empty(shapes[o], dtype=np.float64) for o in range(octaves) for _ in range(intervals + 3)]
        self._check(self.lib.sg_gaussian_pyramid(self.ctx, _dp(base), w, h, octaves, intervals, sigma,
                                                 _level_ptrs(levels)))
        return levels

    def sub(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        out = np.empty_like(a)
        self._check(self.lib.sg_sub(self.ctx, _dp(a), _dp(b), _dp(out), a.shape[1], a.shape[0]))
        return out

    def dog_pyramid(self, gauss, octaves, intervals=3):
        h, w = gauss[0].shape
        shapes = self._octave_shapes(h, w, octaves)
        dog = [np.empty(shapes[o], dtype=np.float64) for o in range(octaves) for _ in range(intervals + 2)]
        self._check(self.lib.sg_dog_pyramid(self.ctx, _level_ptrs(gauss), w, h, octaves, intervals, _level_ptrs(dog)))
        return dog

    def detect_local_extrema(self, dog, octaves, intervals=3, nointerval=False, cap=None):
        h, w = dog[0].shape
        cap = cap or max(4096, w * h // 40)
        out = np.zeros(cap, dtype=KEYPOINT_DTYPE)
        n = C.c_size_t(0)
        ctr = Counters()
        self._check(self.lib.sg_detect_local_extrema(self.ctx, _level_ptrs(dog), w, h, octaves, intervals,
                                                     int(nointerval), out.ctypes.data, cap, C.byref(n), C.byref(ctr)))
        return out[: n.value].copy(), ctr.as_tuple()

    def calculate_scale(self, kps, sigma=1.6, intervals=3):
        kps = np.ascontiguousarray(kps)
        self._check(self.lib.sg_calculate_scale(kps.ctypes.data, len(kps), sigma, intervals))
        return kps

    def half_features(self, kps):
        kps = np.ascontiguousarray(kps)
        self._check(self.lib.sg_half_features(kps.ctypes.data, len(kps)))
        return kps

    def orientation_assignment(self, extrema, gauss, octaves, cap=None):
        extrema = np.ascontiguousarray(extrema)
        h, w = gauss[0].shape
        cap = cap or (2 * len(extrema) + 64)
        out = np.zeros(cap, dtype=KEYPOINT_DTYPE)
        n = C.c_size_t(0)
        self._check(self.lib.sg_orientation_assignment(self.ctx, extrema.ctypes.data, len(extrema), _level_ptrs(gauss),
                                                       w, h, octaves, out.ctypes.data, cap, C.byref(n)))
        return out[: n.value].copy()

    def descriptor_representation(self, feats, gauss, octaves):
        feats = np.ascontiguousarray(feats).copy()
        h, w = gauss[0].shape
        self._check(self.lib.sg_descriptor_representation(self.ctx, feats.ctypes.data, len(feats), _level_ptrs(gauss),
                                                          w, h, octaves))
        return feats

    def sort_features(self, feats):
        feats = np.ascontiguousarray(feats).copy()
        self._check(self.lib.sg_sort_features(self.ctx, feats.ctypes.data, len(feats)))
        return feats

    # ---- descriptor matching (the reference demo's loop, sift/源.cpp:88-156) ------------------------------
    def match(self, f1, f2, ratio=0.5):
        """Brute-force matching with ratio test on Keypoint records; returns int32 pairs [n, 2] = (index_1, index_2)."""
        f1 = np.ascontiguousarray(f1)
        f2 = np.ascontiguousarray(f2)
        assert f1.dtype == KEYPOINT_DTYPE and f2.dtype == KEYPOINT_DTYPE
        out = np.zeros((max(len(f1), 1), 2), dtype=np.int32)
        n = C.c_size_t(0)
        self._check(self.lib.sg_match(self.ctx, f1.ctypes.data, len(f1), f2.ctypes.data, len(f2), ratio,
                                      out.ctypes.data, len(f1), C.byref(n)))
        return out[: n.value]

    def match_packed_device(self, d_f1, n1, d_f2, n2, d_index2, d_dist2=None, ratio=0.5):
        """Device-pointer form on packed features (ints); asynchronous. d_index2: n1 int32; d_dist2: 2*n1 uint32 or None."""
        self._check(self.lib.sg_match_packed_device(self.ctx, d_f1, n1, d_f2, n2, ratio, d_index2, d_dist2))

    def match_packed_device_noalloc(self, d_f1, n1, d_f2, n2, d_index2, d_dist2=None, ratio=0.5):
        """match_packed_device without any device allocation (static scratch; at most 32768 features per set)."""
        self._check(self.lib.sg_match_packed_device_noalloc(self.ctx, d_f1, n1, d_f2, n2, ratio, d_index2, d_dist2))

    # ---- multi-GPU exchange ------------------------------------------------------------------------------------
    def comm_create(self, comm_id, rank, n_ranks):
        """RCCL communicator on this context's device from the 128-byte id of comm_unique_id(); returns a handle."""
        comm = C.c_void_p()
        buf = C.create_string_buffer(bytes(comm_id), 128)
        self._check(self.lib.sg_comm_create(self.ctx, buf, rank, n_ranks, C.byref(comm)))
        return comm

    def comm_destroy(self, comm):
        self._check(self.lib.sg_comm_destroy(comm))

    def allgather_features(self, comm, d_feats, d_counts, n_img, cap_per_image, d_all, cap_all, d_all_counts, n_ranks):
        """All-gather-v of packed features (device pointers as ints); returns rank_first (n_ranks + 1 feature indices).
        comm None = single rank.  Asynchronous after the counts; sync() before reading d_all."""
        first = (C.c_size_t * (n_ranks + 1))()
        self._check(self.lib.sg_allgather_features(self.ctx, comm, d_feats, d_counts, n_img, cap_per_image, d_all,
                                                   cap_all, d_all_counts, first))
        return [int(v) for v in first]

    def compact_features(self, d_feats, d_counts, n_img, cap_per_image, d_out):
        self._check(self.lib.sg_compact_features(self.ctx, d_feats, d_counts, n_img, cap_per_image, d_out))

    # ---- inspection ------------------------------------------------------------------------------------
    def last_octaves(self):
        return self.lib.sg_last_octaves(self.ctx)

    def last_pair_octaves(self):
        """Bit mask of the octaves whose levels 1 and 2 came from one launch in the last call."""
        return int(self.lib.sg_last_pair_octaves(self.ctx))

    def last_tail_octave(self):
        """First octave of the last call that came from the pyramid-tail launch (the octave count: none did)."""
        return int(self.lib.sg_last_tail_octave(self.ctx))

    def read_level(self, octave, level, image=0):
        r, c = C.c_int(0), C.c_int(0)
        self._check(self.lib.sg_level_dims(self.ctx, octave, C.byref(r), C.byref(c)))
        out = np.empty((r.value, c.value), dtype=np.float64)
        self._check(self.lib.sg_read_level(self.ctx, image, octave, level, _dp(out)))
        return out

    def read_counters(self, image=0):
        ctr = Counters()
        self._check(self.lib.sg_read_counters(self.ctx, image, C.byref(ctr)))
        return ctr.as_tuple()

    def read_work_counters(self, n_img=1):
        """(descriptor drain trips, orientation walk trips) of the last call, summed over its first n_img images."""
        d, o = C.c_ulonglong(0), C.c_ulonglong(0)
        self._check(self.lib.sg_read_work_counters(self.ctx, n_img, C.byref(d), C.byref(o)))
        return d.value, o.value

    def read_keypoints(self, image=0, cap=1 << 20):
        out = np.zeros(cap, dtype=KEYPOINT_DTYPE)
        n = C.c_size_t(0)
        self._check(self.lib.sg_read_keypoints(self.ctx, image, out.ctypes.data, cap, C.byref(n)))
        return out[: n.value].copy()

    def read_features_unsorted(self, image=0, cap=1 << 20, prequant=False):
        out = np.zeros(cap, dtype=KEYPOINT_DTYPE)
        n = C.c_size_t(0)
        pq = np.zeros((cap, 128), dtype=np.float64) if prequant else None
        self._check(self.lib.sg_read_features_unsorted(self.ctx, image, out.ctypes.data, cap, C.byref(n),
                                                       _dp(pq) if prequant else None))
        if prequant:
            return out[: n.value].copy(), pq[: n.value].copy()
        return out[: n.value].copy()

    # ---- timing -----------------------------------------------------------------------------------------
    def set_profiling(self, on):
        self._check(self.lib.sg_set_profiling(self.ctx, int(on)))

    def reset_stage_times(self):
        self._check(self.lib.sg_reset_stage_times(self.ctx))

    def stage_times(self):
        ms = (C.c_double * len(STAGES))()
        ln = (C.c_ulonglong * len(STAGES))()
        self._check(self.lib.sg_get_stage_times(self.ctx, ms, ln))
        return {s: (ms[i], ln[i]) for i, s in enumerate(STAGES)}
