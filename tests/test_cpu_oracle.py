"""CPU-side checks: the oracle against recorded known answers, host helpers, and the C ABI's export list."""
import ctypes
import hashlib
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_keypoint_layout(oracle):
    import sift_amd
    assert sift_amd.KEYPOINT_DTYPE.itemsize == 1120
    assert oracle.KEYPOINT_DTYPE == sift_amd.KEYPOINT_DTYPE
    assert oracle.lib().so_sizeof_keypoint() == 1120


def test_numpy_synth_matches_oracle(oracle):
    import sift_amd
    assert np.array_equal(sift_amd.vnoise5(133, 71, 1234), oracle.vnoise5(133, 71, 1234))


def test_blur_taps_of_default_pyramid(oracle):
    # sigma = 1.6, 3 intervals: 11 taps for the base blur, then 11, 13, 17, 21, 27
    s = oracle.pyramid_sigmas(1.6, 3)
    assert [len(oracle.gaussian_kernel(x)) for x in s[1:]] == [11, 13, 17, 21, 27]
    assert len(oracle.gaussian_kernel(np.sqrt(1.6 * 1.6 - 1.0))) == 11
    k = oracle.gaussian_kernel(s[5])
    assert abs(k.sum() - 1.0) < 1e-15 and np.array_equal(k, k[::-1])


def test_blur_reflect101_small_plane(oracle):
    # a 3x2 plane under a 27-tap kernel: repeated reflection; a constant plane stays constant
    src = np.full((2, 3), 0.25)
    assert np.allclose(oracle.gaussian_blur(src, 3.09), 0.25, atol=1e-15)
    # a 1-D check against an explicit reflect-101 convolution
    row = np.random.default_rng(0).random((1, 15))
    k = oracle.gaussian_kernel(3.09)
    h = len(k) // 2
    idx = np.arange(-h, 15 + h)
    idx = np.abs(idx)
    period = 2 * (15 - 1)
    idx = idx % period
    idx = np.where(idx >= 15, period - idx, idx)
    ext = row[0][idx]
    ref = np.array([sum(k[t] * ext[x + t] for t in range(len(k))) for x in range(15)])
    assert np.allclose(oracle.gaussian_blur(row, 3.09)[0], ref, atol=1e-15)


def test_blur_against_scipy_mirror_correlation(oracle):
    """An independent evaluation of the documented cv::GaussianBlur semantics (taps n = cvRound(8 sigma + 1) | 1,
    exp(-x^2 / 2 sigma^2) normalised, BORDER_REFLECT_101 = scipy's 'mirror', rows then columns): scipy.ndimage's
    separable correlation agrees with the oracle's blur to rounding, on planes larger and smaller than the kernel."""
    ndimage = pytest.importorskip("scipy.ndimage")
    rng = np.random.default_rng(5)
    for shape, sigma in (((48, 64), 1.249), ((37, 53), 3.090016), ((9, 40), 3.090016), ((30, 7), 2.452547)):
        src = rng.random(shape)
        n = int(np.rint(8 * sigma + 1)) | 1
        x = np.arange(n) - (n - 1) / 2
        k = np.exp(-(x * x) / (2 * sigma * sigma))
        k /= k.sum()
        assert np.allclose(oracle.gaussian_kernel(sigma), k, rtol=0, atol=1e-16)
        want = ndimage.correlate1d(ndimage.correlate1d(src, k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
        got = oracle.gaussian_blur(src, sigma)
        assert np.allclose(got, want, rtol=0, atol=4e-15), (shape, sigma, np.abs(got - want).max())


def test_upsample_edges(oracle):
    src = np.arange(12, dtype=np.float64).reshape(3, 4)
    up = oracle.upsample(src)
    assert up.shape == (6, 8)
    assert np.array_equal(up[:, 6], up[:, 5]) and np.array_equal(up[:, 7], up[:, 5])
    assert np.array_equal(up[4], up[3]) and np.array_equal(up[5], up[3])
    assert up[0, 0] == 0 and up[1, 0] == 2.0 and up[0, 1] == 0.5 and up[1, 1] == 2.5


def test_oracle_known_counters_640x480(oracle):
    """Counters and counts recorded from the reference's own pipeline on vnoise5(640,480,1234)."""
    img = oracle.vnoise5(640, 480, 1234)
    feats, nk, ctr, _ = oracle.sift(img)
    assert ctr == (4783650, 1630291, 6001, 2724, 2700)
    assert nk == 2700 and len(feats) == 3248
    assert (np.diff(feats["scale"]) >= 0).all()
    assert abs(feats["scale"][0] - 1.12425) < 1e-5 and abs(feats["scale"][-1] - 42.7676) < 1e-4
    d = feats["descriptor"]
    assert d.min() == 0 and d.max() == 159


def test_oracle_output_checksum_640x480(oracle, tmp_path):
    """md5 of the reference's write_features text on vnoise5(640,480,1234), recorded when the
    reference's own pipeline was run on that input; ties ordered by libstdc++ std::sort as there."""
    oracle.build()
    exe = os.path.join(ROOT, "oracle", "pin_check")
    subprocess.check_call(["g++", "-O2", "-std=c++14", "-o", exe, os.path.join(ROOT, "oracle", "pin_check.cpp"),
                           "-L" + os.path.join(ROOT, "oracle"), "-lsift_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    out = str(tmp_path / "d.txt")
    subprocess.check_call([exe, "640", "480", "1234", out], stdout=subprocess.DEVNULL)
    data = open(out, "rb").read()
    assert data.count(b"\n") == 32481
    assert hashlib.md5(data).hexdigest() == "3db9fc42b44c64398677f3042e8059e2"


def test_flat_image(oracle):
    img = np.full((48, 64, 3), 200, dtype=np.uint8)
    feats, nk, ctr, _ = oracle.sift(img)
    assert len(feats) == 0 and nk == 0 and ctr[1] == 0


def test_parity_matcher():
    import sift_amd
    from sift_amd.parity import descriptor_agreement, match_features
    a = np.zeros(3, dtype=sift_amd.KEYPOINT_DTYPE)
    a["dx"], a["dy"], a["scale"], a["ori"] = [1, 5, 9], [1, 5, 9], [2, 2, 2], [0.1, 3.1, -3.1]
    b = a.copy()[::-1]
    b["dx"] += 0.3
    b["ori"][0] = 3.18 - 2 * np.pi  # wraps to within 1 degree of -3.1? no: 3.18 rad vs -3.1 rad = 0.003 rad apart
    pairs = match_features(a, b)
    assert len(pairs) == 3 and sorted(pairs[:, 1].tolist()) == [0, 1, 2]
    assert descriptor_agreement(a["descriptor"], b["descriptor"][::-1])[0] == 1.0


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads without a GPU and exports every function include/siftgpu.h declares."""
    import re
    import sift_amd
    hdr = open(os.path.join(ROOT, "include", "siftgpu.h")).read()
    declared = set(re.findall(r"\b(sg_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(sift_amd.api.EXPORTS)
    lib = ctypes.CDLL(sift_amd.api.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), name
    sift_amd.load_library()


def test_no_cpu_fallback_without_device():
    """Without a HIP device the product refuses to construct a context (no silent CPU path)."""
    import sift_amd
    lib = sift_amd.load_library()
    if lib.sg_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(sift_amd.SiftGpuError):
        sift_amd.SiftGPU()


def test_parameter_limits_are_reported_before_any_device_work():
    """Argument checks of sg_create that need no device: an image size whose base plane would pass 4 GB (the keypoint
    kernels' 32-bit pixel offsets), intervals other than 3, a base sigma of 1 or less."""
    import sift_amd
    from sift_amd import api
    for kw, want in (({"max_width": 16384, "max_height": 8192}, api.SG_E_UNSUPPORTED),
                     ({"max_width": 40000, "max_height": 40000}, api.SG_E_UNSUPPORTED),
                     ({"intervals": 4}, api.SG_E_UNSUPPORTED),
                     ({"sigma": 1.0}, api.SG_E_INVALID),
                     ({"max_batch": 0}, api.SG_E_INVALID)):
        with pytest.raises(sift_amd.SiftGpuError) as e:
            sift_amd.SiftGPU(**kw)
        assert e.value.status == want, kw


def test_match_known_answer(oracle):
    """The demo's matching loop (sift/源.cpp:88-156) on hand-made descriptors."""
    import numpy as np
    from oracle.oracle import KEYPOINT_DTYPE
    f1 = np.zeros(2, dtype=KEYPOINT_DTYPE)
    f2 = np.zeros(4, dtype=KEYPOINT_DTYPE)
    f1["descriptor"][0, :4] = [10, 0, 0, 0]
    f1["descriptor"][1, :4] = [0, 50, 0, 0]
    f2["descriptor"][0, :4] = [13, 4, 0, 0]    # distance 5 from f1[0]
    f2["descriptor"][1, :4] = [10, 0, 12, 5]   # distance 13 from f1[0]
    f2["descriptor"][2, :4] = [0, 47, 4, 0]    # distance 5 from f1[1]
    f2["descriptor"][3, :4] = [0, 50, 6, 0]    # distance 6 from f1[1]: ratio 5/6 fails 0.5
    pairs, best2 = oracle.match(f1, f2, 0.5)
    assert pairs.tolist() == [[0, 0]]
    assert best2[0].tolist() == [25.0, 169.0]
    assert best2[1].tolist() == [25.0, 36.0]
    assert oracle.match(f1, f2, 0.9)[0].tolist() == [[0, 0], [1, 2]]
    # ties: the first index wins, and the second smallest equals the smallest (ratio 1)
    f2["descriptor"][1] = f2["descriptor"][0]
    assert oracle.match(f1[:1], f2, 1.01)[0].tolist() == [[0, 0]]
    assert oracle.match(f1[:1], f2, 1.0)[0].tolist() == []
    # one candidate only: the second distance stays DBL_MAX and the ratio test passes
    assert oracle.match(f1[:1], f2[:1], 0.5)[0].tolist() == [[0, 0]]
    assert oracle.match(f1[:1], f2[:0], 0.5)[0].tolist() == []


def test_match_demo_accumulation_differs(oracle):
    """The demo never resets its distance variable (源.cpp:97-110); the literal form sees sqrt(previous + d^2)."""
    import numpy as np
    from oracle.oracle import KEYPOINT_DTYPE
    f1 = np.zeros(1, dtype=KEYPOINT_DTYPE)
    f2 = np.zeros(3, dtype=KEYPOINT_DTYPE)
    f2["descriptor"][0, 0] = 200   # far
    f2["descriptor"][1, 0] = 3     # near: d = 3; literal form: sqrt(200 + 9) = 14.46
    f2["descriptor"][2, 0] = 8     # d = 8; literal form: sqrt(14.46 + 64) = 8.86
    assert oracle.match(f1, f2, 0.5)[0].tolist() == [[0, 1]]                        # 3 / 8 < 0.5
    assert oracle.match(f1, f2, 0.5, accumulate_bug=True)[0].tolist() == []         # 8.86 / 14.46 > 0.5
