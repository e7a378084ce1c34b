"""Parity of single stages through the C ABI against the CPU oracle (bit-exact in double)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _img(oracle, w, h, seed):
    return oracle.vnoise5(w, h, seed)


def test_device_synth_matches_oracle(oracle, gpu64):
    w, h = 333, 77
    d = gpu64.malloc(w * h * 3)
    try:
        gpu64.synth_vnoise5(d, w, h, 99)
        got = np.empty((h, w, 3), dtype=np.uint8)
        gpu64.d2h(got, d)
    finally:
        gpu64.free(d)
    assert np.array_equal(got, oracle.vnoise5(w, h, 99))


@pytest.mark.parametrize("w,h", [(64, 48), (97, 31), (2, 2), (5, 3)])
def test_gray_and_upsample(oracle, gpu64, w, h):
    img = _img(oracle, w, h, 3)
    # make the channels differ so that the b,g,r weights are exercised
    img[..., 1] = np.roll(img[..., 1], 1, axis=1)
    img[..., 2] = 255 - img[..., 2]
    g_ref = oracle.convert_to_gray(img)
    g = gpu64.convert_to_gray(img)
    assert np.array_equal(g, g_ref)
    up_ref = oracle.upsample(g_ref)
    up = gpu64.upsample(g_ref)
    assert np.array_equal(up, up_ref)


def test_four_channel_input_and_stride(oracle, gpu64):
    img = _img(oracle, 40, 30, 5)
    img4 = np.concatenate([img, np.full((30, 40, 1), 7, np.uint8)], axis=2)
    assert np.array_equal(gpu64.convert_to_gray(img4), oracle.convert_to_gray(img4))


@pytest.mark.parametrize("w,h", [(64, 48), (31, 17), (15, 8), (3, 2)])
def test_downsample(oracle, gpu64, w, h):
    src = np.random.default_rng(1).random((h, w))
    assert np.array_equal(gpu64.downsample(src), oracle.downsample(src))


# sigma values of the default pyramid (11, 11, 13, 17, 21, 27 taps), one that pads into a wider
# kernel, and one beyond the fast kernels; planes smaller than the kernel exercise repeated reflection.
SIGMAS = [1.2489995996796797, 1.2262734984654078, 1.5450077936447955, 1.9465878414647133, 2.4525469969308156,
          3.0900155872895909, 0.8, 2.0, 5.0]


@pytest.mark.parametrize("sigma", SIGMAS)
@pytest.mark.parametrize("w,h", [(300, 200), (15, 8), (3, 2), (257, 33), (1, 9), (9, 1)])
def test_gaussian_blur_bitexact(oracle, gpu64, sigma, w, h):
    src = np.random.default_rng(7).random((h, w))
    ref = oracle.gaussian_blur(src, sigma)
    got = gpu64.gaussian_smooth(src, sigma)
    assert np.array_equal(got, ref)


def test_gaussian_blur_float_close(oracle, gpu32):
    src = np.random.default_rng(8).random((120, 200))
    ref = oracle.gaussian_blur(src, 3.09)
    got = gpu32.gaussian_smooth(src, 3.09)
    assert np.abs(got - ref).max() < 5e-7  # float pyramid: tolerance of one rounding per pass


def test_init_smooth_gray(oracle, gpu64):
    img = _img(oracle, 80, 60, 11)
    assert np.array_equal(gpu64.create_init_smooth_gray(img, 1.6), oracle.create_init_smooth_gray(img, 1.6))


def test_inverse3d_known_answer(oracle):
    # the matrix of the reference's testInverse3D (sift/sift.cpp:1936): H * H^-1 = I
    H = np.array([3, 2, 3, 4, 5, 6, 7, 8, 9], dtype=np.float64).reshape(3, 3)
    ok, Hi = oracle.inverse3d(H)
    assert ok and np.allclose(H @ Hi, np.eye(3), atol=1e-12)
