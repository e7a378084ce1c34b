"""Full-size configurations: known answers recorded from the reference's pipeline on the same seeded inputs,
the oracle beside the HIP path at 1080p, and size-independent properties where the oracle is too slow."""
import os

import numpy as np
import pytest

from sift_amd.parity import descriptor_agreement, match_features

pytestmark = pytest.mark.gpu

# vnoise5 inputs; (cc00, cc0, cc1, cc2, cc3), features, keypoints per octave -- recorded from the reference's own run
KNOWN_1080P_S1234 = ((32820570, 11196758, 40757, 18437, 18276), 21955, [7090, 8269, 2240, 534, 131, 8, 4, 0, 0])
KNOWN_1080P_S7 = (18386, 22151, [7092, 8408, 2228, 536, 105, 17, 0, 0, 0])
KNOWN_8K_S1234 = (298170, 360706, [115219, 135575, 36559, 8562, 1930, 300, 24, 1, 0, 0, 0])


@pytest.fixture(scope="module")
def ctx1080():
    import sift_amd
    c = sift_amd.SiftGPU(max_width=1920, max_height=1080, max_batch=2)
    yield c
    c.close()


def _per_octave(kp, octaves):
    return np.bincount(kp["octave"], minlength=octaves).tolist()


def test_1080p_known_answers_and_oracle(oracle, ctx1080):
    import sift_amd
    img = sift_amd.vnoise5(1920, 1080, 1234)
    got = ctx1080.extract(img)
    counters, n_feat, per_oct = KNOWN_1080P_S1234
    assert ctx1080.read_counters() == counters
    assert len(got) == n_feat
    assert _per_octave(ctx1080.read_keypoints(), 9) == per_oct
    assert (np.diff(got["scale"]) >= 0).all()
    # the oracle on the same image (about 7 s of CPU)
    ref, nk, ctr, _ = oracle.sift(img)
    assert ctr == counters and len(ref) == n_feat
    for f in ("x", "y", "octave", "interval"):
        assert np.array_equal(got[f], ref[f]), f
    assert np.array_equal(got["descriptor"], ref["descriptor"])
    pairs = match_features(ref, got)
    assert len(pairs) == len(ref)  # every feature within 0.5 px, 1 % scale, 1 degree
    assert descriptor_agreement(ref["descriptor"][pairs[:, 0]], got["descriptor"][pairs[:, 1]]) == (1.0, 1.0, 0.0)


def test_1080p_second_seed_counts(ctx1080):
    import sift_amd
    got = ctx1080.extract(sift_amd.vnoise5(1920, 1080, 7))
    n_kp, n_feat, per_oct = KNOWN_1080P_S7
    assert len(got) == n_feat
    kp = ctx1080.read_keypoints()
    assert len(kp) == n_kp and _per_octave(kp, 9) == per_oct


def test_1080p_batch_is_deterministic_and_equals_single(ctx1080):
    import sift_amd
    imgs = np.stack([sift_amd.vnoise5(1920, 1080, 1000 + i) for i in range(2)])
    a = ctx1080.extract_batch(imgs, 32768)
    b = ctx1080.extract_batch(imgs, 32768)
    for i in range(2):
        assert a[i].tobytes() == b[i].tobytes()
        single = ctx1080.extract(imgs[i])
        assert len(single) == len(a[i])
        assert np.array_equal(single["descriptor"].astype(np.uint8), a[i]["desc"])
        assert (np.diff(a[i]["scale"]) >= 0).all()


def test_8k_image_known_counts():
    """7680x4320 with the upsampled base octave (11 octaves, 15360x8640 base): counts recorded from the reference."""
    import sift_amd
    n_kp, n_feat, per_oct = KNOWN_8K_S1234
    with sift_amd.SiftGPU(max_width=7680, max_height=4320) as ctx:
        d_img = ctx.malloc(7680 * 4320 * 3)
        cap = 400000
        d_out = ctx.malloc(cap * 144)
        d_cnt = ctx.malloc(4)
        try:
            ctx.synth_vnoise5(d_img, 7680, 4320, 1234)
            ctx.extract_batch_device(d_img, 1, 7680, 4320, 3, d_out, cap, d_cnt)
            ctx.sync()
            cnt = np.zeros(1, dtype=np.uint32)
            ctx.d2h(cnt, d_cnt)
            out = np.zeros(cap, dtype=sift_amd.PACKED_DTYPE)
            ctx.d2h(out, d_out)
        finally:
            ctx.free(d_img)
            ctx.free(d_out)
            ctx.free(d_cnt)
        assert ctx.last_octaves() == 11
        assert int(cnt[0]) == n_feat
        kp = ctx.read_keypoints()
        assert len(kp) == n_kp and _per_octave(kp, 11) == per_oct
        f = out[:n_feat]
        assert (np.diff(f["scale"]) >= 0).all()
        assert f["x"].min() >= 0 and f["x"].max() < 7680 and f["y"].min() >= 0 and f["y"].max() < 4320
        assert f["desc"].max() <= 255 and f["desc"].any(axis=1).all()


def test_8k_image_equals_oracle_digest(oracle):
    """The 8K configuration against the oracle's full output, through a digest committed under tests/golden/ (the
    oracle takes minutes at this size; tests/golden/make_golden_8k.py wrote the digest): counters, the
    integer fields, dx / dy bit for bit and all descriptor bins of all 360 706 features in final order by hash, every
    8th orientation and scale by value."""
    import hashlib
    import sift_amd
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sift_7680x4320_s1234.npz"))
    img = oracle.vnoise5(7680, 4320, 1234)
    with sift_amd.SiftGPU(max_width=7680, max_height=4320) as ctx:
        f = ctx.extract(img, cap=400000)
        assert list(ctx.read_counters()) == [int(v) for v in g["counters"]]
        assert len(ctx.read_keypoints()) == int(g["keypoints"])
    assert len(f) == int(g["features"]) == KNOWN_8K_S1234[1]
    ints = np.ascontiguousarray(np.stack([f[k] for k in ("x", "y", "octave", "interval")], axis=1).astype("<i4"))
    f64 = np.ascontiguousarray(np.stack([f[k] for k in ("dx", "dy")], axis=1).astype("<f8"))
    desc = np.ascontiguousarray(f["descriptor"]).astype(np.uint8)
    assert hashlib.sha256(ints.tobytes()).hexdigest() == str(g["sha_ints"])
    assert hashlib.sha256(f64.tobytes()).hexdigest() == str(g["sha_f64"])
    assert hashlib.sha256(desc.tobytes()).hexdigest() == str(g["sha_desc"])
    d = np.abs(f["ori"][::8] - g["ori_every_8th"])
    d = np.minimum(d, 2 * np.pi - d)
    assert d.max() < 1e-9
    assert np.allclose(f["scale"][::8], g["scale_every_8th"], rtol=1e-14, atol=0)


def test_1080p_batch256_equals_single(oracle):
    """The configuration bench.py times: one context of 256 1080p frames (17 GB per pyramid level, image offsets
    beyond 2^32 elements).  Every frame of the batch must come out byte-for-byte as the same frame run alone
    (Sift() is a pure function of one frame's pixels, reference sift/sift.cpp:1646-1707); frames 0, 127 and 255
    are also checked against the oracle.  The batch is halved while the context does not fit, as bench.py does."""
    import sift_amd
    W, H, cap = 1920, 1080, 26624
    B = 256
    big = None
    while big is None:
        try:
            big = sift_amd.SiftGPU(max_width=W, max_height=H, max_batch=B)
        except sift_amd.SiftGpuError as e:
            if e.status != sift_amd.api.SG_E_NOMEM or B <= 8:
                raise
            B //= 2
    one = sift_amd.SiftGPU(max_width=W, max_height=H, max_batch=1)
    fb = W * H * 3
    d_img = big.malloc(B * fb)
    d_out = big.malloc(B * cap * 144)
    d_cnt = big.malloc(B * 4)
    d_out1 = one.malloc(cap * 144)
    d_cnt1 = one.malloc(4)
    try:
        for i in range(B):
            big.synth_vnoise5(d_img + i * fb, W, H, 1000 + i)
        big.sync()
        big.extract_batch_device(d_img, B, W, H, 3, d_out, cap, d_cnt)
        big.sync()  # raises on any capacity overflow
        counts = np.zeros(B, dtype=np.uint32)
        big.d2h(counts, d_cnt)
        assert counts.min() > 15000 and counts.max() < cap, (counts.min(), counts.max())
        check = {0, B // 2 - 1, B - 1}
        for i in range(B):
            one.extract_batch_device(d_img + i * fb, 1, W, H, 3, d_out1, cap, d_cnt1)
            one.sync()
            c1 = np.zeros(1, dtype=np.uint32)
            one.d2h(c1, d_cnt1)
            assert int(c1[0]) == int(counts[i]), (i, int(c1[0]), int(counts[i]))
            a = np.zeros(int(c1[0]), dtype=sift_amd.PACKED_DTYPE)
            b = np.zeros(int(c1[0]), dtype=sift_amd.PACKED_DTYPE)
            one.d2h(a, d_out1)
            big.d2h(b, d_out + i * cap * 144)
            assert a.tobytes() == b.tobytes(), "frame %d of the batch differs from the same frame run alone" % i
            if i in check:
                ctr_b = big.read_counters(image=i)
                assert ctr_b == one.read_counters()
                ref, nk, ctr, _ = oracle.sift(sift_amd.vnoise5(W, H, 1000 + i))
                assert ctr == ctr_b, (i, ctr, ctr_b)
                assert len(ref) == len(b)
                assert np.array_equal(ref["descriptor"].astype(np.uint8), b["desc"]), i
                assert np.array_equal(ref["dx"].astype(np.float32), b["x"]) and \
                    np.array_equal(ref["dy"].astype(np.float32), b["y"])
                assert np.allclose(ref["scale"], b["scale"], rtol=1e-6) and np.abs(ref["ori"] - b["ori"]).max() < 1e-6
    finally:
        for c, p in ((big, d_img), (big, d_out), (big, d_cnt), (one, d_out1), (one, d_cnt1)):
            c.free(p)
        big.close()
        one.close()


def test_frame_next_to_the_size_limit_equals_its_crop_locally(oracle):
    """15360x8640 input: the base plane has 4.25e9 bytes, just under the 4 GB the 32-bit pixel offsets of the keypoint
    kernels reach (larger sizes are refused).  No oracle at this size; instead the bottom 2048 rows are run again as
    an image of their own: away from the cut (the blurs are local) the same features must come out, with equal integer
    coordinates, offsets and descriptor bins -- which they only do if every address up to the last row is right."""
    import sift_amd
    W, H, CROP = 15360, 8640, 2048
    img = oracle.vnoise5(W, H, 77)
    top = H - CROP  # a multiple of 64: the octaves' sampling grids of frame and crop coincide
    assert top % 64 == 0
    with sift_amd.SiftGPU(max_width=W, max_height=H) as c:
        full = c.extract(img, cap=1600000)
        crop = c.extract(np.ascontiguousarray(img[top:]), cap=500000)
    assert len(full) > 1300000 and (np.diff(full["scale"]) >= 0).all()
    # features of octaves 0..2 whose windows stay 400 input pixels below the cut
    def pick(f, y0):
        m = (f["octave"] <= 2) & (f["dy"] > y0 + 400)
        f = f[m]
        order = np.lexsort((f["ori"], f["x"], f["y"], f["interval"], f["octave"]))
        return f[order]
    a, b = pick(full, top), pick(crop, 0)
    assert len(a) == len(b) > 250000
    assert np.array_equal(a["x"], b["x"]) and np.array_equal(a["octave"], b["octave"])
    assert np.array_equal(a["y"], b["y"] + (top * 2 >> a["octave"]))  # y is in octave pixels of the doubled image
    for f in ("offset_x", "offset_y", "offset_interval", "scale", "val", "dx"):
        assert np.array_equal(a[f], b[f]), f
    assert np.abs(a["dy"] - (b["dy"] + top)).max() < 1e-9  # (y + offset) rounds once, with y counted from another row
    assert np.abs(a["ori"] - b["ori"]).max() < 1e-9
    assert np.array_equal(a["descriptor"], b["descriptor"])
