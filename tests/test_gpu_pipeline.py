"""Parity of the whole path through the C ABI against the CPU oracle on the same seeded inputs."""
import ctypes as C

import numpy as np
import pytest

from sift_amd.parity import descriptor_agreement, match_features

pytestmark = pytest.mark.gpu

FIELDS_EXACT = ("octave", "interval", "x", "y")
FIELDS_F64 = ("offset_interval", "offset_x", "offset_y", "dx", "dy", "val")


@pytest.fixture(scope="module", params=[(96, 64, 1), (200, 150, 1234), (320, 240, 7)])
def case(request, oracle, gpu64):
    w, h, seed = request.param
    img = oracle.vnoise5(w, h, seed)
    dbg = oracle.Debug(img, prequant=True)
    feats = gpu64.extract(img)
    return dict(img=img, dbg=dbg, feats=feats, ctx=gpu64)


def test_pyramid_levels_bitexact(case):
    dbg, ctx = case["dbg"], case["ctx"]
    assert ctx.last_octaves() == dbg.octaves
    for o in range(dbg.octaves):
        for l in range(6):
            got = ctx.read_level(o, l)
            assert np.array_equal(got, dbg.gauss[o * 6 + l]), "octave %d level %d" % (o, l)


def test_counters_and_keypoints(case):
    dbg, ctx = case["dbg"], case["ctx"]
    assert ctx.read_counters() == dbg.counters
    kps = ctx.read_keypoints()
    ref = dbg.keypoints
    assert len(kps) == len(ref)
    for f in FIELDS_EXACT:
        assert np.array_equal(kps[f], ref[f]), f
    for f in FIELDS_F64:  # same operation order in double, no contraction: identical bits
        assert np.array_equal(kps[f], ref[f]), f
    # pow() comes from different math libraries: allow an ulp
    assert np.allclose(kps["scale"], ref["scale"], rtol=1e-14, atol=0)
    assert np.allclose(kps["octave_scale"], ref["octave_scale"], rtol=1e-14, atol=0)
    # 2^y is evaluated correctly rounded on the device; the C library's pow() is too, but for about one argument in a
    # thousand: bit-equal scales but for a handful
    allowed = max(2, len(ref) // 200)
    assert (kps["scale"] != ref["scale"]).sum() <= allowed and (kps["octave_scale"] != ref["octave_scale"]).sum() <= allowed


def test_features_unsorted(case):
    dbg, ctx = case["dbg"], case["ctx"]
    got, pq = ctx.read_features_unsorted(prequant=True)
    ref = dbg.features
    assert len(got) == len(ref)
    for f in FIELDS_EXACT:
        assert np.array_equal(got[f], ref[f]), f
    d = np.abs(got["ori"] - ref["ori"])
    d = np.minimum(d, 2 * np.pi - d)
    assert d.max() < 1e-9
    # float descriptor before quantisation: L2 distance after normalisation
    l2 = np.linalg.norm(pq - dbg.prequant, axis=1)
    assert l2.max() < 1e-9
    # integer bins identical
    assert np.array_equal(got["descriptor"], ref["descriptor"])
    assert (got["descr_length"] == 128).all()


def test_sorted_output_order(case):
    dbg, feats = case["dbg"], case["feats"]
    ref = dbg.sorted_features
    assert len(feats) == len(ref)
    assert (np.diff(feats["scale"]) >= 0).all()
    for f in FIELDS_EXACT:
        assert np.array_equal(feats[f], ref[f]), f
    assert np.array_equal(feats["descriptor"], ref["descriptor"])


def test_north_star_tolerances(case):
    dbg, feats = case["dbg"], case["feats"]
    ref = dbg.sorted_features
    pairs = match_features(ref, feats, pos_tol=0.5, scale_tol=0.01, ori_tol_deg=1.0)
    assert len(pairs) >= 0.99 * len(ref)
    same_desc, same_bins, l2 = descriptor_agreement(ref["descriptor"][pairs[:, 0]], feats["descriptor"][pairs[:, 1]])
    assert same_desc >= 0.99 and l2 < 1e-3


def test_determinism(case):
    ctx, img = case["ctx"], case["img"]
    a = ctx.extract(img)
    b = ctx.extract(img)
    # field by field (numpy does not carry a record's padding bytes through a copy)
    for f in a.dtype.names:
        assert a[f].tobytes() == b[f].tobytes(), f
        assert a[f].tobytes() == case["feats"][f].tobytes(), f


def test_float_pyramid_meets_tolerances(oracle, gpu32):
    img = oracle.vnoise5(320, 240, 7)
    ref, _, _, _ = oracle.sift(img)
    got = gpu32.extract(img)
    pairs = match_features(ref, got)
    assert len(pairs) >= 0.99 * len(ref)
    assert len(got) <= 1.01 * len(ref) + 2
    same_desc, same_bins, l2 = descriptor_agreement(ref["descriptor"][pairs[:, 0]], got["descriptor"][pairs[:, 1]])
    # a float pyramid flips a few (int)(512 v) truncations by one: bins agree to 99.9 %, descriptors to > 95 %
    assert same_bins >= 0.999
    assert same_desc >= 0.95


def test_nointerval_variant(oracle):
    import sift_amd
    img = oracle.vnoise5(200, 150, 21)
    ref, nk, ctr, _ = oracle.sift(img, nointerval=True)
    with sift_amd.SiftGPU(max_width=200, max_height=150, nointerval=True) as ctx:
        got = ctx.extract(img)
        assert ctx.read_counters() == ctr
    assert len(got) == len(ref)
    assert np.array_equal(got["descriptor"], ref["descriptor"])
    assert np.array_equal(got["x"], ref["x"]) and np.array_equal(got["y"], ref["y"])


@pytest.mark.parametrize("w,h,seed", [(640, 480, 3), (131, 77, 3), (23, 310, 9)])
def test_staged_and_register_forms_agree(oracle, gpu64, gpu32, w, h, seed, monkeypatch):
    """The blur and the extrema scan each have two forms: input rows staged through LDS by direct-to-LDS loads (the
    ones normally launched) and rows loaded into registers (for planes whose rows are not 16-byte aligned;
    SIFTGPU_BLUR_DIRECT=1 / SIFTGPU_DETECT_DIRECT=1 select them everywhere).  Both give the oracle's pyramid, counters
    and features in double, and identical features to each other in float and without the neighbouring-layer
    comparison."""
    import sift_amd
    img = oracle.vnoise5(w, h, seed)
    ref, _, ctr, _ = oracle.sift(img)
    results = []
    for direct in (False, True):
        for var in ("SIFTGPU_DETECT_DIRECT", "SIFTGPU_BLUR_DIRECT"):
            if direct:
                monkeypatch.setenv(var, "1")
            else:
                monkeypatch.delenv(var, raising=False)
        got = gpu64.extract(img)
        assert gpu64.read_counters() == ctr, direct
        if direct:
            dbg = oracle.Debug(img)
            for o in range(dbg.octaves):
                for l in range(6):
                    assert np.array_equal(gpu64.read_level(o, l), dbg.gauss[o * 6 + l]), "octave %d level %d" % (o, l)
        assert len(got) == len(ref)
        for f in FIELDS_EXACT:
            assert np.array_equal(got[f], ref[f]), (f, direct)
        assert np.array_equal(got["descriptor"], ref["descriptor"])
        f32 = gpu32.extract(img)
        c32 = gpu32.read_counters()
        with sift_amd.SiftGPU(max_width=w, max_height=h, nointerval=True) as ctx:
            ni = ctx.extract(img)
            cni = ctx.read_counters()
        results.append((f32, c32, ni, cni))
    (a32, ac32, ani, acni), (b32, bc32, bni, bcni) = results
    assert ac32 == bc32 and acni == bcni
    for f in a32.dtype.names:
        assert a32[f].tobytes() == b32[f].tobytes(), f
        assert ani[f].tobytes() == bni[f].tobytes(), f


@pytest.mark.parametrize("sigma", [3.2, 8.0])
def test_wide_windows_through_stage_entry_points(oracle, gpu64, sigma):
    """Keypoints whose orientation / descriptor windows outgrow the common kernels' tables (orientation
    radius >= 64 and descriptor radius > 127 at the larger sigma) go through the wide-window kernel
    variants, here through the OrientationAssignment / DescriptorRepresentation entry points on the oracle's pyramid
    and keypoints at that sigma."""
    img = oracle.vnoise5(256, 192, 5)
    dbg = oracle.Debug(img, sigma=sigma)
    assert len(dbg.features) > 20
    feats = gpu64.orientation_assignment(dbg.keypoints, dbg.gauss, dbg.octaves)
    ref = dbg.features
    assert len(feats) == len(ref)
    for f in FIELDS_EXACT:
        assert np.array_equal(feats[f], ref[f]), f
    d = np.abs(feats["ori"] - ref["ori"])
    assert np.minimum(d, 2 * np.pi - d).max() < 1e-9
    feats = gpu64.descriptor_representation(feats, dbg.gauss, dbg.octaves)
    assert np.array_equal(feats["descriptor"], ref["descriptor"])


def test_batch_device_matches_single(oracle, gpu64):
    import sift_amd
    w, h, n = 160, 120, 3
    imgs = np.stack([oracle.vnoise5(w, h, 100 + i) for i in range(n)])
    singles = [gpu64.extract(imgs[i]) for i in range(n)]
    cap = 4096
    packed = gpu64.extract_batch(imgs, cap)
    for i in range(n):
        s = singles[i]
        p = packed[i]
        assert len(p) == len(s)
        assert np.array_equal(p["desc"], s["descriptor"].astype(np.uint8))
        assert np.allclose(p["x"], s["dx"], atol=1e-3) and np.allclose(p["scale"], s["scale"], rtol=1e-6)
    # device-pointer form with on-device synthetic input
    d_img = gpu64.malloc(n * w * h * 3)
    d_out = gpu64.malloc(n * cap * 144)
    d_cnt = gpu64.malloc(n * 4)
    try:
        for i in range(n):
            gpu64.synth_vnoise5(d_img + i * w * h * 3, w, h, 100 + i)
        gpu64.extract_batch_device(d_img, n, w, h, 3, d_out, cap, d_cnt)
        gpu64.sync()
        cnt = np.zeros(n, dtype=np.uint32)
        gpu64.d2h(cnt, d_cnt)
        out = np.zeros((n, cap), dtype=sift_amd.PACKED_DTYPE)
        gpu64.d2h(out, d_out)
    finally:
        gpu64.free(d_img)
        gpu64.free(d_out)
        gpu64.free(d_cnt)
    for i in range(n):
        assert cnt[i] == len(singles[i])
        assert out[i, : cnt[i]].tobytes() == packed[i].tobytes()


@pytest.mark.parametrize("pinned", [False, True])
def test_batch_wire_records_equal_single_image_records(oracle, pinned):
    """sg_extract_batch_begin / sg_extract_batch_fetch_wire / sg_expand_wire: several host images in one batch come
    back as the same 1120-byte records, byte for byte, that sg_extract returns for each alone (an image without
    features in the middle of the batch included)."""
    import sift_amd
    w, h, n = 150, 113, 4
    imgs = np.stack([oracle.vnoise5(w, h, 300 + i) for i in range(n)])
    imgs[2] = 90  # flat: no features
    with sift_amd.SiftGPU(max_width=w, max_height=h, max_batch=n) as g:
        singles = [g.extract(imgs[i]) for i in range(n)]
        got = g.extract_batch_wire(imgs, pinned=pinned)
        assert [len(x) for x in got] == [len(x) for x in singles] and len(singles[2]) == 0 and len(singles[0]) > 50
        def same(a, b):  # field by field: a copy of a numpy record array leaves the padding bytes undefined
            return len(a) == len(b) and all(np.array_equal(a[f], b[f]) for f in a.dtype.names)
        for a, b in zip(got, singles):
            assert same(a, b)
        # the same records when they come down in pieces (a piece per 37 records: images end inside pieces' ranges)
        for a, b in zip(g.extract_batch_wire(imgs, pinned=pinned, piece_records=37), singles):
            assert same(a, b)
        n_pieces = C.c_int(0)
        assert g.lib.sg_extract_batch_fetch_wire_wait(g.ctx, 10 ** 6) == sift_amd.api.SG_E_INVALID
        # fewer images than the context's batch, and a fetch into too small an array
        got = g.extract_batch_wire(imgs[:2], pinned=pinned)
        assert same(got[1], singles[1])
        wire = np.zeros(4, dtype=sift_amd.WIRE_DTYPE)
        assert g.lib.sg_extract_batch_fetch_wire(g.ctx, wire.ctypes.data, 4) == sift_amd.api.SG_E_CAPACITY
        assert g.lib.sg_extract_batch_fetch_wire_begin(g.ctx, wire.ctypes.data, 4, 16, C.byref(n_pieces)) == sift_amd.api.SG_E_CAPACITY
        # a single-image call in between drops the pending batch
        g.extract(imgs[0])
        assert g.lib.sg_extract_batch_fetch_wire(g.ctx, wire.ctypes.data, 4) == sift_amd.api.SG_E_INVALID
        assert g.lib.sg_extract_batch_fetch_wire_begin(g.ctx, wire.ctypes.data, 4, 16, C.byref(n_pieces)) == sift_amd.api.SG_E_INVALID
        assert g.lib.sg_extract_batch_fetch_wire_wait(g.ctx, 0) == sift_amd.api.SG_E_INVALID


def test_batch_device_output_at_unaligned_address(oracle, gpu64):
    """Packed records are 4-byte aligned by type; an output buffer that does not start on a 16-byte boundary takes
    the packing kernel's other form and gives the same records."""
    import sift_amd
    w, h, n, cap = 160, 120, 2, 2048
    imgs = np.stack([oracle.vnoise5(w, h, 300 + i) for i in range(n)])
    packed = gpu64.extract_batch(imgs, cap)
    d_img = gpu64.malloc(n * w * h * 3)
    d_out = gpu64.malloc(n * cap * 144 + 16)
    d_cnt = gpu64.malloc(n * 4)
    try:
        gpu64.h2d(d_img, imgs)
        gpu64.extract_batch_device(d_img, n, w, h, 3, d_out + 4, cap, d_cnt)
        gpu64.sync()
        cnt = np.zeros(n, dtype=np.uint32)
        gpu64.d2h(cnt, d_cnt)
        out = np.zeros((n, cap), dtype=sift_amd.PACKED_DTYPE)
        gpu64.d2h(out, d_out + 4)
    finally:
        gpu64.free(d_img)
        gpu64.free(d_out)
        gpu64.free(d_cnt)
    for i in range(n):
        assert cnt[i] == len(packed[i])
        assert out[i, : cnt[i]].tobytes() == packed[i].tobytes()


def test_flat_image_has_no_features(gpu64):
    img = np.full((64, 64, 3), 128, dtype=np.uint8)
    assert len(gpu64.extract(img)) == 0
    assert gpu64.read_counters()[1:] == (0, 0, 0, 0)


def test_capacity_overflow_is_reported(oracle):
    import sift_amd
    img = oracle.vnoise5(200, 150, 5)
    with sift_amd.SiftGPU(max_width=200, max_height=150, max_features=16) as ctx:
        with pytest.raises(sift_amd.SiftGpuError) as e:
            ctx.extract(img)
        assert e.value.status == sift_amd.api.SG_E_CAPACITY


def test_stage_entry_points_chain(oracle, gpu64):
    """GaussianPyramid -> DogPyramid -> DetectionLocalExtrema -> CalculateScale/HalfFeatures ->
    OrientationAssignment -> DescriptorRepresentation -> sort, each through its own entry point."""
    img = oracle.vnoise5(160, 120, 77)
    dbg = oracle.Debug(img)
    base = gpu64.create_init_smooth_gray(img, 1.6)
    octaves = oracle.num_octaves(*base.shape)
    gauss = gpu64.gaussian_pyramid(base, octaves)
    for a, b in zip(gauss, dbg.gauss):
        assert np.array_equal(a, b)
    dog = gpu64.dog_pyramid(gauss, octaves)
    for a, b in zip(dog, dbg.dog):
        assert np.array_equal(a, b)
    ext, ctr = gpu64.detect_local_extrema(dog, octaves)
    assert ctr == dbg.counters
    ext = gpu64.half_features(gpu64.calculate_scale(ext))
    for f in ("x", "y", "octave", "interval", "dx", "dy", "scale", "octave_scale", "val"):
        assert np.array_equal(ext[f], dbg.keypoints[f]), f
    feats = gpu64.orientation_assignment(ext, gauss, octaves)
    assert len(feats) == len(dbg.features)
    feats = gpu64.descriptor_representation(feats, gauss, octaves)
    assert np.array_equal(feats["descriptor"], dbg.features["descriptor"])
    srt = gpu64.sort_features(feats)
    assert np.array_equal(srt["descriptor"], dbg.sorted_features["descriptor"])


@pytest.mark.parametrize("w,h,seed", [(131, 77, 3), (257, 129, 4), (64, 64, 5), (33, 47, 6), (500, 36, 8), (23, 310, 9),
                                      (16, 16, 10), (640, 479, 11)])
def test_ragged_sizes_match_oracle(oracle, gpu64, w, h, seed):
    """Odd widths and heights (strips that cross the right border, rows that are not a multiple of the kernels'
    row groups, octaves that shrink to a few pixels, planes smaller than a blur radius)."""
    img = oracle.vnoise5(w, h, seed)
    dbg = oracle.Debug(img)
    got = gpu64.extract(img)
    assert gpu64.last_octaves() == dbg.octaves
    for o in range(dbg.octaves):
        for l in range(6):
            assert np.array_equal(gpu64.read_level(o, l), dbg.gauss[o * 6 + l]), "octave %d level %d" % (o, l)
    assert gpu64.read_counters() == dbg.counters
    ref = dbg.sorted_features
    assert len(got) == len(ref)
    for f in FIELDS_EXACT:
        assert np.array_equal(got[f], ref[f]), f
    assert np.array_equal(got["descriptor"], ref["descriptor"])


def test_context_reuse_across_sizes_and_batches(oracle):
    """One context, a sequence of different image sizes and batch sizes: every result equals the oracle's and a
    repeated input gives the same bytes again (no state carried from a larger geometry into a smaller one)."""
    import sift_amd
    sizes = [(320, 240, 7), (96, 64, 1), (640, 480, 3), (131, 77, 3), (320, 240, 7)]
    refs = {}
    with sift_amd.SiftGPU(max_width=640, max_height=480, max_batch=3) as ctx:
        first = None
        for w, h, seed in sizes:
            img = oracle.vnoise5(w, h, seed)
            if (w, h, seed) not in refs:
                refs[(w, h, seed)] = oracle.sift(img)[0]
            ref = refs[(w, h, seed)]
            got = ctx.extract(img)
            assert len(got) == len(ref), (w, h)
            assert np.array_equal(got["descriptor"], ref["descriptor"]), (w, h)
            assert np.array_equal(got["x"], ref["x"]) and np.array_equal(got["y"], ref["y"])
            if (w, h, seed) == sizes[0]:
                if first is None:
                    first = got
                else:
                    for f in got.dtype.names:
                        assert got[f].tobytes() == first[f].tobytes(), f
        # a batch of three different frames of one size after all that
        imgs = np.stack([oracle.vnoise5(200, 150, s) for s in (11, 12, 13)])
        outs = ctx.extract_batch(imgs, 4096)
        for img, out in zip(imgs, outs):
            ref = oracle.sift(img)[0]
            assert len(out) == len(ref)
            assert np.array_equal(out["desc"], ref["descriptor"].astype(np.uint8))


@pytest.mark.parametrize("sigma", [1.2, 2.0, 3.2, 8.0])
def test_other_base_sigmas(oracle, sigma):
    """Sift(src, features, sigma) with a sigma other than the default: narrower blurs pad into the unrolled kernels,
    wider ones take the any-radius kernels (and, at the largest, the wide-window orientation / descriptor variants)."""
    import sift_amd
    img = oracle.vnoise5(256, 192, 5)
    dbg = oracle.Debug(img, sigma=sigma)
    with sift_amd.SiftGPU(max_width=256, max_height=192, sigma=sigma) as ctx:
        got = ctx.extract(img)
        for o in range(dbg.octaves):
            for l in range(6):
                assert np.array_equal(ctx.read_level(o, l), dbg.gauss[o * 6 + l]), "octave %d level %d" % (o, l)
        assert ctx.read_counters() == dbg.counters
    ref = dbg.sorted_features
    assert len(ref) > 20 and len(got) == len(ref)
    for f in FIELDS_EXACT:
        assert np.array_equal(got[f], ref[f]), f
    assert np.array_equal(got["descriptor"], ref["descriptor"])


def test_sigma_not_above_one_is_rejected():
    import sift_amd
    with pytest.raises(sift_amd.SiftGpuError):
        sift_amd.SiftGPU(max_width=64, max_height=64, sigma=1.0)


@pytest.mark.parametrize("w,h", [(2, 2), (3, 5), (4, 4), (7, 9), (8, 8), (12, 10)])
def test_tiny_images(oracle, gpu64, w, h):
    """Images with zero, one or two octaves (the octave count may even be negative): same counters as the oracle,
    no features, no error."""
    img = oracle.vnoise5(w, h, 3)
    dbg = oracle.Debug(img)
    got = gpu64.extract(img)
    assert len(got) == len(dbg.features) == 0
    assert gpu64.last_octaves() == max(dbg.octaves, 0)
    assert gpu64.read_counters() == dbg.counters


def test_one_pixel_dimension_is_rejected(oracle, gpu64):
    """UpSample copies column 2C-3 and row 2R-3 (sift.cpp:163-173): with a 1-pixel dimension the reference reads
    out of bounds there, so such an image is an invalid argument here."""
    import sift_amd
    with pytest.raises(sift_amd.SiftGpuError):
        gpu64.extract(oracle.vnoise5(1, 1, 3))
    with pytest.raises(sift_amd.SiftGpuError):
        gpu64.extract(oracle.vnoise5(40, 1, 3))


def test_pair_blur_forced_on_small_and_ragged_planes():
    """Levels 1 and 2 from one launch (blur_pair_k) wherever the plane allows, short segments included: the pyramid
    must stay bit-equal to the oracle's, borders, odd widths and batches included.  Run in a process of its own because
    the switch is read once per process."""
    import subprocess
    import sys
    code = r'''
import numpy as np, sift_amd
from oracle import oracle as O
for (w, h, seed) in [(200, 150, 3), (333, 211, 4), (64, 48, 5), (117, 116, 6), (640, 480, 7)]:
    img = sift_amd.vnoise5(w, h, seed)
    dbg = O.Debug(img)
    with sift_amd.SiftGPU(max_width=w, max_height=h, max_batch=2) as c:
        got = c.extract(img)
        assert c.last_pair_octaves() & 1, (w, h)
        for o in range(dbg.octaves):
            for l in range(6):
                assert np.array_equal(c.read_level(o, l), dbg.gauss[o * 6 + l]), (w, h, o, l)
        assert c.read_counters() == dbg.counters
        ref, _, _, _ = O.sift(img)
        assert len(got) == len(ref) and np.array_equal(got["descriptor"], ref["descriptor"])
        two = c.extract_batch(np.stack([img, img[::-1].copy()]), 8192)
        assert np.array_equal(two[0]["desc"], ref["descriptor"].astype(np.uint8))
print("pair ok")
'''
    import os
    env = dict(os.environ, SIFTGPU_BLUR_PAIR="2")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode == 0 and "pair ok" in r.stdout, r.stderr[-3000:] + r.stdout[-500:]


def test_pyramid_tail_launch_and_level_by_level_agree(oracle, gpu64):
    """The smallest octaves come from one launch that keeps their planes in LDS (pyr_tail_k).  In this process the
    launch is on (default): it must have been taken, second image of a batch included; in a process with
    SIFTGPU_PYR_TAIL=0 the same octaves go level by level through the row / column kernels.  Both must give the
    oracle's pyramid bit for bit, planes narrower than the widest kernel included."""
    import subprocess
    import sys
    import sift_amd
    img = oracle.vnoise5(640, 480, 21)
    with sift_amd.SiftGPU(max_width=640, max_height=480, max_batch=2) as c:
        c.extract(img)
        assert 1 <= c.last_tail_octave() < c.last_octaves()
        dbg = oracle.Debug(img)
        for o in range(c.last_tail_octave(), dbg.octaves):
            for l in range(6):
                assert np.array_equal(c.read_level(o, l), dbg.gauss[o * 6 + l]), (o, l)
        flipped = np.ascontiguousarray(img[::-1])
        c.extract_batch(np.stack([img, flipped]), 8192)
        dbg2 = oracle.Debug(flipped)
        for o in range(dbg2.octaves):
            for l in range(6):
                assert np.array_equal(c.read_level(o, l, image=1), dbg2.gauss[o * 6 + l]), (o, l)
    code = r'''
import numpy as np, sift_amd
from oracle import oracle as O
for (w, h, seed) in [(200, 150, 3), (333, 211, 4), (64, 48, 5), (640, 480, 7)]:
    img = sift_amd.vnoise5(w, h, seed)
    dbg = O.Debug(img)
    with sift_amd.SiftGPU(max_width=w, max_height=h) as c:
        got = c.extract(img)
        assert c.last_tail_octave() == c.last_octaves(), (w, h)
        for o in range(dbg.octaves):
            for l in range(6):
                assert np.array_equal(c.read_level(o, l), dbg.gauss[o * 6 + l]), (w, h, o, l)
        assert c.read_counters() == dbg.counters
print("no tail ok")
'''
    import os
    env = dict(os.environ, SIFTGPU_PYR_TAIL="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode == 0 and "no tail ok" in r.stdout, r.stderr[-3000:] + r.stdout[-500:]


def test_fma_precision_mode_stays_within_tolerance(oracle):
    """SG_F64_FMA (fused multiply-adds in the Gaussian filters; reference sift/sift.cpp:323-326 rounds the product and
    the sum separately): a tolerance mode.  On value-noise frames the pyramid moves by ~1e-16 relative and every
    discrete result stays what the oracle gives: counters, keypoints, all descriptor bins.  Tolerances written out:
    levels within 1e-14 absolute (pixel range 0..1), orientation within 1e-9 rad, scale within 1e-12 relative."""
    import sift_amd
    img = sift_amd.vnoise5(320, 240, 77)
    ref = oracle.Debug(img)
    with sift_amd.SiftGPU(max_width=320, max_height=240, precision=sift_amd.SG_F64_FMA) as ctx, \
            sift_amd.SiftGPU(max_width=320, max_height=240) as exact:
        got = ctx.extract(img)
        assert ctx.read_counters() == ref.counters
        differs = 0
        for o in range(ctx.last_octaves()):
            for l in range(6):
                lv = ctx.read_level(o, l)
                assert np.abs(lv - ref.gauss[o * 6 + l]).max() < 1e-14
                differs += int((lv != ref.gauss[o * 6 + l]).sum())
        assert differs > 0  # the mode really takes the fused form (the default mode is bit-identical)
        want = ref.sorted_features
        assert len(got) == len(want)
        for f in ("octave", "interval", "x", "y"):
            assert np.array_equal(got[f], want[f]), f
        assert np.array_equal(got["descriptor"], want["descriptor"])
        d = np.abs(got["ori"] - want["ori"])
        assert np.minimum(d, 2 * np.pi - d).max() < 1e-9
        assert np.allclose(got["scale"], want["scale"], rtol=1e-12, atol=0)
        # the default mode beside it: bit-identical levels
        exact.extract(img)
        assert np.array_equal(exact.read_level(0, 5), ref.gauss[5])


def test_descriptor_entry_point_checks_the_histogram_range(oracle, gpu64):
    """DescriptorRepresentation on planes handed in from outside: the fixed-point histograms hold sums below 2^14, so
    planes whose range times window area could exceed that, and planes with values that are not finite, are refused
    (the path's own pyramids, pixels 0..1, are far inside); a large-scale feature on ordinary planes is taken."""
    import sift_amd
    from sift_amd import api
    img = oracle.vnoise5(192, 144, 9)
    dbg = oracle.Debug(img)
    feats = dbg.features[:40].copy()
    big = feats.copy()
    big["octave_scale"] = 12.0  # radius 127: the widest window of the common kernel variant; pixels 0..1 are fine
    out = gpu64.descriptor_representation(big, dbg.gauss, dbg.octaves)
    assert out["descr_length"].min() == 128 and out["descriptor"].max() <= 255
    scaled = [g * 1000.0 for g in dbg.gauss]  # range 1000 x radius^2 16129: beyond the histograms
    with pytest.raises(sift_amd.SiftGpuError) as e:
        gpu64.descriptor_representation(big, scaled, dbg.octaves)
    assert e.value.status == api.SG_E_UNSUPPORTED
    # the same planes with the features' own (small) windows are inside the range: taken, bins as for the plain planes
    # (the descriptor is normalised, so a common factor of the pixels cancels)
    small = gpu64.descriptor_representation(feats, [g * 16.0 for g in dbg.gauss], dbg.octaves)
    assert np.array_equal(small["descriptor"], dbg.features[:40]["descriptor"])
    bad = [g.copy() for g in dbg.gauss]
    bad[1][3, 3] = np.nan
    with pytest.raises(sift_amd.SiftGpuError) as e:
        gpu64.descriptor_representation(feats, bad, dbg.octaves)
    assert e.value.status == api.SG_E_INVALID
