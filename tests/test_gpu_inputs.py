"""Inputs beyond smooth value noise, whole path against the oracle: flat regions (exact ties in the non-strict
extremum test, reference sift/sift.cpp:479-523; Hessians with |det| < 1e-10, sift.cpp:738-739), straight edges
(Det <= 0 and the Tr^2/Det test, sift.cpp:616-626), saturated blocks, white noise, a 3-pixel checkerboard (features far denser than
the default buffer sizing), 4-channel and padded inputs; and a sweep of 1080p seeds with recall and bin equality
reported as numbers."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from sift_amd.parity import descriptor_agreement, match_features

pytestmark = pytest.mark.gpu


def _cases():
    from sift_amd import synth
    return [
        ("checkerboard16", synth.checkerboard(256, 192, 16)),
        ("checkerboard3", synth.checkerboard(160, 120, 3)),
        ("checkerboard7_gray", synth.checkerboard(200, 150, 7, 40, 215)),
        ("step_edges", synth.step_edges(240, 180, 5)),
        ("step_edges_big", synth.step_edges(400, 300, 6, n=80)),
        ("blobs", synth.blobs(256, 192, 9)),
        ("blobs_odd", synth.blobs(301, 203, 10, n=90)),
        ("saturated_blocks", synth.saturated_blocks(320, 240, 11)),
        ("white_noise", synth.white_noise(160, 120, 12)),
        ("flat", np.full((96, 128, 3), 77, dtype=np.uint8)),
    ]


@pytest.fixture(scope="module")
def roomy():
    """A context with room for the densest case."""
    import sift_amd
    c = sift_amd.SiftGPU(max_width=400, max_height=300, max_features=40000, keep_prequant=True)
    yield c
    c.close()


@pytest.mark.parametrize("name,img", _cases(), ids=[c[0] for c in _cases()])
def test_stress_inputs_match_oracle(oracle, roomy, name, img):
    dbg = oracle.Debug(img, prequant=True)
    got = roomy.extract(img, cap=60000)
    assert roomy.read_counters() == dbg.counters, name
    kp = roomy.read_keypoints()
    ref_kp = dbg.keypoints
    assert len(kp) == len(ref_kp)
    for f in ("octave", "interval", "x", "y", "offset_x", "offset_y", "offset_interval", "dx", "dy", "val"):
        assert np.array_equal(kp[f], ref_kp[f]), (name, f)  # refinement is bit-exact, singular Hessians included
    feats, pq = roomy.read_features_unsorted(prequant=True)
    ref = dbg.features
    assert len(feats) == len(ref), name
    if len(ref):
        for f in ("octave", "interval", "x", "y"):
            assert np.array_equal(feats[f], ref[f]), (name, f)
        # orientations agree as angles: a histogram that is mirror-symmetric to the last bit (checkerboards) puts a
        # peak's parabola offset at +-1e-17 around bin 0, i.e. at -pi or just below +pi
        d = np.abs(feats["ori"] - ref["ori"])
        assert np.minimum(d, 2 * np.pi - d).max() < 1e-9, name
        assert np.array_equal(feats["descriptor"], ref["descriptor"]), name
        assert np.abs(pq - dbg.prequant).max() < 1e-9, name
    want, _, _, _ = oracle.sift(img)
    assert len(got) == len(want)
    if len(want):
        assert np.array_equal(got["descriptor"], want["descriptor"]) and np.array_equal(got["x"], want["x"])
    if name.startswith("checker") or name.startswith("step"):
        assert dbg.counters[2] > 0  # the case does reach the detector


def test_dense_image_reports_needed_capacity(oracle):
    """A checkerboard of 3-pixel squares overflows the default sizing (width*height/40 + 4096 features): the call fails with SG_E_CAPACITY,
    sg_required_features names a size that works, and the result then equals the oracle's."""
    import sift_amd
    img = sift_amd.synth.checkerboard(320, 240, 3)
    want, _, ctr, _ = oracle.sift(img)
    assert len(want) > 320 * 240 // 40 + 4096
    with sift_amd.SiftGPU(max_width=320, max_height=240) as c:
        with pytest.raises(sift_amd.SiftGpuError) as e:
            c.extract(img, cap=100000)
        assert e.value.status == sift_amd.api.SG_E_CAPACITY
        need = c.required_features()
        assert need > 320 * 240 // 40 + 4096
        # the report is sticky until read and then cleared: a frame that fits runs normally afterwards
        small = c.extract(sift_amd.vnoise5(320, 240, 4))
        assert len(small) > 100
    for _ in range(3):  # a later stage may overflow once the first one fits
        with sift_amd.SiftGPU(max_width=320, max_height=240, max_features=need) as c:
            try:
                got = c.extract(img, cap=200000)
            except sift_amd.SiftGpuError as e2:
                assert e2.status == sift_amd.api.SG_E_CAPACITY
                need = max(need + 1, c.required_features())
                continue
            assert c.read_counters() == ctr
            assert len(got) == len(want) and np.array_equal(got["descriptor"], want["descriptor"])
            break
    else:
        pytest.fail("no capacity reported by sg_required_features was enough")


def test_second_batch_does_not_erase_first_overflow():
    """Two batches enqueued on one context before a sync: the first one's overflow is still reported."""
    import sift_amd
    W, H, cap = 320, 240, 40000
    with sift_amd.SiftGPU(max_width=W, max_height=H) as c:
        d_a, d_b = c.malloc(W * H * 3), c.malloc(W * H * 3)
        d_out, d_cnt = c.malloc(cap * 144), c.malloc(4)
        try:
            c.h2d(d_a, sift_amd.synth.checkerboard(W, H, 3))
            c.synth_vnoise5(d_b, W, H, 5)
            c.extract_batch_device(d_a, 1, W, H, 3, d_out, cap, d_cnt)  # overflows
            c.extract_batch_device(d_b, 1, W, H, 3, d_out, cap, d_cnt)  # fits
            with pytest.raises(sift_amd.SiftGpuError) as e:
                c.sync()
            assert e.value.status == sift_amd.api.SG_E_CAPACITY
            c.extract_batch_device(d_b, 1, W, H, 3, d_out, cap, d_cnt)
            c.sync()  # clean again
        finally:
            for p in (d_a, d_b, d_out, d_cnt):
                c.free(p)


def test_records_outside_the_pyramid_are_rejected(oracle):
    """Stage entry points take caller-built records: one that would index outside the pyramid is refused, not run."""
    import sift_amd
    img = sift_amd.vnoise5(160, 120, 21)
    dbg = oracle.Debug(img)
    with sift_amd.SiftGPU(max_width=160, max_height=120) as c:
        ok = c.orientation_assignment(dbg.keypoints, dbg.gauss, dbg.octaves)
        assert len(ok) == len(dbg.features)
        for field, value in (("octave", dbg.octaves), ("octave", -1), ("interval", 6), ("x", 100000), ("y", -3),
                             ("octave_scale", float("nan")), ("octave_scale", 0.0), ("octave_scale", 1e9)):
            bad = dbg.keypoints.copy()
            bad[field][len(bad) // 2] = value
            with pytest.raises(sift_amd.SiftGpuError) as e:
                c.orientation_assignment(bad, dbg.gauss, dbg.octaves)
            assert e.value.status == sift_amd.api.SG_E_INVALID, field
        feats = dbg.features.copy()
        feats["ori"][0] = float("inf")
        with pytest.raises(sift_amd.SiftGpuError):
            c.descriptor_representation(feats, dbg.gauss, dbg.octaves)


def test_1080p_seed_sweep_against_oracle(oracle):
    """Eight more 1080p seeds, the oracle run on host threads: recall within the north-star tolerances (0.5 px, 1 %
    scale, 1 degree) and descriptor bin equality, as numbers."""
    import sift_amd
    seeds = [2, 3, 5, 8, 13, 21, 34, 55]
    imgs = [sift_amd.vnoise5(1920, 1080, s) for s in seeds]
    with ThreadPoolExecutor(max_workers=8) as pool:
        refs = list(pool.map(lambda im: oracle.sift(im), imgs))
    total = matched = same_desc = 0
    bins_total = bins_same = 0
    with sift_amd.SiftGPU(max_width=1920, max_height=1080) as c:
        for s, img, (ref, nk, ctr, _) in zip(seeds, imgs, refs):
            got = c.extract(img)
            assert c.read_counters() == ctr, s
            pairs = match_features(ref, got)
            total += len(ref)
            matched += len(pairs)
            a, b = ref["descriptor"][pairs[:, 0]], got["descriptor"][pairs[:, 1]]
            same_desc += int((a == b).all(axis=1).sum())
            bins_total += a.size
            bins_same += int((a == b).sum())
            assert len(got) == len(ref), s
    recall, desc_eq, bin_eq = matched / total, same_desc / total, bins_same / bins_total
    print("1080p sweep: %d features, recall %.6f, descriptors identical %.6f, bins identical %.8f"
          % (total, recall, desc_eq, bin_eq))
    assert recall >= 0.9999 and desc_eq >= 0.9999 and bin_eq >= 0.999999


def test_fetch_after_the_buffers_were_reused_is_refused():
    """sg_extract_begin leaves its records for sg_extract_fetch in the context's working buffers; a batch call (or a
    stage entry point) in between overwrites them, and the fetch must then fail instead of packing the new contents
    with the old count."""
    import ctypes as C
    import sift_amd
    from sift_amd import api
    img = sift_amd.vnoise5(160, 120, 31)
    other = sift_amd.vnoise5(160, 120, 32)
    with sift_amd.SiftGPU(max_width=160, max_height=120, max_batch=2) as ctx:
        L = ctx.lib
        n = C.c_size_t(0)
        rc = L.sg_extract_begin(ctx.ctx, img.ctypes.data_as(C.POINTER(C.c_uint8)), 160, 120, 3, 160 * 3, C.byref(n))
        assert rc == api.SG_OK and n.value > 50
        recs = np.zeros(n.value, dtype=sift_amd.KEYPOINT_DTYPE)
        got = C.c_size_t(0)
        assert L.sg_extract_fetch(ctx.ctx, recs.ctypes.data, n.value, C.byref(got)) == api.SG_OK  # may be fetched twice
        assert L.sg_extract_fetch(ctx.ctx, recs.ctypes.data, n.value, C.byref(got)) == api.SG_OK
        ctx.extract_batch(np.stack([other, other]), 4096)  # reuses the working buffers
        assert L.sg_extract_fetch(ctx.ctx, recs.ctypes.data, n.value, C.byref(got)) == api.SG_E_INVALID
        # and the same for the batch form's fetch after a single-image call
        ptrs = (C.POINTER(C.c_uint8) * 2)(img.ctypes.data_as(C.POINTER(C.c_uint8)), other.ctypes.data_as(C.POINTER(C.c_uint8)))
        counts = (C.c_uint32 * 2)()
        assert L.sg_extract_batch_begin(ctx.ctx, ptrs, 2, 160, 120, 3, 160 * 3, 0, counts) == api.SG_OK
        ctx.extract(img)
        wire = np.zeros(counts[0] + counts[1], dtype=sift_amd.WIRE_DTYPE)
        assert L.sg_extract_batch_fetch_wire(ctx.ctx, wire.ctypes.data, len(wire)) == api.SG_E_INVALID
