"""Descriptor matching (the step after the path in the reference's demo, sift/源.cpp:88-156) against the oracle."""
import numpy as np
import pytest

import sift_amd

pytestmark = pytest.mark.gpu


def _two_views(oracle, w, h, seed, shift=(7, 5)):
    """Two overlapping crops of one synthetic scene, so that true matches exist."""
    big = oracle.vnoise5(w + shift[0], h + shift[1], seed)
    a = np.ascontiguousarray(big[:h, :w])
    b = np.ascontiguousarray(big[shift[1]:shift[1] + h, shift[0]:shift[0] + w])
    return a, b


def _cat(*arrs):
    """Concatenate Keypoint record arrays keeping the record layout (np.concatenate repacks the fields)."""
    out = np.zeros(sum(len(a) for a in arrs), dtype=arrs[0].dtype)
    pos = 0
    for a in arrs:
        out[pos:pos + len(a)] = a
        pos += len(a)
    return out


@pytest.fixture(scope="module")
def views(oracle, gpu64):
    a, b = _two_views(oracle, 320, 240, 3)
    return gpu64.extract(a), gpu64.extract(b)


def test_match_equals_oracle(oracle, gpu64, views):
    f1, f2 = views
    ref, best2 = oracle.match(f1, f2, 0.5)
    got = gpu64.match(f1, f2, 0.5)
    assert len(ref) > 50
    assert np.array_equal(got, ref)
    # a looser ratio exercises many more near-ties of the test itself
    for ratio in (0.8, 0.95, 1.0):
        ref_r, _ = oracle.match(f1, f2, ratio)
        assert np.array_equal(gpu64.match(f1, f2, ratio), ref_r), ratio


def test_matches_are_the_true_correspondences(oracle, gpu64, views):
    f1, f2 = views
    m = gpu64.match(f1, f2, 0.5)
    dx = f1["dx"][m[:, 0]] - f2["dx"][m[:, 1]]
    dy = f1["dy"][m[:, 0]] - f2["dy"][m[:, 1]]
    ok = (np.abs(dx - 7) < 1.0) & (np.abs(dy - 5) < 1.0)
    assert ok.mean() > 0.95


def test_squared_distances_exact(oracle, gpu64, views):
    """Packed device form: smallest / second smallest squared distance are exact integers."""
    f1, f2 = views
    _, best2 = oracle.match(f1, f2, 0.5)

    def pack(f):
        p = np.zeros(len(f), dtype=sift_amd.PACKED_DTYPE)
        p["x"], p["y"], p["scale"], p["ori"] = f["dx"], f["dy"], f["scale"], f["ori"]
        p["desc"] = f["descriptor"].astype(np.uint8)
        return p

    p1, p2 = pack(f1), pack(f2)
    d1, d2 = gpu64.malloc(p1.nbytes), gpu64.malloc(p2.nbytes)
    di, dd = gpu64.malloc(4 * len(p1)), gpu64.malloc(8 * len(p1))
    try:
        gpu64.h2d(d1, p1)
        gpu64.h2d(d2, p2)
        gpu64.match_packed_device(d1, len(p1), d2, len(p2), di, dd, 0.5)
        gpu64.sync()
        idx = np.empty(len(p1), dtype=np.int32)
        dist = np.empty((len(p1), 2), dtype=np.uint32)
        gpu64.d2h(idx, di)
        gpu64.d2h(dist, dd)
    finally:
        for p in (d1, d2, di, dd):
            gpu64.free(p)
    assert np.array_equal(dist.astype(np.float64), best2)
    ref, _ = oracle.match(f1, f2, 0.5)
    assert np.array_equal(np.flatnonzero(idx >= 0), ref[:, 0])
    assert np.array_equal(idx[idx >= 0], ref[:, 1])


@pytest.mark.parametrize("n1,n2", [(0, 5), (5, 0), (1, 1), (3, 1), (1, 2), (33, 31), (64, 65), (100, 7)])
def test_edge_sizes(oracle, gpu64, views, n1, n2):
    f1, f2 = views
    a, b = f1[:n1].copy(), f2[:n2].copy()
    ref, _ = oracle.match(a, b, 0.9)
    assert np.array_equal(gpu64.match(a, b, 0.9), ref)


def test_ties_and_duplicates(oracle, gpu64, views):
    """Identical descriptors: first index wins, a zero second distance gives 0/0 and no match."""
    f1, f2 = views
    a = f1[:40].copy()
    b = _cat(f1[:40], f1[:40], f2[:50])  # every a[i] has two exact copies in b
    ref, _ = oracle.match(a, b, 0.9)
    assert np.array_equal(gpu64.match(a, b, 0.9), ref)
    b2 = _cat(f2[:37], f1[:40])           # one exact copy each, at a ragged offset
    ref2, _ = oracle.match(a, b2, 0.5)
    got2 = gpu64.match(a, b2, 0.5)
    # (integer shifts make true correspondences bit-identical, so some a[i] also have a twin in f2[:37]: 0/0, no match)
    assert np.array_equal(got2, ref2) and len(ref2) > 0 and np.array_equal(ref2[:, 1], ref2[:, 0] + 37)


def test_non_integer_descriptors_are_rejected(gpu64, views):
    f1, f2 = views
    bad = f2[:10].copy()
    bad["descriptor"][3, 7] = 0.5
    with pytest.raises(sift_amd.SiftGpuError):
        gpu64.match(f1[:10], bad)
    # the context stays usable
    assert len(gpu64.match(f1[:10], f1[:10], 0.5)) == 10
