"""Committed golden vectors (tests/golden/, written by make_golden.py): the oracle on CPU, the HIP path on GPU."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
INDEX = json.load(open(os.path.join(GOLD, "index.json")))


@pytest.mark.parametrize("case", INDEX, ids=[c["name"] for c in INDEX])
def test_oracle_reproduces_golden(oracle, case):
    g = np.load(os.path.join(GOLD, case["name"] + ".npz"))
    dbg = oracle.Debug(oracle.vnoise5(case["width"], case["height"], case["seed"]))
    assert list(dbg.counters) == case["counters"] == g["counters"].tolist()
    f = dbg.sorted_features
    assert len(f) == case["features"] and len(dbg.keypoints) == case["keypoints"]
    assert np.array_equal(np.stack([f[k] for k in ("x", "y", "octave", "interval")], axis=1), g["feat_xyoi"])
    assert np.array_equal(f["descriptor"].astype(np.uint8), g["feat_desc"])
    assert np.allclose(np.stack([f[k] for k in ("dx", "dy", "scale", "ori")], axis=1), g["feat_f64"], rtol=1e-13, atol=1e-13)


def test_oracle_blur_small_planes(oracle):
    g = np.load(os.path.join(GOLD, "blur_small_planes.npz"))
    for key in g.files:
        if key.startswith("src_"):
            out = oracle.gaussian_blur(g[key], 3.0900155872895909)
            assert np.array_equal(out, g["blur27_" + key[4:]])


@pytest.mark.gpu
@pytest.mark.parametrize("case", INDEX, ids=[c["name"] for c in INDEX])
def test_gpu_reproduces_golden(gpu64, case):
    import sift_amd
    g = np.load(os.path.join(GOLD, case["name"] + ".npz"))
    img = sift_amd.vnoise5(case["width"], case["height"], case["seed"])
    f = gpu64.extract(img)
    assert list(gpu64.read_counters()) == case["counters"]
    kp = gpu64.read_keypoints()
    assert np.array_equal(np.stack([kp[k] for k in ("x", "y", "octave", "interval")], axis=1), g["kp_xyoi"])
    exact = np.stack([kp[k] for k in ("offset_x", "offset_y", "offset_interval", "dx", "dy")], axis=1)
    assert np.array_equal(exact, g["kp_f64"][:, :5])
    assert np.array_equal(kp["val"], g["kp_f64"][:, 7])
    assert np.allclose(kp["scale"], g["kp_f64"][:, 5], rtol=1e-14) and np.allclose(kp["octave_scale"], g["kp_f64"][:, 6], rtol=1e-14)
    assert len(f) == case["features"]
    assert np.array_equal(np.stack([f[k] for k in ("x", "y", "octave", "interval")], axis=1), g["feat_xyoi"])
    assert np.array_equal(f["descriptor"].astype(np.uint8), g["feat_desc"])
    got = np.stack([f[k] for k in ("dx", "dy", "scale", "ori")], axis=1)
    assert np.allclose(got, g["feat_f64"], rtol=1e-12, atol=1e-9)
    for i in range(gpu64.last_octaves() * 6):
        lv = gpu64.read_level(i // 6, i % 6)
        assert float(lv.sum()) == g["level_checksums"][i, 0] and float(np.abs(lv).max()) == g["level_checksums"][i, 1]
    if g["base_level"].size:
        assert np.array_equal(gpu64.read_level(0, 0), g["base_level"])


@pytest.mark.gpu
def test_gpu_blur_small_planes(gpu64):
    g = np.load(os.path.join(GOLD, "blur_small_planes.npz"))
    for key in g.files:
        if key.startswith("src_"):
            assert np.array_equal(gpu64.gaussian_smooth(g[key], 3.0900155872895909), g["blur27_" + key[4:]])


def _match_fixture():
    return np.load(os.path.join(GOLD, "match_200x150_s77.npz"))


def _records(desc):
    from oracle.oracle import KEYPOINT_DTYPE
    f = np.zeros(len(desc), dtype=KEYPOINT_DTYPE)
    f["descriptor"] = desc
    f["descr_length"] = 128
    return f


def test_oracle_match_reproduces_golden(oracle):
    g = _match_fixture()
    pairs, best2 = oracle.match(_records(g["desc_a"]), _records(g["desc_b"]), 0.5)
    assert np.array_equal(pairs, g["pairs_05"]) and np.array_equal(best2.astype(np.uint32), g["best2"])
    assert np.array_equal(oracle.match(_records(g["desc_a"]), _records(g["desc_b"]), 0.8)[0], g["pairs_08"])


@pytest.mark.gpu
def test_gpu_match_reproduces_golden(gpu64):
    g = _match_fixture()
    a, b = _records(g["desc_a"]), _records(g["desc_b"])
    assert len(g["pairs_05"]) > 50
    assert np.array_equal(gpu64.match(a, b, 0.5), g["pairs_05"])
    assert np.array_equal(gpu64.match(a, b, 0.8), g["pairs_08"])
