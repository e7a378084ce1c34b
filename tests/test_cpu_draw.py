"""Drawing and pyramid-dump helpers of the host C++ layer (reference sift/sift.cpp:1772-1884): host code, no GPU.
sift_amd/host/sift_draw_check draws one feature arrow and two key-point circles on a 64x48 canvas and dumps a
two-level pyramid; the files are checked here against the formulas of the reference."""
import math
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "sift_amd", "host", "sift_draw_check")


def read_pnm(path):
    data = open(path, "rb").read()
    fields, pos = [], 0
    while len(fields) < 4:
        end = pos
        while data[end:end + 1] not in (b" ", b"\n"):
            end += 1
        fields.append(data[pos:end])
        pos = end + 1
    magic, w, h, _ = fields[0], int(fields[1]), int(fields[2]), int(fields[3])
    ch = 3 if magic == b"P6" else 1
    a = np.frombuffer(data[pos:], dtype=np.uint8)
    assert a.size == w * h * ch
    return a.reshape(h, w, ch) if ch == 3 else a.reshape(h, w)


@pytest.fixture(scope="module")
def outputs(tmp_path_factory):
    if not os.path.exists(EXE):
        pytest.skip("sift_draw_check has not been built")
    d = tmp_path_factory.mktemp("draw")
    r = subprocess.run([EXE, str(d)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    return d


def seg_dist(px, py, a, b):
    ax, ay, bx, by = a[0], a[1], b[0], b[1]
    vx, vy = bx - ax, by - ay
    t = 0.0 if vx == 0 and vy == 0 else max(0.0, min(1.0, ((px - ax) * vx + (py - ay) * vy) / (vx * vx + vy * vy)))
    return math.hypot(px - (ax + t * vx), py - (ay + t * vy))


def test_feature_arrow(outputs):
    """DrawSiftFeature: shaft of cvRound(5 scale) pixels from (cvRound(dx), cvRound(dy)) along (cos ori, -sin ori), two
    head strokes of length len - cvRound(0.75 scale) at +-10 degrees, in green."""
    img = read_pnm(os.path.join(outputs, "canvas.ppm"))
    green = np.argwhere((img[:, :, 0] == 0) & (img[:, :, 1] == 255) & (img[:, :, 2] == 0))
    dx, dy, scale, ori = 30.4, 24.6, 3.0, 0.5
    sx, sy = round(dx), round(dy)
    ln, hl = round(scale * 5.0), round(scale * 0.75)
    bl = ln - hl
    end = (round(ln * math.cos(ori)) + sx, round(ln * -math.sin(ori)) + sy)
    h1 = (round(bl * math.cos(ori + math.pi / 18)) + sx, round(bl * -math.sin(ori + math.pi / 18)) + sy)
    h2 = (round(bl * math.cos(ori - math.pi / 18)) + sx, round(bl * -math.sin(ori - math.pi / 18)) + sy)
    pts = {(int(x), int(y)) for y, x in green}
    for p in ((sx, sy), end, h1, h2):
        assert p in pts
    for x, y in pts:  # every green pixel lies on one of the three strokes
        assert min(seg_dist(x, y, (sx, sy), end), seg_dist(x, y, end, h1), seg_dist(x, y, end, h2)) <= 0.75
    assert len(pts) >= ln + 1  # an 8-connected shaft has max(|dx|, |dy|) + 1 pixels


def test_keypoint_circles(outputs):
    """DrawKeyPoints / DrawKeyPointsRight: radius-3 circles in blue (CV_RGB(0, 0, 255)), the second one shifted by the left image's width
    and clipped at the border."""
    img = read_pnm(os.path.join(outputs, "canvas.ppm"))
    blue = {(int(x), int(y)) for y, x in np.argwhere((img[:, :, 0] == 0) & (img[:, :, 1] == 0) & (img[:, :, 2] == 255))}
    ring = {(x, y) for x in range(-3, 4) for y in range(-3, 4) if abs(math.hypot(x, y) - 3) < 0.6}
    first = {(x - 12, y - 11) for x, y in blue if x < 32}
    assert first == ring and len(ring) == 16
    second = {(x - 60, y - 40) for x, y in blue if x >= 32}
    assert second == {(x, y) for x, y in ring if x + 60 < 64}
    # nothing else was touched
    other = (img != 10).any(axis=2).sum()
    green = ((img[:, :, 1] == 255) & (img[:, :, 0] == 0)).sum()
    assert other == len(blue) + green


def test_pyramid_dump(outputs):
    """write_pyr -> writecv64f -> cv64f_to_cv8U: min-max normalisation to 0..255, rounded; a constant level gives 0."""
    a = read_pnm(os.path.join(outputs, "lvl0.pgm"))
    src = 0.01 * np.arange(48, dtype=np.float64).reshape(6, 8) - 0.1
    want = np.rint((src - src.min()) * (255.0 / (src.max() - src.min()))).astype(np.uint8)
    assert a.shape == (6, 8) and np.array_equal(a, want)
    assert a.min() == 0 and a.max() == 255
    b = read_pnm(os.path.join(outputs, "lvl1.pgm"))
    assert b.shape == (3, 4) and (b == 0).all()


def test_legacy_helpers_link_and_run():
    """The reference's off-path helpers (8-bit legacy blurs, GetFileName, display_pyr, testInverse3D; sift.h:121-123,
    149, 153, 159) exist in the host library with the reference's signatures; host code only, no GPU needed."""
    import subprocess
    exe = os.path.join(ROOT, "sift_amd", "host", "sift_host_check")
    if not os.path.exists(exe):
        pytest.fail("%s is not built (run __graft_entry__.build())" % exe)
    out = subprocess.run([exe, "legacy"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr + out.stdout
    assert "legacy ok" in out.stdout and "name: dog 1" in out.stdout
    # H * inverse(H) printed by testInverse3D is the identity to rounding
    rows = [l.split() for l in out.stdout.splitlines() if len(l.split()) == 3 and "ok" not in l and "name" not in l]
    m = np.array(rows[-3:], dtype=float)
    assert np.allclose(m, np.eye(3), atol=1e-9)
