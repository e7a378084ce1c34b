"""A 4K frame, a 12-megapixel frame through the in-layer detector and frames of extreme aspect ratio against the CPU
oracle, field by field (run by hand on a GPU box: python profiles/large_frames_check.py; the oracle needs half a minute
for the largest).  The GPU suite covers 1080p, 8K by digest and sizes up to 420 pixels by fuzz; this fills the gap."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import sift_amd
from oracle import oracle as O
def check(w, h, seed, ni=False):
    img = O.vnoise5(w, h, seed)
    t0 = time.time()
    ref, n_kp, ctr, _ = O.sift(img, nointerval=ni)
    t1 = time.time()
    with sift_amd.SiftGPU(max_width=w, max_height=h, nointerval=ni, max_features=max(w * h // 20, 8192)) as c:
        got = c.extract(img, cap=max(w * h // 10, 8192))
        ok = tuple(c.read_counters()) == tuple(ctr) and len(got) == len(ref)
        for f in ("octave", "interval", "x", "y", "dx", "dy", "offset_x", "offset_y", "offset_interval", "val"):
            ok = ok and np.array_equal(got[f], ref[f])
        ok = ok and np.array_equal(got["descriptor"], ref["descriptor"])
        d = np.abs(got["ori"] - ref["ori"]); d = np.minimum(d, 2 * np.pi - d)
        ok = ok and (len(ref) == 0 or d.max() < 1e-9) and np.allclose(got["scale"], ref["scale"], rtol=1e-13, atol=0)
    print("%dx%d seed %d nointerval=%d: %d features, oracle %.1f s, equal %s" % (w, h, seed, ni, len(ref), t1 - t0, ok), flush=True)
    return ok
ok = True
for (w, h, s, ni) in [(3840, 2160, 99, False), (4096, 3000, 5, True), (5000, 64, 3, False), (64, 5000, 4, False), (3001, 17, 8, False), (2049, 1025, 6, False)]:
    ok = check(w, h, s, ni) and ok
sys.exit(0 if ok else 1)
