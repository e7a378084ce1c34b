#!/usr/bin/env python3
"""Writes tests/golden/sift_7680x4320_s1234.npz: a digest of the CPU oracle's full output on the 8K configuration
(7680x4320 vnoise5 seed 1234, 11 octaves, 360 706 features).  The oracle needs about two and a half minutes and
17 GB for this image, too much for a test, so the GPU test compares against this digest instead:

  counters, feature count
  sha256 of the exactly reproducible fields in final order: (x, y, octave, interval) as int32, (dx, dy) as
  float64 bit patterns, the 128 descriptor bins as bytes
  every 8th orientation and scale as float64 (orientations agree to ~1e-13 rad and scales, which come out of pow(),
  to an ulp or two: not bit for bit)

    python tests/golden/make_golden_8k.py
"""
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402

W, H, SEED = 7680, 4320, 1234


def digest(f):
    """f: records with the fields of the reference's Keypoint, in final (ascending scale) order."""
    ints = np.stack([f[k] for k in ("x", "y", "octave", "interval")], axis=1).astype("<i4")
    f64 = np.stack([f[k] for k in ("dx", "dy")], axis=1).astype("<f8")
    desc = np.ascontiguousarray(f["descriptor"]).astype(np.uint8)
    return {
        "sha_ints": hashlib.sha256(np.ascontiguousarray(ints).tobytes()).hexdigest(),
        "sha_f64": hashlib.sha256(np.ascontiguousarray(f64).tobytes()).hexdigest(),
        "sha_desc": hashlib.sha256(desc.tobytes()).hexdigest(),
    }


def main():
    t0 = time.time()
    img = O.vnoise5(W, H, SEED)
    feats, n_keypoints, counters, _ = O.sift(img)
    d = digest(feats)
    np.savez_compressed(os.path.join(HERE, "sift_%dx%d_s%d.npz" % (W, H, SEED)),
                        counters=np.array(counters, dtype=np.int64), keypoints=np.int64(n_keypoints),
                        features=np.int64(len(feats)), ori_every_8th=np.asarray(feats["ori"][::8], dtype=np.float64),
                        scale_every_8th=np.asarray(feats["scale"][::8], dtype=np.float64),
                        sha_ints=d["sha_ints"], sha_f64=d["sha_f64"], sha_desc=d["sha_desc"])
    print("wrote 8K digest: %d features, %d keypoints, counters %s, %.0f s" % (len(feats), n_keypoints, list(counters),
                                                                               time.time() - t0))


if __name__ == "__main__":
    main()
