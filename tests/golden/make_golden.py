#!/usr/bin/env python3
"""Writes the golden fixtures of tests/golden/ from the CPU oracle (oracle/sift_oracle.c).

The reference itself cannot be built in this image (it needs OpenCV), so these vectors come from the
oracle, whose full output on vnoise5(640,480,1234) matches the checksum recorded from the reference's own
pipeline (tests/test_cpu_oracle.py).  Fixtures are data only: inputs are regenerated from (width, height, seed).

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402

CASES = [(96, 64, 1), (160, 120, 4321), (200, 150, 1234)]


def main():
    index = []
    for w, h, seed in CASES:
        img = O.vnoise5(w, h, seed)
        dbg = O.Debug(img)
        f = dbg.sorted_features
        name = "sift_%dx%d_s%d" % (w, h, seed)
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            counters=np.array(dbg.counters, dtype=np.int64),
            kp_xyoi=np.stack([dbg.keypoints[k] for k in ("x", "y", "octave", "interval")], axis=1).astype(np.int32),
            kp_f64=np.stack([dbg.keypoints[k] for k in ("offset_x", "offset_y", "offset_interval", "dx", "dy", "scale",
                                                         "octave_scale", "val")], axis=1),
            feat_xyoi=np.stack([f[k] for k in ("x", "y", "octave", "interval")], axis=1).astype(np.int32),
            feat_f64=np.stack([f[k] for k in ("dx", "dy", "scale", "ori")], axis=1),
            feat_desc=f["descriptor"].astype(np.uint8),
            level_checksums=np.array([[float(g.sum()), float(np.abs(g).max())] for g in dbg.gauss]),
            base_level=dbg.gauss[0] if w * h <= 96 * 64 else np.zeros((0, 0)),
        )
        index.append({"name": name, "width": w, "height": h, "seed": seed, "octaves": dbg.octaves,
                      "keypoints": int(len(dbg.keypoints)), "features": int(len(f)), "counters": list(dbg.counters)})
    # blur border vectors: planes smaller than the 27-tap kernel
    rng = np.random.default_rng(42)
    small = {}
    for (hh, ww) in [(8, 15), (2, 3), (16, 30)]:
        src = rng.random((hh, ww))
        small["src_%dx%d" % (ww, hh)] = src
        small["blur27_%dx%d" % (ww, hh)] = O.gaussian_blur(src, 3.0900155872895909)
    np.savez_compressed(os.path.join(HERE, "blur_small_planes.npz"), **small)
    # matching (the demo's loop after the path): two overlapping crops of one scene
    w, h, seed, sx, sy = 200, 150, 77, 7, 5
    big = O.vnoise5(w + sx, h + sy, seed)
    fa, _, _, _ = O.sift(np.ascontiguousarray(big[:h, :w]))
    fb, _, _, _ = O.sift(np.ascontiguousarray(big[sy:sy + h, sx:sx + w]))
    pairs, best2 = O.match(fa, fb, 0.5)
    pairs8, _ = O.match(fa, fb, 0.8)
    np.savez_compressed(os.path.join(HERE, "match_200x150_s77.npz"), desc_a=fa["descriptor"].astype(np.uint8),
                        desc_b=fb["descriptor"].astype(np.uint8), pairs_05=pairs, pairs_08=pairs8,
                        best2=best2.astype(np.uint32), geometry=np.array([w, h, seed, sx, sy]))
    json.dump(index, open(os.path.join(HERE, "index.json"), "w"), indent=1)
    print("wrote", [c["name"] for c in index])


if __name__ == "__main__":
    main()
