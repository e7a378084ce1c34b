#!/usr/bin/env python3
"""Random sizes, generators and seeds through the whole path, every case against the CPU oracle (not collected by
pytest; run by hand on a GPU box):  python tests/fuzz_parity.py [cases] [first seed] [max width] [max height]

Per case: detection counters, refined keypoints bit for bit, feature list (integer fields, descriptor bins) equal,
orientation within 1e-9 rad; both detectors (3x3x3 and in-layer); every eighth case at another base sigma; every tenth case also as the second frame of a
batch; every fifth case's features matched against the previous case's (ratio test), equal match lists.  Prints one line per failure and a summary; exit status 1 on any failure."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sift_amd  # noqa: E402
from sift_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


MAX_W, MAX_H = 420, 320


def make(rng, k):
    w, h = int(rng.integers(8, MAX_W)), int(rng.integers(8, MAX_H))
    seed = int(rng.integers(1, 1 << 30))
    kind = k % 7
    if kind == 0:
        return "vnoise5", synth.vnoise5(w, h, seed)
    if kind == 1:
        return "checker", synth.checkerboard(w, h, int(rng.integers(2, 24)), int(rng.integers(0, 100)), int(rng.integers(150, 256)))
    if kind == 2:
        return "edges", synth.step_edges(w, h, seed, n=int(rng.integers(3, 60)))
    if kind == 3:
        return "blobs", synth.blobs(w, h, seed, n=int(rng.integers(3, 80)))
    if kind == 4:
        return "blocks", synth.saturated_blocks(w, h, seed)
    if kind == 5:
        return "white", synth.white_noise(w, h, seed)
    img = synth.vnoise5(w, h, seed).astype(np.int32) + synth.step_edges(w, h, seed + 1).astype(np.int32) // 3
    return "mixed", np.clip(img, 0, 255).astype(np.uint8)


def same(a, b, fields):
    return len(a) == len(b) and all(np.array_equal(a[f], b[f]) for f in fields)


def main():
    global MAX_W, MAX_H
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    if len(sys.argv) > 4:
        MAX_W, MAX_H = int(sys.argv[3]), int(sys.argv[4])
    ctx = {}

    room = {}  # (detector, sigma) -> features per frame the context was made for

    def context(ni, sigma, need=0):
        """The context of a detector and base sigma; with need: one with room for that many features per frame (an
        image denser than the default sizing -- fine checkerboards at a small sigma -- reports SG_E_CAPACITY and says
        how much it needs, and the case runs again, as the host layer does it)."""
        key = (ni, sigma)
        if key in ctx and need > room[key]:
            ctx.pop(key).close()
        if key not in ctx:
            room[key] = max(MAX_W * MAX_H // 2, need)
            ctx[key] = sift_amd.SiftGPU(max_width=MAX_W, max_height=MAX_H, max_batch=2, max_features=room[key],
                                        nointerval=bool(ni), sigma=sigma)
        return ctx[key]

    bad = 0
    feats_total = 0
    prev = None
    matched = [0]
    only = os.environ.get("FUZZ_ONLY")  # one case number: the others are generated (the sequence stays) but not run
    for k in range(n):
        name, img = make(rng, k)
        if only is not None and k != int(only):
            continue
        h, w = img.shape[:2]
        ni = k % 3 == 2
        sigma = 1.6 if k % 8 else (1.3 if k % 16 else 2.2)  # every eighth case at another base sigma
        c = context(int(ni), sigma)
        try:
            ref, n_kp, ctr, _ = O.sift(img, sigma=sigma, nointerval=ni)
            try:
                got = c.extract(img, cap=MAX_W * MAX_H)
            except sift_amd.SiftGpuError as e:
                if e.status != sift_amd.api.SG_E_CAPACITY:
                    raise
                c = context(int(ni), sigma, need=c.required_features() + 1024)
                got = c.extract(img, cap=max(MAX_W * MAX_H, room[(int(ni), sigma)]))
            problems = []
            if tuple(c.read_counters()) != tuple(ctr):
                problems.append("counters %s != %s" % (c.read_counters(), ctr))
            if not same(got, ref, ("octave", "interval", "x", "y", "dx", "dy", "offset_x", "offset_y", "offset_interval", "val")):
                problems.append("feature list (%d vs %d)" % (len(got), len(ref)))
                if only is not None and len(got) == len(ref):
                    fields = ("octave", "interval", "x", "y", "dx", "dy", "offset_x", "offset_y", "offset_interval", "val")
                    diff = np.zeros(len(ref), bool)
                    for f in fields:
                        diff |= got[f] != ref[f]
                    ka = np.lexsort((np.round(ref["ori"], 6), ref["x"], ref["y"], ref["interval"], ref["octave"]))
                    kb = np.lexsort((np.round(got["ori"], 6), got["x"], got["y"], got["interval"], got["octave"]))
                    ra, gb = ref[ka], got[kb]
                    def bag(a):
                        return sorted((int(o), int(i), int(x), int(y), d.astype(np.uint8).tobytes())
                                      for o, i, x, y, d in zip(a["octave"], a["interval"], a["x"], a["y"], a["descriptor"]))
                    print("    as sets: integer fields equal %s, (position, descriptor) multisets equal %s, scales differing in "
                          "the last bit: %d" % (all(np.array_equal(ra[f], gb[f]) for f in ("octave", "interval", "x", "y")),
                                                bag(ref) == bag(got), int((ra["scale"] != gb["scale"]).sum())))
                    for j in np.nonzero(diff)[0][:8]:
                        print("   ", j, "ref", [ref[f][j] for f in ("octave", "interval", "x", "y")], ref["scale"][j].hex(),
                              ref["ori"][j], "| got", [got[f][j] for f in ("octave", "interval", "x", "y")],
                              got["scale"][j].hex(), got["ori"][j])
            elif len(ref):
                if not np.array_equal(got["descriptor"], ref["descriptor"]):
                    problems.append("descriptor bins differ in %d features" % int((got["descriptor"] != ref["descriptor"]).any(axis=1).sum()))
                d = np.abs(got["ori"] - ref["ori"])
                if np.minimum(d, 2 * np.pi - d).max() > 1e-9:
                    problems.append("orientation off by %g" % np.minimum(d, 2 * np.pi - d).max())
                if not np.allclose(got["scale"], ref["scale"], rtol=1e-13, atol=0):
                    problems.append("scale")
            if k % 10 == 0 and not problems:
                two = c.extract_batch(np.stack([img[::-1].copy(), img]), room[(int(ni), sigma)])[1]
                if len(two) != len(ref) or (len(ref) and not np.array_equal(two["desc"], ref["descriptor"].astype(np.uint8))):
                    problems.append("second frame of a batch differs")
            # every fifth case: the matcher on this case's and the previous checked case's features (checkerboards
            # give many equal distances: first index wins)
            if k % 5 == 0 and prev is not None and len(ref) and len(prev) and len(ref) * len(prev) < 4e7 and not problems:
                for ratio in (0.5, 0.8):
                    want_pairs, _ = O.match(prev, ref, ratio)
                    got_pairs = c.match(prev, ref, ratio)
                    if not np.array_equal(want_pairs, got_pairs):
                        problems.append("matches at ratio %g (%d vs %d)" % (ratio, len(got_pairs), len(want_pairs)))
                matched[0] += 1
            prev = ref
            feats_total += len(ref)
        except Exception as e:  # report and go on
            problems = ["exception %r" % (e,)]
        if problems:
            bad += 1
            print("FAIL case %d %s %dx%d nointerval=%d sigma=%g: %s" % (k, name, w, h, ni, sigma, "; ".join(problems)), flush=True)
        if k % 25 == 24:
            print("... %d cases, %d failures, %d features compared" % (k + 1, bad, feats_total), flush=True)
    print("fuzz: %d cases, %d failures, %d features compared, %d pairs of sets matched" % (n, bad, feats_total, matched[0]))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
