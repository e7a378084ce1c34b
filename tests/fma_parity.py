#!/usr/bin/env python3
"""How far the SG_F64_FMA mode (fused multiply-adds in the Gaussian filters) moves the results away from the oracle's
(the reference's separate multiplies and adds).  Not collected by pytest; run by hand on a GPU box:

    python tests/fma_parity.py [fuzz cases] [--sweep] [--8k] [--out file.json]

Every case runs the same image through a context in SG_F64_FMA mode and through the CPU oracle and counts, instead of
asserting: detection counters equal, keypoint lists equal (integer fields), features present on one side only,
descriptors with any differing bin, largest bin difference, largest differences of the real-valued fields of the
features both sides have.  Features are paired by (octave, interval, x, y) and, among the orientation peaks of one
keypoint, by ascending orientation.  Prints a JSON summary."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sift_amd  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests.fuzz_parity import make  # noqa: E402
import tests.fuzz_parity as FZ  # noqa: E402


class Tally:
    def __init__(self):
        self.cases = self.counters_equal = self.keypoint_lists_equal = self.feature_lists_equal = 0
        self.features_ref = self.features_got = self.paired = self.only_ref = self.only_got = 0
        self.desc_identical = 0
        self.max_bin_diff = 0
        self.bins_differing = 0
        self.max_ori = self.max_scale_rel = self.max_pos = 0.0
        self.cases_all_identical = 0

    def add(self, ref, got, ctr_ref, ctr_got):
        self.cases += 1
        self.features_ref += len(ref)
        self.features_got += len(got)
        ceq = tuple(ctr_ref) == tuple(ctr_got)
        self.counters_equal += ceq

        def keyed(a):
            order = np.lexsort((a["ori"], a["x"], a["y"], a["interval"], a["octave"]))
            a = a[order]
            keys = np.stack([a["octave"], a["interval"], a["y"], a["x"]], axis=1).astype(np.int64)
            return a, keys

        r, rk = keyed(ref)
        g, gk = keyed(got)
        # rank of a feature among those with its key
        def with_rank(keys):
            if len(keys) == 0:
                return np.zeros((0, 5), np.int64)
            new = np.ones(len(keys), bool)
            new[1:] = (keys[1:] != keys[:-1]).any(axis=1)
            start = np.maximum.accumulate(np.where(new, np.arange(len(keys)), 0))
            return np.concatenate([keys, (np.arange(len(keys)) - start)[:, None]], axis=1)

        rk, gk = with_rank(rk), with_rank(gk)
        rset = {tuple(k): i for i, k in enumerate(rk.tolist())}
        pairs = [(rset[tuple(k)], j) for j, k in enumerate(gk.tolist()) if tuple(k) in rset]
        self.paired += len(pairs)
        self.only_ref += len(r) - len(pairs)
        self.only_got += len(g) - len(pairs)
        kp_equal = len(r) == len(g) == len(pairs)
        self.keypoint_lists_equal += kp_equal
        ident_all = ceq and kp_equal
        if pairs:
            pi = np.asarray(pairs)
            rd, gd = r["descriptor"][pi[:, 0]], g["descriptor"][pi[:, 1]]
            diff = np.abs(rd - gd)
            same = (diff == 0).all(axis=1)
            self.desc_identical += int(same.sum())
            self.bins_differing += int((diff != 0).sum())
            self.max_bin_diff = max(self.max_bin_diff, int(diff.max()))
            do = np.abs(r["ori"][pi[:, 0]] - g["ori"][pi[:, 1]])
            do = np.minimum(do, 2 * np.pi - do)
            self.max_ori = max(self.max_ori, float(do.max()))
            self.max_scale_rel = max(self.max_scale_rel, float(np.abs(r["scale"][pi[:, 0]] / g["scale"][pi[:, 1]] - 1).max()))
            self.max_pos = max(self.max_pos, float(np.hypot(r["dx"][pi[:, 0]] - g["dx"][pi[:, 1]],
                                                            r["dy"][pi[:, 0]] - g["dy"][pi[:, 1]]).max()))
            ident_all = ident_all and bool(same.all())
        self.feature_lists_equal += kp_equal and (not pairs or bool(same.all()))
        self.cases_all_identical += ident_all

    def summary(self):
        d = dict(self.__dict__)
        d["descriptors_identical_frac"] = self.desc_identical / self.paired if self.paired else 1.0
        d["recall"] = self.paired / self.features_ref if self.features_ref else 1.0
        return d


def main():
    args = sys.argv[1:]
    n_fuzz = int(args[0]) if args and args[0].isdigit() else 0
    out_path = args[args.index("--out") + 1] if "--out" in args else None
    result = {}
    if n_fuzz:
        rng = np.random.default_rng(1)
        t = Tally()
        ctxs = {}
        for k in range(n_fuzz):
            name, img = make(rng, k)
            ni = k % 3 == 2
            sigma = 1.6 if k % 8 else (1.3 if k % 16 else 2.2)
            if (ni, sigma) not in ctxs:
                ctxs[(ni, sigma)] = sift_amd.SiftGPU(max_width=FZ.MAX_W, max_height=FZ.MAX_H, max_batch=1,
                                                     max_features=FZ.MAX_W * FZ.MAX_H // 2, nointerval=bool(ni), sigma=sigma,
                                                     precision=sift_amd.SG_F64_FMA)
            c = ctxs[(ni, sigma)]
            ref, _, ctr, _ = O.sift(img, sigma=sigma, nointerval=ni)
            got = c.extract(img, cap=FZ.MAX_W * FZ.MAX_H)
            t.add(ref, got, ctr, c.read_counters())
            if k % 100 == 99:
                print("... fuzz %d cases, %d with everything identical" % (k + 1, t.cases_all_identical), flush=True)
        for c in ctxs.values():
            c.close()
        result["fuzz"] = t.summary()
        print("fuzz:", json.dumps(result["fuzz"]), flush=True)
    if "--sweep" in args:
        t = Tally()
        with sift_amd.SiftGPU(max_width=1920, max_height=1080, precision=sift_amd.SG_F64_FMA) as c:
            for seed in (1234, 7, 1000, 1001, 1002, 1003, 1004, 1005):
                img = sift_amd.vnoise5(1920, 1080, seed)
                ref, _, ctr, _ = O.sift(img)
                got = c.extract(img)
                t.add(ref, got, ctr, c.read_counters())
                print("... 1080p seed %d done" % seed, flush=True)
        result["sweep_1080p_8_seeds"] = t.summary()
        print("sweep:", json.dumps(result["sweep_1080p_8_seeds"]), flush=True)
    if "--8k" in args:
        t = Tally()
        img = sift_amd.vnoise5(7680, 4320, 1234)
        print("... 8K oracle running (about two minutes, one thread)", flush=True)
        t0 = time.time()
        import threading
        done = []

        def ticker():
            while not done:
                time.sleep(45)
                print("... %d s" % (time.time() - t0), flush=True)

        th = threading.Thread(target=ticker, daemon=True)
        th.start()
        ref, _, ctr, _ = O.sift(img)
        done.append(1)
        with sift_amd.SiftGPU(max_width=7680, max_height=4320, precision=sift_amd.SG_F64_FMA) as c:
            got = c.extract(img, cap=400000)
            t.add(ref, got, ctr, c.read_counters())
        result["frame_8k_seed_1234"] = t.summary()
        print("8k:", json.dumps(result["frame_8k_seed_1234"]), flush=True)
    if out_path:
        os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
        json.dump(result, open(out_path, "w"), indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
