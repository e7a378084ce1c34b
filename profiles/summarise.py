#!/usr/bin/env python3
"""Summarise gpurun_out/prof_<tag>/ (written by profiles/collect.sh) into profiles/<tag>/:
kernel_stats.csv (rocprofv3 --stats, verbatim), summary.md, traffic.json (HBM-side bytes per launch per kernel).

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports half of the bytes of a coalesced streaming
read (16 B and 8 B per lane alike; calibration in profiles/r01/fetch_calibration.md), so it is doubled for the blur
kernels and the extrema scan, whose reads are of those shapes."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles", tag)
os.makedirs(dst, exist_ok=True)


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


stats = max(glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv")), key=os.path.getmtime)
shutil.copy(stats, os.path.join(dst, "kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench_stats.json"), os.path.join(dst, "bench_under_rocprof.json"))


def pmc(kind):
    f = max(glob.glob(os.path.join(src, kind, "*", "*counter_collection.csv")), key=os.path.getmtime)
    acc = defaultdict(lambda: [0.0, set()])
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        acc[k][0] += float(r["Counter_Value"])
        acc[k][1].add(r["Dispatch_Id"])
    return {k: (v[0] * 1024.0, len(v[1])) for k, v in acc.items()}


fetch, write = pmc("fetch"), pmc("write")
rows = list(csv.DictReader(open(stats)))
try:
    _batch = json.load(open(os.path.join(src, "bench_stats.json")))["config"]["batch"]
except (OSError, ValueError, KeyError):
    _batch = 256
traffic = {"_config": {"batch": _batch, "width": 1920, "height": 1080, "precision": "f64", "command": "python bench.py"}}
lines = ["# rocprofv3 summary (%s): `python bench.py` (4 steps x %d frames 1920x1080, f64, one context)" % (tag, _batch), "",
         "| kernel | calls | avg us | total ms | % | FETCH_SIZE MB/launch (x2 for blur and detect) | WRITE_SIZE MB/launch |", "|---|---|---|---|---|---|---|"]
for r in rows[:24]:
    k = short(r["Name"])
    fb, fn = fetch.get(k, (0.0, 0))
    wb, wn = write.get(k, (0.0, 0))
    mult = 2.0 if ("blur_" in k or "detect_k" in k) else 1.0  # calibrated shapes: r01/fetch_calibration.md
    f_l = mult * fb / fn if fn else 0.0
    w_l = wb / wn if wn else 0.0
    traffic[k] = {"fetch_bytes_per_launch": f_l, "write_bytes_per_launch": w_l, "fetch_correction": mult,
                  "hbm_bytes_per_launch": f_l + w_l}
    lines.append("| `%s` | %s | %.1f | %.2f | %s | %.1f | %.1f |" % (k[:70], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                     float(r["TotalDurationNs"]) / 1e6, r["Percentage"],
                                                                     f_l / 1e6, w_l / 1e6))
open(os.path.join(dst, "summary.md"), "w").write("\n".join(lines) + "\n")
json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print("\n".join(lines))
