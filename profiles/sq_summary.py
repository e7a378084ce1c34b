#!/usr/bin/env python3
"""Per-kernel sums of the SQ counter passes written by profiles/collect_sq.sh (steady-state step only:
the second half of each kernel's dispatches, the first half being the warm-up step)."""
import csv
import glob
import os
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
write = tag != "tmp"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)


def short(name):
    return name.split("(")[0].replace("void ", "").replace("sg::", "").strip()


out = []
for p in ("sq_a", "sq_b"):
    f = max(glob.glob(os.path.join(src, p, "*", "*counter_collection.csv")), key=os.path.getmtime)
    acc = defaultdict(lambda: defaultdict(float))
    nd = defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        nd[k].add(r["Dispatch_Id"])
    names = sorted({c for v in acc.values() for c in v})
    out.append("| kernel | launches | " + " | ".join(n.replace("SQ_", "") for n in names) + " |")
    out.append("|---|---|" + "---|" * len(names))
    for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
        if "vnoise" in k or "rocclr" in k or "at::" in k:
            continue
        out.append("| `%s` | %d | " % (k[:44], len(nd[k])) + " | ".join("%.3g" % acc[k][n] for n in names) + " |")
    out.append("")
txt = "\n".join(out)
print(txt)
dst = os.path.join(root, "profiles", tag)
if write:
    os.makedirs(dst, exist_ok=True)
if write:
    open(os.path.join(dst, "sq_counters.md"), "w").write(
    "# SQ counter sums per kernel over two steps (warm-up + timed) of `python bench.py --steps 1 --warmup 1`\n\n" + txt + "\n")
