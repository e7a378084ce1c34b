#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection CSV: per kernel name, sum of each counter and dispatch count."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
for r in csv.DictReader(open(path)):
    name = r["Kernel_Name"].split("(")[0][:70]
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[name].add(r["Dispatch_Id"])
for name in sorted(acc, key=lambda n: -acc[n].get("SQ_WAVE_CYCLES", acc[n].get("GRBM_GUI_ACTIVE", 0))):
    print("%-72s n=%4d " % (name, len(cnt[name])) + " ".join("%s=%.4g" % (k, v) for k, v in sorted(acc[name].items())))
