#!/usr/bin/env python3
"""Static instruction counts of the histogram kernels' sample loops, from the ISA hipcc emits for gfx950.

  python profiles/isa_count.py [out.json]        (default: profiles/isa_counts.json)

Compiles sift_amd/csrc/kernels_describe.hip to assembly with the library's own flags, finds in descr_k<double,false>
and orient_k<double,false> the innermost loops that hold the LDS histogram adds (ds_add_u64 / ds_add_f64) and counts their
instructions by class.  The descriptor loop takes two 64-sample trips per pass (two sample slots in rotation), so
its counts are divided by two; the orientation kernel has two forms of its loop (windows inside the plane, the common one, and
windows that cross the border): the shorter one is reported.  Branches inside a loop make these counts upper bounds
of what one trip executes.  bench.py multiplies them by the measured trip counts to price the kernels against the
vector issue rate (`valu_roofline`)."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "sift_amd", "csrc", "kernels_describe.hip")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
         "-S", "--cuda-device-only"]


def classify(ins):
    op = ins.split()[0]
    if op.startswith("v_"):
        return "valu_f64" if re.match(r"v_\w+_f64", op) or op in ("v_cvt_f64_i32_e32", "v_cvt_i32_f64_e32") else "valu_other"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def loops_with_hist_adds(body):
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    out = []
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            seg = [x.split(";")[0].strip() for x in body[labels[m.group(1)]: i + 1]]
            seg = [x for x in seg if x and not x.startswith(".") and not x.endswith(":")]
            adds = sum(1 for x in seg if x.startswith(("ds_add_f64", "ds_add_u64")))
            if adds:
                out.append((len(seg), adds, seg))
    return out


def kernel_body(lines, mangled_prefix):
    start = next(i for i, l in enumerate(lines) if l.startswith(mangled_prefix) and l.rstrip().endswith(":") is False and ":" in l)
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    return lines[start:end]


def count(seg, div):
    c = {"valu_f64": 0, "valu_other": 0, "salu": 0, "lds": 0, "vmem": 0, "other": 0}
    for ins in seg:
        c[classify(ins)] += 1
    c["transcendental_f64"] = sum(1 for x in seg if re.match(r"v_(rsq|rcp|sqrt|exp|log)_f64", x))
    c["lds_f64_adds"] = sum(1 for x in seg if x.startswith("ds_add_f64"))
    c["lds_u64_adds"] = sum(1 for x in seg if x.startswith("ds_add_u64"))
    return {k: v / div for k, v in c.items()}


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "isa_counts.json")
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "kd.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + [SRC, "-o", asm], stderr=subprocess.DEVNULL)
        lines = open(asm).read().split("\n")
    res = {"_source": "sift_amd/csrc/kernels_describe.hip", "_flags": " ".join(FLAGS[:4])}
    # descr_k<double,false>: the loop with 16 histogram adds (two trips of 8)
    body = kernel_body(lines, "_ZN2sg7descr_kIdLb0EE")
    cand = [l for l in loops_with_hist_adds(body) if l[1] == 16]
    n, adds, seg = min(cand, key=lambda t: t[0])
    res["descr_k<double,false>"] = dict(count(seg, 2.0), trips_per_loop_pass=2)
    body = kernel_body(lines, "_ZN2sg8orient_kIdLb0EE")
    cand = [l for l in loops_with_hist_adds(body) if l[1] == 1]
    n, adds, seg = min(cand, key=lambda t: t[0])
    res["orient_k<double,false>"] = dict(count(seg, 1.0), trips_per_loop_pass=1)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
