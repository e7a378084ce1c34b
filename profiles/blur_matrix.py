#!/usr/bin/env python3
"""Runs bench.py under a list of environment / precision settings and prints the pyramid-side stage times and rates
of each run side by side (one line per run).  Usage: python profiles/blur_matrix.py out.txt name:ENV=1,ENV2=0:precision ..."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    os.makedirs(os.path.dirname(os.path.abspath(sys.argv[1])), exist_ok=True)
    out = open(sys.argv[1], "a")
    for spec in sys.argv[2:]:
        parts = spec.split(":")
        name, envs, prec = parts[0], parts[1] if len(parts) > 1 else "", parts[2] if len(parts) > 2 else "f64"
        extra = parts[3].split() if len(parts) > 3 else []
        env = dict(os.environ)
        if name.startswith("lib_"):  # lib_<variant>_...: a library built by profiles/build_variant.sh
            env["SIFTGPU_LIB"] = os.path.join(ROOT, "build", "variants", "lib_%s.so" % name.split("_")[1])
        for kv in filter(None, envs.split(",")):
            k, v = kv.split("=")
            env[k] = v
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-extras",
               "--no-cpu-baseline", "--precision", prec] + extra
        p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=400)
        if p.returncode != 0:
            line = "%-28s FAILED rc=%d %s" % (name, p.returncode, p.stderr.strip()[-300:])
        else:
            r = json.loads(p.stdout.strip().splitlines()[-1])
            st = r["stage_ms_per_step"]
            bk = r["blur_kernels"]
            rf = r["roofline"]
            line = ("%-28s %7.1f img/s step %6.1f ms | base %5.2f pair %5.2f L1 %5.2f L2 %5.2f L3 %5.2f L4 %5.2f L5 %5.2f det %5.2f "
                    "| blurGB/s all %5.0f | %s" %
                    (name, r["value"], r["ms_per_step"], st.get("blur_col0", 0), st.get("blur_pair", 0), st.get("blur_col1", 0),
                     st.get("blur_col2", 0), st.get("blur_col3", 0), st.get("blur_col4", 0), st.get("blur_col5", 0),
                     st.get("detect", 0), rf["all_blur_launches_GBps"] or 0,
                     " ".join("%s=%.0f" % (k.replace("blur_", ""), v["GBps"]) for k, v in bk.items()) +
                     " det=%.0f" % r["detect_kernel"]["GBps"] +
                     " | refine %.2f orient %.2f descr %.2f sort %.2f ms" % (st.get("refine", 0), st.get("orient", 0),
                                                                             st.get("descr", 0), st.get("sort", 0))))
        print(line, flush=True)
        out.write(line + "\n")
        out.flush()


if __name__ == "__main__":
    main()
