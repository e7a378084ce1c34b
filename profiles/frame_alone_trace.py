#!/usr/bin/env python3
"""One 1080p frame alone through the device-pointer API, a few dozen times: the program whose kernel trace
(rocprofv3 --kernel-trace) shows the dependent-launch chain of a single frame.
    python profiles/frame_alone_trace.py run [W H] [n] [B]    the workload (put it behind `rocprofv3 ... --`); B frames per call
    python profiles/frame_alone_trace.py read <kernel_trace.csv>    timeline of the last frame: start, duration, queue"""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(w, h, n, B=1):
    import time
    import torch
    import sift_amd
    dev = torch.device("cuda", 0)
    ctx = sift_amd.SiftGPU(max_width=w, max_height=h, max_batch=B, device=0)
    frame = torch.empty(B * w * h * 3, dtype=torch.uint8, device=dev)
    for i in range(B):
        ctx.synth_vnoise5(frame.data_ptr() + i * w * h * 3, w, h, 1234 + i)
    cap = w * h // 40 + 4096
    out = torch.empty(B * cap * 144, dtype=torch.uint8, device=dev)
    cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    ctx.sync()
    ts = []
    for i in range(n):
        t0 = time.perf_counter()
        ctx.extract_batch_device(frame.data_ptr(), B, w, h, 3, out.data_ptr(), cap, cnt.data_ptr())
        ctx.sync()
        ts.append(time.perf_counter() - t0)
    print("features %d, median %.3f ms, best %.3f ms%s" % (int(cnt[0].item()), 1e3 * sorted(ts)[len(ts) // 2], 1e3 * min(ts),
                                                          "" if B == 1 else " per call of %d frames" % B))
    ctx.close()


def read(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last frame starts at the last gray_k
    starts = [i for i, r in enumerate(rows) if "gray_k" in r["Kernel_Name"]]
    first = starts[-1]
    t0 = int(rows[first]["Start_Timestamp"])
    end_prev = t0
    print("%9s %8s %7s  %-5s %s" % ("start us", "dur us", "gap us", "queue", "kernel"))
    for r in rows[first:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sg::", "")
        print("%9.1f %8.1f %7.1f  %-5s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - end_prev) / 1e3, r.get("Queue_Id", "?"), name[:70]))
        end_prev = max(end_prev, e)
    print("frame: %.1f us from the first kernel's start to the last kernel's end" % ((end_prev - t0) / 1e3))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "read":
        read(sys.argv[2])
    else:
        a = sys.argv[2:]
        run(int(a[0]) if len(a) > 1 else 1920, int(a[1]) if len(a) > 1 else 1080, int(a[2]) if len(a) > 2 else 30,
            int(a[3]) if len(a) > 3 else 1)
