"""bench.py end to end on a small workload: the JSON line carries the fields its readers rely on."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_small_workload():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "4",
           "--width", "640", "--height", "480", "--cap", "8192", "--cpu-threads", "2", "--exchange"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["unit"] == "images/s" and line["value"] > 0 and line["n_gpus"] == 1
    assert line["steps"] == 2 and line["warmup"] == 1 and line["higher_is_better"] is True
    assert line["dtype"] == "f64" and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert abs(line["value"] - 4 * 1e3 / line["ms_per_step"]) < 1e-6 * line["value"]
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    assert abs(roof["achieved"] * 1e9 * roof["avg_launch_us"] * 1e-6 / roof["algorithmic_bytes_per_launch"] - 1) < 1e-6
    cpu = line["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] == 2 and cpu["value"] > 0
    assert line["config"]["batch"] == 4 and "640x480" in line["config"]["workload"]
    # the untimed extras: every frame of the last batch equals the frame run alone; Sift() through the host layer
    assert line["batch_check"]["ok"] is True and line["batch_check"]["frames"] == 4
    assert line["host_api"]["images_per_s"] > 0 and line["host_api"]["sift_ms_single"] > 0
    assert line["host_api"]["images_per_s_shared_context"] > 0
    # the exchange step ran through the C ABI (a communicator of one rank) and gathered this rank's features unpadded
    assert "sg_allgather_features" in line["exchange"]
    assert line["features_gathered_last_step"] == line["features_last_step"] > 0
