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
    # ... and checked itself: own slot, counts, checksum
    chk = line["exchange_check"]
    assert chk["ok"] is True and chk["ranks"] == 1 and chk["features_gathered"] == line["features_last_step"]
    assert chk["own_slot_equal"] and chk["counts_equal"] and chk["checksum_equal"]
    # the tolerance mode beside the headline: same keypoints and bins on value-noise frames (640x480 here, so the
    # throughput half of the key, which is measured at 1080p, is absent)
    assert line["f64_fma"] is None or line["f64_fma"].get("bins_identical") in (True, False)


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal_checks_the_exchange():
    """The N > 1 control flow of bench.py on the one GPU of the test box: two ranks launched as the driver launches
    them, sharing the device, the exchange over gloo (RCCL refuses two ranks on one device).  The gathered set of the
    last step must pass its self-check on both ranks: own slot, counts, checksum equal across ranks."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    # SIFT_BENCH_PREFLIGHT=force: the trial of the C ABI's exchange runs although the backend is gloo; its children
    # cannot build an RCCL communicator of two ranks on one device, so the parents must agree on the fallback.
    env = dict(os.environ, SIFT_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", SIFT_BENCH_PREFLIGHT="force",
               SIFT_BENCH_PREFLIGHT_TIMEOUT="90")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--batch", "3", "--pool", "6", "--width", "320", "--height", "240", "--cap", "4096",
           "--no-extras", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and "gloo" in line["exchange"]
    # the children met over their own rendezvous and failed where they must: at the communicator of two ranks on one GPU
    assert line["exchange_preflight"]["ok"] is False and "entering sg_comm_create" in line["exchange_preflight"]["this_rank"]
    assert line["exchange_preflight"]["form"] is None and line["exchange_preflight"]["p2p"]["ok"] is False
    chk = line["exchange_check"]
    assert chk["ok"] is True and chk["ranks"] == 2 and chk["checksum_equal"] and chk["counts_equal"] and chk["own_slot_equal"]
    assert chk["features_gathered"] == line["features_gathered_last_step"] > 2 * 3 * 300


@pytest.mark.gpu
def test_exchange_preflight_child_world_of_one():
    """The trial exchange bench.py runs beside every rank before a multi-rank measurement, here as a world of one:
    communicator through the C ABI's RCCL binding, two all-gathers of patterned records, every byte compared."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--preflight-child"], capture_output=True,
                       text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    forms = {d["form"]: d for d in (json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{"))}  # (RCCL prints too)
    assert forms["bcast"]["ok"] is True and forms["p2p"]["ok"] is True and forms["bcast"]["ms"] > 0
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--preflight-child"], capture_output=True,
                       text=True, timeout=300, cwd=ROOT, env=dict(env, SIFT_BENCH_PREFLIGHT_FAIL="1"))
    assert r.returncode == 5
