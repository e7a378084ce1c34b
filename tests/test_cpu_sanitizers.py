"""The CPU-side code under the sanitizers: the oracle (AddressSanitizer + UndefinedBehaviorSanitizer) over a 640x480
frame and the edge sizes of the parity tests, and the host C++ layer over a stand-in for libsiftgpu
(tests/native/fake_siftgpu.cpp, deterministic records, no GPU): its GPU-free code under ASan + UBSan and its shared
batching contexts (mutex / condition-variable leader-follower hand-over) under ThreadSanitizer.  The programs are built
by `make -C oracle SAN=1` and `make -C sift_amd/host SAN=1`."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BAD = ("ERROR: AddressSanitizer", "runtime error:", "WARNING: ThreadSanitizer", "ERROR: LeakSanitizer")


def _make(directory):
    r = subprocess.run(["make", "-C", os.path.join(ROOT, directory), "SAN=1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def _run(exe, args, timeout=300, env=None):
    e = dict(os.environ)
    e.pop("LD_PRELOAD", None)  # a sanitizer run-time wants to come first in the library list
    e.update(env or {})
    r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=timeout, env=e)
    out = r.stdout + r.stderr
    for marker in BAD:
        assert marker not in out, out[-4000:]
    assert r.returncode == 0, out[-4000:]
    return out


@pytest.fixture(scope="module")
def san_programs():
    _make("oracle")
    _make(os.path.join("sift_amd", "host"))
    return {"oracle": os.path.join(ROOT, "oracle", "san_check"),
            "host": os.path.join(ROOT, "sift_amd", "host", "host_san_check"),
            "tsan": os.path.join(ROOT, "sift_amd", "host", "host_tsan_check")}


def test_oracle_under_asan_ubsan(san_programs, tmp_path):
    out = _run(san_programs["oracle"], [str(tmp_path)])
    assert "640x480 nointerval=0: 3248 features, 2700 keypoints" in out  # the known answer of the 640x480 frame
    assert "oracle sanitizer check ok" in out
    assert (tmp_path / "oracle_san_features.txt").stat().st_size > 1000


def test_host_layer_under_asan_ubsan(san_programs, tmp_path):
    out = _run(san_programs["host"], [str(tmp_path)])
    assert "host sanitizer check ok" in out
    assert (tmp_path / "feat.txt").exists() and (tmp_path / "lvl0.pgm").exists()


@pytest.mark.parametrize("threads,calls,batch", [(8, 25, 4), (12, 10, 16), (3, 30, 2)])
def test_shared_batching_contexts_under_tsan(san_programs, threads, calls, batch):
    out = _run(san_programs["tsan"], [str(threads), str(calls), str(batch)],
               env={"TSAN_OPTIONS": "halt_on_error=1 second_deadlock_stack=1"})
    assert "host thread-sanitizer check ok" in out
