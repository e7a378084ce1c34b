"""The host C++ layer (the reference's sift.h entry points over the C ABI), exercised as compiled programs."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "sift_amd", "host")


def _need(exe):
    path = os.path.join(HOST, exe)
    if not os.path.exists(path):
        pytest.fail("%s is not built (run __graft_entry__.build())" % path)
    return path


def test_demo_flow_matches_oracle_text(oracle, tmp_path):
    """Sift() and Sift_nointerval() + write_features, as in the reference's main(): the text files equal the
    ones the CPU oracle writes for the same images."""
    exe = _need("sift_demo")
    w, h, s1, s2 = 200, 150, 1234, 1234  # the same scene through both detectors, so that matches exist
    o1, o2, om = str(tmp_path / "d1.txt"), str(tmp_path / "d2.txt"), str(tmp_path / "m.txt")
    out = subprocess.run([exe, str(w), str(h), str(s1), str(s2), o1, o2, om], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0, out.stderr
    f1, _, _, _ = oracle.sift(oracle.vnoise5(w, h, s1))
    f2, _, _, _ = oracle.sift(oracle.vnoise5(w, h, s2), nointerval=True)
    r1, r2 = str(tmp_path / "r1.txt"), str(tmp_path / "r2.txt")
    oracle.write_features(f1, r1)
    oracle.write_features(f2, r2)
    assert open(o1, "rb").read() == open(r1, "rb").read()
    assert open(o2, "rb").read() == open(r2, "rb").read()
    assert "features1 %d features2 %d" % (len(f1), len(f2)) in out.stdout
    # MatchFeatures (the demo's matching loop) against the oracle's
    ref, _ = oracle.match(f1, f2, 0.5)
    got = np.array([[int(v) for v in line.split()] for line in open(om)], dtype=np.int32).reshape(-1, 2)
    assert len(ref) > 20 and np.array_equal(got, ref)
    assert "match number: %d" % len(ref) in out.stdout


def test_precision_modes_of_the_host_layer():
    """SiftSetPrecision: the fused and single-precision pyramids give about the exact mode's features, and the exact
    mode afterwards the same records as before, with a context per thread and with the shared contexts."""
    exe = _need("sift_host_check")
    out = subprocess.run([exe, "precision", "400", "300", "4242"], capture_output=True, text=True, timeout=180)
    assert out.returncode == 0, out.stderr + out.stdout
    assert "precision ok" in out.stdout


def test_stage_entry_points_equal_driver():
    exe = _need("sift_stage_chain")
    out = subprocess.run([exe, "160", "120", "77"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr + out.stdout
    assert "stage chain ok" in out.stdout


def test_dense_frame_grows_the_context(oracle, tmp_path):
    """Sift() on a checkerboard of 3-pixel squares: denser in features than the context's default sizing, where the
    reference simply returns its result.  The host layer must size a new context from the reported counts and return the same features."""
    import sift_amd
    exe = _need("sift_host_check")
    w, h, cell = 320, 240, 3
    o = str(tmp_path / "dense.txt")
    out = subprocess.run([exe, "dense", str(w), str(h), str(cell), o], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr + out.stdout
    want, _, _, _ = oracle.sift(sift_amd.synth.checkerboard(w, h, cell))
    assert len(want) > w * h // 40 + 4096
    r = str(tmp_path / "ref.txt")
    oracle.write_features(want, r)
    assert open(o, "rb").read() == open(r, "rb").read()
    assert "dense ok: %d features" % len(want) in out.stdout


def test_four_channel_and_padded_views():
    exe = _need("sift_host_check")
    out = subprocess.run([exe, "views", "200", "150", "31"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr + out.stdout
    assert "views ok" in out.stdout


def test_threads_sharing_one_context_get_their_own_records():
    """SiftSetBatching: 12 threads call Sift()/Sift_nointerval() through one shared context, frames of two sizes and one
    frame denser than the default sizing in the mix; every call returns, byte for byte, what a context of the thread's
    own returned before."""
    exe = _need("sift_host_check")
    out = subprocess.run([exe, "shared", "320", "240", "12", "6", "8"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    assert "shared ok: 12 threads x 6 calls" in out.stdout


def test_shared_context_shrinks_its_batch_for_8k_frames():
    """Two shared contexts of 16 8K frames each do not fit one GPU (an 8K frame takes 10 GB): the layer must come down
    to batch sizes that fit, queue the requests it had collected beyond them again, and still return every caller's
    own records."""
    exe = _need("sift_host_check")
    out = subprocess.run([exe, "shared", "7680", "4320", "5", "1", "16"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr + out.stdout
    assert "shared ok: 5 threads x 1 calls" in out.stdout


@pytest.mark.parametrize("frames,gpus,batch,rehearsal", [(7, 1, 3, 0), (7, 3, 2, 1), (2, 4, 8, 1)])
def test_multi_gpu_driver_equals_sift_per_frame(frames, gpus, batch, rehearsal):
    """SiftBatchMultiGpu (host C++: frames sharded over GPUs, one thread per GPU, all-gather of the feature sets)
    against Sift() on every frame: same features in the same order, dense frames (which overflow the default capacity
    and make every shard run again with a larger one) included.  On the one GPU of the test box: the real path with one
    rank, and the sharding with several ranks in rehearsal mode (ranks share the GPU, sets meet in host memory; the
    last shard short, or more ranks than frames)."""
    exe = _need("sift_multi_check")
    out = subprocess.run([exe, "check", "320", "240", str(frames), str(gpus), str(batch), str(rehearsal)],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    assert "multi ok: %d frames" % frames in out.stdout
