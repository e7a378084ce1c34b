"""A short run of tests/fuzz_parity.py (random sizes, generators, seeds and detector variants, whole path against the
oracle) inside the GPU suite; the long runs are done by hand: 1000 cases / 790 000 features without a difference."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_fuzz_parity_short_run():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_parity.py"), "84", "20261111"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "84 cases, 0 failures" in r.stdout and "pairs of sets matched" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
