"""The double-double routines behind the two discrete decisions that hang on a C-library function's last bit
(sift_amd/csrc/sg_dd.h: 2^y for the scale sort key, atan2 next to an orientation-bin boundary), compiled for the host
and compared with the C library itself: equal on the special arguments, and on all but about one in a thousand random
ones (the C library's own results are not always correctly rounded)."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dd_routines_agree_with_libm(tmp_path):
    exe = str(tmp_path / "dd_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "sift_amd", "csrc"),
                    "-o", exe, os.path.join(ROOT, "tests", "dd_check.cpp")], check=True)
    out = subprocess.run([exe, "400000"], capture_output=True, text=True, check=True).stdout
    m = re.search(r"exp2: (\d+) of (\d+) differ.*atan2: (\d+) of (\d+) differ \((\d+) of (\d+) next.*specials differing (\d+)", out)
    assert m, out
    p_bad, p_n, a_bad, a_n, near_bad, near_n, specials = map(int, m.groups())
    assert specials == 0, out
    assert p_bad <= 0.003 * p_n and a_bad <= 0.003 * a_n and near_bad <= 0.003 * near_n, out
