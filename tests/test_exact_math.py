"""wm_exact.h (the arithmetic the kernels use) against this image's glibc, bit for bit."""
import os
import subprocess

from conftest import ROOT


def test_atan2f_restatement_matches_libm(tmp_path):
    exe = str(tmp_path / "exact_math_check")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tests", "exact_math_check.c"), "-lm"],
                   check=True)
    p = subprocess.run([exe, "20000000"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout
    assert " 0 mismatches" in p.stdout
