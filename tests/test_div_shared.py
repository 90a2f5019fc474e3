"""rt_device.h `divs` (one reciprocal for the three components of a vector division) must give what three IEEE
divisions give: checked on the host with the same arithmetic (tests/native/div_shared_check.c)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shared_reciprocal_division_is_exact(tmp_path):
    exe = str(tmp_path / "div_shared_check")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tests", "native", "div_shared_check.c"), "-lm"],
                   check=True)
    r = subprocess.run([exe, "20000000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    assert ", 0 mismatches" in r.stdout
    assert ", 0 other differences" in r.stdout          # the sign of a zero quotient is the only thing the sign step changes
