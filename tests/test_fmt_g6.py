"""The driver's "%g" writer (gnumap_amd/csrc/gm_fmt.h) against the C library's printf on 8 M values: floats and doubles of the ranges the
XA / XP columns hold, powers of ten and their neighbours, exact ties of the sixth digit, zeros, infinities, NaN, denormals."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_put_g6_equals_printf(tmp_path):
    exe = tmp_path / "fmt_g6_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "gnumap_amd", "csrc"), os.path.join(ROOT, "tests", "fmt_g6_check.cpp"), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert " 0 mismatches" in r.stdout
