"""The C++11 facade (include/fbstab/): compiles and links against the C-ABI
library on CPU; runs the reference-style end-to-end tests on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "facade_tests")


def _build():
    lib = os.path.join(ROOT, "fbstab_amd", "libfbstab_hip.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "fbstab_amd", "csrc")])
    subprocess.check_call(
        ["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-O1", "-I" + os.path.join(ROOT, "include"),
         "-o", EXE, os.path.join(ROOT, "tests", "cpp", "facade_tests.cc"),
         "-L" + os.path.join(ROOT, "fbstab_amd"), "-lfbstab_hip",
         "-Wl,-rpath," + os.path.join(ROOT, "fbstab_amd")])


def test_facade_compiles_as_cxx11():
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_facade_reference_style_tests():
    _build()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ALL FACADE TESTS PASSED" in r.stdout, r.stdout + r.stderr
