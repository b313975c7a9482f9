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
        subprocess.check_call(["make", "-j6", "-C", os.path.join(ROOT, "fbstab_amd", "csrc")])
    subprocess.check_call(
        ["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-O1", "-I" + os.path.join(ROOT, "include"),
         "-o", EXE, os.path.join(ROOT, "tests", "cpp", "facade_tests.cc"),
         "-I/opt/rocm/include", "-L" + os.path.join(ROOT, "fbstab_amd"), "-lfbstab_hip",
         "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + os.path.join(ROOT, "fbstab_amd"),
         "-Wl,-rpath,/opt/rocm/lib"])


def test_facade_compiles_as_cxx11():
    _build()
    assert os.path.exists(EXE)


def test_eigen_branch_of_the_facade_compiles():
    """include/fbstab/dense_types.h switches to the real Eigen types when <Eigen/Dense>
    exists (the reference's ProblemData / Variable are Eigen types,
    fbstab/fbstab_dense.h:55-107).  The image has no Eigen: the branch is compiled,
    syntax only and with -Werror, against tests/cpp/eigen_mock (Eigen's types without
    its arithmetic), with the reference's own call forms - owning Eigen matrices,
    Eigen::Map views, Eigen::Vector4d sizes."""
    subprocess.check_call(
        ["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only",
         "-I" + os.path.join(ROOT, "tests", "cpp", "eigen_mock"), "-I" + os.path.join(ROOT, "include"),
         os.path.join(ROOT, "tests", "cpp", "eigen_branch_compile.cc")])


@pytest.mark.gpu
def test_facade_reference_style_tests():
    _build()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ALL FACADE TESTS PASSED" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_facade_iter_display_is_the_reference_display():
    """Display::FINAL (the reference's default) / ITER / ITER_DETAILED through the facade and a user OutputStream:
    the text equals what the reference prints for the same problems
    (tests/golden/reference_display.json, produced by the reference's own print
    functions), numbers to 5e-4 relative / 1e-7 absolute."""
    import json
    import re
    from tests import helpers as H
    _build()
    r = subprocess.run([EXE, "display"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    blocks = re.findall(r"===BEGIN (\w+) (\w+) (\d)===\n(.*?)===END===", r.stdout, flags=re.S)
    assert len(blocks) == 7
    with open(os.path.join(ROOT, "tests", "golden", "reference_display.json")) as f:
        golden = json.load(f)["cases"]
    for kind, name, level, text in blocks:
        g = [c for c in golden if (c["kind"], c["name"], c["level"]) == (kind, name, int(level))][0]
        ok, why = H.display_texts_agree(H.normalise_time(text), g["text"])
        assert ok, (kind, name, level, why, text)
