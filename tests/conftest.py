import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def kats():
    with open(os.path.join(GOLDEN, "reference_kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (restated loop).  Test infrastructure only."""
    from oracle.oracle_py import Oracle
    return Oracle(False)


@pytest.fixture(scope="session")
def oracle_fma():
    """The same restatement compiled with fused multiply-adds allowed (oracle/Makefile: `make fma`): the
    second member of the pair {oracle, oracle_fma} a count may come from where only the rounding of the
    reference's own compiler decides it.  Test infrastructure only."""
    from oracle.oracle_py import Oracle
    return Oracle(False, fma=True)


@pytest.fixture(scope="session")
def ref_oracle():
    """Oracle components driven by the reference's own FBstabAlgorithm<>
    template; only available where /root/reference exists (or a prebuilt
    oracle/_ref/libfbstab_ref.so travelled with the snapshot)."""
    from oracle import oracle_py
    so = os.path.join(ROOT, "oracle", "_ref", "libfbstab_ref.so")
    if not os.path.exists(so) and not os.path.isdir(oracle_py.REFERENCE_ROOT):
        pytest.skip("reference tree absent and oracle/_ref not prebuilt")
    return oracle_py.Oracle(True)
