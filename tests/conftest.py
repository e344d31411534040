import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref (the reference built from /root/reference)")


@pytest.fixture(scope="session")
def checkers():
    """Build the parity checkers once (oracle always; oracle/_ref where /root/reference exists)."""
    import tntcheck as tc
    tc.build_oracle()
    return tc


@pytest.fixture(scope="session")
def oracle(checkers):
    return checkers.OracleImpl()


@pytest.fixture(scope="session")
def ref(checkers):
    if not checkers.have_ref():
        pytest.skip("oracle/_ref/libtntref.so not present (needs /root/reference to build)")
    return checkers.RefImpl()


@pytest.fixture(scope="session")
def hip_lib():
    """Compile (cross-compile on CPU boxes) and load the HIP library: no compute calls."""
    import __graft_entry__ as g
    g.build_hip()
    from thermonucleotideblast_amd import engine
    return engine.load_library()
