import os
import sys

import pytest

os.environ.setdefault("GNNA_DEBUG_POISON", "1")   # fresh outputs start as NaN: an element left unwritten must show
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _built_native():
    """Build (no-op when fresh) the oracle and the HIP extension once per session."""
    import oracle
    oracle.build()
    oracle.build_rabbit()
    from gnnadvisor_osdi21_amd import build as gbuild
    gbuild.build_all()
    yield
