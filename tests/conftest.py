import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(params=["auto", "mc"])
def stream_family(request, monkeypatch):
    """Which streaming-kernel family a batch runs on.  "auto": the library's rule -- one chain k_stream<1>, more chains k_stream<2> in pairs
    while the problem is small (blocks x chain pairs <= 320: every fixture of these tests), a matrix-core kernel (k_stream_sep / k_stream_mc)
    from three chains up otherwise.  "mc": MAGI_STREAM_FAMILY=mc, every batch on the matrix-core kernel -- so that tests on small fixtures
    cover both families."""
    if request.param == "mc":
        monkeypatch.setenv("MAGI_STREAM_FAMILY", "mc")
    else:
        monkeypatch.delenv("MAGI_STREAM_FAMILY", raising=False)
    return request.param
