import importlib
import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def mi():
    return importlib.import_module("mitsuba-im_amd")


@pytest.fixture(scope="session")
def oracle():
    import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def golden_scenes():
    from tests.golden.make_golden import golden_scenes as gs
    return gs()


GOLDEN = os.path.join(ROOT, "tests", "golden")
