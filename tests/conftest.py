import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the Python-planned 2D pipeline is the comparison route of the suite (GETDIST_AMD_NATIVE_BATCH=0 on the device, the
    # numpy context double on CPU): the product itself has no second orchestration and raises without it
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import planned_route

    planned_route.install()


@pytest.fixture(scope="session")
def zoo():
    from oracle.fixtures import fixture_zoo

    return {fx["name"]: fx for fx in fixture_zoo()}
