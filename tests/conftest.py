import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True, scope="session")
def _cpu_feeder_for_host_tests():
    """The product's datasets only feed from the GPU (HipFeeder); datasets built on a CPU device inside the
    host-logic tests get the torch-CPU stand-in of tests/_cpu_feeder.py."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _cpu_feeder import feeder_for
    from plenoctree_amd.nerf_sh.nerf import datasets
    old = datasets.Dataset.feeder_factory
    datasets.Dataset.feeder_factory = staticmethod(feeder_for)
    yield
    datasets.Dataset.feeder_factory = old
