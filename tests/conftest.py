import os
import sys

import pytest

# The CPU oracle is OpenMP code: without a cap it starts one thread per core the HOST has, which on a GPU box whose cgroup
# grants 16 of many cores oversubscribes badly (the GPU suite took 5x longer on such boxes).  Must be set before libgomp loads.
os.environ.setdefault("OMP_NUM_THREADS", "8")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle
    oracle.build()
    return oracle
