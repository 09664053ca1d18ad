import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "gpflow-slim_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_addoption(parser):
    parser.addoption("--shuffle", type=int, default=None, metavar="SEED",
                     help="run the collected tests in a seeded random order (state-leak hunting; -1 = reversed)")


def pytest_collection_modifyitems(config, items):
    seed = config.getoption("--shuffle")
    if seed is None:
        return
    if seed < 0:
        items.reverse()
    else:
        import random
        random.Random(seed).shuffle(items)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def handle():
    """The product's default device handle.  No fallback: without a GPU this raises."""
    import gpflowSlim
    return gpflowSlim.get_handle()
