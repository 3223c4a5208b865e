import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a checkout without build artefacts (the .so files are git-ignored): compile once, as the driver's build() does
    import __graft_entry__
    __graft_entry__.ensure_built()


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import ref
    ref.build()
    return ref
