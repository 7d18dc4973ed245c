import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # make sure both shared libraries exist before any test imports them
    import __graft_entry__ as g

    lib = os.path.join(ROOT, "hare_amd", "libhare_hip.so")
    ora = os.path.join(ROOT, "oracle", "_build", "libhare_oracle.so")
    if not (os.path.exists(lib) and os.path.exists(ora)):
        g.build()


@pytest.fixture(scope="session")
def gpu_available():
    import hare_amd

    return hare_amd.device_count() > 0
