import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu`)")


@pytest.fixture(scope="session")
def small_cfg():
    from tests.common import small_config
    return small_config()


def pytest_collection_modifyitems(config, items):
    """the GPU leg calls through the C ABI of the in-tree libdwm_hip.so: build it once if the snapshot came without it"""
    if any("gpu" in item.keywords for item in items) and config.getoption("-m") != "not gpu":
        try:
            from opendwm_amd.build import ensure_built
            ensure_built()
        except Exception as e:                          # the tests themselves then fail loudly in _lib.load()
            print(f"[conftest] libdwm_hip.so could not be built: {e}", file=sys.stderr)
