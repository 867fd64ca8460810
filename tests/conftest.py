import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # a wedged kernel or a deadlocked rendezvous must end as a failed test, not as a hung run
    if config.pluginmanager.hasplugin("timeout"):
        for item in items:
            if item.get_closest_marker("timeout") is None:
                item.add_marker(pytest.mark.timeout(300))


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc
    orc.lib()
    return orc
